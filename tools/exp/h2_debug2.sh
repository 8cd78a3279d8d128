V=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_x3pcheck.so
python tools/exp/h2_debug2.py
echo "=== variant R=4"; RVC_HIP_LIB=$V RVC_X3Q_R=4 python tools/exp/h2_debug2.py
echo "=== variant R=3"; RVC_HIP_LIB=$V RVC_X3Q_R=3 python tools/exp/h2_debug2.py
echo "=== variant WGS=1"; RVC_HIP_LIB=$V RVC_X3Q_WGS=1 python tools/exp/h2_debug2.py
echo "=== variant XCD=0"; RVC_HIP_LIB=$V RVC_X3_XCD=0 python tools/exp/h2_debug2.py
