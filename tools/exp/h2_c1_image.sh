V=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_exp.so
RVC_HIP_LIB=$V python tools/exp/h2_c1_image.py
echo "== full pair, sync between the halves"; RVC_HIP_LIB=$V RVC_EXP_PAIR_SYNC=1 python tools/exp/h2_debug2.py 2>&1 | grep -v "   tile"
echo "== full pair, image zeroed first";      RVC_HIP_LIB=$V RVC_EXP_PAIR_ZERO=1 python tools/exp/h2_debug2.py 2>&1 | grep -v "   tile"
