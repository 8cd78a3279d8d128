export RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_exp.so
bash tools/ab.sh "A=0" "RVC_X3Q_R=4" "RVC_X3Q_R=3" "RVC_X3Q_WGS=2" "RVC_X3_WIDE_K3=0" "RVC_X3_WIDE_XS7=0" "RVC_X3PF64=3" "RVC_RBH=0" "A=1" 2>&1 | grep -v amdgpu.ids
