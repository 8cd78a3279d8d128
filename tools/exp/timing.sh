V=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so
RVC_HIP_LIB=$V python tools/time_pair_rbh.py 2>&1 | grep -v amdgpu.ids
RVC_HIP_LIB=$V python tools/time_pair_x3q.py 2>&1 | grep -v amdgpu.ids
