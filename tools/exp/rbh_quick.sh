timeout 600 python tools/bench_pair32.py 2>&1 | grep -v amdgpu.ids | tail -4
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so python tools/time_pair_rbh.py 2>&1 | grep -v amdgpu.ids
