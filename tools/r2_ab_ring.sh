# round 2: unit-pipelined loop (RVC_X3_PIPE), weight-slab ring (RVC_X3_WBUFS), XCD tile remap (RVC_X3_XCD): A/B inside one box
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_models.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r2c_tests.log
cat gpurun_out/r2c_tests.log
for cfg in "RVC_X3_PIPE=0 RVC_X3_WBUFS=2" "RVC_X3_PIPE=0 RVC_X3_WBUFS=3" "RVC_X3_PIPE=1 RVC_X3_WBUFS=2" "RVC_X3_PIPE=1 RVC_X3_WBUFS=3" "RVC_X3_PIPE=1 RVC_X3_WBUFS=4" "RVC_X3_PIPE=0 RVC_X3_WBUFS=2"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/bench_conv.py gen hubert 2>&1 | grep -v "^ \|amdgpu.ids"
done > gpurun_out/r2c_bench_conv.log 2>&1
cat gpurun_out/r2c_bench_conv.log
bash tools/ab.sh "RVC_X3_PIPE=0 RVC_X3_WBUFS=2" "RVC_X3_PIPE=1 RVC_X3_WBUFS=2" "RVC_X3_PIPE=1 RVC_X3_WBUFS=3" "RVC_X3_PIPE=0 RVC_X3_WBUFS=3" "RVC_X3_PIPE=0 RVC_X3_WBUFS=2" "RVC_X3_PIPE=1 RVC_X3_WBUFS=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c_ab.log
