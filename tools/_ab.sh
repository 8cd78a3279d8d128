timeout 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or conv" 2>&1 | tail -2
echo "== pipe"; timeout 120 python tools/bench_conv.py gen hubert 2>&1 | grep -v "^ "
echo "== nopipe"; RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_nopipe.so timeout 120 python tools/bench_conv.py gen hubert 2>&1 | grep -v "^ "
for i in 1 2; do
for v in "" comfy-rvc_amd/csrc/variants/librvc_hip_nopipe.so; do RVC_HIP_LIB=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LIB=$v', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"; done; done
