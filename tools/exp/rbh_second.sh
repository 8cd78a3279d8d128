mkdir -p gpurun_out/rbh4
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "fused_resblock_pair" > gpurun_out/rbh4/pytest_pairs.txt 2>&1; tail -3 gpurun_out/rbh4/pytest_pairs.txt; grep -E "^E  " gpurun_out/rbh4/pytest_pairs.txt | head
timeout 600 python tools/bench_pair32.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rbh4/bench_pair32.txt
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_timing.so python tools/time_pair_rbh.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rbh4/phase_cycles.txt
timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/rbh4/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/rbh4/bench.json').read()); r=d['roofline']
print(d['value'], d['config'].get('one_clip_alone_ms'), r['kernel'], r['kernel_ms_per_clip'], r['frac'], [ (k[0],k[1]) for k in r['others']][:6])
PY
