mkdir -p gpurun_out/par3
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_models.py -m gpu -q -x > gpurun_out/par3/pytest.txt 2>&1; tail -3 gpurun_out/par3/pytest.txt; grep -E "^FAILED|^E  " gpurun_out/par3/pytest.txt | head
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/par3/bench_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/par3/bench_*.json')):
    d=json.loads(open(f).read()); r=d['roofline']
    print(f, d['value'], d['config'].get('one_clip_alone_ms'), r['kernel_ms_per_clip'], r['frac'])
PY
