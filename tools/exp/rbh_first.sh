mkdir -p gpurun_out/rbh
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "fused_resblock_pair" > gpurun_out/rbh/pytest_pairs.txt 2>&1; tail -3 gpurun_out/rbh/pytest_pairs.txt; grep -E "^E  " gpurun_out/rbh/pytest_pairs.txt | head
timeout 600 python tools/bench_pair32.py 2>&1 | tee gpurun_out/rbh/bench_pair32.txt
rm -f gpurun_out/fullsize_parity.json
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_models.py -m gpu -q -k "c3_30s or c4_30s or 45s_clip or heavy" > gpurun_out/rbh/pytest_fullsize.txt 2>&1; tail -3 gpurun_out/rbh/pytest_fullsize.txt; grep -E "^FAILED|^E  " gpurun_out/rbh/pytest_fullsize.txt | head
python - <<'PY'
import json
d=json.load(open('gpurun_out/fullsize_parity.json'))
for k,v in sorted(d.items()): print(k, {a:v[a] for a in ('max','p9999','mean','within','coarse_max_diff','voicing_equal') if a in v})
PY
for m in 0 1; do RVC_H2=$m timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/rbh/bench_h2_${m}.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/rbh/bench_h2_*.json')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['config'].get('one_clip_alone_ms'), r['kernel'], r['kernel_ms_per_clip'], r['frac'], [ (k[0],k[1]) for k in r['others']][:5])
    except Exception as e: print(f, 'ERR', e)
PY
