mkdir -p gpurun_out/reorder
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "resblock_pair or fp16x2" > gpurun_out/reorder/pytest_pairs.txt 2>&1; tail -2 gpurun_out/reorder/pytest_pairs.txt; grep -E "^E  " gpurun_out/reorder/pytest_pairs.txt | head -5
rm -f gpurun_out/fullsize_parity.json
timeout 1500 python -m pytest tests/test_hip_pipeline.py -m gpu -q -k "c3_30s or c4_30s or 45s_clip or heavy" > gpurun_out/reorder/pytest_fullsize.txt 2>&1; tail -2 gpurun_out/reorder/pytest_fullsize.txt; grep -E "^FAILED|^E  " gpurun_out/reorder/pytest_fullsize.txt | head
for rep in 1 2; do
  RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_exp.so timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/reorder/bench_old_$rep.json
  timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/reorder/bench_new_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/reorder/bench_*.json')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['config'].get('one_clip_alone_ms'), r['kernel_ms_per_clip'], r['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
