#!/bin/bash
# round 6, first GPU call of the fp16x2 pair arithmetic: unit tests (both arithmetics), full-size parity per arithmetic, bench A/B in one box
mkdir -p gpurun_out/h2
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "resblock_pair or fp16x2" > gpurun_out/h2/pytest_pairs.txt 2>&1; echo "pairs rc $?" >> gpurun_out/h2/pytest_pairs.txt
for m in 1 0; do
  rm -f gpurun_out/fullsize_parity.json
  RVC_H2=$m timeout 900 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q -k "c3_30s or c4_30s or 45s_clip" > gpurun_out/h2/pytest_fullsize_h2_$m.txt 2>&1; echo "rc $?" >> gpurun_out/h2/pytest_fullsize_h2_$m.txt
  cp gpurun_out/fullsize_parity.json gpurun_out/h2/fullsize_parity_h2_$m.json
done
for rep in 1 2; do for m in 0 1; do
  RVC_H2=$m timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/h2/bench_h2_${m}_$rep.json
done; done
tail -3 gpurun_out/h2/pytest_pairs.txt
for m in 1 0; do tail -2 gpurun_out/h2/pytest_fullsize_h2_$m.txt; cat gpurun_out/h2/fullsize_parity_h2_$m.json | tr -d '\n '; echo; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/h2/bench_h2_*.json')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['config'].get('one_clip_alone_ms'), r['kernel'], r['kernel_ms_per_clip'], r['frac'], [ (k[0],k[1]) for k in r['others']][:4])
    except Exception as e: print(f, 'ERR', e)
PY
