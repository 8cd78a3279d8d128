#!/bin/bash
mkdir -p gpurun_out/h2c
rm -f gpurun_out/fullsize_parity.json
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "resblock_pair or fp16x2 or register_direct" > gpurun_out/h2c/pytest_pairs.txt 2>&1; echo "pairs rc $?" >> gpurun_out/h2c/pytest_pairs.txt
RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_x3pcheck.so timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -k "resblock_pair or fp16x2 or register_direct" > gpurun_out/h2c/pytest_pairs_check.txt 2>&1; echo "pairs(check build) rc $?" >> gpurun_out/h2c/pytest_pairs_check.txt
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_models.py -m gpu -q -k "c3_30s or c4_30s or 45s_clip or heavy" > gpurun_out/h2c/pytest_fullsize.txt 2>&1; echo "rc $?" >> gpurun_out/h2c/pytest_fullsize.txt
cp gpurun_out/fullsize_parity.json gpurun_out/h2c/fullsize_parity.json
for m in 0 1; do
  RVC_H2=$m timeout 400 python bench.py --no-cpu-baseline --no-traffic --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/h2c/bench_h2_${m}.json
done
tail -4 gpurun_out/h2c/pytest_pairs.txt; tail -4 gpurun_out/h2c/pytest_pairs_check.txt; grep -E "^FAILED|^E  " gpurun_out/h2c/pytest_pairs*.txt | head; tail -5 gpurun_out/h2c/pytest_fullsize.txt; grep -E "^FAILED|^E  " gpurun_out/h2c/pytest_fullsize.txt | head -20
python - <<'PY'
import json,glob
d=json.load(open('gpurun_out/h2c/fullsize_parity.json'))
for k,v in sorted(d.items()): print(k, {a:v[a] for a in ('max','p9999','mean','within','coarse_max_diff','voicing_equal') if a in v})
for f in sorted(glob.glob('gpurun_out/h2c/bench_h2_*.json')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['config'].get('one_clip_alone_ms'), r['kernel'], r['kernel_ms_per_clip'], r['frac'], [ (k[0],k[1]) for k in r['others']][:4])
    except Exception as e: print(f, 'ERR', e)
PY
