#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in 1 3; do
echo "== RVC_X3PF64=$v rep $rep: $(RVC_X3PF64=$v timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["one_clip_alone_ms"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_per_clip"))')"
done
done
