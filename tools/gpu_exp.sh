#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/prof_busy
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_busy -o b -- python3 bench.py --lanes 3 --steps 16 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
f=$(find gpurun_out/prof_busy -name "*kernel_trace.csv" | head -1)
echo "lanes 3: $(python tools/busy_fraction.py $f 20)"
rm -rf gpurun_out/prof_busy
