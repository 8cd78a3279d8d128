#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4d_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r4d_pytest_gpu.txt
run() { env $@ timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['one_clip_alone_ms'])"; }
run RVC_X3S_UC=2
run RVC_X3S_UC=1
run RVC_X3S_UC=2
run RVC_X3S_UC=1
rm -rf gpurun_out/prof_l1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l1 -o r4d -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r4d_prof.err
f=$(find gpurun_out/prof_l1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4d_kernel_stats_lanes1.csv && head -8 "$f" | cut -c1-200
find gpurun_out/prof_l1 -name "*.csv" -size +8M -delete; find gpurun_out/prof_l1 -name "*.db" -delete
