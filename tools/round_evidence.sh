#!/bin/bash
# full round evidence on one box: GPU test suite, tools/profile_round.sh TAG, the RMVPE-only variant's bench line + kernel stats
tag=${1:-r5}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.txt
bash tools/profile_round.sh $tag 2>&1 | tail -30
timeout 600 python3 bench.py --variant rmvpe_60s > gpurun_out/${tag}_bench_rmvpe_60s.json 2> gpurun_out/${tag}_bench_rmvpe_60s.err; cut -c1-300 gpurun_out/${tag}_bench_rmvpe_60s.json
rm -rf gpurun_out/prof_rmvpe60
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rmvpe60 -o ${tag} -- python3 bench.py --variant rmvpe_60s --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1
f=$(find gpurun_out/prof_rmvpe60 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_rmvpe60.csv && head -8 "$f" | cut -c1-160
find gpurun_out/prof_rmvpe60 -name "*.csv" -size +8M -delete; find gpurun_out/prof_rmvpe60 -name "*.db" -delete
# SQ / L2 counters of the persistent ResBlock kernel (one split-resident pair per class; four --pmc passes each)
for c in pair128k11 pair128k3 pair64k7; do bash tools/pmc_kernels.sh $c conv_x3q_kernel ${tag}_x3q_$c > /dev/null 2>&1; [ -f gpurun_out/pmck_${tag}_x3q_$c.txt ] && cp gpurun_out/pmck_${tag}_x3q_$c.txt gpurun_out/${tag}_sq_counters_x3q_$c.txt; done
head -12 gpurun_out/${tag}_sq_counters_x3q_pair128k11.txt 2>/dev/null
# the UVR chain: kernel stats of one clip at a time (MDX23C is 98 % of it), and what the matrix pipe delivers on bare MFMA loops (context for the roofline fractions)
rm -rf gpurun_out/prof_uvr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_uvr -o ${tag} -- python3 bench.py --variant uvr_48k_v2 --lanes 1 --clips 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic > /dev/null 2>&1
f=$(find gpurun_out/prof_uvr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_uvr.csv && head -6 "$f" | cut -c1-160
rm -rf gpurun_out/prof_uvr
[ -x tools/micro/mfmabench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfmabench tools/micro/mfmabench.hip > /dev/null 2>&1
timeout 300 tools/micro/mfmabench > gpurun_out/${tag}_mfmabench.txt 2>&1
for v in 48k_v2 uvr_48k_v2; do timeout 600 python3 bench.py --variant $v --no-cpu-baseline > gpurun_out/${tag}_bench_$v.json 2>/dev/null; cut -c1-200 gpurun_out/${tag}_bench_$v.json; done
timeout 600 python3 tools/mdx_memory.py > gpurun_out/${tag}_mdx_memory.txt 2>&1; tail -1 gpurun_out/${tag}_mdx_memory.txt
