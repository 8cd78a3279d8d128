# final round-1 numbers: bench line, bench under rocprofv3 kernel-trace, kernel stats
export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=12 timeout 200 python bench.py --lanes 4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline | tail -1 | cut -c1-190
timeout 400 python bench.py --lanes 3 --steps 10 --warmup 3 > gpurun_out/r1m_bench.json 2> gpurun_out/r1m_bench.err
cut -c1-300 gpurun_out/r1m_bench.json
rm -rf gpurun_out/prof_m
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m -o r1m -- python3 bench.py --lanes 3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r1m_bench_under_rocprof.json 2> gpurun_out/r1m_prof.err
cut -c1-300 gpurun_out/r1m_bench_under_rocprof.json
find gpurun_out/prof_m -name "*stats*" | head
f=$(find gpurun_out/prof_m -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r1m_kernel_stats.csv; head -12 gpurun_out/r1m_kernel_stats.csv
find gpurun_out/prof_m -name "*kernel_trace.csv" -size +40M -delete
