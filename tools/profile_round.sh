# round-1 final numbers: bench line (3 lanes), one-clip-at-a-time kernel stats (rocprofv3 --kernel-trace --stats), PMC traffic passes
export TMPDIR=/tmp
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/r1n_bench.json 2> gpurun_out/r1n_bench.err
cut -c1-260 gpurun_out/r1n_bench.json
rm -rf gpurun_out/prof_n gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_n -o r1n -- python3 bench.py --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r1n_bench_lanes1_under_rocprof.json 2> gpurun_out/r1n_prof.err
cut -c1-260 gpurun_out/r1n_bench_lanes1_under_rocprof.json
ls gpurun_out/prof_n
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$c.out 2> gpurun_out/pmc_$c.err
  ls -la gpurun_out/pmc_$c | head -5
done
python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv gpurun_out/r1n_pmc_traffic.json && head -c 600 gpurun_out/r1n_pmc_traffic.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
