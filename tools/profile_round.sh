# round-2 evidence: bench line (3 lanes), kernel stats for lanes 1 AND lanes 3 (rocprofv3 --kernel-trace --stats), per-launch table of the
# event-timed pass, FETCH_SIZE / WRITE_SIZE PMC passes joined per launch class.   bash tools/profile_round.sh r2e
tag=${1:-r2}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RVC_PROF_CSV=gpurun_out/${tag}_launches.csv timeout 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cut -c1-260 gpurun_out/${tag}_bench.json
for lanes in 1 3; do
  rm -rf gpurun_out/prof_l$lanes
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l$lanes -o ${tag} -- python3 bench.py --lanes $lanes --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/${tag}_bench_lanes${lanes}_under_rocprof.json 2> gpurun_out/${tag}_prof_l$lanes.err
  f=$(find gpurun_out/prof_l$lanes -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats_lanes${lanes}.csv && head -12 "$f" | cut -c1-160
  find gpurun_out/prof_l$lanes -name "*.csv" -size +8M -delete; find gpurun_out/prof_l$lanes -name "*.db" -delete
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  RVC_PROF_CSV=gpurun_out/${tag}_launches_pmc_$c.csv timeout 1200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py --lanes 1 --steps 1 --warmup 0 --clips 1 --no-cpu-baseline --no-traffic > gpurun_out/pmc_$c.out 2> gpurun_out/pmc_$c.err
  ls -la gpurun_out/pmc_$c/*/ 2>/dev/null | head -5
done
fc=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); wc=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$fc" "$wc" gpurun_out/${tag}_pmc_traffic.json | head -8
# (us from the un-profiled event-timed pass of the bench run above; bytes from the counter passes, joined by launch order per kernel)
python3 tools/launch_classes.py gpurun_out/${tag}_launches.csv "$fc" "$wc" > gpurun_out/${tag}_launch_classes.md; head -14 gpurun_out/${tag}_launch_classes.md | cut -c1-250
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete
