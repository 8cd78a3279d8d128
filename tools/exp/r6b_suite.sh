mkdir -p gpurun_out/r6b
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6b/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r6b/pytest_gpu.txt; grep -E "^FAILED|^ERROR" gpurun_out/r6b/pytest_gpu.txt | head -20
timeout 600 python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err; cut -c1-1500 gpurun_out/r6b/bench.json
for l in 2 4 5; do timeout 300 python bench.py --lanes $l --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $l', d['value'], d['config']['one_clip_alone_ms'])"; done
