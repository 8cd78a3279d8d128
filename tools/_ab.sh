for r in 0 1; do RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RVC_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -2 | cut -c1-400 & done; wait
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
