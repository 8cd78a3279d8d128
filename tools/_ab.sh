timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do RVC_ENCP_FUSED=$v python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSED=$v', d['value'], d['ms_per_step'])"; done
