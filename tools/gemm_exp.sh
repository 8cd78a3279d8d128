export TMPDIR=/tmp
for rs in 3 4 6; do echo "== RS $rs T x1"; RVC_X3S_RS=$rs BENCH_QUICK=1 python tools/bench_gemm.py ffn1 | grep -v best; done
for rs in 3 4 6; do echo "== RS $rs T x4"; RVC_X3S_RS=$rs BENCH_QUICK=1 BENCH_TMUL=4 python tools/bench_gemm.py ffn1 | grep -v best; done
