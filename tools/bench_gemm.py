"""Micro-benchmark of the split-resident bf16x3 GEMM (csrc/conv_x3s.hip) on the projection shapes of HuBERT / the text encoder / the flow:
tile and K-split sweep per shape, microseconds and algorithmic TFLOP/s per launch (HIP events around back-to-back launches).
    python tools/bench_gemm.py [filter ...]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
# name, Ci, Co, T, w2d (> 0: 3 x 3 over a padded image of that width, T = H (w2d + 2))
SHAPES = [("hubert ffn1 768->3072", 768, 3072, 1599, 0), ("hubert ffn2 3072->768", 3072, 768, 1599, 0), ("hubert qkv 768->2304", 768, 2304, 1599, 0),
          ("hubert out 768->768", 768, 768, 1599, 0), ("hubert proj 512->768", 512, 768, 1599, 0), ("encp qkv 192->576", 192, 576, 3198, 0),
          ("flow 192->192", 192, 192, 3198, 0), ("rmvpe fc 512->360", 512, 360, 3232, 0),
          ("rmvpe L5 512 3x3", 512, 512, 101 * 6, 4), ("rmvpe L4 256 3x3", 256, 256, 202 * 10, 8), ("rmvpe L3 128 3x3", 128, 128, 404 * 18, 16),
          ("rmvpe L2 64 3x3", 64, 64, 808 * 34, 32)]
NLAYERS = int(__import__("os").environ.get("BENCH_NLAYERS", "1"))      # > 1: distinct weight sets cycled through (cold weights, as in a model)
import os
sel = sys.argv[1:]
TMUL = int(os.environ.get("BENCH_TMUL", "1"))          # columns x TMUL: the asymptotic rate of a tile without the small-grid effects
QUICK = os.environ.get("BENCH_QUICK") == "1"
MODES = [int(m) for m in os.environ.get("BENCH_MODES", "1,2").split(",")]      # 1 = LDS ring, 2 = register-direct reduction loop
for name, Ci, Co, T, w2d in SHAPES:
    if sel and not any(x in name for x in sel):
        continue
    T = T * TMUL
    fl = 2.0 * Ci * Co * T * (9 if w2d else 1)
    for split_out in ((0,) if QUICK else (0, 1)):
        best = None
        for mode in MODES:
            L.check(L.lib.rvc_debug_set_x3s_mode(mode))
            for am, an in ((0, 0), (2, 2), (2, 1), (1, 2), (1, 1)):
                for ks in ((0,) if QUICK else (0, 1, 2, 3, 4, 6, 8, 12)):
                    if ks > 1 and (Ci // 16 * (9 if w2d else 1)) % ks:
                        continue
                    us = C.c_float()
                    L.check(L.lib.rvc_debug_gemm_split_bench(None, Ci, Co, T, ks, am, an, split_out, 40, C.byref(us), w2d, NLAYERS))
                    tag = f"{'ring  ' if mode == 1 else 'direct'} tile {'auto' if am == 0 else f'{64 * am}x{64 * an}'} split {'auto' if ks == 0 else ks}"
                    print(f"{name:24s} {'split-out+gelu' if split_out else 'fp32-out+res  '} {tag:36s} {us.value:8.1f} us {fl / us.value / 1e6:7.1f} TFLOP/s")
                    if am and ks and (best is None or us.value < best[0]):
                        best = (us.value, tag)
        if best:
            print(f"{name:24s} best: {best[1]}  {best[0]:.1f} us  {fl / best[0] / 1e6:.1f} TFLOP/s\n")
