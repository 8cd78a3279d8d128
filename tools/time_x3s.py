"""Per-phase cycles of conv_x3s_kernel (a -DRVC_CONV_TIMING build: RVC_HIP_LIB=.../librvc_hip_timing.so) on the projection / 3 x 3 shapes of the path,
cold weights (12 distinct layers cycled through).   python tools/time_x3s.py [filter ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
SHAPES = [("hubert ffn1 768->3072", 768, 3072, 1599, 0), ("hubert ffn2 3072->768", 3072, 768, 1599, 0), ("hubert qkv 768->2304", 768, 2304, 1599, 0),
          ("hubert out 768->768", 768, 768, 1599, 0), ("flow 192->192", 192, 192, 3198, 0),
          ("rmvpe L5 512 3x3", 512, 512, 101 * 6, 4), ("rmvpe L4 256 3x3", 256, 256, 202 * 10, 8), ("rmvpe L3 128 3x3", 128, 128, 404 * 18, 16),
          ("mdx L0 128 3x3", 128, 128, 256 * 1026, 1024), ("mdx L1 256 3x3", 256, 256, 128 * 514, 512), ("mdx L2 384 3x3", 384, 384, 64 * 258, 256)]
sel = sys.argv[1:]
NL = int(os.environ.get("BENCH_NLAYERS", "12"))
for name, Ci, Co, T, w2d in SHAPES:
    if sel and not any(x in name for x in sel):
        continue
    fl = 2.0 * Ci * Co * T * (9 if w2d else 1)
    us = C.c_float()
    tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
    L.check(L.lib.rvc_debug_gemm_split_bench(None, Ci, Co, T, 0, 0, 0, 0, 24, C.byref(us), w2d, NL))
    L.lib.rvc_debug_conv_timing(tm, 1)
    print(f"{name:24s} {us.value:8.1f} us {fl / us.value / 1e6:7.1f} TFLOP/s", end="")
    if tm[0]:
        nb = tm[0]
        print("   per workgroup cycles: " + "  ".join(f"{n} {tm[i] / nb:.0f}" for i, n in ((1, "prologue"), (2, "reads+mfma"), (3, "dma wait"), (4, "barrier"), (5, "splitK+epi"), (6, "total"))) + f"   wgs/launch {nb / 25:.0f}")
    else:
        print()
