"""Micro-benchmark of the 3x3 Conv2d launches of the RMVPE U-Net levels (rvc_op_conv2d3x3 includes host packing: timed with the profile hooks)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
import os
L.check(L.lib.rvc_set_conv_precision(int(os.environ.get('BENCH_PRECISION', 2))))
CASES = [("L0 16->16 3232x128", 16, 16, 3232, 128), ("L1 32->32 1616x64", 32, 32, 1616, 64), ("L2 64->64 808x32", 64, 64, 808, 32),
         ("L3 128->128 404x16", 128, 128, 404, 16), ("L4 256->256 202x8", 256, 256, 202, 8), ("L5 512->512 101x4", 512, 512, 101, 4),
         ("dec 32->16 3232x128", 32, 16, 3232, 128)]
for name, Ci, Co, H, W in CASES:
    w = (np.random.randn(Co, Ci, 3, 3) / np.sqrt(Ci * 9)).astype(np.float32); b = np.zeros(Co, np.float32)
    x = torch.randn(Ci, H, W, device="cuda"); r = torch.randn(Co, H, W, device="cuda"); y = torch.empty(Co, H, W, device="cuda")
    ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
    L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), L.ptr(y), Ci, Co, H, W, 1))
    L.check(L.lib.rvc_prof_enable(1))
    for _ in range(5):
        L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), L.ptr(y), Ci, Co, H, W, 1))
    L.check(L.lib.rvc_prof_collect(ms, fl, ln)); L.check(L.lib.rvc_prof_enable(0))
    t = sum(ms) / 5; f = sum(fl) / 5
    cfg = [L.lib.rvc_prof_cfg_name(i).decode() for i in range(24) if ln[i]]
    print(f"{name:24s} {t*1e3:8.1f} us  {f/t/1e9:6.1f} TFLOP/s  {cfg}")
