"""The MDX23C recipe's product shapes (model_2_stem_full_band_8k: 128 + 128 i channels, planes 256 x 1024 >> i) on the two kernel families:
the staged fp32-in conv2d (what model_mdx23.hip runs through conv2d_run) and conv_x3s_kernel on padded split-resident images (what RMVPE's
deep levels run on).  us per launch, TFLOP/s.     python tools/bench_mdx_shapes.py [x3s|staged|lin]"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
sel = sys.argv[1:] or ["x3s", "staged", "lin"]
SC = [(128 + 128 * i, 256 >> i, 1024 >> i) for i in range(6)]
if "x3s" in sel:
    print("== 3x3 on conv_x3s_kernel (padded split image in, fp32 out + residual)")
    for Cc, H, W in SC:
        for Ci in ((Cc, 2 * Cc) if Cc <= 640 else (Cc,)):
            T = H * (W + 2)
            for am, an in ((0, 0), (2, 2)):
                us = C.c_float()
                L.check(L.lib.rvc_debug_gemm_split_bench(None, Ci, Cc, T, 0, am, an, 0, 6, C.byref(us), W, 2))
                fl = 2.0 * Ci * Cc * 9 * H * W
                print(f"  {Ci:4d}->{Cc:4d} {H:3d}x{W:4d} tile {'auto' if not am else '128x128'}  {us.value:8.1f} us {fl / us.value / 1e6:7.1f} TFLOP/s")
if "staged" in sel:
    print("== 3x3 on the staged kernel (conv2d_run, fp32 in / out + residual)")
    for Cc, H, W in SC:
        for Ci in ((Cc, 2 * Cc) if Cc <= 640 else (Cc,)):
            w = (np.random.randn(Cc, Ci, 3, 3) / np.sqrt(Ci * 9)).astype(np.float32); b = np.zeros(Cc, np.float32)
            x = torch.randn(Ci, H, W, device="cuda"); r = torch.randn(Cc, H, W, device="cuda"); y = torch.empty(Cc, H, W, device="cuda")
            ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
            L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), L.ptr(y), Ci, Cc, H, W, 1))
            L.check(L.lib.rvc_prof_enable(1))
            for _ in range(4):
                L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), L.ptr(y), Ci, Cc, H, W, 1))
            L.check(L.lib.rvc_prof_collect(ms, fl, ln)); L.check(L.lib.rvc_prof_enable(0))
            t = sum(ms) / 4; f = sum(fl) / 4
            cfg = [L.lib.rvc_prof_cfg_name(i).decode() for i in range(24) if ln[i]]
            print(f"  {Ci:4d}->{Cc:4d} {H:3d}x{W:4d}  {t * 1e3:8.1f} us {f / t / 1e9:7.1f} TFLOP/s  {cfg}")
            del x, r, y
if "lin" in sel:
    print("== TDF linears / shortcuts as k = 1 products on conv_x3s_kernel (image in, fp32 out)")
    for Cc, H, W in SC:
        R = Cc * H
        for name, Ci, Co, T in (("lin1", W, W // 4, R), ("lin2", W // 4, W, R), ("shortcut", Cc, Cc, H * W)):
            if Ci % 16 or Co % 16:
                continue
            us = C.c_float()
            L.check(L.lib.rvc_debug_gemm_split_bench(None, Ci, Co, T, 0, 0, 0, 0, 6, C.byref(us), 0, 2))
            fl = 2.0 * Ci * Co * T
            print(f"  scale C={Cc:4d} {name:9s} {Ci:4d}->{Co:4d} N={T:7d}  {us.value:8.1f} us {fl / us.value / 1e6:7.1f} TFLOP/s")
