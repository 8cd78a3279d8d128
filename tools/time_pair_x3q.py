"""Per-phase cycle counters of the persistent ResBlock-pair kernel (conv_x3q_kernel) in both pair arithmetics; needs a -DRVC_CONV_TIMING -DRVC_EXPERIMENTS build
(RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_timing.so).  Slots: [0] tiles, [1] prologue (once per workgroup), [2] compute between barriers, [3] weight wait,
[4] barrier, [5] epilogue, [6] total per workgroup."""
import sys, ctypes as C, os
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
for Cc, T in ((128, 319800), (64, 639600), (256, 31980)):
    for k, d in ((3, 1), (7, 3), (11, 5)):
        if Cc == 64 and k == 3: continue
        plans = []
        for dd in (d, 1):
            w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = np.zeros(Cc, np.float32)
            pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
        x = torch.randn(Cc, T, device="cuda"); y = torch.empty_like(x)
        for arith in (0, 1):
            L.check(L.lib.rvc_set_pair_arithmetic(arith))
            fn = lambda: L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
            fn(); torch.cuda.synchronize()
            tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            L.lib.rvc_debug_conv_timing(tm, 1)
            nt = max(tm[0], 1)
            print(f"C{Cc} k{k} d{d} arith {arith}: pair {us:7.1f} us {4.0*Cc*Cc*k*T/us/1e6:6.1f} TFLOP/s | per tile cycles: compute {tm[2]/nt:7.0f} wwait {tm[3]/nt:6.0f} barrier {tm[4]/nt:6.0f} epilogue {tm[5]/nt:6.0f} | total/tile {tm[6]/nt:7.0f} tiles {nt/10:.0f} per launch", flush=True)
        for pl in plans: L.lib.rvc_conv1d_plan_destroy(pl)
