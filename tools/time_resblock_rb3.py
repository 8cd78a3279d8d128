"""Per-phase cycle counters of conv_rb3_kernel (a whole 32-channel ResBlock per launch, fp16x2); needs a -DRVC_CONV_TIMING -DRVC_EXPERIMENTS build
(bash tools/build_variant.sh timing -DRVC_CONV_TIMING; RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_timing.so)."""
import sys, ctypes as C, os
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2)); L.check(L.lib.rvc_set_pair_arithmetic(1))
Cc, T = 32, int(os.environ.get("RB_T", 1279200))
for k in (3, 7, 11):
    plans = []
    for i in range(6):
        dd = (1, 3, 5)[i // 2] if i % 2 == 0 else 1
        w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = np.zeros(Cc, np.float32)
        pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
    arr = (C.c_void_p * 6)(*[pl.value for pl in plans]); ran = C.c_int(-1)
    x = torch.randn(Cc, T, device="cuda"); y = torch.zeros_like(x)
    for acc in (0, 1):
        fn = lambda: L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(x), T, L.ptr(y), 1.0 / 3, acc, C.byref(ran), None, None, None))
        fn(); torch.cuda.synchronize()
        tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        L.lib.rvc_debug_conv_timing(tm, 1)
        nt = max(tm[0], 1)
        print(f"C32 k{k:2d} acc {acc} (ran {ran.value}): ResBlock {us:7.1f} us | per tile cycles (wave 0): x image + requests {tm[1]/nt:6.0f} first convs {tm[2]/nt:6.0f} images {tm[3]/nt:6.0f} "
              f"second convs {tm[4]/nt:6.0f} epilogue {tm[5]/nt:6.0f} (barrier waits inside those {tm[7]/nt:6.0f}) | total per workgroup {tm[6]/(5*256):8.0f} tiles/launch {nt/5:.0f}", flush=True)
    for pl in plans: L.lib.rvc_conv1d_plan_destroy(pl)
