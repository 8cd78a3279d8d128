"""Fused ResBlock pair (PAIR_C = 32 / 64) against the two unfused launches (fp32 / split-resident intermediate)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
import os
Cc = int(os.environ.get("PAIR_C", 32))
T = 1279200 * 32 // Cc
for k, d in ((3, 1), (3, 5), (7, 3), (11, 5)):
    plans = []
    for dd in (d, 1):
        w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = np.zeros(Cc, np.float32)
        pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
    x = torch.randn(Cc, T, device="cuda"); t1 = torch.empty_like(x); y = torch.empty_like(x)
    def unfused():
        L.check(L.lib.rvc_conv1d_plan_run(plans[0], None, L.ptr(x), T, None, L.ptr(t1), 1, 0.1, 0, 0.0))
        L.check(L.lib.rvc_conv1d_plan_run(plans[1], None, L.ptr(t1), T, L.ptr(x), L.ptr(y), 1, 0.1, 0, 0.0))
    def fused():
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
    def split():
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
    variants = [("unfused", unfused), ("fused", fused)] + ([("split", split)] if Cc >= 64 else [])
    for name, fn in variants:
        try:
            fn(); torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 - a variant that does not exist for this shape (the fused pair at C = 128, ...)
            print(f"C{Cc} k{k} d{d} {name:8s} n/a ({str(e)[-60:]})")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"C{Cc} k{k} d{d} {name:8s} {ms*1e3:8.1f} us   {4.0*Cc*Cc*k*T/ms/1e9:6.1f} TFLOP/s  {3*4.0*Cc*T/ms/1e6:7.1f} GB/s (x, res, y)")
