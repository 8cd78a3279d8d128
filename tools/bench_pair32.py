"""32-channel ResBlock pairs at the generator's stage length in both pair arithmetics: conv_x3pf_kernel (bf16x3) vs conv_rbh_kernel (fp16x2, LDS-resident weights)."""
import sys, ctypes as C, os
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
Cc, T = 32, int(os.environ.get("PAIR_T", 1279200))
tot = {0: 0.0, 1: 0.0}
for k in (3, 7, 11):
    for d in (1, 3, 5):
        plans = []
        for dd in (d, 1):
            w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = np.zeros(Cc, np.float32)
            pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
        x = torch.randn(Cc, T, device="cuda"); y = torch.empty_like(x)
        line = f"C{Cc} k{k} d{d}"
        for arith in (0, 1):
            L.check(L.lib.rvc_set_pair_arithmetic(arith))
            fn = lambda: L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            tot[arith] += us
            line += f" | arith {arith} ({'fp16x2' if L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == 1 else 'bf16x3'}): {us:7.1f} us {4.0*Cc*Cc*k*T/us/1e6:6.1f} TFLOP/s {2*4.0*Cc*T/us/1e3:7.1f} GB/s"
        print(line, flush=True)
        for pl in plans: L.lib.rvc_conv1d_plan_destroy(pl)
print(f"sum of the nine pairs: bf16x3 {tot[0]:.0f} us, fp16x2 {tot[1]:.0f} us")
