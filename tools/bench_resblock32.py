"""The three ResBlocks of the generator's 32-channel stage (k = 3 / 7 / 11, dilations 1 / 3 / 5) at the stage's length, fp16x2 arithmetic:
the chain of three fused-pair launches (conv_rbh_kernel) against one launch per ResBlock (conv_rb3_kernel); the third pair / the fused launch
scales by 1/3 and the second and third ResBlock accumulate, as in the generator.  RB_T=<length> overrides the length; RB_C=64: the 64-channel stage's
3-tap ResBlock (its chain is three conv_x3pf_kernel launches in bf16x3: a different arithmetic, not bit-comparable)."""
import sys, ctypes as C, os
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
L.check(L.lib.rvc_set_pair_arithmetic(1))
Cc = int(os.environ.get("RB_C", 32)); T = int(os.environ.get("RB_T", 1279200 if Cc == 32 else 639600))
REP = 10
rng = np.random.default_rng(0)
tot = [0.0, 0.0]
x = torch.randn(Cc, T, device="cuda"); ya = torch.empty_like(x); yb = torch.empty_like(x); y = torch.zeros_like(x); y2 = torch.zeros_like(x)
for j, k in enumerate((3, 7, 11) if Cc == 32 else (3, 7)):
    plans = []
    for i in range(6):
        dd = (1, 3, 5)[i // 2] if i % 2 == 0 else 1
        w = (rng.standard_normal((Cc, Cc, k)) / np.sqrt(Cc * k)).astype(np.float32); b = (rng.standard_normal(Cc) * 0.1).astype(np.float32)
        pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
    arr = (C.c_void_p * 6)(*[pl.value for pl in plans])
    ran = C.c_int(-1)
    acc = int(j > 0)

    pair_run = L.lib.rvc_conv1d_plan_pair_split_run if (Cc == 64 and k > 3) else L.lib.rvc_conv1d_plan_pair_run      # (64 channels x 7 taps: two launches per pair on conv_x3q_kernel, the intermediate image through HBM)

    def chain(out):
        L.check(pair_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(ya), 1.0, 0))
        L.check(pair_run(plans[2], plans[3], None, L.ptr(ya), T, L.ptr(yb), 1.0, 0))
        L.check(pair_run(plans[4], plans[5], None, L.ptr(yb), T, L.ptr(out), 1.0 / 3, acc))

    def fused(out):
        L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(x), T, L.ptr(out), 1.0 / 3, acc, C.byref(ran), None, None, None))

    y.zero_(); y2.zero_()
    chain(y); fused(y2); torch.cuda.synchronize()
    same = bool(torch.equal(y, y2))
    line = f"C{Cc} k{k:2d}: fused ran {ran.value}, bit-identical to the chain: {same}"
    for idx, fn in enumerate((chain, fused)):
        fn(y); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP): fn(y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / REP * 1e3
        tot[idx] += us
        line += f" | {'chain of 3 pairs' if idx == 0 else 'one launch'}: {us:7.1f} us {3 * 4.0 * Cc * Cc * k * T / us / 1e6:6.1f} TFLOP/s {(2 + acc) * 4.0 * Cc * T / us / 1e3:7.1f} GB/s (x once + y{' + previous y' if acc else ''})"
    print(line, flush=True)
    for pl in plans: L.lib.rvc_conv1d_plan_destroy(pl)
print(f"the stage's three ResBlocks: chain {tot[0]:.0f} us, one launch each {tot[1]:.0f} us")
