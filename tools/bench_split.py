"""Split-resident ResBlock pair (c1 -> image -> c2 + residual) per generator class: us per pair and TFLOP/s.  RVC_X3Q=0/1 selects the kernel family
(read once per process).  python tools/bench_split.py [C ...]"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
L.check(L.lib.rvc_set_conv_precision(2))
T0 = int(os.environ.get('BENCH_T', 3198))
sel = [int(a) for a in sys.argv[1:]] or [256, 128, 64]
reps = int(os.environ.get('BENCH_REPS', 5))
tot = 0.0
for Cc in sel:
    T = T0 * {256: 10, 128: 100, 64: 200, 32: 400}[Cc]
    for k, d in ((3, 1), (3, 5), (7, 1), (7, 3), (11, 1), (11, 5)):
        if os.environ.get('BENCH_K') and int(os.environ['BENCH_K']) != k:
            continue
        plans = []
        for dd in (d, 1):
            w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = (np.random.randn(Cc) * 0.1).astype(np.float32)
            pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
        x = torch.randn(Cc, T, device="cuda"); y = torch.empty_like(x)
        fn = lambda: L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
        try:
            fn()
        except L.RvcHipError as e:
            print(f"C{Cc} k{k} d{d}: {e}"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot += ms
        print(f"C{Cc} k{k} d{d} pair {ms*1e3:8.1f} us   {4.0*Cc*Cc*k*T/ms/1e9:6.1f} TFLOP/s", flush=True)
        tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
        if tm[0]:
            nb = tm[0]; print("      per tile cycles: " + "  ".join(f"{n} {tm[i]/nb:.0f}" for i, n in ((1,"prologue"),(2,"compute"),(3,"wwait"),(4,"barrier"),(5,"epilogue"),(6,"total"))) + f"   tiles {nb/(reps+1):.0f}")
        bad = L.lib.rvc_debug_x3p_check()
        if bad > 0: print(f"      !!! wait check: {bad} waits with a too large compile-time count")
        for pl in plans: L.lib.rvc_conv1d_plan_destroy(pl)
print(f"sum {tot*1e3:.1f} us")
