"""Micro-benchmark of the MFMA conv kernel on the generator / HuBERT shapes (HIP events around plan launches)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
import os
T = int(os.environ.get('BENCH_T', 3198))
NORES = os.environ.get('BENCH_NORES') == '1'
L.check(L.lib.rvc_set_conv_precision(int(os.environ.get('BENCH_PRECISION', 2))))   # 0: fp32 MFMA, 2: bf16x3 where eligible
CASES = [  # name, Ci, Co, Tin, k, stride, dil
    ("gen s1 C256 k3", 256, 256, 10 * T, 3, 1, 1), ("gen s1 C256 k11 d5", 256, 256, 10 * T, 11, 1, 5),
    ("gen s2 C128 k3", 128, 128, 100 * T, 3, 1, 1), ("gen s2 C128 k7 d3", 128, 128, 100 * T, 7, 1, 3), ("gen s2 C128 k11", 128, 128, 100 * T, 11, 1, 1), ("gen s2 C128 k11 d5", 128, 128, 100 * T, 11, 1, 5), ("gen s3 C64 k11 d5", 64, 64, 200 * T, 11, 1, 5), ("gen s3 C64 k3", 64, 64, 200 * T, 3, 1, 1),
    ("gen s3 C64 k7", 64, 64, 200 * T, 7, 1, 1), ("gen s4 C32 k3", 32, 32, 400 * T, 3, 1, 1), ("gen s4 C32 k7", 32, 32, 400 * T, 7, 1, 1), ("gen s4 C32 k11 d5", 32, 32, 400 * T, 11, 1, 5),
    ("hubert ffn1 768->3072", 768, 3072, 1599, 1, 1, 1), ("hubert ffn2 3072->768", 3072, 768, 1599, 1, 1, 1), ("hubert qk 768->1536", 768, 1536, 1599, 1, 1, 1), ("hubert out 768->768", 768, 768, 1599, 1, 1, 1), ("hubert qkv 768->2304", 768, 2304, 1599, 1, 1, 1), ("flow 192->192", 192, 192, 3198, 1, 1, 1),
    ("hubert conv1 s2", 512, 512, 102399, 3, 2, 1), ("hubert conv2 s2", 512, 512, 51199, 3, 2, 1),
]
sel = sys.argv[1:] 
reps = 5
for name, Ci, Co, Tin, k, s, d in CASES:
    if sel and not any(x in name for x in sel): continue
    w = (np.random.randn(Co, Ci, k) / np.sqrt(Ci * k)).astype(np.float32); b = np.zeros(Co, np.float32)
    pad = (k * d - d) // 2 if s == 1 else 0
    plan = C.c_void_p()
    L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Ci, Co, k, s, pad, d, 1, C.byref(plan)))
    x = torch.randn(Ci, Tin, device="cuda")
    Tout = (Tin + 2 * pad - d * (k - 1) - 1) // s + 1
    y = torch.empty(Co, Tout, device="cuda"); r = torch.randn(Co, Tout, device="cuda")
    run = lambda: L.check(L.lib.rvc_conv1d_plan_run(plan, None, L.ptr(x), Tin, None if NORES else L.ptr(r), L.ptr(y), 1, 0.1, 0, 0.0))
    run(); torch.cuda.synchronize()
    if os.environ.get('BENCH_CHECK') == '1' and s == 1:
        # value check of the first / last 3000 output columns against a float64 CPU convolution of the matching input slices
        import torch.nn.functional as F
        W = min(3000, Tout); halo = pad + 8
        xc = x.double().cpu(); wc = torch.from_numpy(w).double(); rc = r.double().cpu(); yc = y.double().cpu()
        for lo, hi in ((0, W), (Tout - W, Tout)):
            a0, a1 = max(lo - halo, 0), min(hi + halo, Tin)
            ref = F.conv1d(F.leaky_relu(xc[None, :, a0:a1], 0.1), wc, padding=pad, dilation=d)[0]   # plan_run: pre-activation leaky ReLU 0.1
            ref = ref[:, lo - a0: lo - a0 + (hi - lo)] + (0 if NORES else rc[:, lo:hi])
            err = (yc[:, lo:hi] - ref).abs().max().item() / ref.abs().max().item()
            print(f"      check cols {lo}:{hi}  rel err {err:.2e}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * Co * Tout * Ci * k
    print(f"{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({fl/1e9:.1f} GFLOP)")
    tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
    if tm[0]:
        nb = tm[0]; print("      per block cycles: " + "  ".join(f"{n} {tm[i]/nb:.0f}" for i, n in ((1,"prologue+barrier"),(2,"xstore"),(7,"dmawait"),(3,"issue"),(4,"mfma"),(5,"epilogue"),(6,"total"))) + f"  blocks/launch {nb/(reps+1):.0f}")
    if tm[0] and os.environ.get('BENCH_X3P_TIMING') == '1':
        nb = tm[0]; print("      x3p per tile cycles: " + "  ".join(f"{n} {tm[i]/nb:.0f}" for i, n in ((1,"prologue"),(2,"compute"),(3,"wwait"),(4,"barrier"),(5,"epilogue"),(6,"total"))))
    bad = L.lib.rvc_debug_x3p_check()
    if bad > 0: print(f"      !!! x3p wait check: {bad} waits with a too large compile-time count")
    elif bad == 0: print("      x3p wait check ok")
    L.lib.rvc_conv1d_plan_destroy(plan)
