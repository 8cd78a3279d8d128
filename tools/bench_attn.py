"""Micro-benchmark of the fused attention kernel (HuBERT shape: 12 heads x 64, T = 1599)."""
import sys
sys.path.insert(0, '.')
import torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
for heads, T in ((12, 1599), (12, 3199)):
    q = torch.randn(heads * 64, T, device="cuda") * 0.3; k = torch.randn(heads * 64, T, device="cuda"); v = torch.randn(T, heads * 64, device="cuda")
    bv = torch.zeros(heads * 64, device="cuda"); out = torch.empty(heads * 64, T, device="cuda")
    run = lambda: L.check(L.lib.rvc_op_attention(None, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(bv), L.ptr(out), heads, T))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * heads * 64 * T * T
    print(f"attention heads {heads} T {T}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s")
    import ctypes as C
    tm = (C.c_uint64 * 8)(); L.lib.rvc_debug_conv_timing(tm, 1)
    if tm[0]:
        nb = tm[0]; print("   per block (100 MHz ticks): " + "  ".join(f"{n} {tm[i]/nb:.0f}" for i, n in ((1, "kv-store"), (2, "S"), (3, "softmax"), (4, "PV"), (6, "total"))))

# synthesizer text encoder: 2 heads x 96 with the relative-position band (30 s clip: T = 3001)
for heads, T in ((2, 3001), (2, 1000)):
    q = torch.randn(heads * 96, T, device="cuda") * 0.3; k = torch.randn(heads * 96, T, device="cuda"); v = torch.randn(T, heads * 96, device="cuda")
    bv = torch.zeros(heads * 96, device="cuda"); out = torch.empty(heads * 96, T, device="cuda")
    rel = torch.randn(heads, 21, T, device="cuda"); pb = torch.empty(heads, 21, T, device="cuda")
    run = lambda: L.check(L.lib.rvc_op_attention_rel(None, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(bv), L.ptr(rel), L.ptr(pb), L.ptr(out), heads, T))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * heads * 96 * T * T
    print(f"rel attention heads {heads} T {T}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s")
