"""Runs one free-standing op of the library a few times (for rocprofv3 counter passes: tools/pmc_kernels.sh):
    python tools/run_ops.py att | attrel | cbr16 | cbr32 | pair128k11 | pair128k3 | pair64k7   (pairNNNkK: one split-resident ResBlock pair = conv_x3q_kernel c1 + c2)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
what = sys.argv[1]
g = torch.Generator().manual_seed(1)
if what in ("att", "attrel"):
    heads, T, D = (12, 1599, 64) if what == "att" else (2, 3198, 96)
    q = (torch.randn(heads * D, T, generator=g) * 0.3).cuda(); k = torch.randn(heads * D, T, generator=g).cuda(); v = torch.randn(heads * D, T, generator=g).cuda()
    bv = torch.randn(heads * D, generator=g).cuda(); out = torch.empty(heads * D, T, device="cuda")
    ek = torch.randn(21, D, generator=g) * 0.5; ev = torch.randn(21, D, generator=g) * 0.5
    for _ in range(8):
        if what == "att":
            L.check(L.lib.rvc_op_attention_split(None, q.data_ptr(), k.data_ptr(), v.data_ptr(), bv.data_ptr(), out.data_ptr(), None, heads, T))
        else:
            L.check(L.lib.rvc_op_attention_split_rel(None, q.data_ptr(), k.data_ptr(), v.data_ptr(), bv.data_ptr(), ek.data_ptr(), ev.data_ptr(), out.data_ptr(), None, heads, T, 0))
elif what.startswith("pair"):
    import ctypes as C
    Cc, k = [int(v) for v in what[4:].split("k")]
    T = 3198 * {256: 10, 128: 100, 64: 200}[Cc]
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    for dd in (1, 1):
        w = (np.random.randn(Cc, Cc, k) / np.sqrt(Cc * k)).astype(np.float32); b = (np.random.randn(Cc) * 0.1).astype(np.float32)
        pl = C.c_void_p(); L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w), L.ptr(b), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl))); plans.append(pl)
    x = torch.randn(Cc, T, device="cuda"); y = torch.empty_like(x)
    for _ in range(4):
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(x), T, L.ptr(y), 1.0, 0))
else:
    Cc, H, W = (16, 3232, 128) if what == "cbr16" else (32, 1616, 64)
    x = torch.randn(Cc, H, W, generator=g).cuda(); y = torch.empty_like(x)
    w1 = (torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc)).contiguous(); b1 = torch.randn(Cc, generator=g)
    w2 = (torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc)).contiguous(); b2 = torch.randn(Cc, generator=g)
    for _ in range(8):
        L.check(L.lib.rvc_op_cbr2_small(None, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), Cc, H, W))
torch.cuda.synchronize()
