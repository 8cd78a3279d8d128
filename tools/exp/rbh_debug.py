import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from comfy_rvc_amd import _lib as L
Cc, k, d, T = 32, int(os.environ.get("K", 11)), int(os.environ.get("D", 5)), 600003
g = torch.Generator().manual_seed(1)
x = torch.randn(Cc, T, generator=g)
w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
L.check(L.lib.rvc_set_conv_precision(2))
plans = []
for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
    pl = C.c_void_p()
    L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
    plans.append(pl)
xg = x.cuda()
outs = {}
for arith in (0, 1, 1):
    L.check(L.lib.rvc_set_pair_arithmetic(arith))
    y = torch.zeros(Cc, T, device="cuda")
    L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
    torch.cuda.synchronize()
    outs.setdefault(arith, []).append(y.cpu())
ref = outs[0][0]
NO = 256 - (k - 1)
for i, y in enumerate(outs[1]):
    bad = ((y - ref).abs() > 1e-2).nonzero()
    print(f"run {i}: bad {bad.shape[0]}", end="")
    if bad.shape[0]:
        ch, t = bad[:, 0], bad[:, 1]
        print(f" rows {sorted(set(ch.tolist()))[:32]} col%NO hist(32) {torch.bincount((t % NO) // 32, minlength=8).tolist()} tiles {sorted(set((t // NO).tolist()))[:12]} first {bad[:5].tolist()} vals {y[ch[0], t[0]].item():.4f} ref {ref[ch[0], t[0]].item():.4f} x {x[ch[0], t[0]].item():.4f}", end="")
    print()
print("repeat equal", torch.equal(outs[1][0], outs[1][1]))
