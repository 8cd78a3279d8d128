"""Where do two runs of the same fp16x2 pair differ?  (round 6 debugging)"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from comfy_rvc_amd import _lib as L

def run(Cc, k, d, T, arith, reps=3):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    L.check(L.lib.rvc_set_pair_arithmetic(arith))
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
        pl = C.c_void_p()
        L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
        plans.append(pl)
    L.check(L.lib.rvc_set_conv_precision(1))
    xg = x.cuda()
    outs = []
    for r in range(reps):
        y = torch.empty(Cc, T, device="cuda")
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
        torch.cuda.synchronize()
        outs.append(y)
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)
    return outs

if __name__ == "__main__":
    for (Cc, T) in ((256, 31980), (128, 319800), (64, 639600)):
        for k, d in ((3, 1), (3, 3), (3, 5), (7, 1), (7, 5), (11, 1), (11, 5)):
            if Cc == 64 and k == 3: continue
            o1 = run(Cc, k, d, T, 1)
            o0 = run(Cc, k, d, T, 0, reps=1)[0]
            ref_err = float((o1[0] - o0).abs().max() / o0.abs().max())
            msg = f"C{Cc} k{k} d{d} T{T}: h2 vs bf16x3 max rel {ref_err:.2e}"
            for i in (1, 2):
                diff = (o1[i] != o1[0])
                n = int(diff.sum())
                if n:
                    idx = diff.nonzero()
                    ts = idx[:, 1]; cs = idx[:, 0]
                    msg += f" | rep{i}: {n} differ, ch {int(cs.min())}..{int(cs.max())} t {int(ts.min())}..{int(ts.max())} t%256 hist {torch.bincount((ts % 256) // 32, minlength=8).tolist()} first {idx[:6].tolist()} maxabs {float((o1[i]-o1[0]).abs().max()):.3g}"
            bad = ((o1[0] - o0).abs() > 1e-3 * o0.abs().max())
            if int(bad.sum()):
                idx = bad.nonzero()
                msg += f" | vs bf16x3: {int(bad.sum())} outliers first {idx[:6].tolist()} t%256 hist {torch.bincount((idx[:,1] % 256)//32, minlength=8).tolist()}"
            print(msg, flush=True)

