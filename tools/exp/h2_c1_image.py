"""c1 of an fp16x2 pair alone: read the image back, decode hi + lo, compare with fp64 (round 6 debugging).  Needs a -DRVC_EXPERIMENTS build."""
import ctypes as C, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from comfy_rvc_amd import _lib as L
L.lib.rvc_debug_read_scratch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]

def tp_of(T): return (T + 64 + 704 + 63) & ~63

def one(Cc, k, d, T):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    L.check(L.lib.rvc_set_pair_arithmetic(1)); L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
        pl = C.c_void_p()
        L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
        plans.append(pl)
    L.check(L.lib.rvc_set_conv_precision(1))
    xg = x.cuda(); y = torch.empty(Cc, T, device="cuda")
    ref = F.leaky_relu(F.conv1d(F.leaky_relu(x.cuda().double(), 0.1)[None], w1.half().double().cuda(), b1.double().cuda(), padding=(k - 1) // 2 * d, dilation=d)[0], 0.1).cpu()
    tp = tp_of(T); nbytes = (Cc // 16) * 2 * tp * 32
    for rep in range(3):
        os.environ["RVC_EXP_PAIR_C1ONLY"] = "1"
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
        buf = np.empty(nbytes, dtype=np.uint8)
        L.check(L.lib.rvc_debug_read_scratch(None, 5, buf.ctypes.data, nbytes))
        img = buf.view(np.float16).reshape(Cc // 16, 2, 2, tp, 8)          # [chunk][hi|lo][half][row][8 ch]
        val = (img[:, 0].astype(np.float64) + img[:, 1].astype(np.float64))[:, :, 64:64 + T, :]      # [chunk][half][T][8]
        val = np.transpose(val, (0, 1, 3, 2)).reshape(Cc, T)
        err = np.abs(val - ref.numpy())
        bad = ~(err <= 1e-3 * np.abs(ref.numpy()).max())
        print(f"C{Cc} k{k} d{d} T{T} rep{rep}: c1 image max err {np.nanmax(err):.3e}, bad {int(bad.sum())}, NaN {int(np.isnan(val).sum())}", end="")
        if bad.sum():
            ch, t = np.nonzero(bad)
            print(f" | t%256//32 hist {np.bincount((t % 256) // 32, minlength=8).tolist()} t%32 set {sorted(set((t % 32).tolist()))} ch%16 set {sorted(set((ch % 16).tolist()))} tiles {sorted(set((t // 256).tolist()))[:10]}", end="")
        print(flush=True)

print("env", {k: v for k, v in os.environ.items() if k.startswith("RVC_")}, flush=True)
one(128, 7, 1, 319800)
one(64, 7, 1, 639600)
