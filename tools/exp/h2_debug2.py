"""Map of the positions where the fp16x2 pair differs from bf16x3 (round 6 debugging): by channel block, column block of the tile."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.exp.h2_debug import run  # noqa

def analyse(Cc, k, d, T, BM, BN):
    o1 = run(Cc, k, d, T, 1, reps=1)[0].cpu()
    o0 = run(Cc, k, d, T, 0, reps=1)[0].cpu()
    bad = ~((o1 - o0).abs() <= 1e-3 * o0.abs().max())
    idx = bad.nonzero()
    if idx.shape[0] == 0:
        print(f"C{Cc} k{k} d{d}: clean"); return
    ch, t = idx[:, 0], idx[:, 1]
    m = torch.zeros(Cc // 32, BN // 32, dtype=torch.long)
    m.index_put_((ch // 32, (t % BN) // 32), torch.ones_like(ch), accumulate=True)
    tiles = torch.unique(t // BN)
    print(f"C{Cc} k{k} d{d} T{T}: {idx.shape[0]} bad ({int(torch.isnan(o1).sum())} NaN); map [ch/32][col block]:\n{m.tolist()}\n  bad column tiles: {tiles.shape[0]} of {(T + BN - 1) // BN}, first {tiles[:12].tolist()} last {tiles[-6:].tolist()}")
    for t0 in tiles[:3].tolist():
        blk = bad[:, t0 * BN:(t0 + 1) * BN]
        for cb in range(Cc // 32):
            for nb in range(BN // 32):
                sub = blk[cb * 32:(cb + 1) * 32, nb * 32:(nb + 1) * 32]
                if int(sub.sum()):
                    rows = sub.any(dim=1).nonzero().flatten().tolist(); cols = sub.any(dim=0).nonzero().flatten().tolist()
                    vals = o1[:, t0 * BN:(t0 + 1) * BN][cb * 32:(cb + 1) * 32, nb * 32:(nb + 1) * 32][sub][:4].tolist()
                    refs = o0[:, t0 * BN:(t0 + 1) * BN][cb * 32:(cb + 1) * 32, nb * 32:(nb + 1) * 32][sub][:4].tolist()
                    print(f"   tile {t0} block ch{cb} col{nb}: {int(sub.sum())} bad, rows {rows}, cols {cols[:40]}, vals {vals} ref {refs}")

print("env", {k: v for k, v in os.environ.items() if k.startswith("RVC_")}, flush=True)
analyse(128, 7, 1, 319800, 128, 256)
analyse(64, 7, 1, 639600, 64, 256)
