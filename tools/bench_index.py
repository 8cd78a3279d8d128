"""Retrieval index: exact k = 1 search of T_h query frames over N rows (768-d), fp32-MFMA GEMM vs bf16x3 conv layers (RVC_INDEX_X3)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd.lib.feature_index import DeviceIndex
rng = np.random.default_rng(0)
N, T = 100000, 1599
big = rng.standard_normal((N, 768)).astype(np.float32)
q = (big[rng.integers(0, N, T)] + 0.3 * rng.standard_normal((T, 768))).astype(np.float32)
idx = DeviceIndex(big)
f = torch.from_numpy(q).cuda().t().contiguous()
idx.blend_device(f, 0.75); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): idx.blend_device(f, 0.75)
e1.record(); torch.cuda.synchronize()
print(f"index search + blend, N = {N}, T = {T}: {e0.elapsed_time(e1) / 5:.2f} ms")
s, ix = idx.search(q, k=1)
d = ((q[:, None, :].astype(np.float64) - big[ix[:, 0]][:, None, :]) ** 2).sum(-1)[:, 0]
best = np.array([((big.astype(np.float64) - q[t].astype(np.float64)) ** 2).sum(1).min() for t in range(0, T, 16)])
print("max relative excess of the found distance over the true minimum:", float(np.max(d[::16] / best - 1.0)))
