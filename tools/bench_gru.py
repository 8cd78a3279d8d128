"""GRU scan time inside an RMVPE forward (rocprof-free: events around rm.infer and a run with the scan's cost isolated by variants)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
rm = RMVPE(S.rmvpe_state_dict(0))
a = torch.from_numpy(np.pad(S.synth_audio(30.0, seed=1), (16000, 16000), mode="reflect")).cuda()
for _ in range(2): rm.infer(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): rm.infer(a)
torch.cuda.synchronize(); print("rmvpe forward %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
