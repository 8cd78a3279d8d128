import sys; sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
g = dict(np.load("tests/golden/rmvpe_1s.npz"))
m = RMVPE(S.rmvpe_state_dict(0))
r1 = m.infer(g["audio"], want_mel=True, want_salience=True)
s1 = r1["salience"].cpu().numpy()
print("run1 sal err", np.abs(s1 - g["salience"]).max())
r2 = m.infer(g["audio"], want_mel=True, want_salience=True)
s2 = r2["salience"].cpu().numpy()
print("run2 sal err", np.abs(s2 - g["salience"]).max(), "equal", np.array_equal(s1, s2))
f0 = m.infer_from_audio(g["audio"])
print("f0 nonzero", (f0 > 0).sum(), "ref", (g["f0"] > 0).sum(), np.isnan(f0).sum())
r3 = m.infer(g["audio"], want_mel=True, want_salience=True)
print("run3 sal err", np.abs(r3["salience"].cpu().numpy() - g["salience"]).max())
