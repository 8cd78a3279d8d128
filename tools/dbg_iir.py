import sys; sys.path.insert(0, '.')
import numpy as np, torch
from scipy import signal
from comfy_rvc_amd import _lib as L
from comfy_rvc_amd.vc_infer_pipeline import _AH, _BH, _ZI, _SOS, _SOS_ZI, ah, bh
L.get_ctx(0)
for n in (48000, 5003, 1920, 160001):
    rng = np.random.default_rng(n)
    t = np.arange(n) / 16000.0
    x = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.05 * rng.standard_normal(n) + 0.2).astype(np.float32)
    ref = signal.filtfilt(bh, ah, x)
    xd = torch.from_numpy(x).cuda()
    filt = torch.empty(n, dtype=torch.float64, device="cuda")
    padded = torch.empty(n + 2 * 16000, dtype=torch.float32, device="cuda")
    n1 = n // 8000 + 1
    rms1 = torch.empty(n1, dtype=torch.float64, device="cuda")
    L.check(L.lib.rvc_preprocess(None, L.ptr(xd), 0, n, L.ptr(_BH), L.ptr(_AH), L.ptr(_ZI), 16000, L.ptr(filt), L.ptr(padded), L.ptr(rms1), n1, L.ptr(_SOS), L.ptr(_SOS_ZI)))
    torch.cuda.synchronize()
    got = filt.cpu().numpy()
    e = np.abs(got - ref)
    print(n, "max err", e.max() / np.abs(ref).max(), "at", e.argmax(), "nan", np.isnan(got).sum(), "first bad", np.nonzero(e > 1e-6 * np.abs(ref).max())[0][:5])
