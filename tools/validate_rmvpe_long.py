"""One-off: RMVPE on a 60 s clip (BASELINE configs[1]: n = 6201 frames, U-Net input [1,1,6208,128]) against the CPU oracle."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
from oracle import nets
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
audio = np.pad(S.synth_audio(secs, seed=8), (16000, 16000), mode="reflect")
sd = S.rmvpe_state_dict(0)
t0 = time.time(); ref = nets.rmvpe_infer_from_audio(sd, audio, thred=0.03); print("oracle %.1f s" % (time.time() - t0))
rm = RMVPE(sd)
rm.infer_from_audio(audio)
torch.cuda.synchronize(); t0 = time.time(); f0 = rm.infer_from_audio(audio); print("device %.1f ms, frames %d" % (1e3 * (time.time() - t0), f0.shape[0]))
assert f0.shape == ref.shape
voiced = (f0 > 0) == (ref > 0)
ok = np.abs(f0 - ref) <= 1e-3 * np.maximum(ref, 1.0)
print("voiced agreement %.5f, f0 within 1e-3: %.5f, max rel err on agreeing frames %.2e" % (voiced.mean(), ok.mean(), (np.abs(f0 - ref) / np.maximum(ref, 1.0))[ok].max()))
assert voiced.mean() > 0.995 and ok.mean() > 0.995
print("rmvpe long OK")
