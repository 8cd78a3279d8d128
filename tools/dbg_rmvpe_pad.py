"""Debug: RMVPE U-Net output with the padded split-resident levels against the oracle, for several clip lengths."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
from oracle import nets
def rel(a, b): a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
sd = S.rmvpe_state_dict(0)
m = RMVPE(sd)
for secs in (1.0, 2.6):
    audio = S.synth_audio(secs, seed=3)
    taps = {}
    nets.rmvpe_infer_from_audio(sd, audio, taps=taps)
    n = audio.shape[0] // 160 + 1; Tr = 32 * ((n - 1) // 32 + 1)
    dt = {"unet_out": torch.empty(16, Tr, 128, device="cuda"), "gru": torch.empty(512, Tr, device="cuda")}
    r = m.infer(audio, want_mel=True, want_salience=True, taps=dt)
    u = dt["unet_out"].cpu().numpy(); ur = taps["unet_out"].numpy()[0]
    print(secs, "Tr", Tr, "unet_out rel", rel(u, ur), "salience abs", np.abs(r["salience"].cpu().numpy() - taps["salience"]).max())
    d = np.abs(u - ur).max(axis=(0, 2))   # per time row
    print("   worst rows", np.argsort(-d)[:8], d[np.argsort(-d)[:8]] / np.abs(ur).max())
    d2 = np.abs(u - ur).max(axis=(0, 1))
    print("   worst cols", np.argsort(-d2)[:8], d2[np.argsort(-d2)[:8]] / np.abs(ur).max())
