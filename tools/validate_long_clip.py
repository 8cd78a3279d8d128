"""One-off validation of the segmented device path with the real constants (clip > 41 s -> 2 segments) against the CPU oracle."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single
from oracle import pipeline as opl

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
variant = sys.argv[2] if len(sys.argv) > 2 else "40k_v2"          # 40k_v2 | 48k_v2 | 40k_v1
CFG, VER = {"40k_v2": (S.CONFIG_40K_V2, "v2"), "48k_v2": (S.CONFIG_48K_V2, "v2"), "40k_v1": (S.CONFIG_40K_V1, "v1")}[variant]
audio = S.synth_audio(secs, seed=77)
f0fn = lambda x, **k: S.designed_f0(x.shape[0] // 160 + 1, seed=0).astype(np.float64)
g = torch.Generator().manual_seed(5); tape = []
def rec(shape):
    t = torch.randn(shape, generator=g); tape.append(t); return t
t0 = time.time()
ref = opl.pipeline(S.hubert_state_dict(0), S.rmvpe_state_dict(0), S.synth_state_dict(CFG, VER, 0), CFG, VER, audio,
                   rms_mix_rate=0.25, protect=0.33, noise_fn=rec, f0_override=f0fn)
print("oracle %.1f s, %d noise draws" % (time.time() - t0, len(tape)))
import os
from comfy_rvc_amd import _lib
_lib.get_ctx(0)
_lib.check(_lib.lib.rvc_set_conv_precision(int(os.environ.get("PRECISION", "1"))))
cfg = Config()
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
vcd = get_vc(S.synth_checkpoint(CFG, VER, 0), config=cfg)
vc = VC(CFG[-1], cfg); vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0)); vc.f0_method_dict["pm"] = f0fn
it = iter(tape); vc.noise_fn = lambda shape: next(it)
out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio, 16000), sid=0, f0_up_key=0, f0_method="pm",
                index_rate=0.0, rms_mix_rate=0.25, protect=0.33)
d = np.abs(out[0].astype(np.int32) - ref.astype(np.int32))
print("shape", out[0].shape, ref.shape, "max |diff| LSB", d.max(), "frac <= 33:", (d <= 33).mean())
idx = np.argsort(d)[-5:]
print("worst samples", idx, d[idx], "ref", ref[idx], "segment boundary near", [int(i) for i in np.where(d > 33)[0][:5]], "count >33:", int((d > 33).sum()))
fl = vc.last_float.cpu().numpy() if hasattr(vc.last_float, "cpu") else vc.last_float
print("float peak", np.abs(fl).max())
assert out[0].shape == ref.shape and d.max() <= 33
print("long clip OK")
