"""One-off robustness checks of the device pipeline against the CPU oracle: very short clip, silence, loud clip, stereo input."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single
from comfy_rvc_amd.lib.audio import remix_audio
from oracle import pipeline as opl

cfg = Config()
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg)
rm = RMVPE(S.rmvpe_state_dict(0))
sds = (S.hubert_state_dict(0), S.rmvpe_state_dict(0), S.synth_state_dict(S.CONFIG_40K_V2, "v2", 0))

def run(name, audio_in, sr=16000, **kw):
    a16, _ = remix_audio((audio_in, sr), target_sr=16000)            # what vc_single does first
    g = torch.Generator().manual_seed(3); tape = []
    def rec(shape):
        t = torch.randn(shape, generator=g); tape.append(t); return t
    args = dict(rms_mix_rate=0.25, protect=0.33); args.update(kw)
    try:
        ref = opl.pipeline(*sds, S.CONFIG_40K_V2, "v2", a16, noise_fn=rec, **args)
    except Exception as e:   # noqa: BLE001
        ref = None; print(f"{name}: oracle raised {type(e).__name__}: {e}")
    vc = VC(40000, cfg); vc.model_rmvpe = rm
    it = iter(tape) if ref is not None else None
    vc.noise_fn = (lambda shape: next(it)) if it is not None else None
    out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio_in, sr), sid=0, f0_up_key=0, f0_method="rmvpe",
                    index_rate=0.0, **args)
    if ref is None:
        print(f"{name}: product returned {None if out is None else out[0].shape}")
        return
    if out is None:
        print(f"{name}: product returned None but oracle gave {ref.shape}"); return
    d = np.abs(out[0].astype(np.int32) - ref.astype(np.int32))
    print(f"{name}: shape {out[0].shape} vs {ref.shape}, max diff {d.max() if d.size else 0} LSB, frac<=33 {(d <= 33).mean() if d.size else 1:.5f}, nan {np.isnan(out[0].astype(np.float64)).any()}")

run("short 0.4 s", S.synth_audio(0.4, seed=1))
run("short 0.12 s", S.synth_audio(0.12, seed=1))
run("silence 1 s", np.zeros(16000, np.float32))
run("loud (peak 3.0) 1 s", (S.synth_audio(1.0, seed=2) * 6).astype(np.float32))
run("stereo 1 s", np.stack([S.synth_audio(1.0, seed=3), S.synth_audio(1.0, seed=4)], 0))
run("rms_mix 1.0 / protect 0.5", S.synth_audio(1.0, seed=5), rms_mix_rate=1.0, protect=0.5)
run("float64 input", S.synth_audio(1.0, seed=6).astype(np.float64))
