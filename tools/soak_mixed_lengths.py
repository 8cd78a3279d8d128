"""Soak: clips of mixed lengths through one set of loaded models; every repeat of a length must reproduce its first result bit for bit
(stale arena / scratch contents, buffer growth and re-use), and device memory must stop growing once the largest clip has been seen."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single
cfg = Config()
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg)
vc = VC(40000, cfg); vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0))
first = {}
lengths = [30, 2, 45, 0.4, 10, 30, 45, 2, 0.4, 61, 10, 30, 61]
for k, secs in enumerate(lengths):
    audio = S.synth_audio(secs, seed=int(secs * 10))
    g = torch.Generator(device="cpu").manual_seed(int(secs * 10))
    vc.noise_fn = lambda shape, g=g: torch.randn(shape, generator=g)
    out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio, 16000), sid=0, f0_up_key=0, f0_method="rmvpe",
                    index_rate=0.0, rms_mix_rate=0.25, protect=0.33)
    assert out is not None and np.isfinite(out[0].astype(np.float64)).all()
    free, total = torch.cuda.mem_get_info()
    same = None
    if secs in first:
        same = bool(np.array_equal(first[secs], out[0]))
    else:
        first[secs] = out[0].copy()
    print(f"{k:2d} {secs:5.1f} s -> {out[0].shape[0]:8d} samples, peak {np.abs(out[0]).max():5d}, repeat identical: {same}, device used {(total - free) / 2**30:.2f} GiB")
    assert same is not False
print("soak OK")
