"""Clip lengths around the thresholds at which the generator's narrow stages switch between the pair kernels and one launch per ResBlock (conv_rb3_kernel wants two rounds of its
tiles: 4.7 - 6.3 s of 40 kHz output depending on stage and kernel size), each converted in BOTH pair arithmetics: fp16x2 (fused ResBlocks where eligible) and bf16x3 (no fused
ResBlock at all) must agree to a few LSB of the int16 result on every sample, and a repeat must reproduce the bits."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L, synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single
cfg = Config()
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg)
vc = VC(40000, cfg); vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0))
prev = L.lib.rvc_get_pair_arithmetic()
worst = 0
for secs in (4.5, 4.75, 5.0, 5.3, 5.9, 6.0, 6.2, 6.3, 6.5, 7.0):
    audio = S.synth_audio(secs, seed=int(secs * 100))
    outs = {}
    for arith in (1, 0, 1):
        L.check(L.lib.rvc_set_pair_arithmetic(arith))
        g = torch.Generator(device="cpu").manual_seed(7)
        vc.noise_fn = lambda shape, g=g: torch.randn(shape, generator=g)
        out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio, 16000), sid=0, f0_up_key=0, f0_method="rmvpe",
                        index_rate=0.0, rms_mix_rate=0.25, protect=0.33)
        o = out[0].astype(np.int64)
        if arith in outs:
            assert np.array_equal(outs[arith], o), f"{secs} s: a repeat in arithmetic {arith} differs"
        outs[arith] = o
    d = np.abs(outs[1] - outs[0])
    worst = max(worst, int(d.max()))
    print(f"{secs:5.2f} s -> {len(o):7d} samples: fp16x2 vs bf16x3 max {int(d.max()):3d} LSB, mean {d.mean():.2f}; repeat identical", flush=True)
    assert d.max() <= 40, (secs, int(d.max()))
L.check(L.lib.rvc_set_pair_arithmetic(prev))
print(f"threshold lengths OK (worst {worst} LSB between the arithmetics)")
