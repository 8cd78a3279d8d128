"""Container experiment (CPU, oracle): what does rounding the ResBlock weights of the generator to ONE fp16 / bf16 term cost in int16 LSB on a
full-size golden?  The 2-MFMA split keeps the activations as hi + lo (22 bits), so the weight rounding is the whole error of the mode.
usage: python tools/exp/fp16_weight_rounding.py [fp16|bf16|none] [stages e.g. 012] [golden] [family]"""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from comfy_rvc_amd import synthetic as S
from oracle import nets, pipeline as opl
from conftest import golden, golden_clip, parity_stats

mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
stages = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "012")]
gname = sys.argv[3] if len(sys.argv) > 3 else "pipeline_30s_40k_v2.npz"
cfg = S.CONFIG_48K_V2 if "48k" in gname else S.CONFIG_40K_V2
g = golden(gname)
audio = golden_clip(g)
sd = S.synth_state_dict(cfg, "v2", 0)
sd = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in sd.items()}

def rnd(w):
    w = torch.as_tensor(w, dtype=torch.float32)
    if mode == "none":
        return w
    s = w.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)      # per-output-channel scale (folded back in the epilogue)
    q = (w / s).to(torch.float16 if mode == "fp16" else torch.bfloat16).float()
    return q * s

nk = len(cfg[10])
n = 0
for i in stages:
    for j in range(nk):
        for m in range(3):
            for c in ("convs1", "convs2"):
                p = f"dec.resblocks.{i * nk + j}.{c}.{m}."
                w = nets.weight_norm_fold(torch.as_tensor(sd[p + "weight_v"]), torch.as_tensor(sd[p + "weight_g"]))
                wq = rnd(w)
                sd[p + "weight_v"] = wq.numpy()
                sd[p + "weight_g"] = wq.flatten(1).norm(dim=1).view(-1, 1, 1).numpy()
                n += 1
gen = torch.Generator().manual_seed(int(g["noise_seed"]))
t0 = time.time()
out = opl.pipeline(S.hubert_state_dict(0), S.rmvpe_state_dict(0), sd, cfg, "v2", audio, noise_fn=lambda shp: torch.randn(tuple(shp), generator=gen))
print(mode, "stages", stages, "convs rounded", n, gname, parity_stats(out, g["out_i16"], 33), f"{time.time()-t0:.0f} s", flush=True)
