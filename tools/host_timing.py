"""Stage timing of VC.pipeline on the GPU box (host DSP vs device stages) for a 30 s clip."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from scipy import signal
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.lib.model_utils import change_rms
from comfy_rvc_amd import vc_infer_pipeline as P

cfg = Config()
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
vcd = P.get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg)
vc = P.VC(40000, cfg); vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0)); vc.noise_on_device = True
audio = S.synth_audio(30.0, seed=100)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t = [T()]
    a = signal.filtfilt(P.bh, P.ah, audio); t.append(T())
    ap = np.pad(a, (16000, 16000), mode="reflect"); t.append(T())
    f0 = vc.model_rmvpe.infer_from_audio(ap); t.append(T())
    pitch, pitchf = vc.get_f0(ap, 0, "pm" if False else "rmvpe", f0_min=50, f0_max=1600) if False else (None, None)
    vc.f0_method_dict["x"] = lambda **k: f0.copy()
    pitch, pitchf = vc.get_f0(ap, 0, "x", f0_min=50, f0_max=1600); t.append(T())
    pt = torch.from_numpy(pitch.astype(np.int64))[None]; pf = torch.from_numpy(pitchf.astype(np.float32))[None]; t.append(T())
    out = vc.vc(hub, vcd["net_g"], torch.tensor([0]), ap, pt, pf, [0,0,0], None, None, 0.0, "v2", 0.33); t.append(T())
    out = out[40000:-40000]
    o2 = change_rms(a, 16000, out, 40000, 0.25); t.append(T())
    m = np.abs(o2).max() / 0.99; i16 = (o2 * 32768 / m).astype(np.int16); t.append(T())
    names = ["filtfilt", "pad", "rmvpe(+d2h)", "f0 post", "to torch", "vc fused(+noise,+d2h)", "change_rms", "int16"]
    print(" | ".join(f"{n} {1e3*(t[i+1]-t[i]):.1f}" for i, n in enumerate(names)), "| total %.1f ms" % (1e3*(t[-1]-t[0])))
# finer: inside vc
import ctypes as C
from comfy_rvc_amd import _lib
L = ap.shape[0]; Th = hub.num_frames(L); Tn = 2*Th
t0=T(); nz = torch.randn((1,192,Tn), device="cuda"); ns = torch.randn((1,Tn*400,1), device="cuda"); t1=T()
a_d = torch.from_numpy(ap).float().cuda(); t2=T()
outd = torch.empty(Tn*400, device="cuda"); t3=T(); h = outd.cpu().numpy(); t4=T()
print("noise gen %.2f ms, h2d audio %.2f ms, d2h out %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t4-t3)))
for nm, fn in (("hubert", lambda: hub.extract_features(a_d[None], version="v2", channel_major=True)),
               ("rmvpe", lambda: vc.model_rmvpe.infer(a_d))):
    fn(); t0=T(); fn(); t1=T(); print(nm, "%.2f ms" % (1e3*(t1-t0)))
feats = hub.extract_features(a_d[None], version="v2", channel_major=True)
fu = feats.repeat_interleave(2, dim=1).contiguous()
net = vcd["net_g"]
pc = pt[0,:Tn].cuda(); pfc = pf[0,:Tn].cuda()
def syn(): return net.infer(fu, torch.LongTensor([Tn]), pc[None], pfc[None], torch.LongTensor([0]), noise=(nz, ns), phone_channel_major=True)
syn(); t0=T(); syn(); t1=T(); print("synth %.2f ms" % (1e3*(t1-t0)))
