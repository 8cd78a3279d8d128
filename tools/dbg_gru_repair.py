"""healthy scan vs faulty scan + serial repair: hidden states of both directions, per time step"""
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S, _lib as L
from comfy_rvc_amd.lib.rmvpe import RMVPE
audio=S.synth_audio(1.0, seed=2)
m=RMVPE(S.rmvpe_state_dict(0))
n=(audio.shape[0]+0)//160+1
def run():
    r=m.infer(audio,want_salience=True); torch.cuda.synchronize()
    Tr=r["salience"].shape[0]
    return r
def run_taps():
    a=torch.from_numpy(audio)
    # padded length inside the pipeline is unknown here: read Tr from a first run
    r=m.infer(audio,want_salience=True); n=r["salience"].shape[0]; Tr=32*((n-1)//32+1)
    dt={"gru":torch.empty(512,Tr,device="cuda")}
    r=m.infer(audio,want_salience=True,taps=dt); torch.cuda.synchronize()
    return dt["gru"].cpu(), r["salience"].cpu(), r["f0"].cpu()
g0,s0,f0=run_taps()
L.check(L.lib.rvc_rmvpe_debug_fault(m._h, 1, 1<<10))
g1,s1,f1=run_taps()
print("repaired flag", L.lib.rvc_rmvpe_repaired(m._h, None))
L.check(L.lib.rvc_rmvpe_debug_fault(m._h, 0, 0))
d=(g0-g1).abs()
print("hidden max diff fwd/bwd:", d[:256].max().item(), d[256:].max().item(), "equal:", torch.equal(g0,g1))
T=g0.shape[1]
print("per step fwd first 6:", d[:256].max(0).values[:6].numpy()); print("per step bwd last 6 (its first steps):", d[256:].max(0).values[-6:].numpy())
print("units fwd with diff:", (d[:256].max(1).values>0).sum().item(), " bwd:", (d[256:].max(1).values>0).sum().item())
print("f0 equal:", torch.equal(f0,f1), "sal max diff", (s0-s1).abs().max().item())
