import sys; sys.path.insert(0,'.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
from oracle import nets
def rel(a,b): a=np.asarray(a,np.float64); b=np.asarray(b,np.float64); return np.abs(a-b).max()/max(np.abs(b).max(),1e-30)
g=np.load("tests/golden/rmvpe_1s.npz")
audio=g["audio"]
sd=S.rmvpe_state_dict(0)
taps={}
f0_ref=nets.rmvpe_infer_from_audio(sd,audio,taps=taps)
m=RMVPE(sd)
n=audio.shape[0]//160+1; Tr=32*((n-1)//32+1)
dt={"unet_out":torch.empty(16,Tr,128,device="cuda"),"gru":torch.empty(512,Tr,device="cuda")}
r=m.infer(audio,want_mel=True,want_salience=True,taps=dt)
mel=r["mel"].cpu().numpy(); melr=taps["mel"][0].numpy()
print("mel abs err max",np.abs(mel-melr).max(),"mean",np.abs(mel-melr).mean(), "linear-domain rel", rel(np.exp(mel),np.exp(melr)))
k=np.unravel_index(np.abs(mel-melr).argmax(),mel.shape); print(" worst mel at",k,mel[k],melr[k])
print("unet_out rel",rel(dt["unet_out"].cpu().numpy()[None],taps["unet_out"].numpy()))
print("gru rel",rel(dt["gru"].cpu().numpy().T,taps["gru"].numpy()), "max|gru|", np.abs(taps["gru"].numpy()).max())
print("salience abs",np.abs(r["salience"].cpu().numpy()-taps["salience"]).max())
# feed oracle mel into... compare with fp64-ish reference: run oracle e2e in float64 to see CPU fp32's own error
sd64={k:torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype.kind=='f' else torch.from_numpy(np.asarray(v)) for k,v in sd.items()}
