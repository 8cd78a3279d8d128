import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.rmvpe import RMVPE
from oracle import nets
g=np.load("tests/golden/rmvpe_1s.npz"); audio=g["audio"]
sd=S.rmvpe_state_dict(0); m=RMVPE(sd)
n=audio.shape[0]//160+1; Tr=32*((n-1)//32+1)
def run():
    dt={"unet_out":torch.empty(16,Tr,128,device="cuda"),"gru":torch.empty(512,Tr,device="cuda")}
    r=m.infer(audio,want_salience=True,taps=dt); torch.cuda.synchronize()
    return dt["unet_out"].cpu(), dt["gru"].cpu(), r["salience"].cpu()
a=run(); b=run(); c=run()
for nm,x,y,z in zip(("unet","gru","sal"),a,b,c): print(nm,"deterministic:",torch.equal(x,y) and torch.equal(y,z), (x-y).abs().max().item())
# truth in fp64 from the device's own unet_out
tsd=nets.tensors(sd)
x=F.conv2d(a[0][None].double(), tsd["cnn.weight"].double(), tsd["cnn.bias"].double(), padding=1)
x=x.transpose(1,2).flatten(-2)[0]
def scan(x,wi,wh,bi,bh,rev,dt):
    T=x.shape[0]; H=256; gi=F.linear(x.to(dt),wi.to(dt),bi.to(dt)); h=torch.zeros(H,dtype=dt); out=torch.empty(T,H,dtype=dt); wt=wh.to(dt).t().contiguous()
    for t in (range(T-1,-1,-1) if rev else range(T)):
        gh=h@wt+bh.to(dt); r=torch.sigmoid(gi[t,:H]+gh[:H]); z=torch.sigmoid(gi[t,H:2*H]+gh[H:2*H]); nn=torch.tanh(gi[t,2*H:]+r*gh[2*H:]); h=(1-z)*nn+z*h; out[t]=h
    return out
res={}
for dt in (torch.float64, torch.float32):
    f=scan(x,tsd["fc.0.gru.weight_ih_l0"],tsd["fc.0.gru.weight_hh_l0"],tsd["fc.0.gru.bias_ih_l0"],tsd["fc.0.gru.bias_hh_l0"],False,dt)
    bw=scan(x,tsd["fc.0.gru.weight_ih_l0_reverse"],tsd["fc.0.gru.weight_hh_l0_reverse"],tsd["fc.0.gru.bias_ih_l0_reverse"],tsd["fc.0.gru.bias_hh_l0_reverse"],True,dt)
    res[dt]=torch.cat([f,bw],1).double()
dev=a[1].t().double()
print("cpu fp32 vs fp64 truth:", (res[torch.float32]-res[torch.float64]).abs().max().item())
print("device    vs fp64 truth:", (dev-res[torch.float64]).abs().max().item())
d=(dev-res[torch.float64]).abs()
print("err by direction fwd/bwd:", d[:,:256].max().item(), d[:,256:].max().item())
print("err per time (fwd) first 10:", d[:10,:256].max(1).values.numpy().round(6))
print("err per time (fwd) last 5:", d[-5:,:256].max(1).values.numpy().round(6))
ts=d[:,:256].max(1).values; print("first t with err>1e-4:", int((ts>1e-4).nonzero()[0]) if (ts>1e-4).any() else None)
u=d[:, :256].max(0).values; print("units with err>1e-3:", (u>1e-3).nonzero().view(-1)[:20].tolist())
