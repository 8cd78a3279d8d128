import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from comfy_rvc_amd import _lib as L, synthetic as S
from oracle import nets
L.get_ctx(0)
upp,sr,T=400,40000,320
f0 = torch.from_numpy(S.designed_f0(T, seed=0)).view(1, T)
noise = torch.zeros(1, T*upp, 1)
sd = {"dec.m_source.l_linear.weight": torch.tensor([[0.9]]), "dec.m_source.l_linear.bias": torch.tensor([0.01])}
# oracle internals
f = f0[:, None].transpose(1, 2)
rad_o = (f / sr) % 1
tmp_o = torch.cumsum(rad_o, 1); tmp_o *= upp
tint = F.interpolate(tmp_o.transpose(2,1), scale_factor=float(upp), mode="linear", align_corners=True).transpose(2,1)
radu = F.interpolate(rad_o.transpose(2,1), scale_factor=float(upp), mode="nearest").transpose(2,1)
tm = tint % 1
idx = (tm[:,1:,:]-tm[:,:-1,:])<0
sh = torch.zeros_like(radu); sh[:,1:,:] = idx*-1.0
c_o = torch.cumsum(radu+sh, dim=1)[0,:,0]
N=T*upp
fd, nd = f0.view(-1).cuda(), noise.view(-1).cuda()
har=torch.empty(N,device="cuda"); sine=torch.empty(N,device="cuda"); rad=torch.empty(T,device="cuda"); tmp=torch.empty(T,device="cuda"); ph=torch.empty(N,device="cuda")
L.check(L.lib.rvc_op_sine_source(None, L.ptr(fd), L.ptr(nd), L.ptr(har), L.ptr(sine), T, upp, float(sr), 0.9, 0.01, L.ptr(rad), L.ptr(tmp), L.ptr(ph)))
print("rad equal", torch.equal(rad.cpu(), rad_o[0,:,0]), (rad.cpu()-rad_o[0,:,0]).abs().max().item())
print("tmp equal", torch.equal(tmp.cpu(), tmp_o[0,:,0]), (tmp.cpu()-tmp_o[0,:,0]).abs().max().item())
d=(ph.cpu()-c_o)
dr = d - d.round()
print("phase diff: max |d|", d.abs().max().item(), "max frac diff", dr.abs().max().item(), "n nonint", int((dr.abs()>1e-6).sum()))
nz = torch.nonzero(dr.abs()>1e-6)[:5].view(-1)
for i in nz.tolist(): print(i, ph[i].item(), c_o[i].item(), (radu+sh)[0,i,0].item())
# shifts compare
wr_o = torch.nonzero(sh[0,:,0]<0).view(-1)
print("oracle wraps", wr_o.numel())
