import sys; sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from comfy_rvc_amd import _lib as L, synthetic as S
L.get_ctx(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4100
upp, sr = 400, 40000
f0 = torch.from_numpy(S.designed_f0(T, seed=0)).view(1, T)
N = T * upp
# oracle pieces
f = f0[:, None].transpose(1, 2)
rad = (f / sr) % 1
tmp = torch.cumsum(rad, 1); tmp *= upp
tmpi = F.interpolate(tmp.transpose(2, 1), scale_factor=float(upp), mode="linear", align_corners=True).transpose(2, 1)
radu = F.interpolate(rad.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
tm1 = tmpi % 1
idx = (tm1[:, 1:, :] - tm1[:, :-1, :]) < 0
shift = torch.zeros_like(radu); shift[:, 1:, :] = idx * -1.0
c = torch.cumsum(radu + shift, dim=1)[0, :, 0]
# device
noise = torch.zeros(N)
har = torch.empty(N, device="cuda"); sine = torch.empty(N, device="cuda"); ph = torch.empty(N, device="cuda")
radd = torch.empty(T, device="cuda"); tmpd = torch.empty(T, device="cuda")
fd, nd = f0.view(-1).cuda(), noise.cuda()
L.check(L.lib.rvc_op_sine_source(None, L.ptr(fd), L.ptr(nd), L.ptr(har), L.ptr(sine), T, upp, float(sr), 0.9, 0.01, L.ptr(radd), L.ptr(tmpd), L.ptr(ph)))
torch.cuda.synchronize()
pd = ph.cpu()
print("tmp frame cumsum equal:", torch.equal(tmpd.cpu(), tmp[0, :, 0]), "max tmp", float(tmp.max()))
d = (pd - c)
nz = torch.nonzero(d.abs() > 0.5).view(-1)
print("phase max abs diff", float(d.abs().max()), "first integer divergence at sample", int(nz[0]) if nz.numel() else None, "of", N)
if nz.numel():
    i = int(nz[0]); fr = i // upp
    print(" frame", fr, "f0 around", f0[0, max(fr-2,0):fr+3].tolist(), "oracle shifts so far", int(shift[0, :i+1, 0].sum()), "c oracle", float(c[i]), "c dev", float(pd[i]))
    # device-side shift count is not exported; compare interpolated tmp mod 1 near i with a float64 evaluation
    j = i
    print(" oracle tm1 around", tm1[0, j-2:j+2, 0].tolist())
sw = torch.sin(c * 2 * np.pi) * 0.1
uvu = F.interpolate((f > 0).float().transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)[0, :, 0]
print("sine max abs diff (voiced)", float(((sine.cpu() - sw) * uvu).abs().max()))
