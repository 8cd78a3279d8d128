import sys; sys.path.insert(0,'.')
import torch
from comfy_rvc_amd import _lib as L
L.get_ctx(0)
for Cc,T in ((768,1599),(192,3001),(512,1599)):
    x=torch.randn(Cc,T,device="cuda"); g=torch.ones(Cc,device="cuda"); b=torch.zeros(Cc,device="cuda"); y=torch.empty_like(x)
    run=lambda: L.check(L.lib.rvc_op_layernorm_c(None,L.ptr(x),None,L.ptr(g),L.ptr(b),L.ptr(y),Cc,T))
    run(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"LN C{Cc} T{T}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
