"""Phase timing of attention_dma_kernel<96, .., REL> with a -DRVC_CONV_TIMING build (see tools/time_cbr2.py)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
heads, T, D = 2, 3198, 96
g = torch.Generator().manual_seed(1)
q = (torch.randn(heads * D, T, generator=g) * 0.3).cuda(); k = torch.randn(heads * D, T, generator=g).cuda(); v = torch.randn(heads * D, T, generator=g).cuda()
bv = torch.randn(heads * D, generator=g).cuda(); ek = torch.randn(21, D, generator=g) * 0.5; ev = torch.randn(21, D, generator=g) * 0.5
out = torch.empty(heads * D, T, device="cuda")
t = (C.c_uint64 * 8)()
for kz in (1, 2, 5):
    for rep in range(2):
        L.check(L.lib.rvc_debug_conv_timing(t, 1))
        L.check(L.lib.rvc_op_attention_split_rel(None, q.data_ptr(), k.data_ptr(), v.data_ptr(), bv.data_ptr(), ek.data_ptr(), ev.data_ptr(), out.data_ptr(), None, heads, T, kz))
        L.check(L.lib.rvc_debug_conv_timing(t, 1))
    n = max(t[0], 1)
    names = ["wgs", "prologue", "tiles", "store+ticket", "merge(all since tiles)", "epilogue", "total"]
    print(f"kz={kz}: " + ", ".join(f"{names[i]} {t[i] / n:.0f}" for i in range(1, 7)) + f" (100 MHz ticks summed over {n} workgroups / workgroups)")
