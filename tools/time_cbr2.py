"""Phase timing of the fused ConvBlockRes kernel (csrc/conv_cbr2.hip) with a -DRVC_CONV_TIMING build:
    bash tools/build_variant.sh timing -DRVC_CONV_TIMING && RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_timing.so python tools/time_cbr2.py"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import _lib as L
L.require_experiments()      # (reads rvc_debug_* hooks: variant builds only)
L.get_ctx(0)
for Cc, H, W in ((16, 3232, 128), (32, 1616, 64)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Cc, H, W, generator=g).cuda(); y = torch.empty_like(x)
    w1 = (torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc)).contiguous(); b1 = torch.randn(Cc, generator=g)
    w2 = (torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc)).contiguous(); b2 = torch.randn(Cc, generator=g)
    out = (C.c_uint64 * 8)()
    for rep in range(3):
        L.check(L.lib.rvc_debug_conv_timing(out, 1))
        L.check(L.lib.rvc_op_cbr2_small(None, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), Cc, H, W))
        L.check(L.lib.rvc_debug_conv_timing(out, 1))
    n = max(out[0], 1)
    names = ["wgs", "staging", "conv1", "y1->lds", "conv2", "epilogue", "total"]
    print(f"C={Cc} H={H} W={W}: " + ", ".join(f"{names[i]} {out[i] / n:.0f}" for i in range(1, 7)) + f" (100 MHz ticks per workgroup, {n} workgroups)")
