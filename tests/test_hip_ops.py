"""GPU parity of the single HIP ops (through the C ABI) against the same torch-CPU fp32 ops the oracle is built from."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, x3p_check_count

pytestmark = pytest.mark.gpu

ACT = {"none": 0, "lrelu": 1, "relu": 2, "gelu": 3, "tanh": 4, "sigmoid": 5}


def _act(x, name, slope):
    return {"none": lambda v: v, "lrelu": lambda v: F.leaky_relu(v, slope), "relu": F.relu, "gelu": F.gelu,
            "tanh": torch.tanh, "sigmoid": torch.sigmoid}[name](x)


@pytest.fixture(scope="module")
def L():
    from comfy_rvc_amd import _lib
    _lib.get_ctx(0)
    return _lib


def dev(x):
    return torch.as_tensor(x, dtype=torch.float32).contiguous().cuda()


CONV1D = [
    # Ci, Co, T, k, s, pad, dil, groups, pre, act, res, out_scale, accumulate
    (32, 32, 1000, 3, 1, 1, 1, 1, "lrelu", "none", True, 1.0, False),
    (32, 32, 777, 11, 1, 25, 5, 1, "lrelu", "none", False, 1.0, False),
    (64, 64, 600, 7, 1, 9, 3, 1, "lrelu", "none", True, 1.0 / 3, True),
    (128, 128, 300, 3, 1, 1, 1, 1, "none", "none", False, 1.0, False),
    (256, 256, 130, 11, 1, 5, 1, 1, "none", "lrelu", False, 1.0, False),
    (192, 384, 100, 5, 1, 2, 1, 1, "none", "none", False, -1.0, True),
    (512, 512, 401, 3, 2, 0, 1, 1, "none", "gelu", False, 1.0, False),
    (512, 512, 200, 2, 2, 0, 1, 1, "none", "gelu", False, 1.0, False),
    (768, 768, 50, 128, 1, 64, 1, 16, "none", "gelu", False, 1.0, False),
    (10, 512, 3000, 1, 1, 0, 1, 1, "none", "none", False, 1.0, False),
    (32, 1, 2000, 7, 1, 3, 1, 1, "lrelu", "tanh", False, 1.0, False),
    (21, 96, 50, 1, 1, 0, 1, 1, "none", "none", False, 1.0, True),
    (1, 32, 500, 1, 1, 0, 1, 1, "none", "none", False, 1.0, False),
    (513, 128, 131, 1, 1, 0, 1, 1, "none", "sigmoid", False, 1.0, False),
    (768, 3072, 49, 1, 1, 0, 1, 1, "none", "gelu", False, 1.0, False),
    (32, 32, 70000, 7, 1, 3, 1, 1, "lrelu", "none", True, 1.0, False),
]


@pytest.mark.parametrize("case", CONV1D, ids=[f"c{i}" for i in range(len(CONV1D))])
def test_conv1d(L, case):
    Ci, Co, T, k, s, pad, dil, groups, pre, act, res, scale, accum = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)     # (hash() of a tuple with strings changes per process)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci // groups, k, generator=g) / np.sqrt(Ci // groups * k)
    b = torch.randn(Co, generator=g) * 0.1
    xin = _act(x, pre, 0.1)
    ref = F.conv1d(xin[None], w, b, stride=s, padding=pad, dilation=dil, groups=groups)[0]
    Tout = ref.shape[1]
    r = torch.randn(Co, Tout, generator=g) if res else None
    if r is not None:
        ref = ref + r
    ref = _act(ref, act, 0.1) * scale
    y0 = torch.randn(Co, Tout, generator=g)
    if accum:
        ref = ref + y0
    y = dev(y0)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    xd, rd = dev(x), (dev(r) if res else None)      # keep device tensors alive across the call
    L.check(L.lib.rvc_op_conv1d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(rd), L.ptr(y), Ci, Co, T, k, s,
                                pad, dil, groups, ACT[pre], 0.1, ACT[act], 0.1, 0, scale, int(accum)))
    assert rel_err(y.cpu(), ref) < 2e-5


X3_CONV1D = [
    # Ci, Co, T, k, pad, dil, pre, act, res, out_scale, accumulate      (grids >= 400 workgroups so that the bf16x3 kernel is chosen)
    (128, 128, 60001, 11, 25, 5, "lrelu", "none", False, 1.0, False),
    (128, 128, 55555, 3, 1, 1, "lrelu", "none", True, 1.0, False),
    (256, 256, 26000, 7, 9, 3, "lrelu", "none", True, 1.0 / 3, True),
    (64, 64, 131000, 11, 5, 1, "lrelu", "lrelu", False, 1.0, False),
    (64, 64, 140001, 7, 15, 5, "none", "none", True, 1.0, False),
    (32, 32, 300007, 11, 25, 5, "lrelu", "none", True, 1.0 / 3, True),
    (32, 32, 260000, 3, 3, 3, "lrelu", "none", False, 1.0, False),
    (192, 192, 52000, 5, 2, 1, "none", "none", False, 1.0, False),
    (16, 96, 120000, 4, 1, 1, "none", "relu", False, 1.0, False),
    (768, 768, 1499, 1, 0, 1, "none", "none", True, 1.0, False),        # HuBERT projections: k = 1, 4 chunks per stage
    (768, 3072, 1499, 1, 0, 1, "none", "none", False, 1.0, False),
    (3072, 768, 1499, 1, 0, 1, "none", "none", True, 1.0, False),
    (512, 768, 1499, 1, 0, 1, "none", "none", False, 1.0, False),
    (96, 192, 40000, 1, 0, 1, "none", "none", False, 1.0, False),       # 6 chunks: 2 per stage
    (80, 128, 40000, 3, 2, 2, "lrelu", "none", False, 1.0, False),      # 5 chunks: 1 per stage
    (192, 384, 3000, 5, 2, 1, "none", "none", False, 1.0, False),       # flow WaveNet in_layer
    (128, 128, 52001, 3, 3, 3, "lrelu", "none", True, 1.0, False),      # k = 3: 2 chunks per stage
    (64, 64, 140000, 3, 0, 1, "none", "none", False, 1.0, False, 2),    # stride 2 (HuBERT feature encoder): phase sub-planes in LDS
    (128, 128, 70001, 2, 0, 1, "none", "gelu", False, 1.0, False, 2),
    (512, 512, 20001, 3, 0, 1, "none", "gelu", False, 1.0, False, 2),
    (48, 64, 100000, 5, 2, 1, "lrelu", "none", False, 1.0, False, 3),
    (128, 128, 160001, 7, 9, 3, "lrelu", "none", True, 1.0, False),     # >= 600 workgroups of 128 x 256: wide tile 2x2x2x4
    (64, 64, 310000, 11, 25, 5, "lrelu", "none", True, 1.0 / 3, True),  # wide tile 1x4x2x4 (64 x 512)
    (48, 128, 70000, 7, 3, 1, "lrelu", "none", True, 1.0, False),       # pipelined kernel, three chunks: first / last-but-one / last chunk rules all at once
    (64, 128, 66001, 11, 5, 1, "none", "lrelu", True, 0.5, True),       # four chunks, activation after the sum (residual through the epilogue, not the accumulators)
    (512, 512, 60001, 3, 0, 1, "none", "none", False, 1.0, False, 2),   # stride 2 on the pipelined kernel (phase sub-planes), odd length
    (768, 192, 3198, 3, 1, 1, "none", "none", True, 1.0, False),        # text encoder FFN (k = 3, 100 fps): the GEMM kernel with tap-shifted input pieces, split K
    (192, 768, 3001, 3, 1, 1, "none", "relu", False, 1.0, False),
    (192, 384, 3198, 5, 4, 2, "lrelu", "none", False, 1.0, False),      # flow WaveNet in_layer, dilated
]


@pytest.mark.parametrize("case", X3_CONV1D, ids=[f"x{i}" for i in range(len(X3_CONV1D))])
def test_conv1d_bf16x3(L, case):
    """bf16x3 split-MFMA Conv1d (conv_x3.hip) against an fp64 torch reference: error of a few 1e-6, far inside the 1e-3 budget."""
    Ci, Co, T, k, pad, dil, pre, act, res, scale, accum = case[:11]
    stride = case[11] if len(case) > 11 else 1
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)     # (hash() of a tuple with strings changes per process)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / np.sqrt(Ci * k)
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv1d(_act(x, pre, 0.1).double()[None], w.double(), b.double(), stride=stride, padding=pad, dilation=dil)[0]
    Tout = ref.shape[1]
    r = torch.randn(Co, Tout, generator=g) if res else None
    if r is not None:
        ref = ref + r
    ref = _act(ref, act, 0.1) * scale
    y0 = torch.randn(Co, Tout, generator=g)
    if accum:
        ref = ref + y0
    y = dev(y0)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_op_conv1d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(rd), L.ptr(y), Ci, Co, T, k, stride,
                                    pad, dil, 1, ACT[pre], 0.1, ACT[act], 0.1, 0, scale, int(accum)))
        L.check(L.lib.rvc_prof_collect(ms, fl, ln))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
        L.check(L.lib.rvc_set_conv_precision(1))
    assert sum(ln[14:24]) == 1 and sum(ln[:14]) == 0, "the launch did not go through conv_x3_kernel"
    assert rel_err(y.cpu().double(), ref) < 2e-5
    # the same layer on the fp32 kernel: both must agree with the reference, bf16x3 within a small factor of fp32's own error
    L.check(L.lib.rvc_set_conv_precision(0))
    y32 = dev(y0)
    try:
        L.check(L.lib.rvc_op_conv1d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(rd), L.ptr(y32), Ci, Co, T, k, stride,
                                    pad, dil, 1, ACT[pre], 0.1, ACT[act], 0.1, 0, scale, int(accum)))
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    assert rel_err(y32.cpu().double(), ref) < 4e-6


TCONV1D = [(512, 256, 30, 16, 10, 3), (64, 32, 100, 4, 2, 1), (128, 64, 33, 24, 12, 6), (256, 128, 40, 20, 10, 5)]


@pytest.mark.parametrize("case", TCONV1D)
def test_conv_transpose1d(L, case):
    Ci, Co, T, k, u, pad = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Ci, Co, k, generator=g) / np.sqrt(Ci * k / u)
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1)[None], w, b, stride=u, padding=pad)[0]
    y0 = torch.randn(ref.shape, generator=g)
    y, xd, wc, bc = dev(y0), dev(x), w.contiguous().numpy(), b.numpy()
    L.check(L.lib.rvc_op_conv_transpose1d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(y), Ci, Co, T, k, u,
                                          pad, 1, 0.1, 1))
    assert y.shape[1] == T * u
    assert rel_err(y.cpu(), ref + y0) < 2e-5


@pytest.mark.parametrize("case", [(128, 64, 60000, 4, 2, 1), (256, 128, 30001, 16, 10, 3), (64, 32, 300000, 4, 2, 1), (512, 256, 3000, 16, 10, 3)])
def test_conv_transpose1d_bf16x3(L, case):
    """The generator's upsampling layers on the bf16x3 kernel (polyphase rows, interleaved store)."""
    Ci, Co, T, k, u, pad = case
    g = torch.Generator().manual_seed(9)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Ci, Co, k, generator=g) / np.sqrt(Ci * k / u)
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).double()[None], w.double(), b.double(), stride=u, padding=pad)[0]
    y, xd, wc, bc = torch.zeros(ref.shape, device="cuda"), dev(x), w.contiguous().numpy(), b.numpy()
    ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_op_conv_transpose1d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(y), Ci, Co, T, k, u, pad, 1, 0.1, 0))
        L.check(L.lib.rvc_prof_collect(ms, fl, ln))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
        L.check(L.lib.rvc_set_conv_precision(1))
    assert sum(ln[14:24]) == 1 and sum(ln[:14]) == 0, "the launch did not go through conv_x3_kernel"
    assert rel_err(y.cpu().double(), ref) < 2e-5


CONV2D = [(1, 16, 64, 128), (16, 16, 96, 128), (32, 64, 48, 32), (128, 128, 12, 16), (256, 512, 6, 4), (16, 3, 64, 128), (512, 512, 3, 4)]


@pytest.mark.parametrize("case", CONV2D)
def test_conv2d3x3_relu_residual(L, case):
    Ci, Co, H, W = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / np.sqrt(Ci * 9)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, H, W, generator=g)
    ref = F.relu(F.conv2d(x[None], w, b, padding=1)[0]) + r
    y, xd, rd, wc, bc = torch.empty(Co, H, W, device="cuda"), dev(x), dev(r), w.contiguous().numpy(), b.numpy()
    L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(rd), L.ptr(y), Ci, Co, H, W, 1))
    assert rel_err(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("case", [(16, 16, 3232, 128), (32, 32, 1616, 64), (64, 32, 1617, 64), (64, 64, 808, 32), (128, 64, 803, 32), (16, 32, 1000, 128),
                                  (48, 16, 2000, 128),
                                  # deep U-Net levels: the pipelined GEMM kernel with tap-shifted input pieces, split over K
                                  (128, 128, 404, 16), (256, 256, 202, 8), (512, 512, 101, 4), (256, 512, 101, 4), (512, 256, 203, 8), (64, 128, 401, 16)])
def test_conv2d3x3_bf16x3(L, case):
    """RMVPE U-Net 3x3 convolutions on the bf16x3 kernel (halo patch staged per tile of image rows)."""
    Ci, Co, H, W = case
    g = torch.Generator().manual_seed(13)
    x = torch.randn(Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / np.sqrt(Ci * 9)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double()[None], w.double(), b.double(), padding=1)[0]) + r
    y, xd, rd, wc, bc = torch.empty(Co, H, W, device="cuda"), dev(x), dev(r), w.contiguous().numpy(), b.numpy()
    ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_op_conv2d3x3(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(rd), L.ptr(y), Ci, Co, H, W, 1))
        L.check(L.lib.rvc_prof_collect(ms, fl, ln))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
        L.check(L.lib.rvc_set_conv_precision(1))
    assert sum(ln[14:24]) == 1 and sum(ln[:14]) == 0, "the launch did not go through conv_x3_kernel"
    assert rel_err(y.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("case", [(512, 256, 3, 4), (32, 16, 32, 64), (64, 32, 16, 32)])
def test_conv_transpose2d(L, case):
    Ci, Co, H, W = case
    g = torch.Generator().manual_seed(13)
    x = torch.randn(Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 3, 3, generator=g) / np.sqrt(Ci * 9 / 4)
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x[None], w, b, stride=2, padding=1, output_padding=1)[0])
    y, xd, wc, bc = torch.empty(Co, 2 * H, 2 * W, device="cuda"), dev(x), w.contiguous().numpy(), b.numpy()
    L.check(L.lib.rvc_op_conv_transpose2d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(y), Ci, Co, H, W, 1))
    assert rel_err(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("case", [(512, 256, 101, 4), (256, 128, 202, 8), (128, 64, 404, 16), (64, 32, 808, 32), (32, 16, 1616, 64)])
def test_conv_transpose2d_bf16x3(L, case):
    """RMVPE decoder up-convolutions on the bf16x3 kernel: dense phase-major 3x3 conv (split-K on the deep levels) + 2x2 interleave."""
    Ci, Co, H, W = case
    g = torch.Generator().manual_seed(14)
    x = torch.randn(Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 3, 3, generator=g) / np.sqrt(Ci * 9 / 4)
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.double()[None], w.double(), b.double(), stride=2, padding=1, output_padding=1)[0])
    y, xd, wc, bc = torch.empty(Co, 2 * H, 2 * W, device="cuda"), dev(x), w.contiguous().numpy(), b.numpy()
    ms = (C.c_double * 24)(); fl = (C.c_double * 24)(); ln = (C.c_int64 * 24)()
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_op_conv_transpose2d(None, L.ptr(xd), L.ptr(wc), L.ptr(bc), L.ptr(y), Ci, Co, H, W, 1))
        L.check(L.lib.rvc_prof_collect(ms, fl, ln))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
        L.check(L.lib.rvc_set_conv_precision(1))
    assert sum(ln[14:24]) == 1 and sum(ln[:14]) == 0, "the launch did not go through conv_x3_kernel"
    assert rel_err(y.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("case", [(49, 49, 64, 12), (64, 49, 49, 12), (100, 192, 192, 1), (1536, 70, 384, 1), (96, 333, 333, 2), (333, 333, 96, 2)])
def test_gemm_tn(L, case):
    M, N, K, B = case
    g = torch.Generator().manual_seed(17)
    a = torch.randn(B, K, M, generator=g)
    b = torch.randn(B, K, N, generator=g)
    ref = torch.einsum("zkm,zkn->zmn", a, b)
    y, ad, bd = torch.empty(B, M, N, device="cuda"), dev(a), dev(b)
    L.check(L.lib.rvc_op_gemm_tn(None, L.ptr(ad), L.ptr(bd), L.ptr(y), M, N, K, B))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("heads,T", [(12, 1599), (12, 100), (3, 64), (2, 777)])
def test_fused_attention(L, heads, T):
    """softmax(K^T Q) V + bias without the score matrix (attention.hip) against torch's attention in float64."""
    g = torch.Generator().manual_seed(23)
    D = 64
    q = torch.randn(heads * D, T, generator=g) * 0.35          # pre-scaled queries, logits of a few units like HuBERT's
    k = torch.randn(heads * D, T, generator=g)
    v = torch.randn(T, heads * D, generator=g)
    bv = torch.randn(heads * D, generator=g) * 0.1
    qh = q.double().view(heads, D, T); kh = k.double().view(heads, D, T); vh = v.double().view(T, heads, D).permute(1, 0, 2)
    p = torch.softmax(torch.einsum("hdq,hdk->hqk", qh, kh), dim=-1)
    ref = torch.einsum("hqk,hkd->hdq", p, vh).reshape(heads * D, T) + bv.double()[:, None]
    out, qd, kd, vd, bd = torch.empty(heads * D, T, device="cuda"), dev(q), dev(k), dev(v), dev(bv)
    L.check(L.lib.rvc_op_attention(None, L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(bd), L.ptr(out), heads, T))
    torch.cuda.synchronize()
    assert rel_err(out.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("heads,T", [(12, 1599), (2, 64), (2, 65), (3, 1000), (2, 130), (1, 7), (2, 1540)])
def test_attention_on_split_resident_operands(L, heads, T):
    """attention_dma.hip: Q / K / V^T handed over as bf16 hi / lo images (DMA-staged key tiles, softmax and probabilities in registers);
    fp32 rows AND the output image against torch float64.  The images' rows past T hold NaN bit patterns: masked keys must not leak."""
    g = torch.Generator().manual_seed(31 + T)
    D = 64
    q = torch.randn(heads * D, T, generator=g) * 0.35
    k = torch.randn(heads * D, T, generator=g)
    v = torch.randn(heads * D, T, generator=g)
    bv = torch.randn(heads * D, generator=g) * 0.1
    qh = q.double().view(heads, D, T); kh = k.double().view(heads, D, T); vh = v.double().view(heads, D, T)
    p = torch.softmax(torch.einsum("hdq,hdk->hqk", qh, kh), dim=-1)
    ref = torch.einsum("hqk,hdk->hdq", p, vh).reshape(heads * D, T) + bv.double()[:, None]
    out = torch.empty(heads * D, T, device="cuda"); oimg = torch.empty(heads * D, T, device="cuda")
    qd, kd, vd, bd = dev(q), dev(k), dev(v), dev(bv)
    L.check(L.lib.rvc_op_attention_split(None, L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(bd), L.ptr(out), L.ptr(oimg), heads, T))
    torch.cuda.synchronize()
    assert rel_err(out.cpu().double(), ref) < 2e-5
    assert rel_err(oimg.cpu().double(), ref) < 2e-5
    assert (oimg - out).abs().max().item() <= 2.0 ** -15 * ref.abs().max().item()      # the image is hi + lo of the same fp32 values


@pytest.mark.parametrize("heads,T,kz", [(2, 3198, 0), (2, 3198, 1), (2, 1000, 3), (2, 150, 0), (2, 20, 0), (3, 129, 1), (2, 5, 0), (2, 700, 5)])
def test_relative_attention_on_split_resident_operands(L, heads, T, kz):
    """The text encoder's attention on images (attention_dma_kernel<96, .., REL>): rel-k bias by an in-kernel MFMA block, band probabilities
    x E_v by another, key tiles cut into kz slices merged by the last arriver.  float64 torch restatement of attentions.py:230-267;
    repeated launches are bit-identical (the merge order is the slice order)."""
    g = torch.Generator().manual_seed(37 + T)
    D, W = 96, 10
    q = torch.randn(heads * D, T, generator=g) * 0.3
    k = torch.randn(heads * D, T, generator=g)
    v = torch.randn(heads * D, T, generator=g)
    bv = torch.randn(heads * D, generator=g) * 0.1
    ek = torch.randn(2 * W + 1, D, generator=g) * 0.5
    ev = torch.randn(2 * W + 1, D, generator=g) * 0.5
    qh = q.double().view(heads, D, T); kh = k.double().view(heads, D, T); vh = v.double().view(heads, D, T)
    rel = torch.einsum("rd,hdq->hrq", ek.double(), qh)
    sc = torch.einsum("hdq,hdk->hqk", qh, kh)
    qi = torch.arange(T)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        sc[:, qi[ok], ki[ok]] += rel[:, r, qi[ok]]
    p = torch.softmax(sc, dim=-1)
    ref = torch.einsum("hqk,hdk->hdq", p, vh)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        ref[:, :, qi[ok]] += p[:, qi[ok], ki[ok]][:, None, :] * ev.double()[r][None, :, None]
    ref = ref.reshape(heads * D, T) + bv.double()[:, None]
    qd, kd, vd, bd = dev(q), dev(k), dev(v), dev(bv)
    outs = []
    for _ in range(3):
        out = torch.empty(heads * D, T, device="cuda"); oimg = torch.empty(heads * D, T, device="cuda")
        L.check(L.lib.rvc_op_attention_split_rel(None, L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(bd), ek.contiguous().data_ptr(), ev.contiguous().data_ptr(),
                                                 L.ptr(out), L.ptr(oimg), heads, T, kz))
        torch.cuda.synchronize()
        outs.append(out)
    assert rel_err(outs[0].cpu().double(), ref) < 2e-5
    assert rel_err(oimg.cpu().double(), ref) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("C,H,W", [(16, 64, 128), (16, 37, 128), (16, 5, 64), (32, 48, 64), (32, 33, 64), (32, 3, 128), (16, 200, 192)])
def test_conv_block_res_small_channels_fused(L, C, H, W):
    """conv_cbr2.hip: relu(conv3x3(relu(conv3x3(x) + b1)) + b2) + x in one launch (RMVPE's 16 / 32-channel ConvBlockRes, lib/rmvpe.py:233-268):
    the intermediate lives in LDS as a bf16 hi / lo image, image borders and tile halos are zeros / recomputed.  float64 torch reference."""
    g = torch.Generator().manual_seed(41 + H)
    x = torch.randn(C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / np.sqrt(9 * C); b1 = torch.randn(C, generator=g) * 0.2
    w2 = torch.randn(C, C, 3, 3, generator=g) / np.sqrt(9 * C); b2 = torch.randn(C, generator=g) * 0.2
    xd = x.double()[None]
    y1 = F.relu(F.conv2d(xd, w1.double(), b1.double(), padding=1))
    ref = (F.relu(F.conv2d(y1, w2.double(), b2.double(), padding=1)) + xd)[0]
    y = torch.full((C, H, W), 9.0, device="cuda"); xdev = dev(x)
    L.check(L.lib.rvc_op_cbr2_small(None, L.ptr(xdev), w1.contiguous().data_ptr(), b1.data_ptr(), w2.contiguous().data_ptr(), b2.data_ptr(), L.ptr(y), C, H, W))
    torch.cuda.synchronize()
    assert rel_err(y.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("Ci,C,H,W", [(32, 16, 40, 128), (16, 32, 19, 64), (16, 16, 33, 128), (32, 32, 9, 64), (16, 3, 50, 128), (16, 32, 7, 192)])
def test_small_channel_conv3x3_with_merged_shortcut(L, Ci, C, H, W):
    """conv3_small_kernel: relu(conv3x3(x) + b) and - as a second group of rows whose 3 x 3 kernel is zero except for the centre tap - the block's 1 x 1
    shortcut, one launch, two outputs (RMVPE's channel-changing ConvBlockRes, lib/rmvpe.py:233-268); then the plain form with a residual.  float64 reference."""
    g = torch.Generator().manual_seed(43 + H)
    x = torch.randn(Ci, H, W, generator=g)
    w1 = torch.randn(C, Ci, 3, 3, generator=g) / np.sqrt(9 * Ci); b1 = torch.randn(C, generator=g) * 0.2
    xd = dev(x)
    merged = 2 * C <= (64 if Ci == 16 else 32)
    if merged:
        ws = torch.randn(C, Ci, generator=g) / np.sqrt(Ci); bs = torch.randn(C, generator=g) * 0.2
        wm = torch.zeros(2 * C, Ci, 3, 3); wm[:C] = w1; wm[C:, :, 1, 1] = ws
        bm = torch.cat([b1, bs])
        y = torch.full((C, H, W), 9.0, device="cuda"); y2 = torch.full((C, H, W), 9.0, device="cuda")
        L.check(L.lib.rvc_op_conv3_small(None, L.ptr(xd), wm.contiguous().data_ptr(), bm.data_ptr(), None, L.ptr(y), L.ptr(y2), Ci, 2 * C, H, W, C, C))
        torch.cuda.synchronize()
        ref1 = F.relu(F.conv2d(x.double()[None], w1.double(), b1.double(), padding=1))[0]
        ref2 = torch.einsum("oc,chw->ohw", ws.double(), x.double()) + bs.double()[:, None, None]
        assert rel_err(y.cpu().double(), ref1) < 2e-5 and rel_err(y2.cpu().double(), ref2) < 2e-5
    # plain: act(conv + b) + residual (C <= 3: no activation, no residual - the output convolution)
    r = torch.randn(C, H, W, generator=g); rd = dev(r)
    y = torch.full((C, H, W), 9.0, device="cuda")
    act = C > 3
    L.check(L.lib.rvc_op_conv3_small(None, L.ptr(xd), w1.contiguous().data_ptr(), b1.data_ptr(), L.ptr(rd) if act else None, L.ptr(y), None, Ci, C, H, W, C, C if act else 0))
    torch.cuda.synchronize()
    ref = F.conv2d(x.double()[None], w1.double(), b1.double(), padding=1)[0]
    ref = F.relu(ref) + r.double() if act else ref
    assert rel_err(y.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("Ci,Co,T,row0,rows", [(768, 2304, 1599, 1536, 768), (768, 2304, 100, 1536, 768), (64, 256, 333, 128, 128), (192, 192, 1000, 0, 192)])
def test_gemm_split_swapped_product(L, Ci, Co, T, row0, rows):
    """out[t][j] = sum_c x[c][t] w[row0 + j][c] written as the image of the transposed tensor (the attention's V^T operand): rows t < T
    against float64, rows T .. ceil64(T) exact zeros although the input image holds NaN patterns there."""
    g = torch.Generator().manual_seed(5 + T)
    x = torch.randn(Ci, T, generator=g); w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci)
    ref = x.double().t() @ w.double()[row0:row0 + rows].t()
    T64 = (T + 63) // 64 * 64
    yt = torch.full((T64, rows), 3.0, device="cuda")
    xd = dev(x); wh = w.contiguous()
    L.check(L.lib.rvc_op_gemm_split_swapped(None, L.ptr(xd), wh.data_ptr(), L.ptr(yt), Ci, Co, T, row0, rows))
    torch.cuda.synchronize()
    y = yt.cpu().double()
    assert rel_err(y[:T], ref) < 2e-5
    assert torch.equal(y[T:], torch.zeros(T64 - T, rows, dtype=torch.float64))


@pytest.mark.parametrize("Ci,H,T,k", [(192, 192, 3198, 5), (192, 192, 200, 5), (64, 32, 1000, 3), (32, 48, 77, 1)])
def test_wn_in_layer_with_gate_in_the_epilogue(L, Ci, H, T, k):
    """The flow's WaveNet layer (reference lib/infer_pack/modules.py WN.forward, commons.fused_add_tanh_sigmoid_multiply): acts = tanh(a_t + g_t) * sigmoid(a_s + g_s) with
    a = in_layer(x), computed in the epilogue of the split-resident GEMM from rows packed so that a lane holds both halves of a channel - against float64."""
    g = torch.Generator().manual_seed(21 + T + H)
    x = torch.randn(Ci, T, generator=g); w = torch.randn(2 * H, Ci, k, generator=g) / np.sqrt(Ci * k); b = torch.randn(2 * H, generator=g) * 0.1
    gc = torch.randn(2 * H, generator=g) * 0.5
    a = F.conv1d(x.double()[None], w.double(), b.double(), padding=(k - 1) // 2)[0] + gc.double()[:, None]
    ref = torch.tanh(a[:H]) * torch.sigmoid(a[H:])
    y = torch.full((H, T), 9.0, device="cuda")
    xd, gd = dev(x), dev(gc)
    L.check(L.lib.rvc_op_wn_in_gate_split(None, L.ptr(xd), w.contiguous().data_ptr(), b.contiguous().data_ptr(), L.ptr(gd), L.ptr(y), Ci, H, T, k))
    torch.cuda.synchronize()
    assert rel_err(y.cpu().double(), ref) < 2e-5


@pytest.mark.parametrize("Ci,Co,T,vt_row0", [(768, 2304, 1599, 1536), (192, 576, 3198, 384), (768, 2304, 100, 1536), (192, 576, 61, 384), (64, 256, 333, 128), (256, 384, 1000, 256)])
def test_gemm_split_fused_qkv_with_transposed_v(L, Ci, Co, T, vt_row0):
    """q | k | v as ONE launch of the split-resident GEMM (HuBERT: 768 -> 2304, text encoder: 192 -> 576): rows below vt_row0 as their image, the rows from vt_row0 on
    through the transposing epilogue into the attention's V^T image - both against float64; the transposed rows T .. ceil64(T) exact zeros although the input image holds
    NaN patterns past T."""
    g = torch.Generator().manual_seed(11 + T + Co)
    x = torch.randn(Ci, T, generator=g); w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci); b = torch.randn(Co, generator=g) * 0.1
    b[vt_row0:] = 0.0                                                      # (v's bias is added behind the attention)
    ref = w.double() @ x.double() + b.double()[:, None]
    rows = Co - vt_row0
    T64 = (T + 63) // 64 * 64
    yq = torch.full((vt_row0, T), 3.0, device="cuda"); yt = torch.full((T64, rows), 3.0, device="cuda")
    xd = dev(x)
    L.check(L.lib.rvc_op_gemm_split_qkv(None, L.ptr(xd), w.contiguous().data_ptr(), b.contiguous().data_ptr(), L.ptr(yq), L.ptr(yt), Ci, Co, T, vt_row0))
    torch.cuda.synchronize()
    assert rel_err(yq.cpu().double(), ref[:vt_row0]) < 2e-5
    y = yt.cpu().double()
    assert rel_err(y[:T], ref[vt_row0:].t()) < 2e-5
    assert torch.equal(y[T:], torch.zeros(T64 - T, rows, dtype=torch.float64))


@pytest.mark.parametrize("Ci,Co,T,ld,off", [(256, 1024, 700, 1026, 1), (32, 128, 333, 130, 1), (64, 192, 129, 192, 0)])
def test_gemm_split_swapped_product_with_residual(L, Ci, Co, T, ld, off):
    """y[t][off + j] = sum_c x[c][t] w[j][c] + res[t][off + j], rows of pitch ld (MDX23C's second TDF linear lands in the padded plane layout with
    x1 added: tfc_tdf.py:142): against float64; the columns outside [off, off + Co) keep what they held."""
    g = torch.Generator().manual_seed(9 + T)
    x = torch.randn(Ci, T, generator=g); w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci); r = torch.randn(T, ld, generator=g)
    y = torch.full((T, ld), 7.0, device="cuda")
    xd = dev(x); rd = dev(r); wh = w.contiguous()
    L.check(L.lib.rvc_op_gemm_split_swapped_res(None, L.ptr(xd), wh.data_ptr(), L.ptr(rd), L.ptr(y), Ci, Co, T, ld, off))
    torch.cuda.synchronize()
    got = y.cpu().double()
    ref = x.double().t() @ w.double().t() + r.double()[:, off:off + Co]
    assert rel_err(got[:, off:off + Co], ref) < 2e-5
    keep = torch.ones(ld, dtype=torch.bool); keep[off:off + Co] = False
    assert torch.equal(got[:, keep], torch.full((T, int(keep.sum())), 7.0, dtype=torch.float64))


@pytest.mark.parametrize("Ci1,Ci2,Co,H,W,ksplit", [(128, 128, 128, 96, 1024, 0), (128, 128, 128, 12, 64, 0), (64, 128, 64, 9, 30, 0), (48, 96, 48, 5, 16, 0), (256, 256, 256, 4, 16, 4), (256, 512, 256, 4, 16, 8), (256, 512, 256, 4, 16, 11)])
def test_two_image_product_conv3x3_plus_1x1(L, Ci1, Ci2, Co, H, W, ksplit):
    """conv3x3(x1) + conv1x1(x2) as ONE split-resident product over two padded images (MDX23C's tfc2(x) + shortcut(x0), tfc_tdf.py:137-144): the second
    image's chunks are further units of the reduction with tap offset 0 - also when a K slice starts inside the second image (ksplit) - and the raw
    image written beside the fp32 output holds the same values (bf16 hi + lo: 2^-16 relative).  The first case is a grid of 770 tiles of 128 x 128: the
    three-slot ring with three workgroups per CU."""
    g = torch.Generator().manual_seed(Ci1 + W)
    x1 = torch.randn(Ci1, H, W, generator=g); x2 = torch.randn(Ci2, H, W, generator=g)
    w1 = torch.randn(Co, Ci1, 3, 3, generator=g) / np.sqrt(Ci1 * 9); w2 = torch.randn(Co, Ci2, generator=g) / np.sqrt(Ci2)
    y = torch.full((Co, H, W), 5.0, device="cuda"); yi = torch.full((Co, H, W), 5.0, device="cuda")
    x1d = dev(x1); x2d = dev(x2)
    L.check(L.lib.rvc_op_conv2d3x3_plus_1x1(None, L.ptr(x1d), w1.contiguous().data_ptr(), L.ptr(x2d), w2.contiguous().data_ptr(), L.ptr(y), L.ptr(yi), Ci1, Ci2, Co, H, W, ksplit))
    torch.cuda.synchronize()
    ref = F.conv2d(x1.double()[None], w1.double(), padding=1)[0] + torch.einsum("oc,chw->ohw", w2.double(), x2.double())
    assert rel_err(y.cpu().double(), ref) < 2e-5
    assert rel_err(yi.cpu().double(), y.cpu().double()) < 4e-5


@pytest.mark.parametrize("heads,T", [(2, 3001), (2, 150), (2, 20), (3, 129), (2, 5)])
def test_fused_relative_attention(L, heads, T):
    """The text encoder's windowed relative-position attention in one kernel (reference attentions.py:230-267): scores of keys within
    +-10 of the query get the rel-k bias; the normalised band comes back for the rel-v projection.  float64 torch restatement."""
    g = torch.Generator().manual_seed(29)
    D, W = 96, 10
    q = torch.randn(heads * D, T, generator=g) * 0.3
    k = torch.randn(heads * D, T, generator=g)
    v = torch.randn(T, heads * D, generator=g)
    bv = torch.randn(heads * D, generator=g) * 0.1
    rel = torch.randn(heads, 2 * W + 1, T, generator=g) * 2.0
    qh = q.double().view(heads, D, T); kh = k.double().view(heads, D, T); vh = v.double().view(T, heads, D).permute(1, 0, 2)
    sc = torch.einsum("hdq,hdk->hqk", qh, kh)
    qi = torch.arange(T)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        sc[:, qi[ok], ki[ok]] += rel.double()[:, r, qi[ok]]
    p = torch.softmax(sc, dim=-1)
    ref = torch.einsum("hqk,hkd->hdq", p, vh).reshape(heads * D, T) + bv.double()[:, None]
    pb_ref = torch.zeros(heads, 2 * W + 1, T, dtype=torch.float64)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        pb_ref[:, r, qi[ok]] = p[:, qi[ok], ki[ok]]
    out = torch.empty(heads * D, T, device="cuda")
    pb = torch.full((heads, 2 * W + 1, T), 7.0, device="cuda")            # every entry must be written (zeros outside the sequence)
    qd, kd, vd, bd, rd = dev(q), dev(k), dev(v), dev(bv), dev(rel)
    L.check(L.lib.rvc_op_attention_rel(None, L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(bd), L.ptr(rd), L.ptr(pb), L.ptr(out), heads, T, None, None))
    torch.cuda.synchronize()
    assert rel_err(out.cpu().double(), ref) < 2e-5
    assert (pb.cpu().double() - pb_ref).abs().max() < 2e-6
    # both relative-position projections inside the kernel: rel = Q . E_k (heads share the table), out += P_band . E_v
    ek = torch.randn(2 * W + 1, D, generator=g) * 0.5
    ev = torch.randn(2 * W + 1, D, generator=g) * 0.5
    rel2 = torch.einsum("rd,hdq->hrq", ek.double(), qh)
    sc2 = torch.einsum("hdq,hdk->hqk", qh, kh)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        sc2[:, qi[ok], ki[ok]] += rel2[:, r, qi[ok]]
    p2 = torch.softmax(sc2, dim=-1)
    ref2 = torch.einsum("hqk,hkd->hdq", p2, vh)
    for r in range(2 * W + 1):
        ki = qi + r - W
        ok = (ki >= 0) & (ki < T)
        ref2[:, :, qi[ok]] += p2[:, qi[ok], ki[ok]][:, None, :] * ev.double()[r][None, :, None]
    ref2 = ref2.reshape(heads * D, T) + bv.double()[:, None]
    out2 = torch.empty(heads * D, T, device="cuda")
    ekd, evd = dev(ek), dev(ev)
    L.check(L.lib.rvc_op_attention_rel(None, L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(bd), None, None, L.ptr(out2), heads, T, L.ptr(ekd), L.ptr(evd)))
    torch.cuda.synchronize()
    assert rel_err(out2.cpu().double(), ref2) < 2e-5


def test_layernorm_channels(L):
    g = torch.Generator().manual_seed(19)
    x, r = torch.randn(192, 333, generator=g), torch.randn(192, 333, generator=g)
    ga, be = torch.rand(192, generator=g) + 0.5, torch.randn(192, generator=g)
    ref = F.layer_norm((x + r).t(), (192,), ga, be, 1e-5).t()
    y, xd, rd, gd, bd = torch.empty(192, 333, device="cuda"), dev(x), dev(r), dev(ga), dev(be)
    L.check(L.lib.rvc_op_layernorm_c(None, L.ptr(xd), L.ptr(rd), L.ptr(gd), L.ptr(bd), L.ptr(y), 192, 333))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("C,T", [(192, 333), (768, 1599), (512, 40), (1024, 77), (16, 5)])
def test_layernorm_channels_with_image_output(L, C, T):
    """layernorm_c_split_kernel: fp32 rows and the bf16 hi / lo image of the same values (the three register-caching variants: C <= 256, <= 768, any)."""
    g = torch.Generator().manual_seed(23 + C)
    x = torch.randn(C, T, generator=g) * 2.0 + 0.5
    ga, be = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(x.double().t(), (C,), ga.double(), be.double(), 1e-5).t()
    y = torch.empty(C, T, device="cuda"); yi = torch.empty(C, T, device="cuda")
    xd, gd, bd = dev(x), dev(ga), dev(be)
    L.check(L.lib.rvc_op_layernorm_c_split(None, L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(y), L.ptr(yi), C, T))
    torch.cuda.synchronize()
    assert rel_err(y.cpu().double(), ref) < 1e-5
    assert (yi - y).abs().max().item() <= 2.0 ** -15 * ref.abs().max().item()      # hi + lo of the fp32 result


def test_sine_source_phase_is_exact_on_a_long_segment(L):
    """41 s segment: the running phase (cycles) must equal torch's sample for sample.  The wrap detector of SineGen reacts to single
    ulps of the interpolated frame phase; with `scale * i - floor` contracted into an fma the device made ~8000 different (integer)
    wrap decisions over this segment and the fp32 sine argument lost 3e-3 of accuracy - found on a 45 s clip, 54 LSB end to end."""
    from comfy_rvc_amd import synthetic as S
    T, upp, sr = 4100, 400, 40000
    f0 = torch.from_numpy(S.designed_f0(T, seed=0)).view(1, T)
    f = f0[:, None].transpose(1, 2)
    rad = (f / sr) % 1
    tmp = torch.cumsum(rad, 1); tmp *= upp
    tmpi = F.interpolate(tmp.transpose(2, 1), scale_factor=float(upp), mode="linear", align_corners=True).transpose(2, 1)
    radu = F.interpolate(rad.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    tm1 = tmpi % 1
    shift = torch.zeros_like(radu); shift[:, 1:, :] = ((tm1[:, 1:, :] - tm1[:, :-1, :]) < 0) * -1.0
    c = torch.cumsum(radu + shift, dim=1)[0, :, 0]
    N = T * upp
    har, sine, ph = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    fd, nd = dev(f0.view(-1)), torch.zeros(N, device="cuda")
    L.check(L.lib.rvc_op_sine_source(None, L.ptr(fd), L.ptr(nd), L.ptr(har), L.ptr(sine), T, upp, float(sr), 0.9, 0.01, None, None, L.ptr(ph)))
    torch.cuda.synchronize()
    assert torch.equal(ph.cpu(), c)


@pytest.mark.parametrize("upp,sr,T", [(400, 40000, 320), (480, 48000, 200)])
def test_sine_source_matches_oracle(L, upp, sr, T):
    from oracle import nets
    from comfy_rvc_amd import synthetic as S
    f0 = torch.from_numpy(S.designed_f0(T, seed=0)).view(1, T)
    noise = torch.randn(1, T * upp, 1, generator=torch.Generator().manual_seed(3))
    sd = {"dec.m_source.l_linear.weight": torch.tensor([[0.9]]), "dec.m_source.l_linear.bias": torch.tensor([0.01])}
    taps = {}
    ref = nets.sine_source(sd, f0, upp, sr, noise, taps)[0, :, 0]
    har = torch.empty(T * upp, device="cuda")
    sine = torch.empty(T * upp, device="cuda")
    fd, nd = dev(f0.view(-1)), dev(noise.view(-1))
    L.check(L.lib.rvc_op_sine_source(None, L.ptr(fd), L.ptr(nd), L.ptr(har), L.ptr(sine), T, upp, float(sr), 0.9, 0.01, None, None, None))
    # 1e-3 of the sine amplitude (0.1); the phase itself is an fp64 running sum on both sides
    assert float((sine.cpu() - taps["sine_waves"][0, :, 0]).abs().max()) < 1e-4
    assert float((har.cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("n,dtype,t_pad", [(48000, np.float32, 16000), (160001, np.float64, 16000), (5003, np.float32, 800), (700000, np.float32, 16000),
                                           (16000, np.float32, 16000), (1920, np.float32, 16000)])     # pad wider than the clip: repeated reflection
def test_preprocess_filtfilt_pad_rms(L, n, dtype, t_pad):
    """rvc_preprocess (overlap-discard float64 IIR on the device) against scipy.signal.filtfilt + np.pad + the oracle's framed RMS.
    The filter amplifies float64 rounding noise to ~4e-8 of full scale (a literal Python transcription of scipy's loop differs
    from scipy by that much), so that is the agreement any re-ordered evaluation can reach: about one float32 ulp."""
    from scipy import signal
    from comfy_rvc_amd.vc_infer_pipeline import _AH, _BH, _ZI, ah, bh
    rng = np.random.default_rng(n)
    t = np.arange(n) / 16000.0
    x = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.05 * rng.standard_normal(n) + 0.2).astype(dtype)     # DC offset: the high-pass has work to do
    ref = signal.filtfilt(bh, ah, x)
    xd = torch.from_numpy(x).cuda()
    filt = torch.empty(n, dtype=torch.float64, device="cuda")
    padded = torch.empty(n + 2 * t_pad, dtype=torch.float32, device="cuda")
    n1 = n // 8000 + 1
    rms1 = torch.empty(n1, dtype=torch.float64, device="cuda")
    from comfy_rvc_amd.vc_infer_pipeline import _SOS, _SOS_ZI
    L.check(L.lib.rvc_preprocess(None, L.ptr(xd), int(dtype == np.float64), n, L.ptr(_BH), L.ptr(_AH), L.ptr(_ZI), t_pad, L.ptr(filt),
                                 L.ptr(padded), L.ptr(rms1), n1, L.ptr(_SOS), L.ptr(_SOS_ZI)))
    torch.cuda.synchronize()
    got = filt.cpu().numpy()
    # the transfer-function form (overlap-discard kernels, no sections given) is still there and agrees with both
    filt_tf = torch.empty(n, dtype=torch.float64, device="cuda")
    L.check(L.lib.rvc_preprocess(None, L.ptr(xd), int(dtype == np.float64), n, L.ptr(_BH), L.ptr(_AH), L.ptr(_ZI), t_pad, L.ptr(filt_tf),
                                 None, None, 0, None, None))
    torch.cuda.synchronize()
    assert np.abs(filt_tf.cpu().numpy() - ref).max() <= 3e-7 * np.abs(ref).max()
    assert np.abs(got - signal.sosfiltfilt(_SOS, x, padtype="odd", padlen=18)).max() <= 1e-11 * np.abs(ref).max()     # the cascade itself: float64-exact
    assert np.abs(got - ref).max() <= 3e-7 * np.abs(ref).max()
    ref_pad = np.pad(ref, (t_pad, t_pad), mode="reflect").astype(np.float32)
    gp = padded.cpu().numpy()
    assert np.array_equal(gp, np.pad(got, (t_pad, t_pad), mode="reflect").astype(np.float32))      # padding + cast are exact
    assert np.abs(gp - ref_pad).max() <= 4e-7 * np.abs(ref_pad).max()
    yp = np.pad(ref, 8000, mode="constant")
    cols = 8000 * np.arange(n1)[None, :] + np.arange(16000)[:, None]
    ref_rms = np.sqrt(np.mean(np.abs(yp[cols]) ** 2, axis=0))
    assert np.allclose(rms1.cpu().numpy(), ref_rms, rtol=1e-6, atol=0)


@pytest.mark.parametrize("orig,target,n", [(44100, 16000, 30000), (48000, 16000, 4801), (40000, 48000, 20000), (16000, 44100, 7000), (44100, 16000, 37)])
def test_resample_kernel_matches_polyphase_definition(L, orig, target, n):
    """rvc_resample against the float64 definition y[n] = sum_m x[m] h[m U - n D + half] (zero extension), output length ceil(n t / o);
    a pass-band sine keeps its amplitude and phase, a tone above the new Nyquist frequency is removed."""
    from comfy_rvc_amd.lib.audio import design_resample_filter, resample_audio
    h, half, up, down = design_resample_filter(orig, target)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n).astype(np.float32)
    y = resample_audio(x, orig, target)
    n_out = int(np.ceil(n * target / orig))
    assert y.shape == (n_out,) and y.dtype == np.float32
    ref = np.zeros(n_out)
    for i in range(n_out):
        c = i * down
        m0, m1 = max(0, -((half - c) // up)), min(n - 1, (c + half) // up)
        m = np.arange(m0, m1 + 1)
        ref[i] = np.dot(x[m].astype(np.float64), h[m * up - c + half])
    assert np.abs(y - ref).max() < 1e-6 * max(1.0, np.abs(ref).max())
    if n > 10000:
        t_in, t_out = np.arange(n) / orig, np.arange(n_out) / target
        f_pass, f_stop = 0.4 * min(orig, target) / 2, 0.5 * (min(orig, target) / 2 + max(orig, target) / 2)
        yp = resample_audio(np.sin(2 * np.pi * f_pass * t_in).astype(np.float32), orig, target)
        mid = slice(n_out // 4, 3 * n_out // 4)
        assert np.abs(yp[mid] - np.sin(2 * np.pi * f_pass * t_out[mid])).max() < 2e-5
        if target < orig:
            ys = resample_audio(np.sin(2 * np.pi * f_stop * t_in).astype(np.float32), orig, target)
            assert np.abs(ys[mid]).max() < 1e-5
    # multi-channel input is resampled along the last axis, like librosa
    x2 = np.stack([x, -0.5 * x])
    y2 = resample_audio(x2, orig, target)
    assert y2.shape == (2, n_out) and np.array_equal(y2[0], y) and np.allclose(y2[1], -0.5 * y, atol=1e-7)


@pytest.mark.parametrize("Cc,k,d,T,scale,accum", [(128, 11, 5, 70003, 1.0 / 3, True), (128, 7, 3, 66001, 1.0, False), (128, 3, 1, 80000, 1.0, False),
                                                  (64, 11, 1, 131000, 1.0, False), (64, 7, 5, 140001, 1.0 / 3, True), (64, 3, 3, 140000, 1.0, False),
                                                  (192, 7, 1, 60000, 1.0, False), (256, 11, 3, 32000, 1.0, True), (256, 3, 5, 30001, 1.0, False)])
@pytest.mark.parametrize("arith", [0, 1])
def test_split_resident_resblock_pair(L, pair_arith, arith, Cc, k, d, T, scale, accum):
    """One ResBlock1 pair as the wide generator stages run it: conv1 writes its output as the hi / lo image (split-resident), conv2
    stages that image by DMA - both on the persistent software-pipelined kernel (conv_x3q.hip; conv_x3p.hip where the persistent kernel declines; channel counts from three
    chunks up, every kernel size of the generator, sequence ends inside / at tile borders) - against fp64 torch, in both pair arithmetics (rvc_set_pair_arithmetic):
    0 = bf16x3, 1 = fp16x2.  fp16x2 is held to the SAME 2e-5 against fp64 with the weights rounded to fp16 (that rounding is the mode's whole definition; the
    activations keep 22 bits) and to 6e-4 against the exact weights (2^-12 per weight)."""
    pair_arith(arith)
    g = torch.Generator().manual_seed(2000 + 37 * k + d + Cc)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    xd = x.double()
    y0 = torch.randn(Cc, T, generator=g)

    def pair_ref(wa, wb):
        h = F.conv1d(F.leaky_relu(xd, 0.1)[None], wa.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
        r = (F.conv1d(F.leaky_relu(h, 0.1), wb.double(), b2.double(), padding=(k - 1) // 2)[0] + xd) * scale
        return r + y0 if accum else r
    y, xg = dev(y0), dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    # what the library will do for this pair at this length (192 channels: no 128-row tiling -> the per-tile kernel, which only knows bf16x3)
    h2 = L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == 1
    assert h2 == (arith == 1 and Cc != 192), (arith, Cc, h2)
    ref = pair_ref(w1.half().float(), w2.half().float()) if h2 else pair_ref(w1, w2)
    try:
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), scale, int(accum)))
    except L.RvcHipError as e:
        if "not eligible" in str(e):
            pytest.skip("the tile heuristics keep this shape off the split-resident paths")
        raise
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs()
    assert rel_err(y.cpu().double(), ref) < 2e-5, (float(err.max()), int(err.argmax()) % T)
    for c0 in (0, 126, 254, T // 2, T - 40):                      # sequence ends and tile seams carry the same error as the interior
        assert float(err[:, c0:c0 + 40].max()) < 1e-4 * float(ref.abs().max())
    if h2:
        assert rel_err(y.cpu().double(), pair_ref(w1, w2)) < 6e-4
    bad = x3p_check_count(L)                             # (a -DRVC_X3P_CHECK build counts waits whose compile-time vmcnt was too large)
    assert bad <= 0, f"{bad} waits of the pipelined kernel with a too large compile-time count"
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("Cc,k,d,T", [(128, 7, 1, 319800), (128, 3, 5, 319800), (128, 11, 3, 319800), (64, 7, 5, 639600), (64, 11, 1, 639600), (256, 7, 3, 31980)])
def test_fp16x2_pair_whole_tensor_at_generator_lengths(L, pair_arith, Cc, k, d, T):
    """The window checks of test_persistent_resblock_pair cannot see a fault that strikes a few tiles out of a thousand, differently in every run - which is
    what round 6's first fp16x2 build did at the generator's REAL stage lengths only (profiles/r6_sdwa_pk_hazard.txt: a packed fp32 add fed by an SDWA
    conversion read stale registers beside in-flight MFMAs under memory load; a latent wait-count race of the 3-tap image-in launches sat next to it).
    Here every element of three fp16x2 runs is compared: with each other (bit-identical) and with the bf16x3 result of the same pair computed on the device
    (no NaN, 6e-4: the weight rounding), at the three wide stages' true lengths."""
    g = torch.Generator().manual_seed(9000 + Cc + 13 * k + d)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    xg = dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    outs = {}
    for arith, reps in ((0, 1), (1, 3)):
        pair_arith(arith)
        assert L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == arith
        outs[arith] = []
        for _ in range(reps):
            y = torch.full((Cc, T), float("nan"), device="cuda")
            L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
            torch.cuda.synchronize()
            outs[arith].append(y)
    ref = outs[0][0]
    assert bool(torch.isfinite(ref).all())
    for y in outs[1]:
        assert bool(torch.isfinite(y).all())
        assert torch.equal(y, outs[1][0])
    assert float((outs[1][0] - ref).abs().max() / ref.abs().max()) < 6e-4
    assert x3p_check_count(L) <= 0
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("Ci,Co,T,k,act", [(512, 512, 15999, 3, "gelu"), (512, 512, 16000, 3, "gelu"), (512, 512, 7999, 2, "gelu"), (512, 512, 8002, 2, "none"),
                                           (64, 128, 1001, 3, "none"), (32, 48, 200, 3, "gelu"), (16, 32, 3, 3, "none"), (512, 512, 159999, 3, "gelu")])
def test_conv1d_stride2_on_deinterleaved_image(L, Ci, Co, T, k, act):
    """HuBERT's feature-encoder layers 1 .. 6 (Conv1d k = 3 / 2, stride 2, no padding, GELU; modeling_hubert.py conv_layers through lib/infer_pack/loaders.py:55-61) on the
    split-resident GEMM: the input as a de-interleaved (even | odd positions) bf16 hi / lo image, a tap = a row offset, the exact-erf GELU in the epilogue; result as fp32
    rows and as the de-interleaved image of the next layer - both against fp64 torch, odd and even lengths, the unwritten parts of the input image poisoned with NaN."""
    g = torch.Generator().manual_seed(Ci + Co + T + k)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / np.sqrt(Ci * k); b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv1d(x.double()[None], w.double(), b.double(), stride=2)[0]
    if act == "gelu":
        ref = F.gelu(ref)
    Tout = ref.shape[1]
    assert Tout == (T - k) // 2 + 1
    xg = dev(x)
    y, yi = torch.full((Co, Tout), float("nan")).cuda(), torch.full((Co, Tout), float("nan")).cuda()
    L.check(L.lib.rvc_op_conv1d_s2_split(None, L.ptr(xg), L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), L.ptr(y), L.ptr(yi), Ci, Co, T, k, ACT[act]))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(yi).all())
    assert rel_err(y.cpu().double(), ref) < 2e-5
    assert rel_err(yi.cpu().double(), ref) < 2e-5                  # (the image keeps hi + lo = 16 bits of mantissa: 4e-6 of each value)
    err = (y.cpu().double() - ref).abs()
    assert float(err[:, :8].max()) < 1e-4 * float(ref.abs().max()) and float(err[:, -8:].max()) < 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("Cc,k,T,scale,accum", [(32, 3, 270003, 1.0 / 3, True), (32, 3, 2 * 256 * 510, 1.0, False), (32, 7, 300001, 1.0 / 3, True), (32, 7, 2 * 256 * 506 + 2, 1.0 / 3, False),
                                                (32, 11, 262144, 1.0 / 3, True), (32, 11, 2 * 256 * 502 + 1, 1.0, False), (32, 11, 1279200, 1.0 / 3, True),
                                                # (32 channels: from two rounds of the PAIR kernel's 512 - (k - 1)-column tiles on - below that the chain runs in bf16x3)
                                                (64, 3, 130001, 1.0 / 3, False), (64, 3, 2 * 256 * 232, 1.0, True), (64, 3, 639600, 1.0 / 3, True),
                                                (64, 7, 100001, 1.0 / 3, True), (64, 7, 2 * 256 * 184, 1.0, False), (64, 7, 639600, 1.0 / 3, True)])
def test_fused_resblock_matches_pair_chain(L, pair_arith, tmp_path, Cc, k, T, scale, accum):
    """A whole ResBlock1 of a narrow generator stage (dilations 1, 3, 5; reference lib/infer_pack/modules.py:295-308) in ONE launch of conv_rb3_kernel.
    32 channels (3 / 7 / 11 taps): bit-identical to the chain of three fused-pair launches (conv_rbh_kernel) it replaces - same unit order, same accumulator
    initial values - on every element, at lengths with ragged last tiles.  64 channels (3 taps: the pairs would run on conv_x3pf_kernel in bf16x3; 7 taps: on conv_x3q_kernel as two launches each): held to the
    arithmetic's definition.  Both: within 2e-5 of fp64 torch on the fp16-rounded weights, seams and ends included, repeatable to the bit.  The stages' real
    lengths (1 279 200 and 639 600) are among the cases."""
    pair_arith(1)
    dils = (1, 3, 5)
    g = torch.Generator().manual_seed(4000 + k + Cc)
    x = torch.randn(Cc, T, generator=g)
    ws, bs = [], []
    for i in range(6):
        ws.append(torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k)); bs.append(torch.randn(Cc, generator=g) * 0.1)
    y0 = torch.randn(Cc, T, generator=g)
    xg = dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for i in range(6):
            dd = dils[i // 2] if i % 2 == 0 else 1
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(ws[i].contiguous().numpy()), L.ptr(bs[i].numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    yf = dev(y0)
    arr = (C.c_void_p * 6)(*[pl.value for pl in plans])
    ran = C.c_int(-1)
    csv_path = str(tmp_path / "launches.csv")
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(xg), T, L.ptr(yf), scale, int(accum), C.byref(ran), None, None, None))
        torch.cuda.synchronize()
        L.check(L.lib.rvc_prof_dump_csv(csv_path.encode()))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
    assert ran.value == 1
    kernels = [ln.split(",")[1] for ln in open(csv_path).read().strip().split("\n")[1:]]
    assert kernels == ["conv_rb3_kernel"], kernels
    assert bool(torch.isfinite(yf).all())
    if Cc == 32:
        # the chain of pairs, the way the generator ran it before: x -> a -> b -> y (the last pair scales and accumulates)
        ya, yb, yc = torch.empty_like(xg), torch.empty_like(xg), dev(y0)
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(ya), 1.0, 0))
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[2], plans[3], None, L.ptr(ya), T, L.ptr(yb), 1.0, 0))
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[4], plans[5], None, L.ptr(yb), T, L.ptr(yc), scale, int(accum)))
        torch.cuda.synchronize()
        assert torch.equal(yf, yc), (int((yf != yc).sum()), float((yf - yc).abs().max()))
    # a second run gives the same bits (the tile walk has no run-to-run freedom)
    yf2 = dev(y0)
    L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(xg), T, L.ptr(yf2), scale, int(accum), C.byref(ran), None, None, None))
    torch.cuda.synchronize()
    assert torch.equal(yf, yf2)
    if T <= 401000:                                                                   # fp64 torch on the fp16-rounded weights (the arithmetic's definition)
        h = x.double()
        for i in range(3):
            w1, w2 = ws[2 * i].half().double(), ws[2 * i + 1].half().double()
            t = F.conv1d(F.leaky_relu(h, 0.1)[None], w1, bs[2 * i].double(), padding=(k - 1) // 2 * dils[i], dilation=dils[i])
            h = h + F.conv1d(F.leaky_relu(t, 0.1), w2, bs[2 * i + 1].double(), padding=(k - 1) // 2)[0]
        ref = h * scale + (y0.double() if accum else 0.0)
        err = (yf.cpu().double() - ref).abs()
        assert rel_err(yf.cpu().double(), ref) < 2e-5, (float(err.max()), int(err.argmax()) % T)
        NO = (512 if Cc == 32 else 256) - 24 * ((k - 1) // 2)                             # columns stored per tile
        for c0 in (0, NO - 20, 7 * NO - 20, 256 * NO - 20, T - 40):                     # sequence ends and tile seams carry the same error as the interior
            assert float(err[:, c0:c0 + 40].max()) < 1e-4 * float(ref.abs().max())
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


def test_fused_resblock_folds_the_noise_branch(L, pair_arith):
    """The last stage's noise branch (reference models.py GeneratorNSF.forward: x = ups(x) + noise_convs[-1](har), a Conv1d(1, 32, 1)) folded into the ResBlock's
    read of x: the same result as the ResBlock of the summed tensor (the sum formed in fp64 and rounded once differs from the kernel's fmaf + add by an ulp of x
    here and there: 1e-6 of the output's scale)."""
    pair_arith(1)
    Cc, k, T = 32, 7, 300007
    g = torch.Generator().manual_seed(77)
    x = torch.randn(Cc, T, generator=g); har = torch.randn(T, generator=g) * 0.3
    nw = torch.randn(Cc, generator=g) * 0.5; nb = torch.randn(Cc, generator=g) * 0.1
    plans = []
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        for i in range(6):
            dd = (1, 3, 5)[i // 2] if i % 2 == 0 else 1
            w = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b = torch.randn(Cc, generator=g) * 0.1
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    arr = (C.c_void_p * 6)(*[pl.value for pl in plans])
    ran = C.c_int(-1)
    xs = dev((x.double() + nw.double()[:, None] * har.double()[None, :] + nb.double()[:, None]).float())
    xg, hg, wg, bg = dev(x), dev(har), dev(nw), dev(nb)
    y1, y2 = torch.zeros(Cc, T).cuda(), torch.zeros(Cc, T).cuda()
    L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(xs), T, L.ptr(y1), 1.0 / 3, 0, C.byref(ran), None, None, None))
    assert ran.value == 1
    L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(xg), T, L.ptr(y2), 1.0 / 3, 0, C.byref(ran), L.ptr(hg), L.ptr(wg), L.ptr(bg)))
    torch.cuda.synchronize()
    assert ran.value == 1
    d = (y1 - y2).abs()
    assert float(d.max()) < 2e-6 * float(y1.abs().max()), (float(d.max()), int(d.argmax()) % T)
    assert float(d[:, :600].max()) < 2e-6 * float(y1.abs().max()) and float(d[:, -600:].max()) < 2e-6 * float(y1.abs().max())
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


def test_fused_resblock_declines_what_it_cannot_run(L, pair_arith):
    """Short sequences (fewer than two rounds of tiles), the bf16x3 pair arithmetic and unequal kernel sizes are left to the pair kernels: ran = 0, y untouched."""
    Cc, k, T = 32, 7, 50000
    g = torch.Generator().manual_seed(9)
    plans = []
    L.check(L.lib.rvc_set_conv_precision(2))
    try:
        for i in range(6):
            dd = (1, 3, 5)[i // 2] if i % 2 == 0 else 1
            w = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k)
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), None, Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    arr = (C.c_void_p * 6)(*[pl.value for pl in plans])
    ran = C.c_int(-1)
    for arith, TT in ((1, T), (0, 300000)):
        pair_arith(arith)
        x, y = dev(torch.randn(Cc, TT, generator=g)), torch.full((Cc, TT), 7.0).cuda()
        L.check(L.lib.rvc_conv1d_plan_resblock_run(arr, None, L.ptr(x), TT, L.ptr(y), 1.0, 0, C.byref(ran), None, None, None))
        torch.cuda.synchronize()
        assert ran.value == 0 and bool((y == 7.0).all())
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("case", ["tiny", "huge"])
def test_fp16x2_is_not_offered_to_layers_outside_fp16s_range(L, pair_arith, case):
    """A layer whose weights are all below 2^-10 (fp16 would keep only a few bits of them) or reach beyond 60000 (fp16 overflows at 65504) gets no fp16 image:
    the pair keeps bf16x3 whatever rvc_set_pair_arithmetic says, and its accuracy."""
    pair_arith(1)
    Cc, k, d, T = 128, 7, 1, 70001
    g = torch.Generator().manual_seed(5)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.zeros(Cc)
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.zeros(Cc)
    if case == "tiny":
        w1 = w1 * (0.9 * 2.0 ** -10 / float(w1.abs().max()))     # every weight below 2^-10
    else:
        w1[3, 5, 2] = 7.0e4                                       # one weight beyond fp16's range
    xd = x.double()
    h = F.conv1d(F.leaky_relu(xd, 0.1)[None], w1.double(), None, padding=(k - 1) // 2 * d, dilation=d)
    branch = F.conv1d(F.leaky_relu(h, 0.1), w2.double(), None, padding=(k - 1) // 2)[0]
    y, xg = torch.empty(Cc, T, device="cuda"), dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    assert L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == 0
    L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    if case == "huge":
        assert rel_err(y.cpu().double(), xd + branch) < 2e-5
    else:
        # the branch is ~1e-3 of the skip path here: compared on its own scale, with the fp32 rounding of y = x + branch (2.4e-7 of |x| <= 5) allowed for
        err = float(((y.cpu().double() - xd) - branch).abs().max() / branch.abs().max())
        assert err < 2e-3, err
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("xscale,tol", [(0.03, 2e-5), (1e-3, 1.5e-4), (3e-5, 3e-3), (300.0, 2e-5)])
def test_fp16x2_pair_keeps_its_accuracy_at_small_and_large_activations(L, pair_arith, xscale, tol):
    """fp16x2's activation split x = hi + lo leaves lo below fp16's normal range (6.1e-5) for every |x| < 0.125: the mode relies on the matrix cores
    taking fp16 SUBNORMAL operands at full value (they do on gfx950; a flush would cost 2^-11 of every such activation, 5e-4 here).  What remains is fp16's
    absolute floor of 2^-24 = 6e-8 per element: invisible at activations of 0.03 and above (2e-5 against fp64 with the fp16-rounded weights, like bf16x3),
    5e-5 at a tensor scale of 1e-3 (the leaky ReLU's negative side is then 1e-4: hi itself is subnormal for a third of it), 1.4e-3 at 3e-5 - measured on
    MI355X; a generator stage carries 0.01 - 10.  Activations of a few hundred (the top of what a vocoder stage carries) keep the 2e-5."""
    pair_arith(1)
    Cc, k, d, T = 128, 7, 3, 70001
    g = torch.Generator().manual_seed(77)
    x = torch.randn(Cc, T, generator=g) * xscale
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1 * xscale
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1 * xscale
    xd = x.double()
    h = F.conv1d(F.leaky_relu(xd, 0.1)[None], w1.half().double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
    ref = F.conv1d(F.leaky_relu(h, 0.1), w2.half().double(), b2.double(), padding=(k - 1) // 2)[0] + xd
    branch = ref - xd                                             # the pair's own contribution (the skip path is exact)
    y, xg = torch.empty(Cc, T, device="cuda"), dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    assert L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == 1
    L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), 1.0, 0))
    torch.cuda.synchronize()
    err = float(((y.cpu().double() - xd) - branch).abs().max() / branch.abs().max())
    assert err < tol, err
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("Cc,k,d,T,scale,accum", [(128, 11, 5, 274489, 1.0 / 3, True), (128, 7, 3, 270001, 1.0, False), (128, 3, 1, 262144 + 256 * 3 + 5, 1.0, True),
                                                  (64, 11, 1, 400003, 1.0, False), (64, 7, 5, 393216 + 77, 1.0 / 3, True)])
@pytest.mark.parametrize("arith", [0, 1])
def test_persistent_resblock_pair(L, pair_arith, arith, tmp_path, Cc, k, d, T, scale, accum):
    """The split-resident ResBlock pair at lengths where the PERSISTENT kernel takes it (conv_x3q.hip: at least two rounds of resident
    workgroups - the sizes of the generator's 128- / 64-channel stages): every workgroup walks over several tiles in one stream of units, the
    residual joins the sum block by block, stores go through the LDS staging area.  The launch profile must name conv_x3q_kernel for both
    halves; values are checked against fp64 torch on windows at both ends of the sequence, at tile seams and in the middle (the whole tensor in
    fp64 on the CPU would take minutes), the -DRVC_X3P_CHECK build's wait bookkeeping must stay clean.  Both pair arithmetics (see test_split_resident_resblock_pair)."""
    pair_arith(arith)
    g = torch.Generator().manual_seed(4000 + 37 * k + d + Cc)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    y0 = torch.randn(Cc, T, generator=g)
    y, xg = dev(y0), dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    csv_path = str(tmp_path / "launches.csv")
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_conv1d_plan_pair_split_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), scale, int(accum)))
        torch.cuda.synchronize()
        L.check(L.lib.rvc_prof_dump_csv(csv_path.encode()))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
    kernels = [ln.split(",")[1] for ln in open(csv_path).read().strip().split("\n")[1:]]
    assert kernels == ["conv_x3q_kernel", "conv_x3q_kernel"], kernels
    assert L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == arith
    yc = y.cpu().double()
    halo = (k - 1) // 2 * (d + 1) + 8
    BN = 256
    wins = [(0, 1500), (T - 1500, T), (T // 2 - 700, T // 2 + 700)]
    for seam in (BN * 3, BN * 517, (T // BN) * BN):                 # tile seams: an early one, one deep inside a later round, the last (ragged) tile
        wins.append((max(seam - 300, 0), min(seam + 300, T)))
    worst = 0.0
    for lo, hi in wins:
        a0, a1 = max(lo - halo, 0), min(hi + halo, T)
        xd = x[:, a0:a1].double()
        wa, wb = (w1.half().float(), w2.half().float()) if arith == 1 else (w1, w2)
        h = F.conv1d(F.leaky_relu(xd, 0.1)[None], wa.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
        # (positions outside the sequence are the second convolution's zero padding; inside the window's margin they are simply not compared)
        ref = (F.conv1d(F.leaky_relu(h, 0.1), wb.double(), b2.double(), padding=(k - 1) // 2)[0] + xd) * scale
        o = lo - a0
        # trim the part of the window whose halo was cut by the slice (not by the true sequence ends)
        tl = 0 if a0 == 0 else halo
        tr = 0 if a1 == T else halo
        ref = ref[:, o + (tl if lo - a0 < tl else 0): o + (hi - lo)]
        got = yc[:, lo + (tl if lo - a0 < tl else 0): hi]
        if accum:
            ref = ref + y0[:, lo + (tl if lo - a0 < tl else 0): hi].double()
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 2e-5, (lo, hi, err)
    bad = x3p_check_count(L)
    assert bad <= 0, f"{bad} waits of the persistent kernel with a too large compile-time count"
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


@pytest.mark.parametrize("Cc,k,d,T,scale,accum", [(32, 11, 5, 140003, 1.0 / 3, True), (32, 7, 3, 131072, 1.0, False), (32, 3, 1, 200001, 1.0, False),
                                                  (32, 3, 5, 136100, 1.0 / 3, True), (32, 11, 1, 150000, 1.0, False),
                                                  (32, 11, 5, 140004, 1.0 / 3, True), (32, 11, 3, 131076, 1.0, False), (32, 7, 5, 128000, 1.0 / 3, True), (32, 7, 1, 130052, 1.0, False), (32, 3, 3, 160000, 1.0, False),
                                                  (64, 3, 5, 70001, 1.0 / 3, True), (64, 3, 3, 66000, 1.0, False), (64, 3, 1, 80003, 1.0, False),   # (64 channels: k = 3 only by default)
                                                  # lengths of two and more rounds of 512-column tiles: the persistent fp16x2 kernel with LDS-resident weights (conv_rbh.hip) in arithmetic 1
                                                  (32, 11, 5, 600003, 1.0 / 3, True), (32, 11, 1, 530001, 1.0, False), (32, 7, 3, 777777, 1.0, False), (32, 7, 5, 262200, 1.0 / 3, True),
                                                  (32, 3, 1, 700000, 1.0, False), (32, 3, 5, 513 * 510 + 7, 1.0 / 3, True)])
@pytest.mark.parametrize("arith", [0, 1])
def test_fused_resblock_pair(L, pair_arith, arith, tmp_path, Cc, k, d, T, scale, accum):
    """One ResBlock1 pair of the generator's narrow stages in a single launch, the intermediate in LDS (zero outside the sequence like the second conv's
    padding), against fp64 torch: y = (x + c2(lrelu(c1_d(lrelu(x))))) * s [+ y].  bf16x3: conv_x3pf_kernel (32 channels: 256-column tiles; 64 channels:
    128-column tiles).  fp16x2 (arithmetic 1) at 32 channels and two or more rounds of tiles: conv_rbh_kernel - persistent workgroups, both weight sets
    resident in LDS, 512-column tiles - held to 2e-5 against fp64 with the fp16-rounded weights and 6e-4 against the exact ones; shorter sequences and 64
    channels keep conv_x3pf_kernel in either arithmetic.  Lengths with and without a multiple of 4."""
    pair_arith(arith)
    g = torch.Generator().manual_seed(1000 + 37 * k + d)
    x = torch.randn(Cc, T, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, k, generator=g) / np.sqrt(Cc * k); b2 = torch.randn(Cc, generator=g) * 0.1
    xd = x.double()
    y0 = torch.randn(Cc, T, generator=g)

    def pair_ref(wa, wb):
        h = F.conv1d(F.leaky_relu(xd, 0.1)[None], wa.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
        r = (F.conv1d(F.leaky_relu(h, 0.1), wb.double(), b2.double(), padding=(k - 1) // 2)[0] + xd) * scale
        return r + y0 if accum else r
    y, xg = dev(y0), dev(x)
    L.check(L.lib.rvc_set_conv_precision(2))
    plans = []
    try:
        for w, b, dd in ((w1, b1, d), (w2, b2, 1)):
            pl = C.c_void_p()
            L.check(L.lib.rvc_conv1d_plan_create(L.ptr(w.contiguous().numpy()), L.ptr(b.numpy()), Cc, Cc, k, 1, (k - 1) // 2 * dd, dd, 1, C.byref(pl)))
            plans.append(pl)
    finally:
        L.check(L.lib.rvc_set_conv_precision(1))
    h2 = L.lib.rvc_conv1d_plan_pair_arithmetic(plans[0], plans[1], T) == 1
    assert h2 == (arith == 1 and Cc == 32 and T >= 2 * 256 * (512 - (k - 1))), (arith, Cc, T, h2)       # (256 CUs: two rounds of tiles)
    ref = pair_ref(w1.half().float(), w2.half().float()) if h2 else pair_ref(w1, w2)
    csv_path = str(tmp_path / "launches.csv")
    try:
        L.check(L.lib.rvc_prof_enable(1))
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y), scale, int(accum)))
        torch.cuda.synchronize()
        L.check(L.lib.rvc_prof_dump_csv(csv_path.encode()))
    finally:
        L.check(L.lib.rvc_prof_enable(0))
    kernels = [ln.split(",")[1] for ln in open(csv_path).read().strip().split("\n")[1:]]
    assert kernels == (["conv_rbh_kernel"] if h2 else ["conv_x3pf_kernel"]), kernels
    err = (y.cpu().double() - ref).abs()
    assert rel_err(y.cpu().double(), ref) < 2e-5, (float(err.max()), int(err.argmax()) % T)
    # sequence ends and tile seams (tiles of 256 / 512 - (k - 1) columns) carry the same error as the interior
    NO = (512 if h2 else (256 if Cc == 32 else 128)) - (k - 1)
    TN = 256 - 4 * ((k - 1 + 3) // 4)
    for c0 in (0, NO - 2, 7 * NO - 3, 257 * NO - 20, TN - 20, 9 * TN - 20, T - 40):
        if 0 <= c0 < T - 40 or c0 == T - 40:
            assert float(err[:, c0:c0 + 40].max()) < 1e-4 * float(ref.abs().max())
    if h2:
        assert rel_err(y.cpu().double(), pair_ref(w1, w2)) < 6e-4
        y2 = dev(y0)                                              # a second run is bit-identical (fixed summation order, no races between the phases)
        L.check(L.lib.rvc_conv1d_plan_pair_run(plans[0], plans[1], None, L.ptr(xg), T, L.ptr(y2), scale, int(accum)))
        torch.cuda.synchronize()
        assert torch.equal(y2, y)
    for pl in plans:
        L.lib.rvc_conv1d_plan_destroy(pl)


# ------------------------------------------------------------------ split-resident GEMM (csrc/conv_x3s.hip)
GEMM_SPLIT = [
    # Ci, Co, T, act, res, act_before_res, out_scale, ksplit, am, an
    (768, 3072, 1599, "gelu", False, 0, 1.0, 0, 0, 0),       # HuBERT FFN1 (automatic tile: 128 x 64), GELU before the split
    (3072, 768, 1599, "none", True, 0, 1.0, 0, 0, 0),        # FFN2: automatic K split, reduced inside the launch
    (768, 2304, 1599, "none", False, 0, 1.0, 0, 0, 0),       # q / k / v
    (768, 768, 1599, "none", True, 0, 1.0, 4, 2, 2),         # out projection, forced 4-way split on 128 x 128 tiles
    (768, 768, 333, "lrelu", True, 1, 0.5, 2, 1, 2),         # act(Wx + b) + res, scaled; 64 x 128 tiles; ragged columns
    (192, 576, 3198, "relu", False, 0, 1.0, 0, 1, 1),        # text-encoder size, 64 x 64 tiles
    (512, 360, 100, "none", False, 0, 1.0, 2, 2, 1),         # Co not a multiple of the tile (360 = 2 x 128 + 104), few columns
    (64, 32, 70, "gelu", True, 1, 1.0, 0, 0, 0),             # the smallest eligible layer
]


@pytest.mark.parametrize("case", GEMM_SPLIT, ids=[f"g{i}" for i in range(len(GEMM_SPLIT))])
def test_gemm_split_resident(L, case):
    """Y = act(W X + b [+ R]) with X handed over as the bf16 hi / lo image: fp32 rows AND the split output image (read back as hi + lo)
    against torch float64; 3-term bf16 split => ~1e-5 relative.  The K split is summed in slice order by whichever slice arrives last:
    repeated launches are bit-identical."""
    Ci, Co, T, act, res, abr, scale, ks, am, an = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, T, generator=g) if res else None
    v = w.double() @ x.double() + b.double()[:, None]
    if res and not abr:
        v = v + r.double()
    v = _act(v, act, 0.1)
    if res and abr:
        v = v + r.double()
    ref = (v * scale).numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    y = torch.full((Co, T), float("nan"), device="cuda")
    ys = torch.full((Co, T), float("nan"), device="cuda") if Co % 16 == 0 else None
    wn, bn = w.contiguous().numpy(), b.contiguous().numpy()

    def run(yo, yso):
        L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(wn), L.ptr(bn), L.ptr(rd) if res else None, L.ptr(yo) if yo is not None else None,
                                        L.ptr(yso) if yso is not None else None, Ci, Co, T, ACT[act], 0.1, abr, scale, ks, am, an, 1, 1))
    run(y, ys)
    assert rel_err(y.cpu().numpy(), ref) < 2e-5
    if ys is not None:
        assert rel_err(ys.cpu().numpy(), ref) < 2e-5            # hi + lo carries 16 mantissa bits: 2^-17 relative per element
    y2 = torch.empty_like(y)
    for _ in range(3):
        run(y2, None)
        assert torch.equal(y2, y)                               # whoever reduces, the sum order is the slice order


def test_gemm_split_resident_under_load(L):
    """The in-launch K-split hand-off (write-through slabs, ticket, last arriver reduces) while another stream keeps the chip busy with
    streaming traffic: 20 launches of the 3072 -> 768 GEMM must all equal the un-split result bit for bit ... up to the sum order, i.e.
    equal each other exactly and the float64 reference to 2e-5."""
    Ci, Co, T = 3072, 768, 1599
    g = torch.Generator().manual_seed(11)
    x = torch.randn(Ci, T, generator=g); w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci); b = torch.randn(Co, generator=g) * 0.1
    ref = (w.double() @ x.double() + b.double()[:, None]).numpy()
    xd = dev(x)
    wn, bn = w.contiguous().numpy(), b.contiguous().numpy()
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device="cuda")
    outs = []
    with torch.cuda.stream(side):
        for _ in range(40):
            big.mul_(1.0001)                                    # 512 MB of HBM traffic per pass on the other stream
    for i in range(20):
        y = torch.full((Co, T), float("nan"), device="cuda")
        L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(wn), L.ptr(bn), None, L.ptr(y), None, Ci, Co, T, 0, 0.0, 0, 1.0, 3 + (i % 2), 2, 1, 1, 1))
        outs.append(y)
    torch.cuda.synchronize()
    assert rel_err(outs[0].cpu().numpy(), ref) < 2e-5 and rel_err(outs[1].cpu().numpy(), ref) < 2e-5
    for i in range(2, 20):
        assert torch.equal(outs[i], outs[i % 2])


def _x3s_modes(L, fn):
    """fn() under the LDS-ring reduction loop (mode 1) and the register-direct one (mode 2) of csrc/conv_x3s.hip; the process-wide default is restored."""
    if not L.has_experiments:
        pytest.skip("the register-direct loop exists in -DRVC_EXPERIMENTS builds only (RVC_HIP_LIB=<variant>); the product library runs the LDS ring")
    outs = []
    try:
        for mode in (1, 2):
            L.check(L.lib.rvc_debug_set_x3s_mode(mode))
            outs.append(fn())
    finally:
        L.check(L.lib.rvc_debug_set_x3s_mode(0))
    return outs


@pytest.mark.parametrize("case", GEMM_SPLIT, ids=[f"g{i}" for i in range(len(GEMM_SPLIT))])
def test_gemm_split_register_direct_equals_lds_ring(L, case):
    """The register-direct reduction loop (operands loaded straight into the MFMA registers: no LDS, no barrier) against the LDS-ring loop on every
    GEMM case above: same units, same MFMA order per accumulator, same K-split reduction => BIT-identical fp32 rows and split images; and against
    float64 directly."""
    Ci, Co, T, act, res, abr, scale, ks, am, an = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci, generator=g) / np.sqrt(Ci)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, T, generator=g) if res else None
    v = w.double() @ x.double() + b.double()[:, None]
    if res and not abr:
        v = v + r.double()
    v = _act(v, act, 0.1)
    if res and abr:
        v = v + r.double()
    ref = (v * scale).numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    wn, bn = w.contiguous().numpy(), b.contiguous().numpy()

    def run():
        y = torch.full((Co, T), float("nan"), device="cuda")
        ys = torch.full((Co, T), float("nan"), device="cuda") if Co % 16 == 0 else None
        L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(wn), L.ptr(bn), L.ptr(rd) if res else None, L.ptr(y), L.ptr(ys) if ys is not None else None,
                                        Ci, Co, T, ACT[act], 0.1, abr, scale, ks, am, an, 1, 1))
        torch.cuda.synchronize()
        return y, ys
    (y1, ys1), (y2, ys2) = _x3s_modes(L, run)
    assert rel_err(y2.cpu().numpy(), ref) < 2e-5
    assert torch.equal(y1, y2) and (ys1 is None or torch.equal(ys1, ys2))
    for dm, dn in ((1, 1), (2, 1), (1, 2), (2, 2)):           # every register tile of the direct kernel, un-split and split
        for dks in (1, 2):
            if (Ci // 16) % dks:
                continue

            def run_t():
                y = torch.full((Co, T), float("nan"), device="cuda")
                L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(wn), L.ptr(bn), L.ptr(rd) if res else None, L.ptr(y), None,
                                                Ci, Co, T, ACT[act], 0.1, abr, scale, dks, dm, dn, 1, 1))
                torch.cuda.synchronize()
                return y
            a, bb = _x3s_modes(L, run_t)
            assert torch.equal(a, bb), (dm, dn, dks)
            assert rel_err(bb.cpu().numpy(), ref) < 2e-5


CONV_SPLIT_TAPS = [
    # Ci, Co, T, k, dil, act, res, ksplit, am, an
    (192, 768, 3198, 3, 1, "relu", False, 0, 0, 0),        # text-encoder FFN1 (attentions.py:383-395: k = 3 "same")
    (768, 192, 3198, 3, 1, "none", True, 0, 0, 0),         # FFN2 + residual: automatic K split over (chunk, tap) units
    (192, 384, 3198, 5, 1, "none", False, 0, 0, 0),        # flow WaveNet in-layer (modules.py:184-209)
    (192, 512, 700, 7, 1, "none", False, 3, 2, 1),         # conv_pre; ragged length, forced 3-way split
    (64, 64, 515, 3, 5, "lrelu", True, 0, 1, 1),           # dilation 5: offsets -5, 0, 5
]


@pytest.mark.parametrize("case", CONV_SPLIT_TAPS, ids=[f"t{i}" for i in range(len(CONV_SPLIT_TAPS))])
def test_conv1d_taps_on_split_resident_gemm(L, case):
    """1-D "same" convolutions as the split-resident GEMM with (chunk, tap) units: a tap is a row offset into the image (im2col by address),
    the zero padding is the image's zero margin.  Against torch float64, incl. the first / last columns (the padded ones)."""
    Ci, Co, T, k, dil, act, res, ks, am, an = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / np.sqrt(Ci * k)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, T, generator=g) if res else None
    v = F.conv1d(x.double()[None], w.double(), b.double(), padding=(k - 1) // 2 * dil, dilation=dil)[0]
    if res:
        v = v + r.double()
    ref = _act(v, act, 0.1).numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    y = torch.full((Co, T), float("nan"), device="cuda")
    ys = torch.full((Co, T), float("nan"), device="cuda")
    L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(y), L.ptr(ys),
                                    Ci, Co, T, ACT[act], 0.1, 0, 1.0, ks, am, an, k, dil))
    assert rel_err(y.cpu().numpy(), ref) < 2e-5 and rel_err(ys.cpu().numpy(), ref) < 2e-5
    assert rel_err(y.cpu().numpy()[:, :16], ref[:, :16]) < 2e-5 and rel_err(y.cpu().numpy()[:, -16:], ref[:, -16:]) < 2e-5

    def run_mode():          # both reduction loops of the kernel: bit-identical (taps are row offsets in either)
        yy, yys = torch.full((Co, T), float("nan"), device="cuda"), torch.full((Co, T), float("nan"), device="cuda")
        L.check(L.lib.rvc_op_gemm_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(yy), L.ptr(yys),
                                        Ci, Co, T, ACT[act], 0.1, 0, 1.0, ks, am, an, k, dil))
        torch.cuda.synchronize()
        return yy, yys
    (a1, b1), (a2, b2) = _x3s_modes(L, run_mode)
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and rel_err(a2.cpu().numpy(), ref) < 2e-5


CONV2D_SPLIT = [
    # Ci, Co, H, W, act, res, act_before_res, ksplit, am, an
    (512, 512, 101, 4, "relu", True, 1, 0, 0, 0),          # RMVPE intermediate level (30 s clip): 404 positions, automatic K split
    (256, 256, 202, 8, "relu", True, 1, 0, 0, 0),
    (128, 128, 404, 16, "relu", False, 0, 0, 0, 0),
    (64, 64, 203, 32, "relu", True, 1, 2, 1, 2),           # odd height, forced split / tile
    (256, 128, 50, 8, "none", False, 0, 0, 0, 0),          # channel change (first block of a level)
    (64, 48, 64, 32, "relu", False, 0, 0, 1, 1),           # Co not a multiple of the tile
]


@pytest.mark.parametrize("case", CONV2D_SPLIT, ids=[f"p{i}" for i in range(len(CONV2D_SPLIT))])
def test_conv2d_3x3_on_padded_split_images(L, case):
    """3 x 3 / pad 1 convolution over padded split-resident images (row pitch W + 2: taps are constant row offsets, the pad columns and the
    margins are the zero padding) against torch float64 - every border row and column included."""
    Ci, Co, H, W, act, res, abr, ks, am, an = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)
    x = torch.randn(Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / np.sqrt(Ci * 9)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, H, W, generator=g) if res else None
    v = F.conv2d(x.double()[None], w.double(), b.double(), padding=1)[0]
    if res and not abr:
        v = v + r.double()
    v = _act(v, act, 0.1)
    if res and abr:
        v = v + r.double()
    ref = v.numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    y = torch.full((Co, H, W), float("nan"), device="cuda")
    ys = torch.full((Co, H, W), float("nan"), device="cuda") if Co % 16 == 0 else None
    L.check(L.lib.rvc_op_conv2d_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(y),
                                      L.ptr(ys) if ys is not None else None, Ci, Co, H, W, ACT[act], abr, ks, am, an))
    assert rel_err(y.cpu().numpy(), ref) < 2e-5
    if ys is not None:
        assert rel_err(ys.cpu().numpy(), ref) < 2e-5
    for sl in (np.s_[:, 0, :], np.s_[:, -1, :], np.s_[:, :, 0], np.s_[:, :, -1]):
        assert rel_err(y.cpu().numpy()[sl], ref[sl]) < 3e-5

    def run_mode():
        yy = torch.full((Co, H, W), float("nan"), device="cuda")
        L.check(L.lib.rvc_op_conv2d_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(yy),
                                          None, Ci, Co, H, W, ACT[act], abr, ks, am, an))
        torch.cuda.synchronize()
        return yy
    a1, a2 = _x3s_modes(L, run_mode)
    assert torch.equal(a1, a2) and rel_err(a2.cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize("case", [(768, 768, 1599, 128, 64, 1, 16, "gelu", True, 1), (768, 768, 77, 128, 64, 1, 16, "gelu", True, 1),
                                  (192, 96, 333, 9, 4, 1, 2, "none", False, 0), (256, 256, 500, 4, 2, 1, 4, "relu", True, 0)],
                         ids=["posconv", "posconv_short", "g2_k9", "g4_k4_even"])
def test_grouped_long_kernel_conv1d_on_split_resident_gemm(L, case):
    """Grouped stride-1 convolutions with long / even kernels on the split-resident kernel (one 64-row tile per group, tap offsets by
    formula): HuBERT's positional convolution (768 -> 768, k = 128, padding 64, 16 groups, cut to T, GELU, + residual) against torch float64."""
    Ci, Co, T, k, pad, dil, groups, act, res, abr = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 10000)
    x = torch.randn(Ci, T, generator=g)
    w = torch.randn(Co, Ci // groups, k, generator=g) / np.sqrt(Ci // groups * k)
    b = torch.randn(Co, generator=g) * 0.1
    r = torch.randn(Co, T, generator=g) if res else None
    v = F.conv1d(x.double()[None], w.double(), b.double(), padding=pad, dilation=dil, groups=groups)[0][:, :T]
    if res and not abr:
        v = v + r.double()
    v = _act(v, act, 0.1)
    if res and abr:
        v = v + r.double()
    ref = v.numpy()
    xd, rd = dev(x), (dev(r) if res else None)
    y = torch.full((Co, T), float("nan"), device="cuda")
    L.check(L.lib.rvc_op_conv1d_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(y),
                                      Ci, Co, T, k, pad, dil, groups, ACT[act], abr))
    assert rel_err(y.cpu().numpy(), ref) < 2e-5

    def run_mode():
        yy = torch.full((Co, T), float("nan"), device="cuda")
        L.check(L.lib.rvc_op_conv1d_split(None, L.ptr(xd), L.ptr(w.contiguous().numpy()), L.ptr(b.contiguous().numpy()), L.ptr(rd) if res else None, L.ptr(yy),
                                          Ci, Co, T, k, pad, dil, groups, ACT[act], abr))
        torch.cuda.synchronize()
        return yy
    a1, a2 = _x3s_modes(L, run_mode)
    assert torch.equal(a1, a2) and rel_err(a2.cpu().numpy(), ref) < 2e-5
