"""CPU oracle for the MDX23C separation network of the UVR chain (SURVEY 8 f4 / BASELINE config C5).  TEST INFRASTRUCTURE ONLY.

Restates reference lib/karafan/tfc_tdf.py:47-235 (STFT / inverse, TFC_TDF blocks, TFC_TDF_net.forward) and the chunked overlap-add
driver lib/karafan/inference.py:32-74 (demix_mdxv3) with plain torch-CPU functional ops and explicit DFT matrices instead of
torch.stft / torch.istft, so that every step has a named tensor the HIP path can be compared with.  Pinned to the real reference
module by tests/golden/mdx23c_small.npz (oracle/gen_golden.py mdx23c: the reference's own TFC_TDF_net run on a reduced configuration
with the procedural weights of comfy-rvc_amd/synthetic.py::mdx23c_state_dict).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def stft_basis(n_fft, dim_f):
    """[2 * dim_f, n_fft]: rows reim * dim_f + k = hann_periodic[n] * (cos | -sin)(2 pi k n / n_fft)  (torch.stft, onesided, no normalisation)."""
    n = np.arange(n_fft, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    k = np.arange(dim_f, dtype=np.float64)[:, None]
    ang = 2 * np.pi * k * n[None, :] / n_fft
    return np.concatenate([np.cos(ang) * w, -np.sin(ang) * w]).astype(np.float32)


def istft_basis(n_fft, dim_f):
    """[n_fft, 2 * dim_f]: windowed inverse real FFT of a one-sided spectrum whose bins >= dim_f are zero (dim_f <= n_fft / 2):
    y[n] = w[n] / N * (Re X0 + 2 sum_{k>=1} (Re Xk cos(2 pi k n / N) - Im Xk sin(2 pi k n / N)))."""
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    k = np.arange(dim_f, dtype=np.float64)[None, :]
    ang = 2 * np.pi * k * n / n_fft
    scale = np.where(k == 0, 1.0, 2.0) / n_fft
    return np.concatenate([np.cos(ang) * scale * w, -np.sin(ang) * scale * w], axis=1).astype(np.float32)


def stft(x, n_fft, hop, dim_f):
    """STFT.__call__ (tfc_tdf.py:55-64): x [C, L] -> [2 C, dim_f, T] with channel = c * 2 + (re | im), center=True (reflect pad n_fft / 2)."""
    c, L = x.shape
    xp = F.pad(x[None], (n_fft // 2, n_fft // 2), mode="reflect")[0]
    frames = xp.unfold(1, n_fft, hop)                               # [C, T, n_fft]
    spec = torch.einsum("kn,ctn->ckt", _t(stft_basis(n_fft, dim_f)), frames)     # [C, 2 dim_f, T]
    return spec.reshape(c, 2, dim_f, -1).reshape(2 * c, dim_f, -1)


def istft(x, n_fft, hop, length):
    """STFT.inverse (tfc_tdf.py:66-77) for one source: x [2 C, dim_f, T] -> [C, length]; overlap-add of windowed inverse FFTs over the
    squared-window envelope, centre-trimmed (torch.istft, center=True)."""
    c2, dim_f, T = x.shape
    c = c2 // 2
    spec = x.reshape(c, 2 * dim_f, T)
    fr = torch.einsum("nk,ckt->cnt", _t(istft_basis(n_fft, dim_f)), spec)         # [C, n_fft, T]
    full = n_fft + hop * (T - 1)
    y = torch.zeros(c, full)
    env = torch.zeros(full)
    n = torch.arange(n_fft, dtype=torch.float64)
    w2 = ((0.5 - 0.5 * torch.cos(2 * math.pi * n / n_fft)) ** 2).float()
    for t in range(T):
        y[:, t * hop: t * hop + n_fft] += fr[:, :, t]
        env[t * hop: t * hop + n_fft] += w2
    return (y / env)[:, n_fft // 2: n_fft // 2 + length]


def _norm_act(x, sd, key):
    return F.gelu(F.instance_norm(x, weight=sd[key + ".weight"], bias=sd[key + ".bias"], eps=1e-5))


def tfc_tdf(x, sd, prefix, l):
    """TFC_TDF.forward (tfc_tdf.py:137-144)."""
    for i in range(l):
        p = f"{prefix}.blocks.{i}."
        s = F.conv2d(x, sd[p + "shortcut.weight"])
        x = F.conv2d(_norm_act(x, sd, p + "tfc1.0"), sd[p + "tfc1.2.weight"], padding=1)
        h = F.linear(_norm_act(x, sd, p + "tdf.0"), sd[p + "tdf.2.weight"])
        x = x + F.linear(_norm_act(h, sd, p + "tdf.3"), sd[p + "tdf.5.weight"])
        x = F.conv2d(_norm_act(x, sd, p + "tfc2.0"), sd[p + "tfc2.2.weight"], padding=1)
        x = x + s
    return x


def forward(sd, cfg, x, taps=None):
    """TFC_TDF_net.forward (tfc_tdf.py:197-233) for one chunk: x [2, chunk] -> [S, 2, chunk]."""
    sd = {k: _t(v).float() for k, v in sd.items()}
    a, m = cfg["audio"], cfg["model"]
    k, n, l = m["num_subbands"], m["num_scales"], m["num_blocks_per_scale"]
    S = len(cfg["training"]["instruments"])
    x = _t(x).float()
    L = x.shape[-1]
    spec = stft(x, a["n_fft"], a["hop_length"], a["dim_f"])          # [4, dim_f, T]
    c, f, t = spec.shape
    mix = h = spec.reshape(c, k, f // k, t).reshape(1, c * k, f // k, t)          # cac2cws
    first = h = F.conv2d(h, sd["first_conv.weight"])
    h = h.transpose(-1, -2)
    if taps is not None:
        taps["spec"], taps["first_conv"] = spec.numpy().copy(), first[0].numpy().copy()
    enc = []
    for i in range(n):
        h = tfc_tdf(h, sd, f"encoder_blocks.{i}.tfc_tdf", l)
        if taps is not None and i == 0:
            taps["enc0"] = h[0].numpy().copy()
        enc.append(h)
        h = F.conv2d(_norm_act(h, sd, f"encoder_blocks.{i}.downscale.conv.0"), sd[f"encoder_blocks.{i}.downscale.conv.2.weight"], stride=2)
    h = tfc_tdf(h, sd, "bottleneck_block", l)
    if taps is not None:
        taps["bottleneck"] = h[0].numpy().copy()
    for i in range(n):
        h = F.conv_transpose2d(_norm_act(h, sd, f"decoder_blocks.{i}.upscale.conv.0"), sd[f"decoder_blocks.{i}.upscale.conv.2.weight"], stride=2)
        h = torch.cat([h, enc.pop()], 1)
        h = tfc_tdf(h, sd, f"decoder_blocks.{i}.tfc_tdf", l)
    h = h.transpose(-1, -2) * first
    h = F.conv2d(torch.cat([mix, h], 1), sd["final_conv.0.weight"])
    h = F.conv2d(F.gelu(h), sd["final_conv.2.weight"])
    if taps is not None:
        taps["mask_out"] = h[0].numpy().copy()
    b, cc, ff, tt = h.shape
    h = h.reshape(b, cc // k, k, ff, tt).reshape(cc // k, ff * k, tt)             # cws2cac
    h = h.reshape(S, -1, ff * k, tt)
    return torch.stack([istft(h[s], a["n_fft"], a["hop_length"], L) for s in range(S)])


def demix_mdxv3(sd, cfg, mix, overlap):
    """demix_mdxv3 (inference.py:32-74): zero-pad, chunks of C = hop (dim_t - 1) every H = C / overlap, accumulate, divide by overlap."""
    mix = _t(mix).float()
    a = cfg["audio"]
    S = len(cfg["training"]["instruments"])
    C = a["hop_length"] * (cfg["inference"]["dim_t"] - 1)
    H = C // overlap
    L = mix.shape[1]
    pad_size = H - (L - C) % H
    mix = torch.cat([torch.zeros(2, C - H), mix, torch.zeros(2, pad_size + C - H)], 1)
    chunks = mix.unfold(1, C, H).transpose(0, 1)
    X = torch.zeros(S, *mix.shape)
    with torch.no_grad():
        for cnt, ch in enumerate(chunks):
            y = forward(sd, cfg, ch)
            y[torch.isnan(y)] = 0.0
            X[..., cnt * H: cnt * H + C] += y
    est = X[..., C - H: -(pad_size + C - H)] / overlap
    return {k: v for k, v in zip(cfg["training"]["instruments"], est.numpy())}
