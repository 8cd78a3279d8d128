"""demix_mdxv3 (reference lib/karafan/inference.py:32-74): chunked, overlapped separation of a whole stereo track with an MDX23C network."""
import numpy as np
import torch


def _get(cfg, *path):
    for k in path:
        cfg = cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)
    return cfg


def demix_mdxv3(mix, model, device, config, overlap_MDX23):
    """mix [2, L] float -> {instrument: ndarray [2, L]} (or ndarray for a single target).  Zero padding, chunks of C = hop (dim_t - 1)
    samples every H = C / overlap, accumulation and the 1 / overlap normalisation happen on the device; NaNs are zeroed as upstream."""
    mix = torch.as_tensor(np.asarray(mix), dtype=torch.float32)
    S = model.num_target_instruments
    C = _get(config, "audio", "hop_length") * (_get(config, "inference", "dim_t") - 1)
    H = C // overlap_MDX23
    L = mix.shape[1]
    pad_size = H - (L - C) % H
    mix = torch.cat([torch.zeros(2, C - H), mix, torch.zeros(2, pad_size + C - H)], 1).to(model.device)
    n_chunks = (mix.shape[1] - C) // H + 1
    if hasattr(model, "demix_device"):
        # the loop, the NaN handling, the accumulation and the division run behind rvc_mdx23_demix (csrc/model_mdx23.hip) in the reference's order
        X = model.demix_device(mix, H, n_chunks, overlap_MDX23)
        X = X if S > 1 else X[0]
        est = X[..., C - H: -(pad_size + C - H)]
    else:
        chunks = mix.unfold(1, C, H).transpose(0, 1)
        X = torch.zeros(S, *mix.shape, dtype=torch.float32, device=model.device) if S > 1 else torch.zeros_like(mix)
        for cnt in range(chunks.shape[0]):
            x = model(chunks[cnt: cnt + 1].contiguous())
            x = torch.nan_to_num(x, nan=0.0, posinf=float("inf"), neginf=-float("inf"))
            X[..., cnt * H: cnt * H + C] += x[0]
        est = X[..., C - H: -(pad_size + C - H)] / overlap_MDX23
    if S > 1:
        return {k: v for k, v in zip(_get(config, "training", "instruments"), est.cpu().numpy())}
    return est.cpu().numpy()
