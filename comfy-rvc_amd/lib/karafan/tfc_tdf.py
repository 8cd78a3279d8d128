"""TFC_TDF_net (MDX23C) on the HIP path: same constructor argument (the model's config), `load_state_dict`, and `net(batch)` with
batch [B, 2, hop * (dim_t - 1)] -> [B, S, 2, chunk] as reference lib/karafan/tfc_tdf.py:147-235.  All compute - STFT, the TFC / TDF
U-Net, mask head, inverse STFT - runs behind rvc_mdx23_forward (csrc/model_mdx23.hip); this file only marshals the config, the weights
and the three host-built constants (windowed DFT matrices, hann window)."""
import ctypes as C

import numpy as np
import torch

from ... import _lib


def _get(cfg, *path):
    for k in path:
        cfg = cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)
    return cfg


def stft_basis(n_fft, dim_f):
    """[2 dim_f][n_fft]: hann_periodic[n] * (cos | -sin)(2 pi k n / n_fft) (torch.stft: one-sided, unnormalised; float64 then rounded)."""
    n = np.arange(n_fft, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    ang = 2 * np.pi * np.arange(dim_f, dtype=np.float64)[:, None] * n[None, :] / n_fft
    return np.concatenate([np.cos(ang) * w, -np.sin(ang) * w]).astype(np.float32)


def istft_basis(n_fft, dim_f):
    """[n_fft][2 dim_f]: windowed inverse real FFT of a one-sided spectrum whose bins >= dim_f are zero (torch.istft's irfft * window)."""
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    k = np.arange(dim_f, dtype=np.float64)[None, :]
    ang = 2 * np.pi * k * n / n_fft
    scale = np.where(k == 0, 1.0, 2.0) / n_fft
    return np.concatenate([np.cos(ang) * scale * w, -np.sin(ang) * scale * w], axis=1).astype(np.float32)


class TFC_TDF_net:
    def __init__(self, config, device=None):
        self.config = config
        a = lambda k: _get(config, "audio", k)     # noqa: E731
        m = lambda k: _get(config, "model", k)     # noqa: E731
        assert m("norm") == "InstanceNorm" and m("act") == "gelu" and list(m("scale")) == [2, 2], "only the shipped MDX23C recipe (InstanceNorm, GELU, 2x2 scales)"
        target = _get(config, "training", "target_instrument")
        self.num_target_instruments = 1 if target else len(_get(config, "training", "instruments"))
        self.num_subbands = m("num_subbands")
        self.n_fft, self.hop, self.dim_f = a("n_fft"), a("hop_length"), a("dim_f")
        self.dim_t = _get(config, "inference", "dim_t")
        self.chunk_size = self.hop * (self.dim_t - 1)
        if device is None or str(device) == "cpu":
            device = "cuda:0"
        self.device = torch.device(device)
        self._cfg = _lib.Mdx23Config(self.n_fft, self.hop, self.dim_f, self.dim_t, m("num_channels"), m("growth"), m("num_scales"), self.num_subbands,
                                     m("num_blocks_per_scale"), m("bottleneck_factor"), self.num_target_instruments, a("num_channels"))
        self._ctx = _lib.get_ctx(self.device.index or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib.rvc_mdx23_create(self._ctx, C.byref(self._cfg), C.byref(h)))
        self._h = h
        self._ready = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "lib", None) is not None:
            _lib.lib.rvc_mdx23_destroy(h)
            self._h = None

    def load_state_dict(self, sd, strict=True):
        n = np.arange(self.n_fft, dtype=np.float64)
        consts = {"stft.basis": stft_basis(self.n_fft, self.dim_f), "istft.basis": istft_basis(self.n_fft, self.dim_f),
                  "window": (0.5 - 0.5 * np.cos(2 * np.pi * n / self.n_fft)).astype(np.float32)}
        with torch.cuda.device(self.device):
            _lib.set_tensors(_lib.lib.rvc_mdx23_set_tensor, self._h, sd)
            _lib.set_tensors(_lib.lib.rvc_mdx23_set_tensor, self._h, consts)
            _lib.check(_lib.lib.rvc_mdx23_finalize(self._h))
        self._ready = True
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    def __call__(self, x):
        """x [B, 2, chunk] (tensor / array, any device) -> device tensor [B, S, 2, chunk] ([B, 2, chunk] for a single target, as upstream)."""
        assert self._ready, "load_state_dict first"
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(self.device, torch.float32).contiguous()
        assert x.dim() == 3 and x.shape[1] == 2 and x.shape[2] == self.chunk_size, f"expected [B, 2, {self.chunk_size}]"
        S = self.num_target_instruments
        out = torch.empty(x.shape[0], S, 2, self.chunk_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for b in range(x.shape[0]):
                _lib.check(_lib.lib.rvc_mdx23_forward(self._h, _lib.current_stream(), _lib.ptr(x[b]), self.chunk_size, _lib.ptr(out[b])))
        return out if S > 1 else out[:, 0]

    def set_streams(self, k):
        """Chunk streams of demix_device (rvc_mdx23_set_streams): 1 = the reference's order of additions (default), 3 = a single conversion alone on the GPU."""
        _lib.check(_lib.lib.rvc_mdx23_set_streams(self._h, int(k)))
        return self

    def demix_device(self, mix, step, n_chunks, overlap):
        """The chunk loop of demix_mdxv3 (reference lib/karafan/inference.py:52-66) behind one C call: mix [2, Lp] device tensor (already zero-padded) ->
        [S, 2, Lp] = sum over chunks (every `step` samples, in order, NaN as zero) of the separated chunk at its offset, divided by `overlap`."""
        assert self._ready, "load_state_dict first"
        mix = mix.to(self.device, torch.float32).contiguous()
        assert mix.dim() == 2 and mix.shape[0] == 2
        acc = torch.empty(self.num_target_instruments, 2, mix.shape[1], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_mdx23_demix(self._h, _lib.current_stream(), _lib.ptr(mix), int(mix.shape[1]), int(step), int(n_chunks), float(overlap), _lib.ptr(acc)))
        return acc
