"""SynthesizerTrnMs{256,768}NSFsid on the HIP kernel graph.

Drop-in for the inference surface of reference lib/infer_pack/models.py:580-809: same constructor signature (the
`cpt["config"]` list splatted, `is_half` kwarg), `load_state_dict`, `eval/float/half/to`, and
`infer(phone, phone_lengths, pitch, nsff0, sid, rate=None)` returning `(o, x_mask, (z, z_p, m_p, logs_p))`.
The reference draws its noise from the global torch RNG inside `infer` (models.py:801 and :409); here the same two
draws are made on the host with the same shapes and order - or supplied explicitly through `noise=(noise_z, noise_src)`
(precedent: the reference's ONNX twin takes `rnd`, models_onnx.py:634-648).
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib


class _SynthesizerNSFsid:
    FEAT_DIM = 768
    HAS_F0 = True

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, spk_embed_dim, gin_channels, sr, device="cuda:0", **kwargs):
        if isinstance(sr, str):
            sr = {"32k": 32000, "40k": 40000, "48k": 48000}[sr]
        assert str(resblock) == "1", "only ResBlock1 generators are on the RVC inference path"
        assert len(resblock_kernel_sizes) == 3 and all(len(d) == 3 for d in resblock_dilation_sizes)
        assert len(upsample_rates) <= 8
        self.inter_channels, self.hidden_channels, self.sr = inter_channels, hidden_channels, sr
        self.upsample_rates = list(upsample_rates)
        self.upp = int(np.prod(upsample_rates))
        self.device = torch.device(device)
        cfg = _lib.SynthConfig()
        cfg.inter_channels, cfg.hidden_channels, cfg.filter_channels = inter_channels, hidden_channels, filter_channels
        cfg.n_heads, cfg.n_layers, cfg.kernel_size = n_heads, n_layers, kernel_size
        cfg.n_resblock_kernels = 3
        for i in range(3):
            cfg.resblock_kernel_sizes[i] = resblock_kernel_sizes[i]
            for j in range(3):
                cfg.resblock_dilations[i][j] = resblock_dilation_sizes[i][j]
        cfg.n_upsamples = len(upsample_rates)
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.upsample_initial_channel, cfg.spk_embed_dim, cfg.gin_channels, cfg.sr = upsample_initial_channel, spk_embed_dim, gin_channels, sr
        cfg.feat_dim = self.FEAT_DIM
        self._ctx = _lib.get_ctx(self.device.index or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib.rvc_synth_create(self._ctx, C.byref(cfg), C.byref(h)))
        self._h = h
        self._loaded = False
        self.enc_q = None   # `del net_g.enc_q` in get_vc (reference vc_infer_pipeline.py:219) must keep working

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, 'lib', None) is not None:   # (module may be torn down at exit)
            _lib.lib.rvc_synth_destroy(h)
            self._h = None

    def load_state_dict(self, state_dict, strict=False):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("enc_q.")}
        with torch.cuda.device(self.device):
            _lib.set_tensors(_lib.lib.rvc_synth_set_tensor, self._h, sd)
            _lib.check(_lib.lib.rvc_synth_finalize(self._h))
        has_f0 = bool(_lib.lib.rvc_synth_has_f0(self._h))
        if has_f0 != self.HAS_F0:      # the reference would fail in load_state_dict(strict) / at the first infer call
            raise ValueError(f"{type(self).__name__} expects a checkpoint trained {'with' if self.HAS_F0 else 'without'} f0 "
                             f"(cpt['f0'] = {int(self.HAS_F0)}), this one is the other family")
        self._loaded = True
        return self

    def eval(self):
        return self

    def float(self):
        return self

    def half(self):
        return self   # weights stay fp32 on the device: the fp32-MFMA graph is the parity path

    def to(self, device):
        return self

    def infer(self, phone, phone_lengths, pitch, nsff0, sid, rate=None, noise=None, taps=None, phone_channel_major=False):
        assert self._loaded, "load_state_dict first"
        assert rate is None, "`rate` is unused by every caller of the reference (SURVEY 8a9)"
        dev = self.device
        if phone_channel_major:
            T = int(phone.shape[-1])
            ph = phone.to(dev, torch.float32).contiguous()
        else:
            assert phone.dim() == 3 and phone.shape[0] == 1
            T = int(phone.shape[1])
            ph = phone.to(dev, torch.float32).contiguous()
        assert int(phone_lengths.reshape(-1)[0]) == T, "only full-length sequences (the reference always passes p_len = T)"
        if noise is None:   # same draw order and shapes as the reference on its CPU path
            noise_z = torch.randn(1, self.inter_channels, T)
            torch.rand(1, 1)                                   # SineGen rand_ini (zeroed for harmonic_num = 0, models.py:378-381)
            noise_src = torch.randn(1, T * self.upp, 1)
        else:
            noise_z, noise_src = noise
        nz = torch.as_tensor(noise_z).to(dev, torch.float32).contiguous().view(self.inter_channels, T)
        ns = torch.as_tensor(noise_src).to(dev, torch.float32).contiguous().view(T * self.upp)
        pc = pitch.to(dev, torch.int64).contiguous().view(-1)[:T]
        pf = nsff0.to(dev, torch.float32).contiguous().view(-1)[:T]
        assert pc.numel() == T and pf.numel() == T
        out = torch.empty(1, 1, T * self.upp, dtype=torch.float32, device=dev)
        tp, tbuf = None, {}
        if taps is not None:
            C_, N = self.inter_channels, T * self.upp
            sizes = {"enc_p_layer0": (self.hidden_channels, T), "m_p": (C_, T), "logs_p": (C_, T), "z_p": (C_, T), "z": (C_, T),
                     "sine_waves": (N,), "har_source": (N,)}
            for n in taps:
                if n in sizes:
                    tbuf[n] = torch.empty(sizes[n], dtype=torch.float32, device=dev)
                else:
                    tbuf[n] = taps[n]
            tp = _lib.SynthTaps(*[_lib.ptr(tbuf.get(n)) for n, _ in _lib.SynthTaps._fields_])
        sid_i = int(torch.as_tensor(sid).reshape(-1)[0])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.rvc_synth_infer(self._h, _lib.current_stream(), _lib.ptr(ph), 1 if phone_channel_major else 0,
                                                _lib.ptr(pc), _lib.ptr(pf), sid_i, _lib.ptr(nz), _lib.ptr(ns), T, _lib.ptr(out),
                                                C.byref(tp) if tp is not None else None))
        if taps is not None:
            taps.update(tbuf)
        x_mask = torch.ones(1, 1, T, dtype=torch.float32, device=dev)
        return out, x_mask, (tbuf.get("z"), tbuf.get("z_p"), tbuf.get("m_p"), tbuf.get("logs_p"))


class _SynthesizerNSFsid_nono(_SynthesizerNSFsid):
    """No-f0 family (reference lib/infer_pack/models.py:812-1022): text encoder without pitch embedding, plain HiFi-GAN Generator.
    The reference's constructor has `sr=None` last (:833,:939); `infer(phone, phone_lengths, sid, rate=None)` draws one randn_like."""
    HAS_F0 = False

    def __init__(self, *config, sr=None, **kwargs):
        if len(config) == 18:
            config, sr = config[:17], config[17]
        super().__init__(*config, sr if sr is not None else 40000, **kwargs)

    def infer(self, phone, phone_lengths, sid, rate=None, noise=None, taps=None, phone_channel_major=False):
        assert self._loaded, "load_state_dict first"
        assert rate is None, "`rate` is unused by every caller of the reference (SURVEY 8a9)"
        dev = self.device
        T = int(phone.shape[-1]) if phone_channel_major else int(phone.shape[1])
        ph = phone.to(dev, torch.float32).contiguous()
        assert int(torch.as_tensor(phone_lengths).reshape(-1)[0]) == T, "only full-length sequences (the reference always passes p_len = T)"
        noise_z = torch.randn(1, self.inter_channels, T) if noise is None else (noise[0] if isinstance(noise, (tuple, list)) else noise)
        nz = torch.as_tensor(noise_z).to(dev, torch.float32).contiguous().view(self.inter_channels, T)
        out = torch.empty(1, 1, T * self.upp, dtype=torch.float32, device=dev)
        tp, tbuf = None, {}
        if taps is not None:
            C_ = self.inter_channels
            sizes = {"enc_p_layer0": (self.hidden_channels, T), "m_p": (C_, T), "logs_p": (C_, T), "z_p": (C_, T), "z": (C_, T)}
            tbuf = {n: torch.empty(sizes[n], dtype=torch.float32, device=dev) for n in taps if n in sizes}
            tp = _lib.SynthTaps(*[_lib.ptr(tbuf.get(n)) for n, _ in _lib.SynthTaps._fields_])
        sid_i = int(torch.as_tensor(sid).reshape(-1)[0])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.rvc_synth_infer(self._h, _lib.current_stream(), _lib.ptr(ph), 1 if phone_channel_major else 0, None, None, sid_i,
                                                _lib.ptr(nz), None, T, _lib.ptr(out), C.byref(tp) if tp is not None else None))
        if taps is not None:
            taps.update(tbuf)
        x_mask = torch.ones(1, 1, T, dtype=torch.float32, device=dev)
        return out, x_mask, (tbuf.get("z"), tbuf.get("z_p"), tbuf.get("m_p"), tbuf.get("logs_p"))


class SynthesizerTrnMs768NSFsid_nono(_SynthesizerNSFsid_nono):
    FEAT_DIM = 768


class SynthesizerTrnMs256NSFsid_nono(_SynthesizerNSFsid_nono):
    FEAT_DIM = 256


class SynthesizerTrnMs768NSFsid(_SynthesizerNSFsid):
    """v2: 768-d ContentVec features (reference lib/infer_pack/models.py:696-809)."""
    FEAT_DIM = 768


class SynthesizerTrnMs256NSFsid(_SynthesizerNSFsid):
    """v1: 256-d final_proj features (reference lib/infer_pack/models.py:580-693)."""
    FEAT_DIM = 256
