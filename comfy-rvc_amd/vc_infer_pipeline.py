"""VC / get_vc / vc_single on the HIP path (mirror of reference vc_infer_pipeline.py:23-327).

Same call surface as the reference: `get_vc(model_path, file_index=None, config=config, device=None)`,
`vc_single(cpt, net_g, vc, hubert_model, sid, input_audio, ..., f0_method, ...)`, `VC(tgt_sr, config).pipeline(...)`
and `VC.vc(...)`.  Host work is what the reference also does on the host (zero-phase high-pass, reflect padding,
segmentation, f0 post-processing, RMS mixing, int16 normalisation); the three networks run through librvc_hip.so.
When both callees are this package's HIP-backed objects, `VC.vc` uses the fused `rvc_vc_segment` entry point (features
stay channel-major on the device); any other object that honours the callee protocol takes the generic tensor path.
"""
import os
import sys
import traceback

import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

from . import _lib
from .config import config
from .lib.audio import MAX_INT16, remix_audio, resample
from .lib.infer_pack.loaders import HubertModelWithFinalProj
from .lib.infer_pack.models import (SynthesizerTrnMs256NSFsid, SynthesizerTrnMs256NSFsid_nono, SynthesizerTrnMs768NSFsid,
                                    SynthesizerTrnMs768NSFsid_nono, _SynthesizerNSFsid)
from .lib.model_utils import change_rms, load_hubert
from .lib.utils import gc_collect
from .pitch_extraction import FeatureExtractor

bh, ah = signal.butter(N=5, Wn=48, btype="high", fs=16000)   # 48 Hz zero-phase high-pass (reference :21)
_BH, _AH = np.ascontiguousarray(bh, dtype=np.float64), np.ascontiguousarray(ah, dtype=np.float64)
_ZI = np.ascontiguousarray(signal.lfilter_zi(bh, ah), dtype=np.float64)   # filtfilt's initial conditions (device path)
# the same Butterworth design as second-order sections: the device evaluates the filter block-propagated in this (well-conditioned) form
_SOS = np.ascontiguousarray(signal.butter(N=5, Wn=48, btype="high", fs=16000, output="sos"), dtype=np.float64)      # [3][6], a0 = 1
_SOS_ZI = np.ascontiguousarray(signal.sosfilt_zi(_SOS), dtype=np.float64)


class VC(FeatureExtractor):
    noise_fn = None        # optional callable(shape) -> CPU float tensor replacing the global-RNG draws (tests / replay)
    noise_on_device = False  # True: draw the synthesizer noise with the GPU generator (faster, not CPU-replayable)
    overlap_streams = True   # HuBERT on a side stream while RMVPE runs (False: one stream, for per-kernel profiling)
    device_f0_post = True    # plain "rmvpe": transpose + mel quantisation of the pitch on the device, no host round trip before the synthesizer

    def _draw(self, shape):
        if self.noise_fn is not None:
            return self.noise_fn(shape)
        if self.noise_on_device:
            return torch.randn(shape, device=self.device)
        return torch.randn(shape)

    def _noise(self, inter, T, upp):
        nz = self._draw((1, inter, T))
        if self.noise_fn is None and not self.noise_on_device:
            torch.rand(1, 1)       # SineGen's rand_ini draw keeps the global stream aligned with the reference
        ns = self._draw((1, T * upp, 1))
        return nz, ns

    def vc(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, index_rate, version, protect):
        feats = torch.from_numpy(audio0).float()
        if feats.dim() == 2:
            feats = feats.mean(-1)
        assert feats.dim() == 1, feats.dim()
        feats = feats.view(1, -1)
        use_index = index is not None and big_npy is not None and index_rate > 0
        fused = isinstance(model, HubertModelWithFinalProj) and isinstance(net_g, _SynthesizerNSFsid) \
            and pitch is not None and pitchf is not None and not use_index
        if fused:
            L = feats.shape[1]
            Th = model.num_frames(L)
            T = 2 * Th
            p_len = min(audio0.shape[0] // self.window, T)
            assert p_len == T and pitch.shape[1] >= T, "p_len = 2*T_h always holds (SURVEY 9)"
            dev = net_g.device
            a = feats.view(-1).to(dev)
            pc = pitch[0, :T].to(dev, torch.int64).contiguous()
            pf = pitchf[0, :T].to(dev, torch.float32).contiguous()
            nz, ns = self._noise(net_g.inter_channels, T, net_g.upp)
            nz = nz.to(dev, torch.float32).contiguous()
            ns = ns.to(dev, torch.float32).contiguous()
            out = torch.empty(T * net_g.upp, dtype=torch.float32, device=dev)
            sid_i = int(torch.as_tensor(sid).reshape(-1)[0])
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.rvc_vc_segment(model._h, net_g._h, _lib.current_stream(), _lib.ptr(a), L, 1 if version == "v1" else 2,
                                                   _lib.ptr(pc), _lib.ptr(pf), sid_i, float(protect), 1 if protect < 0.5 else 0,
                                                   _lib.ptr(nz), _lib.ptr(ns), _lib.ptr(out)))
            return out.cpu().numpy()
        # ---- generic callee-protocol path (any extract_features / infer implementation)
        dev = self.device
        feats = model.extract_features(version=version, source=feats.to(dev), padding_mask=None, output_layer=9 if version == "v1" else 12)
        feats0 = feats.clone() if (protect < 0.5 and pitch is not None and pitchf is not None) else None
        if use_index:
            # nearest training feature per frame, blended in with weight index_rate (reference :60-75)
            npy = feats[0].cpu().numpy().astype("float32")
            score, ix = index.search(npy, k=1)
            weight = np.square(1 / score)
            weight /= weight.sum(axis=1, keepdims=True)
            npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
            feats = torch.from_numpy(npy.astype("float32")).unsqueeze(0).to(feats.device) * index_rate + (1 - index_rate) * feats
        feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
        if feats0 is not None:
            feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
        p_len = min(audio0.shape[0] // self.window, feats.shape[1])
        if pitch is not None and pitchf is not None:
            pitch, pitchf = pitch[:, :p_len].to(feats.device), pitchf[:, :p_len].to(feats.device)
            if protect < 0.5:
                pitchff = pitchf.clone()
                pitchff[pitchf > 0] = 1
                pitchff[pitchf < 1] = protect
                pitchff = pitchff.unsqueeze(-1)
                feats = (feats * pitchff + feats0 * (1 - pitchff)).to(feats0.dtype)
        p_len_t = torch.tensor([p_len], device=feats.device).long()
        with torch.no_grad():
            if pitch is not None and pitchf is not None:
                kw = {}
                if isinstance(net_g, _SynthesizerNSFsid):
                    kw["noise"] = self._noise(net_g.inter_channels, p_len, net_g.upp)
                audio1 = net_g.infer(feats, p_len_t, pitch, pitchf, sid, **kw)[0][0, 0].data.cpu().float().numpy()
            else:
                kw = {}
                if isinstance(net_g, _SynthesizerNSFsid):      # the no-f0 family draws one randn_like (reference models.py:908,:1014)
                    kw["noise"] = self._draw((1, net_g.inter_channels, p_len))
                audio1 = net_g.infer(feats, p_len_t, sid, **kw)[0][0, 0].data.cpu().float().numpy()
        return audio1

    def _cut_points(self, audio):
        """Cut points of a long clip: the quietest sample (|160-tap moving sum|) within +-t_query of every t_center (reference
        :127-135).  160 sequential adds on purpose: a cumulative-sum shortcut would change the rounding and could move a cut."""
        opt_ts = []
        if audio.shape[0] + self.window > self.t_max:
            audio_pad = np.pad(audio, (self.window // 2, self.window // 2), mode="reflect")
            audio_sum = np.zeros_like(audio)
            for i in range(self.window):
                audio_sum += audio_pad[i: i - self.window]
            for t in range(self.t_center, audio.shape[0], self.t_center):
                seg = np.abs(audio_sum[t - self.t_query: t + self.t_query])
                opt_ts.append(t - self.t_query + np.where(seg == seg.min())[0][0])
        return opt_ts

    def pipeline(self, model, net_g, sid, audio, times, f0_up_key, f0_method, merge_type, file_index, index_rate, if_f0,
                 filter_radius, tgt_sr, resample_sr, rms_mix_rate, version, protect, crepe_hop_length, f0_autotune, rmvpe_onnx,
                 f0_file=None, f0_min=50, f0_max=1600):
        index, big_npy = self.load_index(file_index)
        use_index = index is not None and big_npy is not None and index_rate > 0
        assert bool(if_f0) == bool(getattr(net_g, "HAS_F0", if_f0)), "if_f0 must match the model family (cpt['f0'])"
        device_path = (isinstance(model, HubertModelWithFinalProj) and isinstance(net_g, _SynthesizerNSFsid)
                       and (not use_index or hasattr(index, "blend_device"))
                       and not (resample_sr >= 16000 and tgt_sr != resample_sr) and f0_file is None)
        if device_path:
            return self._pipeline_device(model, net_g, sid, audio, f0_up_key, f0_method, merge_type, filter_radius,
                                         tgt_sr, rms_mix_rate, version, protect, crepe_hop_length, f0_autotune, rmvpe_onnx, f0_min, f0_max,
                                         index if use_index else None, index_rate, bool(if_f0))
        audio = signal.filtfilt(bh, ah, audio)
        opt_ts = self._cut_points(audio)
        s = 0
        audio_opt = []
        t = None
        audio_pad = np.pad(audio, (self.t_pad, self.t_pad), mode="reflect")
        inp_f0 = None
        if f0_file is not None:
            try:
                with open(f0_file.name, "r") as f:
                    inp_f0 = np.array([list(map(float, line.split(","))) for line in f.read().strip("\n").split("\n")], dtype="float32")
            except Exception:   # noqa: BLE001
                traceback.print_exc()
        sid = torch.tensor(sid).unsqueeze(0).long()
        pitch, pitchf = None, None
        if if_f0:
            pitch, pitchf = self.get_f0(audio_pad, f0_up_key, f0_method, merge_type, filter_radius, crepe_hop_length, f0_autotune,
                                        rmvpe_onnx, inp_f0, f0_min, f0_max)
            p_len = min(pitch.shape[0], pitchf.shape[0])
            pitch = torch.from_numpy(pitch[:p_len].astype(np.int64)).unsqueeze(0)
            pitchf = torch.from_numpy(pitchf[:p_len].astype(np.float32)).unsqueeze(0)
        for t in opt_ts:
            t = t // self.window * self.window
            start, end = s, t + self.t_pad2 + self.window
            ps = pitch[:, start // self.window: end // self.window] if if_f0 else None
            pfs = pitchf[:, start // self.window: end // self.window] if if_f0 else None
            audio_opt.append(self.vc(model, net_g, sid, audio_pad[start:end], ps, pfs, times, index, big_npy, index_rate, version,
                                     protect)[self.t_pad_tgt: -self.t_pad_tgt])
            s = t
        ps = pitch[:, t // self.window:] if if_f0 and t is not None else pitch
        pfs = pitchf[:, t // self.window:] if if_f0 and t is not None else pitchf
        audio_opt.append(self.vc(model, net_g, sid, audio_pad[t:], ps, pfs, times, index, big_npy, index_rate, version,
                                 protect)[self.t_pad_tgt: -self.t_pad_tgt])
        audio_opt = np.concatenate(audio_opt)
        if rms_mix_rate < 1:
            audio_opt = change_rms(audio, 16000, audio_opt, tgt_sr, rms_mix_rate)
        if resample_sr >= 16000 and tgt_sr != resample_sr:
            audio_opt = resample(audio_opt, tgt_sr, resample_sr, device=str(self.device))     # reference :185-186 (librosa.resample)
        self.last_float = audio_opt          # float waveform before the int16 normalisation (parity tests compare this too)
        audio_max = np.abs(audio_opt).max() / 0.99
        audio_opt = (audio_opt * MAX_INT16 / audio_max).astype(np.int16)
        return audio_opt


# the base implementations the device-side f0 path stands in for (captured at import: a later class-level patch no longer matches)
_BASE_F0_METHODS = {name: FeatureExtractor.__dict__[name] for name in ("get_f0", "get_rmvpe", "_rmvpe")}


def _is_base_method(obj, name):
    """True when obj.<name> resolves to FeatureExtractor's own function: no instance attribute, no subclass override, no class-level patch."""
    if name in getattr(obj, "__dict__", {}):
        return False
    for klass in type(obj).__mro__:
        if name in klass.__dict__:
            return klass.__dict__[name] is _BASE_F0_METHODS[name]
    return False


def _pipeline_device(self, model, net_g, sid, audio, f0_up_key, f0_method, merge_type, filter_radius, tgt_sr,
                     rms_mix_rate, version, protect, crepe_hop_length, f0_autotune, rmvpe_onnx, f0_min, f0_max, index=None, index_rate=0.0,
                     if_f0=True):
    """VC.pipeline with every per-sample stage on the GPU: the zero-phase high-pass, reflect padding and input RMS frames
    (rvc_preprocess), HuBERT on a side stream while RMVPE produces the pitch, the segments synthesised from device-resident
    features, and change_rms + int16 normalisation (reference vc_infer_pipeline.py:182-189) as kernels; only the cut search of
    long clips and the 100 fps pitch post-processing stay on the host, exactly where the reference has them."""
    import time as _t
    _tr = [("start", _t.perf_counter())] if os.environ.get("RVC_TRACE") else None
    def _mark(n):
        if _tr is not None:
            _tr.append((n, _t.perf_counter()))
    dev = net_g.device
    main = torch.cuda.current_stream(dev)
    audio = np.ascontiguousarray(audio)
    if audio.dtype not in (np.float32, np.float64):
        audio = audio.astype(np.float64)
    n = int(audio.shape[0])
    raw_d = torch.from_numpy(audio).to(dev)
    filt_d = torch.empty(n, dtype=torch.float64, device=dev)            # signal.filtfilt(bh, ah, audio)
    a_dev = torch.empty(n + 2 * self.t_pad, dtype=torch.float32, device=dev)
    rms1 = torch.empty(n // 8000 + 1, dtype=torch.float64, device=dev) if rms_mix_rate < 1 else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib.rvc_preprocess(_lib.current_stream(), _lib.ptr(raw_d), 1 if audio.dtype == np.float64 else 0, n, _lib.ptr(_BH),
                                           _lib.ptr(_AH), _lib.ptr(_ZI), int(self.t_pad), _lib.ptr(filt_d), _lib.ptr(a_dev), _lib.ptr(rms1),
                                           0 if rms1 is None else rms1.numel(), _lib.ptr(_SOS), _lib.ptr(_SOS_ZI)))
    opt_ts = self._cut_points(filt_d.cpu().numpy()) if n + self.window > self.t_max else []
    bounds, s0 = [], 0
    for t in opt_ts:
        t = t // self.window * self.window
        bounds.append((s0, t + self.t_pad2 + self.window))
        s0 = t
    bounds.append((s0, a_dev.shape[0]))
    _mark("preprocess enqueued")
    if getattr(self, "_side", None) is None:
        self._side = torch.cuda.Stream(dev)
    side = self._side if self.overlap_streams else main
    side.wait_stream(main)
    feats, feats0 = [], []
    with torch.cuda.stream(side):
        for (b0, b1) in bounds:
            f = model.extract_features(a_dev[b0:b1].view(1, -1), version=version, channel_major=True)
            if index is not None:
                # feature retrieval on the device, still on the side stream (reference :60-75); feats0 feeds the protect blend
                feats0.append(f if (protect < 0.5 and if_f0) else None)
                f = index.blend_device(f.contiguous(), index_rate)
            else:
                feats0.append(None)
            feats.append(f)
    a_dev.record_stream(side)
    _mark("hubert enqueued")
    # the synthesizer's noise does not depend on the pitch: draw it now, so that nothing but the pitch upload sits between "f0 ready"
    # and the first synthesizer kernel (get_f0 draws no random numbers, the order of the RNG stream is unchanged)
    noises = []
    for f in feats:
        T = 2 * f.shape[1]
        if self.noise_fn is None and not self.noise_on_device:
            for _ in range(12):
                torch.rand([])      # the reference's HuBERT draws one LayerDrop scalar per layer from the same global stream
        if if_f0:
            nz, ns = self._noise(net_g.inter_channels, T, net_g.upp)
            ns = ns.to(dev, torch.float32).contiguous()
        else:
            nz, ns = self._draw((1, net_g.inter_channels, T)), None          # no-f0 family: one draw, no source noise
        noises.append((nz.to(dev, torch.float32).contiguous(), ns, torch.empty(T * net_g.upp, dtype=torch.float32, device=dev)))
    # pitch on the main stream (RMVPE) + host post-processing at 100 fps
    pitch_d = pitchf_d = None
    deferred_status = None
    if if_f0:
        m = f0_method[0] if isinstance(f0_method, (list, tuple)) and len(f0_method) == 1 else f0_method
        # the fused device path replaces get_f0 / get_rmvpe / _rmvpe only when all three ARE the base implementations: an instance attribute,
        # a subclass override or a class-level patch (VC.get_f0 = ...) of any of them sends the clip through self.get_f0 like the reference
        plain_rmvpe = (m == "rmvpe" and not f0_autotune and self.device_f0_post and _is_base_method(self, "get_f0")
                       and _is_base_method(self, "_rmvpe") and _is_base_method(self, "get_rmvpe")
                       and getattr(self.f0_method_dict.get("rmvpe"), "__func__", None) is _BASE_F0_METHODS["get_rmvpe"]
                       and getattr(self.f0_method_dict.get("rmvpe"), "__self__", None) is self)
        if plain_rmvpe:
            # RMVPE with nothing spliced in (the default front end): the 100-fps post-processing of get_f0 (pitch_extraction.py:176-185, reference
            # vc_infer_pipeline.py / pitch_extraction.py: transpose, mel-scale quantisation to 1 .. 255) runs on the device in float64 like the numpy
            # original, so nothing waits for the pitch on the host: the synthesizer is enqueued behind RMVPE.  The GRU scan's status word is
            # checked after the clip (below) instead of before the synthesizer.
            rm = self._rmvpe()
            f0 = rm.infer(a_dev, 0.03)["f0"]
            mel_min, mel_max = 2595 * np.log10(1 + f0_min / 700), 2595 * np.log10(1 + f0_max / 700)      # lib/audio.py hz_to_mel
            pitch_d = torch.empty(f0.numel(), dtype=torch.int64, device=dev)
            pitchf_d = torch.empty(f0.numel(), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.rvc_f0_post(_lib.current_stream(), _lib.ptr(f0), f0.numel(), float(pow(2, f0_up_key / 12)), float(mel_min), float(mel_max),
                                                int(self.f0_bins), _lib.ptr(pitch_d), _lib.ptr(pitchf_d)))
            deferred_status = rm
            _mark("f0 enqueued (rmvpe + device post)")
        else:
            x_f0 = a_dev if f0_method in ("rmvpe", "rmvpe+") else a_dev.cpu().numpy().astype(np.float64)
            pitch, pitchf = self.get_f0(x_f0, f0_up_key, f0_method, merge_type, filter_radius, crepe_hop_length, f0_autotune, rmvpe_onnx, None,
                                        f0_min, f0_max)
            _mark("f0 ready (rmvpe sync + host post)")
            p_len = min(pitch.shape[0], pitchf.shape[0])
            pitch_d = torch.from_numpy(pitch[:p_len].astype(np.int64)).to(dev)
            pitchf_d = torch.from_numpy(pitchf[:p_len].astype(np.float32)).to(dev)
    self.last_pitch = (pitch_d, pitchf_d)      # device tensors (coarse int64, Hz float32) of this clip, for inspection
    main.wait_stream(side)
    sid_i = int(torch.as_tensor(sid).reshape(-1)[0])
    D = 256 if version == "v1" else 768
    outs = []
    for (b0, b1), f, f0c, (nz, ns, out) in zip(bounds, feats, feats0, noises):
        Th = f.shape[1]
        T = 2 * Th
        pc = pf = None
        if if_f0:
            pc = pitch_d[b0 // self.window: b0 // self.window + T].contiguous()
            pf = pitchf_d[b0 // self.window: b0 // self.window + T].contiguous()
            assert pc.numel() == T, "pitch track shorter than the feature sequence"
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.rvc_vc_segment_feats(net_g._h, _lib.current_stream(), _lib.ptr(f), _lib.ptr(f0c), Th, D, _lib.ptr(pc), _lib.ptr(pf), sid_i,
                                                     float(protect), 1 if (protect < 0.5 and if_f0) else 0, _lib.ptr(nz), _lib.ptr(ns), _lib.ptr(out)))
        outs.append(out[self.t_pad_tgt: out.numel() - self.t_pad_tgt])
    _mark("synth enqueued")
    wav = torch.cat(outs) if len(outs) > 1 else outs[0].contiguous()
    N = wav.numel()
    i16 = torch.empty(N, dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib.rvc_postprocess(_lib.current_stream(), _lib.ptr(wav), N, _lib.ptr(rms1), 0 if rms1 is None else rms1.numel(),
                                            int(tgt_sr), float(rms_mix_rate), _lib.ptr(i16)))
    self.last_float = wav          # device tensor: waveform after change_rms, before the int16 normalisation
    self.last_i16 = i16            # device tensor: the int16 result (what a multi-GPU caller hands to the gather without a host round trip)
    _mark("post enqueued")
    res = i16.cpu().numpy()
    _mark("result on host")
    if deferred_status is not None:
        deferred_status.check_status()      # raises RvcHipError if the GRU scan of this clip's RMVPE forward timed out (f0 was NaN then)
    if _tr is not None:
        print("trace ms: " + " | ".join(f"{n} {1e3 * (t - _tr[i][1]):.1f}" for i, (n, t) in enumerate(_tr[1:])))
    return res


VC._pipeline_device = _pipeline_device


def _synth_class(version, if_f0):
    if if_f0 != 1:          # trained without f0: text encoder without pitch embedding + plain Generator (reference :209-218)
        return SynthesizerTrnMs256NSFsid_nono if version == "v1" else SynthesizerTrnMs768NSFsid_nono
    return SynthesizerTrnMs256NSFsid if version == "v1" else SynthesizerTrnMs768NSFsid


def get_vc(model_path, file_index=None, config=config, device=None):
    """`.pth` (or an already loaded cpt dict) -> {vc, cpt, net_g, model_name, file_index, sr} (reference :198-249)."""
    if isinstance(model_path, dict):
        cpt, model_name = model_path, model_path.get("info", "cpt")
    else:
        cpt = torch.load(model_path, map_location="cpu")
        model_name = os.path.basename(model_path).split(".")[0]
    tgt_sr = cpt["config"][-1]
    cpt["config"][-3] = cpt["weight"]["emb_g.weight"].shape[0]   # n_spk
    if_f0 = cpt.get("f0", 1)
    version = cpt.get("version", "v1")
    net_g = _synth_class(version, if_f0)(*cpt["config"], is_half=config.is_half, device=device if device else config.device)
    net_g.load_state_dict(cpt["weight"], strict=False)
    net_g.eval()
    vc = VC(tgt_sr, config)
    # preload the retrieval index (reference :230-246): the dict then carries the (index, big_npy) tuple
    if file_index and os.path.exists(str(file_index)):
        sys.stdout.write(f"Attempting to load {file_index}....\n")
        index, big_npy = vc.load_index(str(file_index))
        file_index = (index, big_npy) if index is not None else ""
    else:
        file_index = ""
    return {"vc": vc, "cpt": cpt, "net_g": net_g, "model_name": model_name, "file_index": file_index, "sr": cpt["config"][-1]}


def vc_single(cpt=None, net_g=None, vc=None, hubert_model=None, sid=0, input_audio=None, input_audio_path=None, f0_up_key=0,
              f0_file=None, f0_method="crepe", merge_type="median", file_index="", index_rate=.75, filter_radius=3, resample_sr=0,
              rms_mix_rate=.25, protect=0.33, crepe_hop_length=160, f0_autotune=False, is_onnx=False, config=config,
              hubert_path=None, **kwargs):
    """(int16 ndarray, sr) or None on any failure, like the reference (:251-327)."""
    if hubert_model is None:
        hubert_model = load_hubert(hubert_path, config)
    if not (cpt and net_g and vc and hubert_model):
        return None
    tgt_sr = cpt["config"][-1]
    version = cpt.get("version", "v1")
    if input_audio is None and input_audio_path is None:
        return None
    f0_up_key = int(f0_up_key)
    try:
        if input_audio is None:
            raise NotImplementedError("file decoding needs ffmpeg/soundfile; pass input_audio=(ndarray, sr)")
        audio, _ = remix_audio((input_audio[0], input_audio[1]), target_sr=16000)
        times = [0, 0, 0]
        if_f0 = cpt.get("f0", 1)
        audio_opt = vc.pipeline(hubert_model, net_g, sid, audio, times, f0_up_key,
                                f0_method if len(f0_method) > 1 else f0_method[0], merge_type, file_index, index_rate, if_f0,
                                filter_radius, tgt_sr, resample_sr, rms_mix_rate, version, protect, crepe_hop_length, f0_autotune,
                                is_onnx, f0_file=f0_file)
        return audio_opt, resample_sr if resample_sr >= 16000 and tgt_sr != resample_sr else tgt_sr
    except Exception as error:   # noqa: BLE001 - reference behaviour: print and return None
        print(error)
        return None
