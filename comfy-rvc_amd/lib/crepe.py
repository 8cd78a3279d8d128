"""CREPE pitch tracking on the HIP path: the part of `torchcrepe.predict` that the reference's f0 methods "crepe" / "mangio-crepe" /
"crepe-tiny" / "mangio-crepe-tiny" rely on (reference pitch_extraction.py:76-150; torchcrepe is a third-party package, requirements.txt:24).

The network (frame normalisation, six convolution blocks, classifier, sigmoid) runs as HIP kernels behind rvc_crepe_forward and the
decoder (bins outside [fmin, fmax) masked, softmax over bins, Viterbi with the triangular transition matrix, periodicity = probability
at the decoded bin) behind rvc_crepe_viterbi (csrc/model_crepe.hip); torchcrepe decodes on the CPU with librosa's Viterbi.  What stays
on the host is what depends on numpy's global RNG - cents + triangular dither (scipy.stats.triang, exactly the call torchcrepe makes) -
and the NaN-aware median / mean filters over the 100 fps tracks.  `viterbi_bins` / `postprocess` below are the same decoder in numpy:
they pin the device kernel in tests/test_hip_crepe.py and are themselves pinned to the oracle in tests/test_host_logic.py.  torchcrepe's exact arithmetic cannot be checked offline: parity-unpinned; the restatement
follows torchcrepe 0.0.23.

Weights: torchcrepe ships `full.pth` / `tiny.pth` inside its package; put them under `models/torchcrepe/` (or pass a state dict).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from . import BASE_MODELS_DIR

SAMPLE_RATE, WINDOW_SIZE, PITCH_BINS, CENTS_PER_BIN = 16000, 1024, 360, 20


class Crepe:
    """The CREPE network on one GPU.  `probabilities(audio, hop)` -> device tensor [360, n]."""

    def __init__(self, model_path=None, model="full", device=None):
        assert model in ("full", "tiny"), f"unknown CREPE capacity {model!r} (torchcrepe knows 'full' and 'tiny')"
        if device is None or str(device) == "cpu":
            device = "cuda:0"
        self.device = torch.device(device)
        self.model = model
        if isinstance(model_path, dict):
            sd = model_path
        else:
            path = model_path or os.path.join(BASE_MODELS_DIR, "torchcrepe", f"{model}.pth")
            if not os.path.isfile(path):
                raise FileNotFoundError(f"{path}: CREPE weights not found (copy torchcrepe's assets/{model}.pth there)")
            sd = torch.load(path, map_location="cpu")
        self._ctx = _lib.get_ctx(self.device.index or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib.rvc_crepe_create(self._ctx, 1 if model == "tiny" else 0, C.byref(h)))
        self._h = h
        with torch.cuda.device(self.device):
            _lib.set_tensors(_lib.lib.rvc_crepe_set_tensor, h, {k: v for k, v in sd.items() if "num_batches_tracked" not in k})
            _lib.check(_lib.lib.rvc_crepe_finalize(h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "lib", None) is not None:
            _lib.lib.rvc_crepe_destroy(h)
            self._h = None

    def probabilities(self, audio, hop_length, pad=True, taps=None):
        a = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).to(self.device, torch.float32).contiguous().view(-1)
        L = a.numel()
        n = int(_lib.lib.rvc_crepe_num_frames(L, int(hop_length), 1 if pad else 0))
        assert n > 0, "audio too short"
        probs = torch.empty(PITCH_BINS, n, dtype=torch.float32, device=self.device)
        tp = None
        if taps is not None:
            tp = _lib.CrepeTaps(*[_lib.ptr(taps.get(nm)) for nm, _ in _lib.CrepeTaps._fields_])
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_crepe_forward(self._h, _lib.current_stream(), _lib.ptr(a), L, int(hop_length), 1 if pad else 0, _lib.ptr(probs),
                                                  C.byref(tp) if tp is not None else None))
        return probs


# ---------------------------------------------------------------------------------------------- decoding (host, 100 fps)
def frequency_to_bins(frequency, ceil=False):
    """torchcrepe.convert.frequency_to_bins (float32 arithmetic like the torch original)."""
    cents = np.float32(1200) * np.log2(np.float32(frequency) / np.float32(10.))
    b = (cents - np.float32(1997.3794084376191)) / np.float32(CENTS_PER_BIN)
    return int(np.ceil(b)) if ceil else int(np.floor(b))


def _log_transition_band():
    """log of the triangular transition matrix max(12 - |i - j|, 0) / row sum, as a band: lt[d + 11][j] = log A[j + d][j], d = -11 .. 11
    (entries outside the band are log(tiny) = -708 in librosa's dense form and can never win: neighbouring states differ by < 6 per step)."""
    idx = np.arange(PITCH_BINS)
    tr = np.maximum(12 - np.abs(idx[None, :] - idx[:, None]), 0).astype(np.float64)
    tr = tr / tr.sum(axis=1, keepdims=True)
    lt = np.full((23, PITCH_BINS), -np.inf)
    for d in range(-11, 12):
        j = idx[(idx + d >= 0) & (idx + d < PITCH_BINS)]
        lt[d + 11, j] = np.log(tr[j + d, j] + np.finfo(np.float64).tiny)
    return lt


_LT = None


def viterbi_bins(seq):
    """librosa.sequence.viterbi(seq [360, n], transition) with a uniform prior: the most likely bin path (first maximum wins, as np.argmax)."""
    global _LT
    if _LT is None:
        _LT = _log_transition_band()
    n = seq.shape[1]
    logp = np.log(seq.T + np.finfo(seq.dtype).tiny).astype(np.float64)
    value = logp[0] + np.log(1.0 / PITCH_BINS + np.finfo(np.float64).tiny)
    ptr = np.zeros((n, PITCH_BINS), dtype=np.int16)
    pad = np.full(11, -np.inf)
    jj = np.arange(PITCH_BINS)
    for t in range(1, n):
        vp = np.concatenate([pad, value, pad])
        cand = np.lib.stride_tricks.sliding_window_view(vp, 23).T + _LT        # [23, 360]: cand[d + 11][j] = value[j + d] + log A[j + d][j]
        k = np.argmax(cand, axis=0)                                            # smallest source index among equal maxima, like librosa
        ptr[t] = k - 11
        value = logp[t] + cand[k, jj]
    state = np.zeros(n, dtype=np.int64)
    state[-1] = int(np.argmax(value))
    for t in range(n - 2, -1, -1):
        state[t] = state[t + 1] + ptr[t + 1, state[t + 1]]
    return state


def bins_to_frequency(bins):
    """torchcrepe.convert.bins_to_frequency: 20 cents per bin + triangular dither of +-20 cents, float32 (scipy's global numpy RNG)."""
    import scipy.stats
    cents = np.float32(CENTS_PER_BIN) * bins.astype(np.float32) + np.float32(1997.3794084376191)
    noise = scipy.stats.triang.rvs(c=0.5, loc=-CENTS_PER_BIN, scale=2 * CENTS_PER_BIN, size=cents.shape)
    cents = cents + noise.astype(np.float32)
    return (np.float32(10) * np.float32(2) ** (cents / np.float32(1200))).astype(np.float32)


def postprocess(probabilities, fmin, fmax, return_periodicity=False):
    """probabilities [360, n] (numpy float32) -> pitch [n] float32 (, periodicity [n])."""
    p = np.array(probabilities, dtype=np.float32, copy=True)
    minidx, maxidx = frequency_to_bins(fmin), frequency_to_bins(fmax, ceil=True)
    p[:minidx] = -np.inf
    p[maxidx:] = -np.inf
    e = np.exp(p - p.max(axis=0, keepdims=True))
    seq = (e / e.sum(axis=0, keepdims=True)).astype(np.float32)
    bins = viterbi_bins(seq)
    pitch = bins_to_frequency(bins)
    if not return_periodicity:
        return pitch
    return pitch, p[bins, np.arange(p.shape[1])]


def filter_median(x, win_length):
    """torchcrepe.filter.median for NaN-free input: reflect-padded values, zero-padded validity mask, element (valid - 1) // 2 of the
    sorted valid values."""
    x = np.asarray(x, dtype=np.float32)
    h = win_length // 2
    xp = np.pad(x, (h, h), mode="reflect")
    valid = np.pad(np.ones_like(x, dtype=bool), (h, h), mode="constant")
    w = np.lib.stride_tricks.sliding_window_view(xp, win_length).copy()
    m = np.lib.stride_tricks.sliding_window_view(valid, win_length)
    w[~m] = np.inf
    w.sort(axis=1)
    idx = np.maximum((m.sum(axis=1) - 1) // 2, 0)
    return w[np.arange(x.shape[0]), idx]


def filter_mean(x, win_length):
    """torchcrepe.filter.mean for NaN-free input: zero-padded window sum over the number of in-range samples; exact zeros become NaN."""
    x = np.asarray(x, dtype=np.float32)
    h = win_length // 2
    s = np.convolve(x, np.ones(win_length, dtype=np.float32), mode="same") if win_length > 1 else x.copy()
    cnt = np.convolve(np.ones_like(x), np.ones(win_length, dtype=np.float32), mode="same")
    avg = (s / np.maximum(cnt, 1)).astype(np.float32)
    avg[avg == 0] = np.nan
    del h
    return avg


_models = {}


def _model_for(model, device):
    key = (model, str(device))
    if key not in _models:
        _models[key] = Crepe(None, model, device)
    return _models[key]


def predict(audio, sample_rate, hop_length=None, fmin=50., fmax=2006., model="full", return_periodicity=False, batch_size=None, device="cuda:0",
            pad=True, crepe=None):
    """The call surface of torchcrepe.predict that the reference uses (decoder = viterbi).  audio: [1, L] tensor / array at 16 kHz.
    Returns pitch [1, n] (and periodicity [1, n]) as torch tensors on the CPU; `batch_size` is accepted and ignored (frames are batched
    by the kernel graph).  `crepe`: a Crepe instance (tests inject procedural weights); default: models/torchcrepe/<model>.pth."""
    assert int(sample_rate) == SAMPLE_RATE, "resample to 16 kHz first (the reference always calls with 16 kHz audio)"
    hop_length = int(sample_rate // 100) if hop_length is None else int(hop_length)
    a = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
    a = a.reshape(1, -1) if a.dim() == 1 else a
    assert a.shape[0] == 1, "one clip per call"
    net = crepe if crepe is not None else _model_for(model, device)
    probs = net.probabilities(a[0], hop_length, pad)
    bins, per = viterbi_device(probs, frequency_to_bins(fmin), frequency_to_bins(fmax, ceil=True))
    pitch = torch.from_numpy(bins_to_frequency(bins))[None]
    return (pitch, torch.from_numpy(per)[None]) if return_periodicity else pitch


def viterbi_device(probs, min_bin, max_bin):
    """Masked softmax + Viterbi path + periodicity gather on the GPU (rvc_crepe_viterbi): probs device tensor [360, n] ->
    (bins int64 [n], periodicity float32 [n]) numpy.  `viterbi_bins` / `postprocess` above are the same algorithm on the host."""
    n = int(probs.shape[1])
    bins = torch.empty(n, dtype=torch.int32, device=probs.device)
    per = torch.empty(n, dtype=torch.float32, device=probs.device)
    with torch.cuda.device(probs.device):
        _lib.check(_lib.lib.rvc_crepe_viterbi(_lib.current_stream(), _lib.ptr(probs), n, int(min_bin), int(max_bin), _lib.ptr(bins), _lib.ptr(per)))
    return bins.cpu().numpy().astype(np.int64), per.cpu().numpy()
