"""RMVPE pitch estimator on the HIP kernel graph.

Drop-in for the host class of reference lib/rmvpe.py:559-685: `RMVPE(model_path, is_half, onnx=False, device=None)`,
`infer_from_audio(audio, thred=0.03)`, `infer_from_audio_with_pitch(...)`, `decode(hidden, thred)`,
`mel2hidden(mel)`-style access through `infer(..., return_all=True)`.  The conv-STFT basis and the HTK/Slaney mel
filterbank are the two constant matrices the reference builds at construction (lib/rmvpe.py:88-109,:492-499);
they are built here with numpy and handed to the library as tensors, everything else is rvc_rmvpe_forward.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def stft_forward_basis(n_fft=1024):
    """[real; imag] rows 0..n_fft/2 of the DFT matrix times a periodic Hann window -> float32 [n_fft + 2, n_fft]."""
    k = np.arange(n_fft // 2 + 1)[:, None].astype(np.float64)
    n = np.arange(n_fft)[None, :].astype(np.float64)
    ang = 2.0 * np.pi * k * n / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)])
    # exact zeros / ones where the reference's np.fft.fft(np.eye(n)) produces them does not matter at fp32
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)
    return (basis.astype(np.float32) * win[None, :]).astype(np.float32)


def mel_filterbank(sr=16000, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0):
    """HTK mel scale, triangular filters, Slaney area normalisation (what librosa.filters.mel(htk=True) returns)."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)

    freqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    fb = np.zeros((n_mels, freqs.size), dtype=np.float32)
    for i in range(n_mels):
        fb[i] = np.maximum(0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb


class RMVPE:
    def __init__(self, model_path, is_half=False, onnx=False, device=None):
        assert not onnx, "the ONNX/DirectML branch of the reference is out of scope"
        if device is None or str(device) == "cpu":
            device = "cuda:0"
        self.device = torch.device(device)
        self.is_half = False
        if isinstance(model_path, dict):
            sd = model_path
        else:
            sd = torch.load(model_path, map_location="cpu")
        self._ctx = _lib.get_ctx(self.device.index or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib.rvc_rmvpe_create(self._ctx, C.byref(h)))
        self._h = h
        consts = {"stft.forward_basis": stft_forward_basis(1024), "mel_basis": mel_filterbank()}
        with torch.cuda.device(self.device):
            _lib.set_tensors(_lib.lib.rvc_rmvpe_set_tensor, h, sd)
            _lib.set_tensors(_lib.lib.rvc_rmvpe_set_tensor, h, consts)
            _lib.check(_lib.lib.rvc_rmvpe_finalize(h))
        self.cents_mapping = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, 'lib', None) is not None:   # (module may be torn down at exit)
            _lib.lib.rvc_rmvpe_destroy(h)
            self._h = None

    def infer(self, audio, thred=0.03, want_mel=False, want_salience=False, taps=None):
        """audio: 1-D float array/tensor (16 kHz).  Returns dict(f0=float64 [n] numpy, mel=[128,n], salience=[n,360])."""
        a = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).to(self.device, torch.float32).contiguous().view(-1)
        L = a.numel()
        n = L // 160 + 1
        f0 = torch.empty(n, dtype=torch.float64, device=self.device)
        mel = torch.empty(128, n, dtype=torch.float32, device=self.device) if want_mel else None
        sal = torch.empty(n, 360, dtype=torch.float32, device=self.device) if want_salience else None
        tp = None
        if taps is not None:
            tp = _lib.RmvpeTaps(*[_lib.ptr(taps.get(nm)) for nm, _ in _lib.RmvpeTaps._fields_])
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_rmvpe_forward(self._h, _lib.current_stream(), _lib.ptr(a), L, float(thred), _lib.ptr(mel),
                                                  _lib.ptr(sal), _lib.ptr(f0), C.byref(tp) if tp is not None else None))
        return {"f0": f0, "mel": mel, "salience": sal}

    def infer_from_audio(self, audio, thred=0.03):
        """reference lib/rmvpe.py:614-623 -> numpy float64 [L // 160 + 1]"""
        f0 = self.infer(audio, thred)["f0"].cpu().numpy()      # audio may already be a device tensor
        self.check_status()
        return f0

    def check_status(self):
        """Raises RvcHipError when the last forward's GRU scan failed (its workgroups poll each other; f0 is NaN then).  Synchronises."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_rmvpe_status(self._h, _lib.current_stream()))

    def infer_from_audio_with_pitch(self, audio, thred=0.03, f0_min=50, f0_max=1100):
        """reference lib/rmvpe.py:649-659 ("rmvpe+"): note the clip turns unvoiced zeros into f0_min, as upstream does."""
        return np.clip(self.infer_from_audio(audio, thred), a_min=f0_min, a_max=f0_max)

    def decode(self, hidden, thred=0.03):
        """reference lib/rmvpe.py:607-612: salience [n, 360] (numpy or tensor) -> f0 [n] float64 numpy."""
        s = torch.as_tensor(np.asarray(hidden) if not torch.is_tensor(hidden) else hidden).to(self.device, torch.float32).contiguous()
        n = s.shape[0]
        f0 = torch.empty(n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_rmvpe_decode(self._h, _lib.current_stream(), _lib.ptr(s), n, float(thred), _lib.ptr(f0)))
        return f0.cpu().numpy()
