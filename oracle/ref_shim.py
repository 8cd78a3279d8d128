"""Container-only import shim for the upstream reference (TEST INFRASTRUCTURE, never shipped, never imported by the product).

The reference (`/root/reference`, SayanoAI/Comfy-RVC) is a ComfyUI node pack whose top-level
`__init__.py` needs ComfyUI.  This module builds a scratch package under a temp directory that
symlinks only the hot-path files (vc_infer_pipeline.py, pitch_extraction.py, config.py, lib/, i18n/)
behind an empty `__init__.py`, stubs the third-party modules that are absent in this container
(librosa, soundfile, ffmpeg, monotonic_align) and returns the imported reference modules.

It is used ONLY by `oracle/gen_golden.py` (to produce `tests/golden/*.npz`) and by the
container-only oracle-vs-reference checks.  `/root/reference` does not exist on the GPU box, so
nothing reachable from `-m gpu` tests, `smoke()` or `bench.py` imports this file.

The stubbed `librosa.filters.mel` / `librosa.feature.rms` / `librosa.util.*` are restatements of
librosa 0.10.2 (reference `requirements.txt:1`) and cannot be checked against the real package
here ("parity unpinned" for those two formulas beyond their published definitions).
"""
import os
import sys
import types
import shutil
import importlib
import tempfile

import numpy as np

REF_ROOT = os.environ.get("RVC_REFERENCE_ROOT", "/root/reference")
PKG = "rvcref"
_state = {}


def reference_available():
    return os.path.isdir(REF_ROOT) and os.path.isfile(os.path.join(REF_ROOT, "vc_infer_pipeline.py"))


# ----------------------------------------------------------------------------- librosa stand-ins
def _hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asanyarray(f, dtype=np.float64) / 700.0)


def _mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asanyarray(m, dtype=np.float64) / 2595.0) - 1.0)


def librosa_mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """librosa.filters.mel (0.10.2) for htk=True / norm='slaney' (the only use: lib/rmvpe.py:492-499)."""
    assert htk, "only the HTK scale is used by the reference"
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz_htk(np.linspace(_hz_to_mel_htk(fmin), _hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == "slaney":
        enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights


def librosa_pad_center(data, *, size, axis=-1, **kwargs):
    kwargs.setdefault("mode", "constant")
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, **kwargs)


def librosa_tiny(x):
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating) or np.issubdtype(x.dtype, np.complexfloating):
        dtype = x.dtype
    else:
        dtype = np.dtype(np.float32)
    return np.finfo(dtype).tiny


def librosa_normalize(S, *, norm=np.inf, axis=0, threshold=None, fill=None):
    if norm is None:
        return S
    mag = np.abs(S).astype(float)
    if norm == np.inf:
        length = np.max(mag, axis=axis, keepdims=True)
    else:
        raise NotImplementedError(norm)
    small = length < (librosa_tiny(S) if threshold is None else threshold)
    length[small] = 1.0
    return S / length


def librosa_rms(*, y, frame_length=2048, hop_length=512, center=True, pad_mode="constant"):
    """librosa.feature.rms (0.10.2): centre-pad, frame, sqrt(mean(|x|^2)); returns [1, n_frames]."""
    y = np.asarray(y)
    if center:
        y = np.pad(y, int(frame_length // 2), mode=pad_mode)
    n_frames = 1 + (y.shape[-1] - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    x = y[idx]
    power = np.mean(np.abs(x) ** 2, axis=-2, keepdims=True)
    return np.sqrt(power)


def _install_stubs():
    librosa = types.ModuleType("librosa")
    librosa.__path__ = []
    filters = types.ModuleType("librosa.filters")
    filters.mel = librosa_mel
    util = types.ModuleType("librosa.util")
    util.pad_center = librosa_pad_center
    util.tiny = librosa_tiny
    util.normalize = librosa_normalize

    def _fix_length(data, *, size, axis=-1, **kw):
        n = data.shape[axis]
        if n > size:
            sl = [slice(None)] * data.ndim
            sl[axis] = slice(0, size)
            return data[tuple(sl)]
        if n < size:
            lengths = [(0, 0)] * data.ndim
            lengths[axis] = (0, size - n)
            return np.pad(data, lengths, **kw)
        return data

    util.fix_length = _fix_length
    util.stack = lambda arrays, axis=0: np.stack(arrays, axis=axis)
    feature = types.ModuleType("librosa.feature")
    feature.rms = librosa_rms

    def _resample(*a, **k):
        raise NotImplementedError("librosa.resample has no faithful stand-in (SURVEY 8c): parity unpinned")

    librosa.resample = _resample
    librosa.filters, librosa.util, librosa.feature = filters, util, feature
    for name, mod in (("librosa", librosa), ("librosa.filters", filters), ("librosa.util", util),
                      ("librosa.feature", feature)):
        sys.modules[name] = mod
    for name in ("soundfile", "ffmpeg", "monotonic_align"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


def load_reference():
    """Returns a namespace with the reference's hot-path modules (imported once per process)."""
    if "ns" in _state:
        return _state["ns"]
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    # transformers must be imported before a fake librosa is visible (it probes for librosa+soxr)
    from transformers import HubertModel, HubertConfig  # noqa: F401

    ws = tempfile.mkdtemp(prefix="rvc_oracle_ws_")
    pkgdir = os.path.join(ws, PKG)
    os.makedirs(pkgdir)
    open(os.path.join(pkgdir, "__init__.py"), "w").close()
    for name in ("lib", "config.py", "vc_infer_pipeline.py", "pitch_extraction.py", "i18n"):
        os.symlink(os.path.join(REF_ROOT, name), os.path.join(pkgdir, name))
    # config.py rewrites configs/*.json in the CWD on CPU -> give it a private writable copy
    shutil.copytree(os.path.join(REF_ROOT, "configs"), os.path.join(ws, "configs"))
    os.makedirs(os.path.join(ws, "models"), exist_ok=True)
    _install_stubs()
    old_cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(ws)
    sys.argv = [old_argv[0]]
    sys.path.insert(0, ws)
    try:
        ns = types.SimpleNamespace()
        ns.ws = ws
        ns.config_mod = importlib.import_module(f"{PKG}.config")
        ns.models = importlib.import_module(f"{PKG}.lib.infer_pack.models")
        ns.modules = importlib.import_module(f"{PKG}.lib.infer_pack.modules")
        ns.attentions = importlib.import_module(f"{PKG}.lib.infer_pack.attentions")
        ns.loaders = importlib.import_module(f"{PKG}.lib.infer_pack.loaders")
        ns.rmvpe = importlib.import_module(f"{PKG}.lib.rmvpe")
        ns.audio = importlib.import_module(f"{PKG}.lib.audio")
        ns.model_utils = importlib.import_module(f"{PKG}.lib.model_utils")
        ns.pitch_extraction = importlib.import_module(f"{PKG}.pitch_extraction")
        ns.vc_infer_pipeline = importlib.import_module(f"{PKG}.vc_infer_pipeline")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    _state["ns"] = ns
    return ns


def load_preprocessing_utils():
    """The reference's preprocessing_utils (FeatureInput), imported on demand: it pulls lib/slicer2 and lib/audio's
    AudioProcessor, which the inference goldens never need."""
    ns = load_reference()
    if not hasattr(ns, "preprocessing_utils"):
        link = os.path.join(ns.ws, PKG, "preprocessing_utils.py")
        if not os.path.exists(link):
            os.symlink(os.path.join(REF_ROOT, "preprocessing_utils.py"), link)
        old = os.getcwd()
        os.chdir(ns.ws)
        try:
            ns.preprocessing_utils = importlib.import_module(f"{PKG}.preprocessing_utils")
        finally:
            os.chdir(old)
    return ns.preprocessing_utils


class chdir_ws:
    """Context manager: run reference code with CWD = the scratch workspace (models/, configs/)."""

    def __enter__(self):
        self.old = os.getcwd()
        os.chdir(load_reference().ws)

    def __exit__(self, *a):
        os.chdir(self.old)
