"""ComfyUI node surface of the RVC inference path (drop-in for the four inference nodes of reference
custom_nodes/rvc_nodes.py:44-206; dataset/training nodes, downloads and audio codecs are out of scope).

Node names, categories, INPUT_TYPES / RETURN_TYPES / FUNCTION and the tuple protocol between nodes follow the reference:
  LoadPitchExtractionParams -> ('PITCH_EXTRACTION',)   = the kwargs dict itself
  LoadHubertModel           -> ('HUBERT_MODEL',)       = zero-argument thunk returning the model
  LoadRVCModelNode          -> ('RVC_MODEL', 'STRING') = thunk returning get_vc(...)'s dict, model name
  RVCNode.convert           -> {"ui": ..., "result": (VHS_AUDIO thunk, AUDIO dict {"waveform": [1, N, C], "sample_rate"})}
The reference re-loads the weights on every execution because the thunks are not cached (rvc_nodes.py:191-192); here they
are memoised per (path, mtime) so repeated graph runs keep the weights resident in HBM.
"""
import hashlib
import os

import numpy as np
import torch

from ..config import config
from ..lib import BASE_MODELS_DIR
from ..lib.audio import audio_to_bytes, get_audio
from ..lib.model_utils import load_hubert
from ..vc_infer_pipeline import get_vc, vc_single

CATEGORY = "🌺RVC-Studio/rvc"
PITCH_EXTRACTION_OPTIONS = ["crepe", "mangio-crepe", "rmvpe", "rmvpe+"]
SUPPORTED_AUDIO = ["mp3", "flac", "wav"]


class MultipleTypeProxy(str):
    """Socket type that matches any of several comma-separated ComfyUI types (reference custom_nodes/utils.py:32-41)."""

    def __eq__(self, other):
        mine, theirs = set(self.split(",")), set(str(other).split(","))
        return bool(mine & theirs) or str(self) == "*"

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = str.__hash__


def to_audio_dict(audio, sr):
    """ndarray [N] or [C, N] -> {"waveform": tensor [1, N, C], "sample_rate"} (reference custom_nodes/audio_nodes.py:17-20)."""
    audio = np.atleast_2d(audio)
    return dict(waveform=torch.from_numpy(audio.reshape((-1, audio.shape[0]))).unsqueeze(0), sample_rate=sr)


def _list_models(folder, exts):
    root = os.path.join(BASE_MODELS_DIR, folder)
    if not os.path.isdir(root):
        return []
    return sorted(f for f in os.listdir(root) if f.rsplit(".", 1)[-1] in exts)


_memo = {}


def _memoised(kind, path, loader, extra=None):
    mt = lambda f: os.path.getmtime(f) if f and os.path.isfile(f) else None   # noqa: E731
    key = (kind, path, mt(path), extra, mt(extra))
    if key not in _memo:
        _memo[key] = loader()
    return _memo[key]


class LoadPitchExtractionParams:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "f0_method": (PITCH_EXTRACTION_OPTIONS, {"default": "rmvpe"}),
            "f0_autotune": ("BOOLEAN",),
            "index_rate": ("FLOAT", {"default": .75, "min": 0., "max": 1., "step": .01}),
            "resample_sr": ([0, 16000, 32000, 40000, 44100, 48000], {"default": 0}),
            "rms_mix_rate": ("FLOAT", {"default": 0.25, "min": 0., "max": 1., "step": .01}),
            "protect": ("FLOAT", {"default": 0.25, "min": 0., "max": .5, "step": .01}),
            "crepe_hop_length": ("INT", {"default": 160, "min": 16, "max": 512, "step": 16}),
        }}

    RETURN_TYPES = ("PITCH_EXTRACTION",)
    RETURN_NAMES = ("pitch_extraction_params",)
    CATEGORY = CATEGORY
    FUNCTION = "load_params"

    def load_params(self, **params):
        return (params,)


class LoadHubertModel:
    @classmethod
    def INPUT_TYPES(cls):
        models = sorted(set(["content-vec-best.safetensors"] + _list_models(".", ("pt", "safetensors"))))
        return {"required": {"model": (models, {"default": "content-vec-best.safetensors"})}}

    RETURN_TYPES = ("HUBERT_MODEL",)
    RETURN_NAMES = ("hubert_model",)
    CATEGORY = CATEGORY
    FUNCTION = "load_model"

    def load_model(self, model):
        path = os.path.join(BASE_MODELS_DIR, model)
        return (lambda: _memoised("hubert", path, lambda: load_hubert(path, config=config)),)


class LoadRVCModelNode:
    @classmethod
    def INPUT_TYPES(cls):
        models = [f"RVC/{m}" for m in _list_models("RVC", ("pth",))] or [""]
        # faiss .index files (readable when faiss is installed) and big_npy .npy matrices (the vectors the index is built from)
        index = [""] + [f"RVC/.index/{m}" for m in _list_models(os.path.join("RVC", ".index"), ("index", "npy"))]
        return {"required": {"model": (models, {"default": models[0]})}, "optional": {"index": (index, {"default": ""})}}

    RETURN_TYPES = ("RVC_MODEL", "STRING")
    RETURN_NAMES = ("model", "model_name")
    CATEGORY = CATEGORY
    FUNCTION = "load_model"

    def load_model(self, model, index=""):
        path = os.path.join(BASE_MODELS_DIR, os.path.dirname(model), os.path.basename(model))
        file_index = os.path.join(BASE_MODELS_DIR, os.path.dirname(model), ".index", os.path.basename(index)) if index else None
        return (lambda: _memoised("rvc", path, lambda: get_vc(path, file_index), extra=file_index), os.path.basename(model).split(".")[0])


class _ByteLRU:
    """Result cache of RVCNode bounded by BYTES (upstream caches on disk by file name, reference rvc_nodes.py:176-183; a long-lived ComfyUI server would
    otherwise keep 2.4 MB per distinct 30 s conversion forever): least recently used entries go first, one entry larger than the bound is not kept."""

    def __init__(self, max_bytes):
        from collections import OrderedDict
        self.max_bytes, self.bytes, self._d = int(max_bytes), 0, OrderedDict()

    def __contains__(self, key):
        return key in self._d

    def __len__(self):
        return len(self._d)

    def get(self, key):
        self._d.move_to_end(key)
        return self._d[key]

    def put(self, key, value):
        n = int(np.asarray(value[0]).nbytes)
        if key in self._d:
            self.bytes -= int(np.asarray(self._d.pop(key)[0]).nbytes)
        if n > self.max_bytes:
            return
        self._d[key] = value
        self.bytes += n
        while self.bytes > self.max_bytes:
            _, old = self._d.popitem(last=False)
            self.bytes -= int(np.asarray(old[0]).nbytes)


class RVCNode:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "audio": (MultipleTypeProxy("AUDIO,VHS_AUDIO"),),
            "model": ("RVC_MODEL",),
            "hubert_model": ("HUBERT_MODEL",),
            "pitch_extraction_params": ("PITCH_EXTRACTION",),
            "f0_up_key": ("INT", {"default": 0, "min": -14, "max": 14, "step": 1, "display": "slider"}),
        }, "optional": {"format": (SUPPORTED_AUDIO, {"default": "flac"}), "use_cache": ("BOOLEAN", {"default": True})}}

    OUTPUT_NODE = True
    RETURN_TYPES = ("VHS_AUDIO", "AUDIO")
    FUNCTION = "convert"
    CATEGORY = CATEGORY
    CACHE_BYTES = 256 << 20                        # ~ 100 conversions of 30 s at 40 kHz int16
    _cache = _ByteLRU(CACHE_BYTES)

    def convert(self, audio, model, hubert_model, pitch_extraction_params, f0_up_key, format="flac", use_cache=True):
        input_audio = get_audio(audio)
        voice_model = model()
        feature_model = hubert_model()
        h = hashlib.md5()
        for part in (feature_model.__class__.__name__, voice_model.get("model_name"), str(voice_model.get("file_index")), f0_up_key,
                     sorted(pitch_extraction_params.items())):
            h.update(str(part).encode())
        h.update(np.ascontiguousarray(input_audio[0]).tobytes())
        widget_id = h.hexdigest()
        if use_cache and widget_id in self._cache:
            output_audio = self._cache.get(widget_id)
        else:
            output_audio = vc_single(hubert_model=feature_model, input_audio=input_audio, f0_up_key=f0_up_key, **voice_model,
                                     **pitch_extraction_params)
            if output_audio is None:
                raise RuntimeError("voice conversion failed (vc_single returned None; see the message printed above)")
            if use_cache:
                self._cache.put(widget_id, output_audio)
        wav, sr = output_audio
        audio_name = f"{widget_id}.{format}"
        ui = {"preview": [{"filename": audio_name, "type": "temp", "subfolder": "preview", "widgetId": widget_id}]}
        # VHS_AUDIO = thunk returning the encoded stream (reference rvc_nodes.py:206: `lambda: audio_to_bytes(*output_audio)`, always WAV:
        # PCM_16 for the int16 result).  `format` only names the preview / cache file upstream; no file is written here (file codecs are
        # out of scope), so flac / mp3 previews are not produced.
        return {"ui": ui, "result": (lambda: audio_to_bytes(wav, sr), to_audio_dict(wav, sr))}


NODE_CLASS_MAPPINGS = {
    "LoadRVCModelNode": LoadRVCModelNode,
    "RVCNode": RVCNode,
    "LoadHubertModel": LoadHubertModel,
    "LoadPitchExtractionParams": LoadPitchExtractionParams,
}
NODE_DISPLAY_NAME_MAPPINGS = {
    "LoadRVCModelNode": "🌺Load RVC Model",
    "RVCNode": "🌺Voice Changer",
    "LoadHubertModel": "🌺Load Hubert Model",
    "LoadPitchExtractionParams": "🌺Load Pitch Extraction Params",
}
