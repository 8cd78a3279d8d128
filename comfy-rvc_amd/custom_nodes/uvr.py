"""ComfyUI node of the vocal / instrumental split that precedes voice conversion in BASELINE config C5 (mirror of reference
custom_nodes/uvr.py:16-100 for the karafan MDX23C models).  Socket protocol as upstream: AUDIO or VHS_AUDIO in, two VHS_AUDIO thunks
(primary stem = vocals, secondary stem = instrumental) out.  Only models the reference routes to karafan's TFC_TDF_net are handled
here ("karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt"); the python-audio-separator / VR branches are third-party packages upstream and raise."""
import os

import numpy as np
import torch
import yaml

from ..lib import BASE_MODELS_DIR
from ..lib.audio import audio_to_bytes, get_audio, resample_audio
from ..lib.karafan.inference import demix_mdxv3
from ..lib.karafan.tfc_tdf import TFC_TDF_net
from .rvc_nodes import MultipleTypeProxy, _memoised

KARAFAN_MODELS = ["karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt"]
# the fields of reference lib/karafan/Data/model_2_stem_full_band_8k.yaml that the network and demix_mdxv3 read
MDX23C_CONFIG = {"audio": {"chunk_size": 261120, "dim_f": 4096, "dim_t": 256, "hop_length": 1024, "n_fft": 8192, "num_channels": 2, "sample_rate": 44100},
                 "model": {"act": "gelu", "bottleneck_factor": 4, "growth": 128, "norm": "InstanceNorm", "num_blocks_per_scale": 2, "num_channels": 128,
                           "num_scales": 5, "num_subbands": 4, "scale": [2, 2]},
                 "training": {"instruments": ["Vocals", "Instrumental"], "target_instrument": None},
                 "inference": {"batch_size": 1, "dim_t": 256, "num_overlap": 8}}


def load_mdx23c(model_path, config=None):
    cfg = config
    if cfg is None:
        side = os.path.splitext(model_path)[0] + ".yaml"            # a config next to the checkpoint wins over the built-in recipe
        cfg = yaml.safe_load(open(side)) if os.path.isfile(side) else MDX23C_CONFIG
    net = TFC_TDF_net(cfg)
    net.load_state_dict(torch.load(model_path, map_location="cpu"))
    net.set_streams(3)          # the node converts one clip at a time: its chunks alternate over three streams (one clip 0.86 -> 0.70 s)
    return net, cfg


class UVR5Node:
    @classmethod
    def INPUT_TYPES(cls):
        root = os.path.join(BASE_MODELS_DIR, "karafan")
        found = [f"karafan/{f}" for f in sorted(os.listdir(root))] if os.path.isdir(root) else []
        models = sorted(set(KARAFAN_MODELS + [f for f in found if f.endswith((".ckpt", ".pth"))]))
        return {"required": {"audio": (MultipleTypeProxy("AUDIO,VHS_AUDIO"),), "model": (models, {"default": models[0]})},
                "optional": {"use_cache": ("BOOLEAN", {"default": True}),
                             "agg": ("INT", {"default": 10, "min": 0, "max": 20, "step": 1, "display": "slider"}),
                             "format": (["wav", "flac", "mp3"], {"default": "flac"})}}

    RETURN_TYPES = ("VHS_AUDIO", "VHS_AUDIO")
    RETURN_NAMES = ("primary_stem", "secondary_stem")
    FUNCTION = "split"
    CATEGORY = "🌺RVC-Studio/uvr"

    def split(self, audio, model, use_cache=True, agg=10, format="flac", overlap=None):
        if "karafan" not in model:
            raise NotImplementedError(f"{model}: only the karafan MDX23C models run on this build (VR / MDX-Net ONNX models need audio_separator)")
        path = os.path.join(BASE_MODELS_DIR, os.path.dirname(model), os.path.basename(model))
        net, cfg = _memoised("mdx23c", path, lambda: load_mdx23c(path))
        wav, sr = get_audio(audio)
        wav = np.atleast_2d(np.asarray(wav, dtype=np.float32))
        if wav.shape[0] == 1:
            wav = np.repeat(wav, 2, axis=0)                         # the network separates stereo; mono is duplicated as upstream's loader does
        msr = cfg["audio"]["sample_rate"]
        if int(sr) != msr:
            wav = resample_audio(wav, sr, msr)
        est = demix_mdxv3(wav[:2], net, net.device, cfg, overlap or cfg["inference"]["num_overlap"])
        vocals, music = est["Vocals"], est["Instrumental"]
        return (lambda: audio_to_bytes(vocals, msr), lambda: audio_to_bytes(music, msr))


NODE_CLASS_MAPPINGS = {"UVR5Node": UVR5Node}
NODE_DISPLAY_NAME_MAPPINGS = {"UVR5Node": "🌺Vocal Removal"}
