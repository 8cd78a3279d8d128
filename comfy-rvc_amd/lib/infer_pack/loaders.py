"""HuBERT / ContentVec feature extractor on the HIP kernel graph.

Drop-in for reference lib/infer_pack/loaders.py:10-61 (`HubertModelWithFinalProj`): same constructor entry
points (`from_safetensors`) and the same `extract_features(source, version=..., **kwargs)` callee protocol that
`VC.vc` uses (reference vc_infer_pipeline.py:51-56).  No transformers / torch.nn modules are involved: the
checkpoint tensors go straight to librvc_hip.so (rvc_hubert_set_tensor) and the forward is rvc_hubert_forward.
"""
import ctypes as C
import json

import torch

from ... import _lib


class HubertModelWithFinalProj:
    def __init__(self, state_dict, config=None, device="cuda:0"):
        self.device = torch.device(device)
        self.config = dict(config or {})
        idx = self.device.index or 0
        self._ctx = _lib.get_ctx(idx)
        h = C.c_void_p()
        _lib.check(_lib.lib.rvc_hubert_create(self._ctx, C.byref(h)))
        self._h = h
        with torch.cuda.device(idx):
            _lib.set_tensors(_lib.lib.rvc_hubert_set_tensor, h, state_dict)
            _lib.check(_lib.lib.rvc_hubert_finalize(h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, 'lib', None) is not None:   # (module may be torn down at exit)
            _lib.lib.rvc_hubert_destroy(h)
            self._h = None

    @staticmethod
    def from_safetensors(path: str, device="cuda:0", framework="pt"):
        """reference lib/infer_pack/loaders.py:19-31 (safetensors file with the HubertConfig JSON in its metadata)."""
        assert path.endswith(".safetensors"), f"{path} must end with '.safetensors'"
        from safetensors import safe_open
        with safe_open(path, framework=framework, device="cpu") as f:
            metadata = f.metadata() or {}
            state_dict = {k: f.get_tensor(k) for k in f.keys()}
        cfg = json.loads(metadata["config"]) if "config" in metadata else {}
        return HubertModelWithFinalProj(state_dict, cfg, device=device)

    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device) == self.device, "weights are resident on the device given at construction"
        return self

    @staticmethod
    def num_frames(n_samples: int) -> int:
        return int(_lib.lib.rvc_hubert_num_frames(int(n_samples)))

    def extract_features(self, source: torch.Tensor, version="v2", channel_major=False, n_layers=0, taps=None, **kwargs):
        """source: float tensor [1, L] -> [1, T_h, 768] (v2: hidden_states[11]) or [1, T_h, 256] (v1: final_proj(hidden_states[8])).

        `padding_mask` / `output_layer` kwargs are accepted and ignored exactly like the reference (loaders.py:55).
        channel_major=True returns [D, T_h] instead (what the synthesizer graph consumes, saving two transposes).
        """
        assert source.dim() == 2 and source.shape[0] == 1, "batch-1 like the reference (vc_infer_pipeline.py:48)"
        src = source.to(self.device, torch.float32).contiguous()
        L = src.shape[1]
        Th = self.num_frames(L)
        D = 256 if version == "v1" else 768
        out = torch.empty((D, Th) if channel_major else (1, Th, D), dtype=torch.float32, device=self.device)
        tp = None
        if taps is not None:
            tp = _lib.HubertTaps(*[_lib.ptr(taps.get(n)) for n, _ in _lib.HubertTaps._fields_])
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.rvc_hubert_forward(self._h, _lib.current_stream(), _lib.ptr(src), L, 1 if version == "v1" else 2,
                                                   int(n_layers), None if channel_major else _lib.ptr(out),
                                                   _lib.ptr(out) if channel_major else None, C.byref(tp) if tp is not None else None))
        return out.to(source.dtype) if source.dtype.is_floating_point else out
