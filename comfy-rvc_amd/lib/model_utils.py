"""load_hubert / change_rms (mirror of reference lib/model_utils.py:19-57)."""
import numpy as np
import torch
import torch.nn.functional as F

from .infer_pack.loaders import HubertModelWithFinalProj


def load_hubert(model_path, config):
    """Returns the HIP-backed HuBERT, or None on failure like the reference (lib/model_utils.py:19-37)."""
    try:
        if isinstance(model_path, dict):
            return HubertModelWithFinalProj(model_path, device=config.device)
        if str(model_path).endswith(".safetensors"):
            return HubertModelWithFinalProj.from_safetensors(model_path, device=config.device)
        raise NotImplementedError("Please use content-vec-best.safetensors!")
    except Exception as e:   # noqa: BLE001 - the reference prints and returns None
        print(e)
        return None


def _frame_rms(y, frame_length, hop_length):
    """RMS per frame with zero centre-padding: the librosa.feature.rms definition the reference relies on."""
    y = np.pad(np.asarray(y), int(frame_length // 2), mode="constant")
    n_frames = 1 + (y.shape[-1] - frame_length) // hop_length
    # frames as columns, mean over the frame axis in the input dtype: same accumulation order as librosa's framed view
    # (a cumulative-sum shortcut would change the float32 rounding)
    cols = hop_length * np.arange(n_frames)[None, :] + np.arange(frame_length)[:, None]
    return np.sqrt(np.mean(np.abs(y[cols]) ** 2, axis=-2, keepdims=True))


def change_rms(data1, sr1, data2, sr2, rate):
    """Blend the RMS envelope of the input (data1) into the output (data2), in place (reference lib/model_utils.py:39-57)."""
    rms1 = torch.from_numpy(_frame_rms(data1, sr1 // 2 * 2, sr1 // 2))
    rms2 = torch.from_numpy(_frame_rms(data2, sr2 // 2 * 2, sr2 // 2))
    rms1 = F.interpolate(rms1.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(rms2.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    data2 *= (torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))).numpy()
    return data2
