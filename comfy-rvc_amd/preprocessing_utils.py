"""Training-prep feature dump on the HIP path: mirror of the reference's `FeatureInput`
(preprocessing_utils.py:102-193).

For every sliced 16 kHz training clip it writes the HuBERT features (`3_feature768/<name>.npy`, float32 [T_h, 768] for v2 /
`3_feature256`, [T_h, 256] for v1), the coarse pitch (`2a_f0/<name>.npy`, int16 [n]) and the NSF pitch (`2b-f0nsf/<name>.npy`,
float64 [n]) - same arrays, dtypes and skip-if-present rule as the reference.  Differences: the networks are this build's HIP
graphs; `go` can shard the file list over the ranks of an initialised process group (files are independent: clip i -> rank
i mod N, no collective); audio files are read with scipy (PCM / float WAV) because soundfile / librosa / ffmpeg are not available
offline - other containers raise.  A WAV at another rate is resampled to 16 kHz on load like the reference does (preprocessing_utils.py:171 ->
load_input_audio(path, 16000) -> librosa.resample), here with the device polyphase kernel of lib/audio.py::resample_audio - PARITY-UNPINNED like the
two other resampling branches (librosa / soxr absent: the kernel is pinned to its own float64 definition, tests/test_hip_ops.py).
Note the reference's quirk that training prep quantises the pitch with f0_max = 1100 Hz (get_f0's default) while inference uses
1600 Hz (vc_infer_pipeline.py:118).
"""
import os
import traceback

import numpy as np
import torch

from .config import Config
from .lib.audio import hz_to_mel
from .pitch_extraction import FeatureExtractor


def load_wav(path, sr, device="cuda:0"):
    """float32 mono / stereo samples in [-1, 1) of a WAV file at the working rate `sr` (resampled on the device when the file has another rate)."""
    from scipy.io import wavfile   # noqa: PLC0415
    rate, data = wavfile.read(path)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if rate != sr:
        from .lib.audio import resample_audio   # noqa: PLC0415
        x = resample_audio(x.T if x.ndim > 1 else x, rate, sr, device=device)       # along the last axis, like librosa.resample
        x = np.ascontiguousarray(x.T) if x.ndim > 1 else x
    return x, sr


class FeatureInput(FeatureExtractor):
    def __init__(self, model, f0_method, exp_dir, samplerate=16000, hop_size=160, device="cuda:0", version="v2", if_f0=False, config=None):
        self.sr = samplerate
        self.hop = hop_size
        self.f0_method = f0_method
        self.exp_dir = exp_dir
        self.version = version
        self.if_f0 = if_f0
        self.f0_bin = 256
        self.f0_max = 1100.0
        self.f0_min = 50.0
        self.f0_mel_min = hz_to_mel(self.f0_min)
        self.f0_mel_max = hz_to_mel(self.f0_max)
        self.model = model
        super().__init__(samplerate, config if config is not None else Config(device=device), onnx=False)
        self.device = device

    def printt(self, strr):
        print(strr)
        if self.exp_dir:
            with open("%s/extract_f0_feature.log" % self.exp_dir, "a+") as f:
                f.write("%s\n" % strr)
                f.flush()

    def compute_feats(self, x):
        feats = torch.from_numpy(np.asarray(x)).float()
        if feats.dim() == 2:  # double channels
            feats = feats.mean(-1)
        assert feats.dim() == 1, feats.dim()
        feats = feats.view(1, -1)
        feats = self.model.extract_features(version=self.version, source=feats, padding_mask=None,
                                            output_layer=9 if self.version == "v1" else 12)
        feats = feats.squeeze(0).float().cpu().numpy()
        if np.isnan(feats).sum() == 0:
            return feats
        return self.printt("==contains nan==")

    def compute_f0(self, x):
        return self.get_f0(x, 0, self.f0_method, crepe_hop_length=self.hop)

    def go(self, paths, shard=True):
        """paths: [(wav, coarse_f0_out, nsf_f0_out, feature_out), ...] (output paths without the .npy suffix).  With an initialised
        torch.distributed process group and shard=True every rank handles paths[rank::world]."""
        if shard and torch.distributed.is_available() and torch.distributed.is_initialized():
            paths = paths[torch.distributed.get_rank()::torch.distributed.get_world_size()]
        if len(paths) == 0:
            self.printt("no-f0-todo")
            return 0
        self.printt("todo-f0-%s" % len(paths))
        done = 0
        for idx, (inp_path, opt_path1, opt_path2, opt_path3) in enumerate(paths):
            try:
                if os.path.exists(opt_path1 + ".npy") and os.path.exists(opt_path2 + ".npy") and os.path.exists(opt_path3 + ".npy"):
                    continue
                x, _ = load_wav(inp_path, self.sr, self.device)
                if self.model:
                    feats = self.compute_feats(x)
                    if feats is not None:
                        np.save(opt_path3, feats, allow_pickle=False)          # features
                        if self.if_f0:                                           # uses pitch
                            coarse_pit, featur_pit = self.compute_f0(x if x.ndim == 1 else x.mean(-1))
                            np.save(opt_path2, featur_pit, allow_pickle=False)  # nsf
                            np.save(opt_path1, coarse_pit, allow_pickle=False)  # ori
                        done += 1
            except Exception:   # noqa: BLE001 - reference behaviour: log and continue
                self.printt("f0fail-%s-%s-%s" % (idx, inp_path, traceback.format_exc()))
        return done
