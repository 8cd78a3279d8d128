"""GPU: the training-prep feature dump (FeatureInput) against the CPU oracle and its on-disk formats."""
import os

import numpy as np
import pytest
import torch

from conftest import f0_frames_ok, record_parity
from comfy_rvc_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_feature_input_dump_matches_oracle(tmp_path):
    from scipy.io import wavfile
    from comfy_rvc_amd.lib.audio import hz_to_mel
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    from comfy_rvc_amd.preprocessing_utils import FeatureInput
    from oracle import nets
    hsd, rsd = S.hubert_state_dict(0), S.rmvpe_state_dict(0)
    hub = HubertModelWithFinalProj(hsd, S.HUBERT_CONFIG)
    fi = FeatureInput(hub, "rmvpe", str(tmp_path), version="v2", if_f0=True)
    fi.model_rmvpe = RMVPE(rsd)
    paths = []
    for i in range(3):
        x = S.synth_audio(1.5 + 0.5 * i, seed=30 + i)
        wav = str(tmp_path / f"clip{i}.wav")
        wavfile.write(wav, 16000, (x * 32767.0).astype(np.int16))
        paths.append((wav, str(tmp_path / f"2a_f0_{i}"), str(tmp_path / f"2b_f0nsf_{i}"), str(tmp_path / f"3_feature768_{i}")))
    assert fi.go(paths) == 3
    assert fi.go(paths) == 0                       # everything present: skipped, like the reference
    for i, (wav, p1, p2, p3) in enumerate(paths):
        _, data = wavfile.read(wav)
        x = data.astype(np.float32) / 32768.0
        feats, coarse, nsf = np.load(p3 + ".npy"), np.load(p1 + ".npy"), np.load(p2 + ".npy")
        ref = nets.hubert_extract_features(hsd, torch.from_numpy(x)[None], "v2")[0].numpy()
        assert feats.dtype == np.float32 and feats.shape == ref.shape
        assert np.abs(feats - ref).max() <= 1e-3 * np.abs(ref).max()
        taps = {}
        f0 = nets.rmvpe_infer_from_audio(rsd, x, thred=0.03, taps=taps)
        assert coarse.dtype == np.int16 and nsf.dtype == np.float64 and nsf.shape == f0.shape
        # every frame equal (voicing included) unless the reference's own arg-max / voicing threshold is a tie within the salience noise between
        # the two implementations (5e-5 measured by the RMVPE model tests; conftest.f0_frames_ok) - a maximum gate, not a percentile
        n_bad, unexplained = f0_frames_ok(nsf, f0, taps["salience"], 5e-5, rtol=1e-3)
        record_parity(f"feature_dump_f0_{i}", {"f0_frames": int(f0.shape[0]), "f0_frames_differing": n_bad, "unexplained": unexplained})
        assert unexplained == 0, (i, n_bad, unexplained)
        mel = (hz_to_mel(nsf) - hz_to_mel(50.0)) * 254 / (hz_to_mel(1100.0) - hz_to_mel(50.0)) + 1   # training prep quantises with f0_max = 1100
        assert np.array_equal(coarse, np.rint(np.clip(mel, 1, 255)).astype(np.int16))
    assert os.path.exists(str(tmp_path / "extract_f0_feature.log"))
