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


def test_feature_input_dump_matches_reference_golden(tmp_path):
    """The files FeatureInput.go writes against the files the REFERENCE's FeatureInput.go wrote for the same clips (tests/golden/featinput.npz,
    generated in the build container with load_input_audio stubbed; reference preprocessing_utils.py:155-193): names, dtypes, shapes, values."""
    from scipy.io import wavfile
    from conftest import golden
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    from comfy_rvc_amd.preprocessing_utils import FeatureInput
    g = golden("featinput.npz")
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    rm = RMVPE(S.rmvpe_state_dict(0))
    for version, D in (("v2", 768), ("v1", 256)):
        d = tmp_path / version
        for sub in ("2a_f0", "2b-f0nsf", f"3_feature{D}"):
            (d / sub).mkdir(parents=True)
        paths = []
        for i, (secs, seed) in enumerate(zip(g["seconds"], g["seeds"])):
            name = f"0_{i}.wav"
            wavfile.write(str(d / name), 16000, S.synth_audio(float(secs), seed=int(seed)).astype(np.float32))      # IEEE-float WAV: the samples survive exactly
            paths.append((str(d / name), str(d / "2a_f0" / name), str(d / "2b-f0nsf" / name), str(d / f"3_feature{D}" / name)))
        fi = FeatureInput(hub, "rmvpe", str(d), version=version, if_f0=True)
        fi.model_rmvpe = rm
        assert fi.go(paths) == 2 and fi.go(paths) == 0
        for i, (_, p1, p2, p3) in enumerate(paths):
            coarse, nsf, feat = np.load(p1 + ".npy"), np.load(p2 + ".npy"), np.load(p3 + ".npy")
            rc, rn, rf = g[f"{version}_0_{i}_coarse"], g[f"{version}_0_{i}_nsf"], g[f"{version}_0_{i}_feat"]
            assert (coarse.dtype, nsf.dtype, feat.dtype) == (rc.dtype, rn.dtype, rf.dtype) and (coarse.shape, nsf.shape, feat.shape) == (rc.shape, rn.shape, rf.shape)
            assert np.abs(feat - rf).max() <= 1e-3 * np.abs(rf).max()
            assert np.array_equal(nsf > 0, rn > 0) and np.allclose(nsf, rn, rtol=1e-3, atol=1e-3)
            dc = np.abs(coarse.astype(int) - rc.astype(int))
            record_parity(f"feature_dump_golden_{version}_{i}", {"frames": int(rc.shape[0]), "coarse_equal": float((dc == 0).mean()), "coarse_max_diff": int(dc.max())})
            assert dc.max() <= 1 and (dc == 0).mean() > 0.99
        log = open(str(d / "extract_f0_feature.log")).read().split("\n")
        ref_log = bytes(g[f"{version}_log"]).decode().split("\n")
        assert log[:2] == ref_log[:2] == ["todo-f0-2", "todo-f0-2"]
