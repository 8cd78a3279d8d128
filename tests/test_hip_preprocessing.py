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
            # equal on every frame, or an EXPLAINED tie: the reference's own mel value sits closer to a rounding boundary (x.5) than a 1e-3 relative change
            # of its f0 can move it - no percentile (measured on MI355X: 0 differing frames of 804)
            from comfy_rvc_amd.lib.audio import hz_to_mel
            scale = 254.0 / (hz_to_mel(1100.0) - hz_to_mel(50.0))
            mel_ref = (hz_to_mel(rn) - hz_to_mel(50.0)) * scale + 1
            reach = 1e-3 * (2595.0 / np.log(10.0)) * (rn / (700.0 + rn)) * scale                # |d mel| for |d f0| / f0 = 1e-3
            near_boundary = np.abs(np.abs(mel_ref - np.floor(mel_ref)) - 0.5) <= reach
            unexplained = int(((dc != 0) & ~((dc == 1) & near_boundary)).sum())
            record_parity(f"feature_dump_golden_{version}_{i}", {"frames": int(rc.shape[0]), "coarse_differing": int((dc != 0).sum()), "unexplained": unexplained})
            assert unexplained == 0, (version, i, int((dc != 0).sum()), int(dc.max()))
        log = open(str(d / "extract_f0_feature.log")).read().split("\n")
        ref_log = bytes(g[f"{version}_log"]).decode().split("\n")
        assert log[:2] == ref_log[:2] == ["todo-f0-2", "todo-f0-2"]


def test_feature_input_resamples_other_rates_on_load(tmp_path):
    """A training clip stored at 40 kHz / 48 kHz is resampled to 16 kHz on load, as the reference's load_input_audio(path, 16000) does (reference
    preprocessing_utils.py:171, lib/audio.py:149-150) - with the device polyphase kernel (parity-unpinned: librosa / soxr absent).  The dump of a
    band-limited clip written at 48 kHz equals the dump of the same clip written at 16 kHz up to the resampler's 2e-5 pass-band error."""
    from scipy.io import wavfile
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    from comfy_rvc_amd.preprocessing_utils import FeatureInput, load_wav
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    fi = FeatureInput(hub, "rmvpe", str(tmp_path), version="v2", if_f0=True)
    fi.model_rmvpe = RMVPE(S.rmvpe_state_dict(0))
    secs = 2.0
    t16, t48 = np.arange(int(16000 * secs)) / 16000.0, np.arange(int(48000 * secs)) / 48000.0
    sig = lambda t: (0.3 * np.sin(2 * np.pi * 220.0 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.1 * np.sin(2 * np.pi * 1800.0 * t)).astype(np.float32)
    wavfile.write(str(tmp_path / "a16.wav"), 16000, sig(t16))
    wavfile.write(str(tmp_path / "a48.wav"), 48000, np.stack([sig(t48), sig(t48)], axis=1))        # stereo at 48 kHz
    x48, sr = load_wav(str(tmp_path / "a48.wav"), 16000)
    assert sr == 16000 and x48.shape == (t16.shape[0], 2)
    mid = slice(2000, -2000)
    assert np.abs(x48[mid, 0] - sig(t16)[mid]).max() < 1e-4
    paths = [(str(tmp_path / f"{n}.wav"), str(tmp_path / f"c_{n}"), str(tmp_path / f"n_{n}"), str(tmp_path / f"f_{n}")) for n in ("a16", "a48")]
    assert fi.go(paths) == 2
    f16, f48 = np.load(paths[0][3] + ".npy"), np.load(paths[1][3] + ".npy")
    assert f16.shape == f48.shape and np.abs(f16[5:-5] - f48[5:-5]).max() <= 2e-3 * np.abs(f16).max()
    n16, n48 = np.load(paths[0][2] + ".npy"), np.load(paths[1][2] + ".npy")
    assert n16.shape == n48.shape and np.mean(np.isclose(n16[10:-10], n48[10:-10], rtol=2e-3, atol=1e-3)) > 0.98
