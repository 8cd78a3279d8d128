"""GPU parity of the three HIP network graphs: against the golden vectors captured from the real reference and against
the CPU oracle on fresh seeded inputs.  Gate: 1e-3 relative (BASELINE.json north_star), fp32."""
import numpy as np
import pytest
import torch

from conftest import f0_frames_ok, golden, record_parity, rel_err
from comfy_rvc_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def hubert():
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    return HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)


@pytest.fixture(scope="module")
def rmvpe():
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    return RMVPE(S.rmvpe_state_dict(0), is_half=False)


def make_synth(config, version):
    from comfy_rvc_amd.lib.infer_pack import models as M
    cls = M.SynthesizerTrnMs768NSFsid if version == "v2" else M.SynthesizerTrnMs256NSFsid
    net = cls(*config, is_half=False)
    net.load_state_dict(S.synth_state_dict(config, version, 0))
    return net


def cm(x):
    """channel-major device tap [C][T] -> numpy [1,C,T]"""
    return x.cpu().numpy()[None]


def test_hubert_matches_reference_golden(hubert):
    g = golden("hubert_1s.npz")
    audio = torch.from_numpy(g["audio"])
    Th = hubert.num_frames(audio.shape[1])
    taps = {"conv_stack": torch.empty(512, Th, device="cuda"), "pos_conv": torch.empty(768, Th, device="cuda"),
            "hidden_0": torch.empty(768, Th, device="cuda"), "hidden_8": torch.empty(768, Th, device="cuda")}
    v2 = hubert.extract_features(audio, version="v2", taps=taps)
    assert tuple(v2.shape) == g["out_v2"].shape
    assert rel_err(cm(taps["conv_stack"]), g["conv_stack"]) < TOL
    assert rel_err(taps["pos_conv"].cpu().numpy().T[None], g["pos_conv"]) < TOL
    assert rel_err(taps["hidden_0"].cpu().numpy().T[None], g["hidden_0"]) < TOL
    assert rel_err(taps["hidden_8"].cpu().numpy().T[None], g["hidden_8"]) < TOL
    assert rel_err(v2.cpu(), g["out_v2"]) < TOL
    v1 = hubert.extract_features(audio, version="v1")
    assert rel_err(v1.cpu(), g["out_v1"]) < TOL
    vcm = hubert.extract_features(audio, version="v2", channel_major=True)
    assert torch.equal(vcm.t()[None], v2)


def test_heavy_family_hubert_and_synth_match_reference_golden(pair_arith):
    """The second weight family (outlier channels: conftest / oracle/gen_golden.py heavy_*) on the stage level: HuBERT's taps with 15 - 30 x channels in
    the residual stream and the FFN intermediate through the bf16 hi / lo images, the synthesizer's taps and waveform with x10 - x30 channels in every
    ResBlock pair's intermediate (T = 16: the per-tile kernels) - same 1e-3 as the plain family."""
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    g = golden("hubert_1s_heavy.npz")
    hub = HubertModelWithFinalProj(S.hubert_state_dict(1, family="heavy"), S.HUBERT_CONFIG)
    audio = torch.from_numpy(g["audio"])
    Th = hub.num_frames(audio.shape[1])
    taps = {"conv_stack": torch.empty(512, Th, device="cuda"), "pos_conv": torch.empty(768, Th, device="cuda"),
            "hidden_0": torch.empty(768, Th, device="cuda"), "hidden_8": torch.empty(768, Th, device="cuda")}
    v2 = hub.extract_features(audio, version="v2", taps=taps)
    for k in ("pos_conv", "hidden_0", "hidden_8"):
        assert rel_err(taps[k].cpu().numpy().T[None], g[k]) < TOL, k
    assert rel_err(v2.cpu(), g["out_v2"]) < TOL
    assert rel_err(hub.extract_features(audio, version="v1").cpu(), g["out_v1"]) < TOL
    g = golden("synth_40k_v2_heavy.npz")
    from comfy_rvc_amd.lib.infer_pack import models as M
    net = M.SynthesizerTrnMs768NSFsid(*S.CONFIG_40K_V2, is_half=False)
    net.load_state_dict(S.synth_state_dict(S.CONFIG_40K_V2, "v2", 1, family="heavy"))
    T = g["phone"].shape[1]
    for arith in (1, 0):
        pair_arith(arith)
        taps = {k: None for k in ("enc_p_layer0", "m_p", "logs_p", "z_p", "z", "har_source", "sine_waves")}
        o, mask, _ = net.infer(torch.from_numpy(g["phone"]), torch.LongTensor([T]), torch.from_numpy(g["pitch"]), torch.from_numpy(g["pitchf"]),
                               torch.LongTensor([int(g["sid"])]), noise=(g["noise_z"], g["noise_src"]), taps=taps)
        for k in ("m_p", "logs_p", "z_p", "z", "enc_p_layer0"):
            assert rel_err(cm(taps[k]), g[k]) < TOL, k
        assert rel_err(o.cpu(), g["wav"]) < TOL, arith


def test_hubert_matches_oracle_other_length(hubert):
    from oracle import nets
    audio = S.synth_audio(2.37, seed=21)[None]
    ref = nets.hubert_extract_features(S.hubert_state_dict(0), audio, "v2")
    out = hubert.extract_features(torch.from_numpy(audio), version="v2")
    assert rel_err(out.cpu(), ref) < TOL


def test_rmvpe_matches_reference_golden(rmvpe):
    g = golden("rmvpe_1s.npz")
    r = rmvpe.infer(g["audio"], want_mel=True, want_salience=True)
    assert np.max(np.abs(r["mel"].cpu().numpy()[None] - g["mel"])) < 2e-3          # log-mel: absolute (values span [-11.5, 3])
    assert np.max(np.abs(r["salience"].cpu().numpy() - g["salience"])) < TOL       # sigmoid outputs in [0, 1]
    f0 = rmvpe.infer_from_audio(g["audio"])
    assert f0.dtype == np.float64 and f0.shape == g["f0"].shape
    assert np.allclose(f0, g["f0"], rtol=TOL)
    assert np.allclose(rmvpe.infer_from_audio_with_pitch(g["audio"], f0_min=50, f0_max=1600), g["f0_plus"], rtol=TOL)
    dec = rmvpe.decode(g["syn_salience"])
    assert np.array_equal(dec == 0, g["syn_f0"] == 0)                     # voiced / unvoiced decisions are exact
    assert np.allclose(dec, g["syn_f0"], rtol=1e-6, atol=0), np.max(np.abs(dec / np.maximum(g["syn_f0"], 1e-30) - 1))


def test_rmvpe_matches_oracle_other_length(rmvpe):
    from oracle import nets
    audio = S.synth_audio(3.21, seed=22)
    taps = {}
    f0_ref = nets.rmvpe_infer_from_audio(S.rmvpe_state_dict(0), audio, taps=taps)
    r = rmvpe.infer(audio, want_salience=True)
    serr = float(np.max(np.abs(r["salience"].cpu().numpy() - taps["salience"])))
    assert serr < TOL
    f0 = r["f0"].cpu().numpy()
    # every frame equal within tolerance, except genuine ties of the reference's own arg-max (conftest.f0_frames_ok): no percentile
    n_bad, unexplained = f0_frames_ok(f0, f0_ref, taps["salience"], serr, rtol=TOL)
    record_parity("rmvpe_3.21s", {"salience_max_err": serr, "f0_frames": int(f0.shape[0]), "f0_frames_differing": n_bad, "unexplained": unexplained})
    assert unexplained == 0, (n_bad, unexplained)


@pytest.mark.parametrize("name,config,version", [("synth_40k_v2.npz", S.CONFIG_40K_V2, "v2"), ("synth_48k_v2.npz", S.CONFIG_48K_V2, "v2"),
                                                 ("synth_40k_v1.npz", S.CONFIG_40K_V1, "v1"), ("synth_32k_v1.npz", S.CONFIG_32K_V1, "v1"),
                                                 ("synth_48k_v1.npz", S.CONFIG_48K_V1, "v1"), ("synth_32k_v2.npz", S.CONFIG_32K_V2, "v2")])
def test_synth_matches_reference_golden(name, config, version):
    g = golden(name)
    net = make_synth(config, version)
    T = g["phone"].shape[1]
    taps = {k: None for k in ("enc_p_layer0", "m_p", "logs_p", "z_p", "z", "har_source", "sine_waves")}
    o, mask, (z, z_p, m_p, logs_p) = net.infer(torch.from_numpy(g["phone"]), torch.LongTensor([T]), torch.from_numpy(g["pitch"]),
                                               torch.from_numpy(g["pitchf"]), torch.LongTensor([int(g["sid"])]),
                                               noise=(g["noise_z"], g["noise_src"]), taps=taps)
    assert tuple(o.shape) == g["wav"].shape and tuple(mask.shape) == (1, 1, T)
    for k in ("m_p", "logs_p", "z_p", "z"):
        assert rel_err(cm(taps[k]), g[k]) < TOL, k
    if "enc_p_layer0" in g:
        assert rel_err(cm(taps["enc_p_layer0"]), g["enc_p_layer0"]) < TOL
        assert rel_err(taps["har_source"].cpu().numpy()[None, None], g["har_source"]) < TOL
    assert rel_err(o.cpu(), g["wav"]) < TOL


@pytest.mark.parametrize("name,config,version", [("synth_40k_v2_nono.npz", S.CONFIG_40K_V2, "v2"), ("synth_40k_v1_nono.npz", S.CONFIG_40K_V1, "v1")])
def test_synth_nono_matches_reference_golden(name, config, version):
    """The no-f0 family (reference models.py:812-1022): infer(phone, phone_lengths, sid), one noise draw, plain Generator."""
    from comfy_rvc_amd.lib.infer_pack.models import SynthesizerTrnMs256NSFsid_nono, SynthesizerTrnMs768NSFsid_nono
    g = golden(name)
    cls = SynthesizerTrnMs768NSFsid_nono if version == "v2" else SynthesizerTrnMs256NSFsid_nono
    net = cls(*config)                                                # reference call: the cpt["config"] list splatted, sr last
    net.load_state_dict(S.synth_state_dict(config, version, 0, f0=False))
    T = g["phone"].shape[1]
    taps = {k: None for k in ("m_p", "logs_p", "z_p", "z")}
    o, mask, _ = net.infer(torch.from_numpy(g["phone"]), torch.LongTensor([T]), torch.LongTensor([int(g["sid"])]), noise=g["noise_z"], taps=taps)
    assert tuple(o.shape) == g["wav"].shape and tuple(mask.shape) == (1, 1, T)
    for k in ("m_p", "logs_p", "z_p", "z"):
        assert rel_err(cm(taps[k]), g[k]) < TOL, k
    assert rel_err(o.cpu(), g["wav"]) < TOL
    # the two families do not accept each other's checkpoints or arguments
    with pytest.raises(ValueError, match="without f0"):
        cls(*config).load_state_dict(S.synth_state_dict(config, version, 0))
    f0net = make_synth(config, version)
    from comfy_rvc_amd import _lib
    out = torch.empty(T * net.upp, device="cuda")
    ph, nz = torch.from_numpy(g["phone"]).cuda().contiguous(), torch.from_numpy(g["noise_z"]).cuda().contiguous()
    with pytest.raises(RuntimeError, match="trained with f0"):        # C ABI: an f0 model without pitch arguments
        _lib.check(_lib.lib.rvc_synth_infer(f0net._h, None, _lib.ptr(ph), 0, None, None, 0, _lib.ptr(nz), None, T, _lib.ptr(out), None))
    with pytest.raises(RuntimeError, match="no-f0 model"):            # ... and a no-f0 model with them
        pc = torch.ones(T, dtype=torch.int64, device="cuda"); pf = torch.ones(T, device="cuda"); ns = torch.zeros(T * net.upp, device="cuda")
        _lib.check(_lib.lib.rvc_synth_infer(net._h, None, _lib.ptr(ph), 0, _lib.ptr(pc), _lib.ptr(pf), 0, _lib.ptr(nz), _lib.ptr(ns), T, _lib.ptr(out), None))


def test_synth_matches_oracle_longer_sequence():
    from oracle import nets
    from oracle.pipeline import f0_postprocess
    config, version, T = S.CONFIG_40K_V2, "v2", 150
    rng = np.random.default_rng(5)
    phone = (rng.standard_normal((1, T, 768)) * 0.5).astype(np.float32)
    f0 = S.designed_f0(T, seed=0)
    coarse, f0f = f0_postprocess(f0.astype(np.float64), 2)
    pitch = torch.from_numpy(coarse.astype(np.int64))[None]
    pitchf = torch.from_numpy(f0f.astype(np.float32))[None]
    gen = torch.Generator().manual_seed(77)
    nz, ns = torch.randn(1, 192, T, generator=gen), torch.randn(1, T * 400, 1, generator=gen)
    sd = S.synth_state_dict(config, version, 0)
    ref = nets.synth_infer(sd, config, phone, pitch, pitchf, 5, nz, ns)
    net = make_synth(config, version)
    o, _, _ = net.infer(torch.from_numpy(phone), torch.LongTensor([T]), pitch, pitchf, torch.LongTensor([5]), noise=(nz, ns))
    assert rel_err(o.cpu(), ref) < TOL


def test_synth_default_noise_follows_global_rng_order():
    """Without explicit noise the two draws follow the reference's order and shapes on the CPU generator."""
    config, T = S.CONFIG_40K_V2, 20
    net = make_synth(config, "v2")
    phone = torch.randn(1, T, 768, generator=torch.Generator().manual_seed(1)) * 0.5
    pitch = torch.full((1, T), 60, dtype=torch.int64)
    pitchf = torch.full((1, T), 220.0)
    torch.manual_seed(1234)
    a, _, _ = net.infer(phone, torch.LongTensor([T]), pitch, pitchf, torch.LongTensor([0]))
    torch.manual_seed(1234)
    nz = torch.randn(1, 192, T); torch.rand(1, 1); ns = torch.randn(1, T * 400, 1)
    b, _, _ = net.infer(phone, torch.LongTensor([T]), pitch, pitchf, torch.LongTensor([0]), noise=(nz, ns))
    assert torch.equal(a, b)
