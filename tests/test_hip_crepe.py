"""GPU: the CREPE f0 front-ends (reference pitch_extraction.py:76-150,:205-248) on the HIP path against the CPU oracle
(oracle/crepe.py: a restatement of third-party torchcrepe, parity-unpinned - torchcrepe itself is not available offline)."""
import numpy as np
import pytest
import torch

from comfy_rvc_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets():
    from comfy_rvc_amd.lib.crepe import Crepe
    return {m: (S.crepe_state_dict(m, 0), Crepe(S.crepe_state_dict(m, 0), m)) for m in ("full", "tiny")}


@pytest.mark.parametrize("model,hop,pad,seconds", [("full", 160, True, 1.3), ("tiny", 160, True, 1.3), ("full", 128, True, 0.7), ("tiny", 64, False, 0.5),
                                                   ("tiny", 160, True, 7.0)])
def test_crepe_probabilities_match_oracle(nets, model, hop, pad, seconds):
    """Frame normalisation + six conv blocks + classifier + sigmoid: [360, n] against the oracle, with the first conv block and the
    flattened last feature map as intermediate taps.  7 s = 701 frames: more than one batch of 512 frames."""
    from oracle import crepe as oc
    sd, net = nets[model]
    x = S.synth_audio(seconds, seed=11)
    taps_o = {}
    with torch.no_grad():
        ref = oc.infer(sd, oc.preprocess(torch.from_numpy(x)[None], hop, pad), model, taps_o).numpy()
    n = ref.shape[0]
    C1, C6 = S.CREPE_CHANNELS[model][0], S.CREPE_CHANNELS[model][5]
    B = min(n, 512)
    dt = {"conv1": torch.empty(C1, 256, device="cuda"), "embed": torch.empty(4 * C6, B, device="cuda")}
    p = net.probabilities(x, hop, pad, taps=dt).cpu().numpy()
    assert p.shape == (360, n)
    c1 = dt["conv1"].cpu().numpy()
    assert np.max(np.abs(c1 - taps_o["conv1"])) / np.max(np.abs(taps_o["conv1"])) < 1e-4
    emb = dt["embed"].cpu().numpy().T                               # [B, 4 * C6], position-major like torchcrepe's permute + reshape
    assert np.max(np.abs(emb - taps_o["embed"][:B])) / np.max(np.abs(taps_o["embed"])) < 1e-3
    assert np.max(np.abs(p.T - ref)) < 1e-3                         # sigmoid outputs in [0, 1]; measured ~1e-5


def test_device_viterbi_matches_host_decoding(nets):
    """rvc_crepe_viterbi (masked softmax, banded Viterbi in float64, back-pointer walk, periodicity gather) against lib/crepe.py's host
    implementation (itself pinned to the oracle's literal librosa restatement in tests/test_host_logic.py), on flat network output
    and on a ridge that crosses most of the bin range, for two frequency windows."""
    from comfy_rvc_amd.lib import crepe as pc
    _, net = nets["tiny"]
    flat = net.probabilities(S.synth_audio(9.0, seed=21), 160)                    # 901 frames
    n = flat.shape[1]
    centre = (180 + 150 * np.sin(np.arange(n) / 17.0)).astype(int)
    ridge = torch.from_numpy(np.stack([0.02 + 0.9 * np.exp(-0.5 * ((np.arange(360) - c) / 2.0) ** 2) for c in centre]).astype(np.float32).T.copy()).cuda()
    for P in (flat, ridge):
        for fmin, fmax in ((50.0, 1600.0), (80.0, 700.0)):
            lo, hi = pc.frequency_to_bins(fmin), pc.frequency_to_bins(fmax, ceil=True)
            bins, per = pc.viterbi_device(P, lo, hi)
            Ph = P.cpu().numpy()
            pm = Ph.copy(); pm[:lo] = -np.inf; pm[hi:] = -np.inf
            e = np.exp(pm - pm.max(axis=0, keepdims=True))
            ref = pc.viterbi_bins((e / e.sum(axis=0, keepdims=True)).astype(np.float32))
            assert bins.shape == ref.shape and (bins == ref).mean() >= 0.995, (bins != ref).sum()
            assert np.array_equal(per, Ph[bins, np.arange(n)]) and bins.min() >= lo and bins.max() < hi


def _fe(nets):
    from comfy_rvc_amd.config import Config
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    from comfy_rvc_amd.pitch_extraction import FeatureExtractor
    fe = FeatureExtractor(40000, Config())
    fe.model_crepe = {m: nets[m][1] for m in nets}
    fe.model_rmvpe = RMVPE(S.rmvpe_state_dict(0))
    return fe


@pytest.mark.parametrize("method", ["crepe", "crepe-tiny", "mangio-crepe", "mangio-crepe-tiny"])
def test_crepe_f0_methods_match_oracle(nets, method):
    """FeatureExtractor.get_f0 for the four CREPE slots of f0_method_dict: coarse pitch and f0 against the oracle's restatement of
    the two reference call sites.  get_f0 always passes model="full" in its parameter dict (reference pitch_extraction.py:268-271), which
    overrides the keyword the -tiny slots bind, so through get_f0 all four slots run the full network - as upstream.  The dither
    torchcrepe adds to the decoded cents comes from numpy's global RNG: seeded on both sides."""
    from oracle import crepe as oc
    from oracle.pipeline import f0_postprocess
    fe = _fe(nets)
    x = np.pad(S.synth_audio(1.5, seed=12).astype(np.float64), (16000, 16000), mode="reflect")
    np.random.seed(7)
    coarse, f0 = fe.get_f0(x.copy(), 2, method, crepe_hop_length=128, f0_min=50, f0_max=1600)
    np.random.seed(7)
    if method.startswith("mangio"):
        ref = oc.get_f0_mangio_crepe(nets["full"][0], x.copy(), 50, 1600, hop_length=128, model="full")
    else:
        ref = oc.get_f0_official_crepe(nets["full"][0], x.copy(), 50, 1600, model="full")
    rc, rf = f0_postprocess(ref.astype(np.float64), 2)
    assert f0.shape == rf.shape and coarse.dtype == np.int16
    ok = np.isclose(f0, rf, rtol=1e-3, atol=1e-3)
    assert ok.mean() >= 0.99, (ok.mean(), np.abs(f0 - rf).max())
    assert (np.abs(coarse.astype(int) - rc.astype(int)) <= 1).mean() >= 0.99


@pytest.mark.parametrize("which", ["official", "mangio"])
def test_crepe_tiny_capacity_matches_oracle(nets, which):
    """The tiny network through the method functions themselves (model="tiny" as a caller of the functions can pass it)."""
    from oracle import crepe as oc
    fe = _fe(nets)
    x = np.pad(S.synth_audio(1.5, seed=15).astype(np.float64), (16000, 16000), mode="reflect")
    np.random.seed(9)
    if which == "mangio":
        f0 = fe.get_f0_crepe_computation(x.copy(), 50, 1100, crepe_hop_length=160, model="tiny")
        np.random.seed(9)
        ref = oc.get_f0_mangio_crepe(nets["tiny"][0], x.copy(), 50, 1100, hop_length=160, model="tiny")
    else:
        f0 = fe.get_f0_official_crepe_computation(x.copy(), 50, 1100, model="tiny")
        np.random.seed(9)
        ref = oc.get_f0_official_crepe(nets["tiny"][0], x.copy(), 50, 1100, model="tiny")
    assert f0.shape == ref.shape
    ok = np.isclose(f0, ref, rtol=1e-3, atol=1e-3)
    assert ok.mean() >= 0.99, (ok.mean(), np.abs(f0 - ref).max())


def test_hybrid_rmvpe_crepe_merge_matches_oracle(nets):
    """f0_method = ["rmvpe", "crepe"] (reference get_f0_hybrid_computation, pitch_extraction.py:205-248): quantile-normalised audio to
    tracks padded to equal length, nan-median across methods."""
    from oracle import crepe as oc, nets as onets
    fe = _fe(nets)
    x = np.pad(S.synth_audio(1.2, seed=13).astype(np.float64), (16000, 16000), mode="reflect")
    np.random.seed(3)
    coarse, f0 = fe.get_f0(x.copy(), 0, ["rmvpe", "crepe"], merge_type="median", f0_min=50, f0_max=1600)
    # (upstream builds the parameter dict BEFORE it normalises its local copy of x, pitch_extraction.py:217-225: every method receives
    # the un-normalised signal; the quantile normalisation there is dead code, mirrored as such)
    a = onets.rmvpe_infer_from_audio(S.rmvpe_state_dict(0), x.astype(np.float32), thred=0.03)
    np.random.seed(3)
    b = oc.get_f0_official_crepe(nets["full"][0], x, 50, 1600)
    m = max(len(a), len(b))
    ref = np.nanmedian(np.stack([np.pad(a, (0, m - len(a))), np.pad(b, (0, m - len(b)))]), axis=0)
    assert f0.shape == ref.shape, (f0.shape, ref.shape)
    ok = np.isclose(f0, ref, rtol=1e-3, atol=1e-3)
    assert ok.mean() >= 0.99, (ok.mean(), np.abs(f0 - ref).max(), f0[:5], ref[:5])


def test_vc_single_with_default_arguments_runs_crepe(nets, tmp_path, monkeypatch):
    """vc_single's own default is f0_method="crepe" (reference vc_infer_pipeline.py:261): with torchcrepe's weight file in
    models/torchcrepe/ a caller that relies on the defaults gets audio, not None."""
    import comfy_rvc_amd.lib as lib
    import comfy_rvc_amd.lib.crepe as tc
    import comfy_rvc_amd.pitch_extraction as pe
    from comfy_rvc_amd.config import Config
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.vc_infer_pipeline import get_vc, vc_single
    models = tmp_path / "models"
    (models / "torchcrepe").mkdir(parents=True)
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in S.crepe_state_dict("full", 0).items()}, str(models / "torchcrepe" / "full.pth"))
    for mod in (lib, pe, tc):
        monkeypatch.setattr(mod, "BASE_MODELS_DIR", str(models), raising=False)
    cfg = Config()
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg)
    audio = S.synth_audio(1.5, seed=14)
    np.random.seed(0); torch.manual_seed(0)
    out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vcd["vc"], hubert_model=hub, input_audio=(audio, 16000))     # every other argument defaulted
    assert out is not None and out[1] == 40000 and out[0].dtype == np.int16 and np.abs(out[0].astype(np.int32)).max() > 1000
    np.random.seed(0); torch.manual_seed(0)
    again = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vcd["vc"], hubert_model=hub, input_audio=(audio, 16000), f0_method="crepe")
    assert np.array_equal(out[0], again[0])
