"""GPU: the MDX23C separation network and the UVR -> VC chain (BASELINE config C5) on the HIP path against golden vectors of
the reference's own TFC_TDF_net (oracle/gen_golden.py mdx23c) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from comfy_rvc_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
    cfg = S.mdx23c_config(**S.MDX23C_SMALL)
    net = TFC_TDF_net(cfg)
    net.load_state_dict(S.mdx23c_state_dict(cfg, 0))
    return cfg, net


def test_mdx23c_chunk_matches_reference_golden(small):
    """One chunk through STFT -> TFC/TDF U-Net -> mask head -> inverse STFT against the reference module's output (1e-3 of the peak)."""
    cfg, net = small
    g = golden("mdx23c_small.npz")
    y = net(g["x"][None])[0].cpu().numpy()
    assert y.shape == g["out"].shape == (2, 2, cfg["audio"]["chunk_size"])
    assert rel_err(y, g["out"]) < 1e-3
    # two chunks in one batch: each equals its own single call (no state between chunks)
    x2 = np.stack([g["x"], g["x"][::-1].copy() * 0.5])
    y2 = net(x2).cpu().numpy()
    assert np.array_equal(y2[0], y) and rel_err(y2[1], net(x2[1:2])[0].cpu().numpy()) < 1e-6


def test_demix_mdxv3_matches_reference_golden(small):
    """demix_mdxv3: zero padding, chunks every C / overlap samples, overlap-add, 1 / overlap - on a clip of 2.6 chunks, overlap 4; then the same
    clip with its chunks over three streams (`set_streams(3)`): the golden again, bit-identical repeats, within re-association of the one-stream sum."""
    from comfy_rvc_amd.lib.karafan.inference import demix_mdxv3
    cfg, net = small
    g = golden("mdx23c_small.npz")
    est = demix_mdxv3(g["clip"], net, net.device, cfg, int(g["overlap"]))
    assert list(est) == ["Vocals", "Instrumental"]
    got = np.stack([est["Vocals"], est["Instrumental"]])
    assert got.shape == g["demix"].shape and rel_err(got, g["demix"]) < 1e-3
    net.set_streams(3)
    try:
        e3 = [demix_mdxv3(g["clip"], net, net.device, cfg, int(g["overlap"])) for _ in range(2)]
    finally:
        net.set_streams(1)
    got3 = np.stack([e3[0]["Vocals"], e3[0]["Instrumental"]])
    assert rel_err(got3, g["demix"]) < 1e-3 and rel_err(got3, got) < 1e-5
    assert np.array_equal(got3, np.stack([e3[1]["Vocals"], e3[1]["Instrumental"]]))


def test_mdx23c_other_geometry_matches_oracle():
    """A second geometry (3 scales, 2 sub-bands, one block per scale, wider bottleneck factor) against the CPU oracle."""
    from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
    from oracle import mdx23c as om
    cfg = S.mdx23c_config(n_fft=1024, hop=128, dim_f=512, dim_t=32, num_channels=48, growth=16, num_scales=3, num_subbands=2, blocks=1, bottleneck=8)
    sd = S.mdx23c_state_dict(cfg, 1)
    net = TFC_TDF_net(cfg)
    net.load_state_dict(sd)
    x = (np.random.default_rng(4).standard_normal((2, cfg["audio"]["chunk_size"])) * 0.2).astype(np.float32)
    with torch.no_grad():
        ref = om.forward(sd, cfg, x).numpy()
    assert rel_err(net(x[None])[0].cpu().numpy(), ref) < 1e-3


_ALT_GRAPH = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from conftest import golden, rel_err
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
from comfy_rvc_amd.lib.karafan.inference import demix_mdxv3
cfg = S.mdx23c_config(**S.MDX23C_SMALL)
net = TFC_TDF_net(cfg); net.load_state_dict(S.mdx23c_state_dict(cfg, 0))
g = golden("mdx23c_small.npz")
e1 = rel_err(net(g["x"][None])[0].cpu().numpy(), g["out"])
est = demix_mdxv3(g["clip"], net, net.device, cfg, int(g["overlap"]))
e2 = rel_err(np.stack([est["Vocals"], est["Instrumental"]]), g["demix"])
print("ERR", e1, e2)
assert e1 < 1e-3 and e2 < 1e-3
"""


@pytest.mark.parametrize("env", [{"RVC_MDX_X3S": "0"}, {"RVC_MDX_FUSE_SC": "0"}, {"RVC_MDX_STREAMS": "3"}])
def test_mdx23c_alternative_graphs_match_reference_golden(env):
    """The graphs behind the switches - fp32 planes on the staged kernels (RVC_MDX_X3S=0: what a network without bf16x3 weight images runs) and the padded graph
    with the shortcuts as launches of their own (RVC_MDX_FUSE_SC=0) - and demix with a clip's chunks over three streams (RVC_MDX_STREAMS=3 = `set_streams(3)`, the node's
    setting) against the same golden chunk and demix (the switches are read once per process)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-c", _ALT_GRAPH, os.path.dirname(here), here], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ERR" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_full_mdx23c_recipe_runs_at_size():
    """The shipped recipe (n_fft 8192, dim_f 4096, dim_t 256, 128 channels + 128 per scale, 5 scales: 112 M parameters) on one 5.9 s
    chunk - too large for the CPU oracle inside the suite, so size-independent properties: shape, finiteness, bit-identical repeats,
    and the silent-input fixed point (every convolution is bias-free and the mask multiplies the first conv's output: zeros in, zeros out)."""
    from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG
    from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
    cfg = MDX23C_CONFIG
    net = TFC_TDF_net(cfg)
    net.load_state_dict(S.mdx23c_state_dict(cfg, 0))
    x = (np.random.default_rng(5).standard_normal((1, 2, 261120)) * 0.1).astype(np.float32)
    a = net(x).cpu().numpy()
    b = net(x).cpu().numpy()
    assert a.shape == (1, 2, 2, 261120) and np.isfinite(a).all() and np.array_equal(a, b) and np.abs(a).max() > 1e-4
    z = net(np.zeros_like(x)).cpu().numpy()
    assert np.abs(z).max() < 1e-6            # all-zero input: the mask multiplies a zero spectrogram branch and every conv is bias-free


def test_full_mdx23c_recipe_matches_reference_golden():
    """VALUES at the shipped recipe: the same 5.9 s stereo chunk the reference's own TFC_TDF_net converted in the build container
    (tests/golden/mdx23c_full_chunk.npz, oracle/gen_golden.py mdx23c_full): every 64th output sample, a dense 4096-sample window and the
    per-(stem, channel) energy within 1e-3 of the peak; first-conv / first-scale / bottleneck / mask-head taps are pinned on the oracle side."""
    import hashlib
    from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG
    from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
    g = golden("mdx23c_full_chunk.npz")
    x = S.mdx23c_full_chunk()
    assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(x).tobytes()).digest(), dtype=np.uint8), g["audio_sha256"])
    net = TFC_TDF_net(MDX23C_CONFIG)
    net.load_state_dict(S.mdx23c_state_dict(MDX23C_CONFIG, 0))
    y = net(x[None])[0].cpu().numpy()
    assert y.shape == (2, 2, 261120)
    assert rel_err(y[..., ::64], g["out_sub"]) < 1e-3 and rel_err(y[..., 100000:104096], g["out_win"]) < 1e-3
    assert rel_err(np.sqrt((y.astype(np.float64) ** 2).sum(-1)), g["out_norm"]) < 1e-3
    assert rel_err(np.abs(y).max(-1), g["out_absmax"]) < 1e-3


@pytest.fixture(scope="module")
def full_net():
    from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG
    from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
    net = TFC_TDF_net(MDX23C_CONFIG)
    net.load_state_dict(S.mdx23c_state_dict(MDX23C_CONFIG, 0))
    return MDX23C_CONFIG, net


def test_full_recipe_demix_matches_the_references_own_demix(full_net):
    """VALUES of the whole demix_mdxv3 at the shipped recipe: three overlapping full-size chunks (2.96 s stereo clip, overlap 2) against what the
    REFERENCE's own demix_mdxv3 + TFC_TDF_net produced in the build container (tests/golden/mdx23c_demix_full.npz, oracle/gen_golden.py
    mdx23c_demix_full; reference lib/karafan/inference.py:32-74): padding, chunk order, overlap-add across the seams, the division."""
    import hashlib
    from comfy_rvc_amd.lib.karafan.inference import demix_mdxv3
    cfg, net = full_net
    g = golden("mdx23c_demix_full.npz")
    L = int(g["n"])
    mix = np.stack([S.synth_audio(L / 44100.0, seed=int(sd), sr=44100)[:L] for sd in g["seeds"]]).astype(np.float32)
    assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(mix).tobytes()).digest(), dtype=np.uint8), g["audio_sha256"])
    w0 = int(g["win0"])
    outs = []
    for k in (1, 3):
        net.set_streams(k)
        try:
            est = demix_mdxv3(mix, net, net.device, cfg, int(g["overlap"]))
        finally:
            net.set_streams(1)
        y = np.stack([est["Vocals"], est["Instrumental"]])
        assert y.shape == (2, 2, L)
        assert rel_err(y[..., ::16], g["out_sub"]) < 1e-3 and rel_err(y[..., w0:w0 + 4096], g["out_win"]) < 1e-3
        assert rel_err(np.sqrt((y.astype(np.float64) ** 2).sum(-1)), g["out_norm"]) < 1e-3 and rel_err(np.abs(y).max(-1), g["out_absmax"]) < 1e-3
        outs.append(y)
    assert rel_err(outs[1], outs[0]) < 1e-5                      # chunks over three streams: the same sums re-associated


def test_c5_full_size_uvr_then_vc_chain(full_net, tmp_path, monkeypatch):
    """BASELINE.json configs[4] at FULL size through the node surface: a 30 s stereo 44.1 kHz clip -> UVR5Node.split (MDX23C at the shipped recipe,
    overlap 8: 48 chunks of 5.9 s) -> vocal stem thunk -> RVCNode.convert with the 48k_v2 synthesizer.  Values of the network and of demix_mdxv3 are
    pinned by the golden tests above; here the size-independent properties at the real size: shapes, finiteness, bit-identical repeats, chunk
    streams 1 vs 3 within re-association, output length and normalisation of the conversion."""
    import comfy_rvc_amd.lib as lib
    import comfy_rvc_amd.pitch_extraction as pe
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N, uvr as U
    from comfy_rvc_amd.lib.audio import bytes_to_audio
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.vc_infer_pipeline import get_vc
    cfg, net = full_net
    models = tmp_path / "models"
    (models / "karafan").mkdir(parents=True)
    (models / "karafan" / "MDX23C-8KFFT-InstVoc_HQ.ckpt").write_bytes(b"")      # (the 450 MB checkpoint file itself is exercised at the reduced recipe)
    as_t = lambda sd: {k: torch.as_tensor(np.ascontiguousarray(v)).clone() for k, v in sd.items()}   # noqa: E731
    torch.save(as_t(S.rmvpe_state_dict(0)), str(models / "rmvpe.pt"))
    for mod in (lib, pe, N, U):
        monkeypatch.setattr(mod, "BASE_MODELS_DIR", str(models), raising=False)

    def loader(path, config=None):
        net.set_streams(3)                                       # what load_mdx23c sets for the node (one clip at a time)
        return net, cfg
    monkeypatch.setattr(U, "load_mdx23c", loader)
    secs = 30.0
    stereo = np.stack([S.synth_audio(secs, seed=100, sr=44100), S.synth_audio(secs, seed=300, sr=44100)]).astype(np.float32)
    n = stereo.shape[1]
    audio = N.to_audio_dict(stereo, 44100)
    try:
        stems = []
        for _ in range(2):
            vocals, music = U.UVR5Node().split(audio, "karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt", format="wav")
            v, sr = bytes_to_audio(vocals())
            m, _ = bytes_to_audio(music())
            assert sr == 44100 and v.shape == m.shape == stereo.shape and np.isfinite(v).all() and np.isfinite(m).all()
            stems.append((v, m))
        assert np.array_equal(stems[0][0], stems[1][0]) and np.array_equal(stems[0][1], stems[1][1])      # three chunk streams: bit-identical repeats
        assert np.abs(stems[0][0]).max() > 1e-4 and not np.array_equal(stems[0][0], stems[0][1])
        net.set_streams(1)
        monkeypatch.setattr(U, "load_mdx23c", lambda path, config=None: (net, cfg))
        N._MEMO.clear() if hasattr(N, "_MEMO") else None
        v1, _ = bytes_to_audio(U.UVR5Node().split(audio, "karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt", format="wav")[0]())
        assert rel_err(v1, stems[0][0]) < 1e-4                 # (wav bytes: int16-quantised stems of the two summation orders)
    finally:
        net.set_streams(1)
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_48K_V2, "v2", 0))
    (params,) = N.LoadPitchExtractionParams().load_params(f0_method="rmvpe", f0_autotune=False, index_rate=0.0, resample_sr=0, rms_mix_rate=0.25,
                                                          protect=0.25, crepe_hop_length=160)
    wavs = []
    for _ in range(2):
        torch.manual_seed(7)
        out = N.RVCNode().convert(vocals, lambda: vcd, lambda: hub, params, f0_up_key=0, format="wav", use_cache=False)
        wav, sr2 = bytes_to_audio(out["result"][0]())
        wavs.append(wav)
        assert sr2 == 48000 and wav.ndim == 1 and np.isfinite(wav).all()
    n16 = int(round(n * 16000 / 44100))
    expect = 2 * ((n16 + 32000 - 400) // 320 + 1) * 480 - 2 * 48000              # 2 T_h upp - 2 t_pad_tgt of the 16 kHz vocal stem
    assert abs(wavs[0].shape[0] - expect) <= 960, (wavs[0].shape, expect)            # (the resampler may round the 16 kHz length by a sample or two)
    assert np.array_equal(wavs[0], wavs[1])                                        # same seed, same audio: bit-identical conversion
    assert abs(np.abs(wavs[0]).max() - 0.99) < 2e-3                                # pipeline normalises the peak to 0.99 (reference vc_infer_pipeline.py:188-189)


def test_uvr_then_vc_chain(small, tmp_path, monkeypatch):
    """BASELINE config C5 in miniature: UVR5Node (karafan MDX23C, reduced recipe with a yaml next to the checkpoint) splits a stereo
    44.1 kHz clip, the vocal stem thunk goes into RVCNode (48k_v2 synthesizer) and comes out as WAV bytes at 48 kHz."""
    import yaml
    import comfy_rvc_amd.lib as lib
    import comfy_rvc_amd.pitch_extraction as pe
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N, uvr as U
    from comfy_rvc_amd.lib.audio import bytes_to_audio
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.vc_infer_pipeline import get_vc
    models = tmp_path / "models"
    (models / "karafan").mkdir(parents=True)
    cfg, _ = small
    as_t = lambda sd: {k: torch.as_tensor(np.ascontiguousarray(v)).clone() for k, v in sd.items()}   # noqa: E731
    torch.save(as_t(S.mdx23c_state_dict(cfg, 0)), str(models / "karafan" / "MDX23C-8KFFT-InstVoc_HQ.ckpt"))
    yaml.safe_dump(cfg, open(models / "karafan" / "MDX23C-8KFFT-InstVoc_HQ.yaml", "w"))
    torch.save(as_t(S.rmvpe_state_dict(0)), str(models / "rmvpe.pt"))
    for mod in (lib, pe, N, U):
        monkeypatch.setattr(mod, "BASE_MODELS_DIR", str(models), raising=False)
    t = np.arange(int(0.8 * 44100)) / 44100.0
    stereo = np.stack([0.3 * np.sin(2 * np.pi * 220 * t), 0.2 * np.sin(2 * np.pi * 330 * t)]).astype(np.float32)
    assert "karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt" in U.UVR5Node.INPUT_TYPES()["required"]["model"][0]
    vocals, music = U.UVR5Node().split(N.to_audio_dict(stereo, 44100), "karafan/MDX23C-8KFFT-InstVoc_HQ.ckpt", overlap=2)
    v, sr = bytes_to_audio(vocals())
    m, _ = bytes_to_audio(music())
    assert sr == 44100 and v.shape == m.shape == stereo.shape and np.isfinite(v).all()
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_48K_V2, "v2", 0))
    (params,) = N.LoadPitchExtractionParams().load_params(f0_method="rmvpe", f0_autotune=False, index_rate=0.0, resample_sr=0, rms_mix_rate=0.25,
                                                          protect=0.25, crepe_hop_length=160)
    out = N.RVCNode().convert(vocals, lambda: vcd, lambda: hub, params, f0_up_key=0, use_cache=False)
    wav, sr2 = bytes_to_audio(out["result"][0]())
    assert sr2 == 48000 and wav.ndim == 1 and wav.shape[0] > 0.5 * 48000 and np.isfinite(wav).all()
    with pytest.raises(NotImplementedError):
        U.UVR5Node().split(N.to_audio_dict(stereo, 44100), "UVR/HP5-vocals+instrumentals.pth")
