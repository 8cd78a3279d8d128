"""Generates tests/golden/*.npz by running the REAL reference (imported via oracle/ref_shim.py).

Container-only (needs /root/reference); the GPU box only sees the committed .npz files.
Weights: comfy-rvc_amd/synthetic.py procedural checkpoints (regenerated, never stored).
Noise:   torch.randn_like is patched to draw from torch.Generator(cpu).manual_seed(seed) in the
         reference's call order, so oracle / HIP runs can replay the identical draws.

    python -m oracle.gen_golden            # writes tests/golden/*.npz (a few MB)
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim                                  # noqa: E402
from comfy_rvc_amd import synthetic as S                      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


class NoiseTape:
    """Replays randn draws from a seeded CPU generator; patched over torch.randn_like."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.shapes = []

    def __call__(self, shape):
        self.shapes.append(tuple(shape))
        return torch.randn(tuple(shape), generator=self.g)

    def randn_like(self, x, **kw):
        return self(tuple(x.shape)).to(x.dtype)


class patched_randn_like:
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.old = torch.randn_like
        torch.randn_like = self.tape.randn_like

    def __exit__(self, *a):
        torch.randn_like = self.old


def np_(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def build_models(ns, config, version, seed=0, family="plain", rmvpe_seed=None):
    from transformers import HubertConfig
    hub = ns.loaders.HubertModelWithFinalProj(HubertConfig())
    hub.load_state_dict(to_torch_sd(S.hubert_state_dict(seed, family=family)))
    hub.eval()
    mdir = os.path.join(ns.ws, "models")
    torch.save(to_torch_sd(S.rmvpe_state_dict(seed if rmvpe_seed is None else rmvpe_seed)), os.path.join(mdir, "rmvpe.pt"))
    cpt = S.synth_checkpoint(config, version, seed, family=family)
    cpt["weight"] = {k: v.half() for k, v in to_torch_sd(cpt["weight"]).items()}
    name = f"synth_{version}_{config[-1]}.pth"
    torch.save(cpt, os.path.join(mdir, name))
    with ref_shim.chdir_ws():
        vcd = ns.vc_infer_pipeline.get_vc(os.path.join(mdir, name), file_index="")
    return hub, vcd


def gen_hubert(ns, seed=0, family="plain", name="hubert_1s"):
    from transformers import HubertConfig
    hub = ns.loaders.HubertModelWithFinalProj(HubertConfig())
    hub.load_state_dict(to_torch_sd(S.hubert_state_dict(seed, family=family)))
    hub.eval()
    audio = torch.from_numpy(S.synth_audio(1.0, seed=3)).view(1, -1)
    with torch.no_grad():
        conv = hub.feature_extractor(audio)
        out = hub(audio, output_hidden_states=True)
        hs = out["hidden_states"]
        h_in = hub.feature_projection(conv.transpose(1, 2))
        pos = hub.encoder.pos_conv_embed(h_in)
    v2 = hub.extract_features(audio, version="v2")
    v1 = hub.extract_features(audio, version="v1")
    assert len(hs) == 13 and torch.equal(v2, hs[11])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), audio=np_(audio), conv_stack=np_(conv), pos_conv=np_(pos),
                        hidden_0=np_(hs[0]), hidden_8=np_(hs[8]), out_v2=np_(v2), out_v1=np_(v1))
    print(name, v2.shape, v1.shape, float(v2.abs().mean()), float(v2.std()), "max |hidden_8|", float(hs[8].abs().max()), "max |v2|", float(v2.abs().max()))


def gen_rmvpe(ns):
    mdir = os.path.join(ns.ws, "models")
    torch.save(to_torch_sd(S.rmvpe_state_dict(0)), os.path.join(mdir, "rmvpe.pt"))
    m = ns.rmvpe.RMVPE(os.path.join(mdir, "rmvpe.pt"), is_half=False, device="cpu")
    audio = S.synth_audio(1.3, seed=5)
    a = torch.from_numpy(audio).float().unsqueeze(0)
    mel = m.mel_extractor(a, center=True)
    hidden = m.mel2hidden(mel).squeeze(0).numpy()
    f0 = m.infer_from_audio(audio, thred=0.03)
    f0p = m.infer_from_audio_with_pitch(audio, thred=0.03, f0_min=50, f0_max=1600)
    # synthetic salience for the decoder: peaks, ties, sub-threshold rows, edge bins
    rng = np.random.default_rng(11)
    sal = rng.uniform(0, 0.02, (64, 360)).astype(np.float32)
    for i in range(48):
        cbin = [0, 1, 3, 4, 356, 358, 359][i % 7] if i < 14 else int(rng.integers(0, 360))
        w = np.exp(-0.5 * ((np.arange(360) - cbin) / 1.5) ** 2).astype(np.float32)
        sal[i] += (0.025 if i % 11 == 10 else rng.uniform(0.1, 0.9)) * w
    f0_syn = m.decode(sal.copy(), thred=0.03)
    np.savez_compressed(os.path.join(OUT, "rmvpe_1s.npz"), audio=audio, mel=np_(mel), salience=hidden, f0=f0, f0_plus=f0p,
                        syn_salience=sal, syn_f0=f0_syn)
    print("rmvpe_1s mel", mel.shape, "sal", hidden.shape, "f0 range", f0.min(), f0.max(), "voiced", (f0 > 0).mean())


def gen_synth_nono(ns, config, version, tag, T=12):
    """SynthesizerTrnMs{256,768}NSFsid_nono.infer(phone, phone_lengths, sid) - the no-f0 model family (one randn_like draw)."""
    cls = ns.models.SynthesizerTrnMs768NSFsid_nono if version == "v2" else ns.models.SynthesizerTrnMs256NSFsid_nono
    net = cls(*config)
    del net.enc_q
    net.load_state_dict(to_torch_sd(S.synth_state_dict(config, version, 0, f0=False)), strict=True)
    net.eval()
    rng = np.random.default_rng(22)
    D = 768 if version == "v2" else 256
    phone = torch.from_numpy(rng.standard_normal((1, T, D)).astype(np.float32) * 0.5)
    tape = NoiseTape(4321)
    with torch.no_grad(), patched_randn_like(tape):
        o, x_mask, (z, z_p, m_p, logs_p) = net.infer(phone, torch.LongTensor([T]), torch.LongTensor([2]))
    assert tape.shapes == [(1, config[2], T)], tape.shapes
    tape2 = NoiseTape(4321)
    np.savez_compressed(os.path.join(OUT, f"synth_{tag}.npz"), phone=np_(phone), sid=np.int64(2), noise_seed=np.int64(4321),
                        noise_z=np_(tape2((1, config[2], T))), m_p=np_(m_p), logs_p=np_(logs_p), z_p=np_(z_p), z=np_(z), wav=np_(o))
    print(f"synth_{tag}", o.shape, "wav rms", float(o.pow(2).mean().sqrt()), "max", float(o.abs().max()))


def gen_synth(ns, config, version, tag, T=16, full_taps=False, seed=0, family="plain"):
    cls = ns.models.SynthesizerTrnMs768NSFsid if version == "v2" else ns.models.SynthesizerTrnMs256NSFsid
    net = cls(*config, is_half=False)
    del net.enc_q
    sd = to_torch_sd(S.synth_state_dict(config, version, seed, family=family))
    missing = net.load_state_dict(sd, strict=True)
    net.eval()
    rng = np.random.default_rng(21)
    D = 768 if version == "v2" else 256
    phone = torch.from_numpy(rng.standard_normal((1, T, D)).astype(np.float32) * 0.5)
    f0 = S.designed_f0(T + 40, seed=0)[20:20 + T].copy()
    f0[T // 2: T // 2 + 3] = 0.0
    from oracle.pipeline import f0_postprocess
    coarse, f0f = f0_postprocess(f0.astype(np.float64))
    pitch = torch.from_numpy(coarse.astype(np.int64)).view(1, -1)
    pitchf = torch.from_numpy(f0f.astype(np.float32)).view(1, -1)
    tape = NoiseTape(1234)
    taps = {}
    hooks = []
    if full_taps:
        hooks.append(net.dec.m_source.register_forward_hook(lambda m, i, o: taps.__setitem__("har_source", o[0].transpose(1, 2))))
        hooks.append(net.dec.ups[0].register_forward_hook(lambda m, i, o: taps.__setitem__("ups0_raw", o)))
        hooks.append(net.dec.conv_pre.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_pre_raw", o)))
        hooks.append(net.enc_p.encoder.norm_layers_2[0].register_forward_hook(lambda m, i, o: taps.__setitem__("enc_p_layer0", o)))
    with torch.no_grad(), patched_randn_like(tape):
        o, x_mask, (z, z_p, m_p, logs_p) = net.infer(phone, torch.LongTensor([T]), pitch, pitchf, torch.LongTensor([3]))
    for h in hooks:
        h.remove()
    upp = int(np.prod(config[12]))
    assert tape.shapes == [(1, config[2], T), (1, T * upp, 1)], tape.shapes
    tape2 = NoiseTape(1234)
    d = dict(phone=np_(phone), pitch=np_(pitch), pitchf=np_(pitchf), sid=np.int64(3), noise_seed=np.int64(1234),
             noise_z=np_(tape2((1, config[2], T))), noise_src=np_(tape2((1, T * upp, 1))),
             m_p=np_(m_p), logs_p=np_(logs_p), z_p=np_(z_p), z=np_(z), wav=np_(o))
    d.update({k: np_(v) for k, v in taps.items()})
    np.savez_compressed(os.path.join(OUT, f"synth_{tag}.npz"), **d)
    print(f"synth_{tag}", o.shape, "wav rms", float(o.pow(2).mean().sqrt()), "max", float(o.abs().max()))


class Cfg:
    """Stand-in for the reference's global `config` singleton with chosen segmentation constants."""

    def __init__(self, x_pad=1, x_query=6, x_center=38, x_max=41):
        self.x_pad, self.x_query, self.x_center, self.x_max = x_pad, x_query, x_center, x_max
        self.is_half = False
        self.device = "cpu"


def run_ref_pipeline(ns, hub, vcd, audio, seed, cfg=None, designed_f0=None, **kw):
    vc = vcd["vc"] if cfg is None else ns.vc_infer_pipeline.VC(vcd["cpt"]["config"][-1], cfg)
    captured = {}
    orig_get_f0 = vc.get_f0

    def get_f0(*a, **k):
        r = orig_get_f0(*a, **k)
        captured["pitch"], captured["pitchf"] = np.array(r[0]), np.array(r[1])
        return r

    vc.get_f0 = get_f0
    if designed_f0 is not None:
        vc.f0_method_dict["pm"] = lambda x, **k: designed_f0(x)
    tape = NoiseTape(seed)
    args = dict(sid=0, f0_up_key=0, f0_method="rmvpe", merge_type="median", file_index="", index_rate=0.0,
                filter_radius=3, resample_sr=0, rms_mix_rate=0.25, protect=0.33, crepe_hop_length=160,
                f0_autotune=False)
    args.update(kw)
    try:
        with ref_shim.chdir_ws(), patched_randn_like(tape), torch.no_grad():
            out = ns.vc_infer_pipeline.vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub,
                                                 input_audio=(audio, 16000), config=cfg, **args)
    finally:
        vc.get_f0 = orig_get_f0          # (a second call on the same VC must not write into this call's `captured` through a nested wrapper)
    assert out is not None, "reference vc_single swallowed an exception"
    return out[0], out[1], captured, tape.shapes


def gen_pipeline(ns):
    hub, vcd = build_models(ns, S.CONFIG_40K_V2, "v2")
    # warm the lazy RMVPE constructor (it consumes global RNG; irrelevant with the patched randn_like)
    audio = S.synth_audio(2.0, seed=7)
    i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=99)
    np.savez_compressed(os.path.join(OUT, "pipeline_2s_rmvpe.npz"), audio=audio, out_i16=i16, sr=np.int64(sr),
                        pitch=cap["pitch"], pitchf=cap["pitchf"], noise_seed=np.int64(99))
    print("pipeline_2s_rmvpe", i16.shape, sr, shapes, "voiced", (cap["pitchf"] > 0).mean())

    # designed f0 + autotune + transposition + protect<0.5 + rms mix (all host-side branches)
    def dz(x):
        n = x.shape[0] // 160 + 1
        return S.designed_f0(n, seed=0).astype(np.float64)

    i16b, srb, capb, _ = run_ref_pipeline(ns, hub, vcd, audio, seed=100, designed_f0=dz, f0_method="pm",
                                          f0_up_key=3, f0_autotune=True, protect=0.2, rms_mix_rate=0.5)
    np.savez_compressed(os.path.join(OUT, "pipeline_2s_designed.npz"), audio=audio, out_i16=i16b, sr=np.int64(srb),
                        pitch=capb["pitch"], pitchf=capb["pitchf"], noise_seed=np.int64(100))
    print("pipeline_2s_designed", i16b.shape)

    # segmentation: short constants so a 7 s clip is cut into 3 segments (x_pad 1, x_query 1, x_center 2, x_max 3)
    cfg = Cfg(1, 1, 2, 3)
    audio7 = S.synth_audio(7.0, seed=8)
    i16c, src, capc, shapes = run_ref_pipeline(ns, hub, vcd, audio7, seed=101, cfg=cfg, designed_f0=dz, f0_method="pm",
                                               rms_mix_rate=1.0, protect=0.5)
    np.savez_compressed(os.path.join(OUT, "pipeline_7s_segmented.npz"), audio=audio7, out_i16=i16c, sr=np.int64(src),
                        pitchf=capc["pitchf"], noise_seed=np.int64(101), n_segments=np.int64(len(shapes) // 2),
                        seg_T=np.array([s[2] for s in shapes[0::2]], dtype=np.int64))
    print("pipeline_7s_segmented", i16c.shape, "segments", len(shapes) // 2, shapes)


def gen_hostdsp(ns):
    from scipy import signal
    audio = S.synth_audio(3.0, seed=9).astype(np.float32)
    filt = signal.filtfilt(ns.vc_infer_pipeline.bh, ns.vc_infer_pipeline.ah, audio)
    rng = np.random.default_rng(5)
    f0 = np.concatenate([np.zeros(5), rng.uniform(40, 1700, 200), [50.0, 1600.0, 49.9, 1601.0, 0.0]])
    fe = ns.pitch_extraction.FeatureExtractor(40000, Cfg())
    fe.f0_method_dict["pm"] = lambda **k: f0.copy()
    res = {}
    for key, tune in ((0, False), (-5, False), (7, True)):
        c, f = fe.get_f0(np.zeros(16000), key, "pm", f0_autotune=tune, f0_min=50, f0_max=1600)
        res[f"coarse_k{key}_a{int(tune)}"] = np.array(c)
        res[f"f0_k{key}_a{int(tune)}"] = np.array(f)
    d1 = S.synth_audio(2.0, seed=1).astype(np.float64)
    d2 = (rng.standard_normal(80000) * 0.1).astype(np.float32)
    mixed = ns.model_utils.change_rms(d1, 16000, d2.copy(), 40000, 0.25)
    st = np.stack([audio[:16000], audio[16000:32000] * 4.0])
    rem, _ = ns.audio.remix_audio((st, 16000), target_sr=16000)
    np.savez_compressed(os.path.join(OUT, "hostdsp.npz"), audio=audio, filtfilt=filt, f0_in=f0, rms_d1=d1, rms_d2=d2,
                        rms_out=mixed, remix_in=st, remix_out=rem, **res)
    print("hostdsp ok")


class StubIndex:
    """The `.search(x, k)` surface of a faiss L2 index (the only member VC.vc touches, reference vc_infer_pipeline.py:65) over `big`: exact
    squared-L2 k = 1 search in float64, ties to the smaller row.  faiss itself is absent from the container; what this pins is everything the
    REFERENCE does around the search - score -> weight, the gather from big_npy, the index_rate blend, the x2 up-sampling, the protect blend."""

    def __init__(self, big):
        self.big = np.asarray(big, dtype=np.float64)
        self.calls = []

    def search(self, x, k=1):
        assert k == 1
        x = np.asarray(x, dtype=np.float64)
        d = (x * x).sum(1)[:, None] - 2.0 * x @ self.big.T + (self.big * self.big).sum(1)[None]
        ix = d.argmin(1)
        score = np.maximum(d[np.arange(x.shape[0]), ix], 0).astype(np.float32)[:, None]
        self.calls.append(ix.astype(np.int64))
        return score, ix.astype(np.int64)[:, None]


def gen_pinned(ns, which):
    """Branches of the hot path that the first goldens left to the restated oracle alone, now through the real
    reference:
      pipeline_2s_index      vc_single with file_index = (index, big_npy) - the "preloaded file index" form load_index accepts
                             (pitch_extraction.py:55-57) - index_rate 0.75, protect 0.33: vc_infer_pipeline.py:58-95
      pipeline_2s_rmvpeplus  f0_method = "rmvpe+" through vc_single (pitch_extraction.py:197-201 -> lib/rmvpe.py:636-659: the clip turns
                             unvoiced frames into 50 Hz)
      pipeline_2s_f0file     f0_file = object with .name -> "time,f0" lines spliced over the extracted pitch (vc_infer_pipeline.py:146-151,
                             pitch_extraction.py:281-291)
      featinput              FeatureInput.go (preprocessing_utils.py:155-193) over two clips with load_input_audio stubbed: the three .npy
                             files per clip (dtype, shape, values)"""
    hub, vcd = build_models(ns, S.CONFIG_40K_V2, "v2")

    def dz(x):
        n = x.shape[0] // 160 + 1
        return S.designed_f0(n, seed=0).astype(np.float64)
    audio = S.synth_audio(2.0, seed=7)
    if "index" in which:
        # the bank: reference-HuBERT features of three other clips plus jitter, stored as float16 (0.6 MB) and used as float32 on both sides
        rng = np.random.default_rng(3)
        with torch.no_grad():
            base = np.concatenate([np_(hub.extract_features(torch.from_numpy(S.synth_audio(2.0, seed=50 + i)).view(1, -1), version="v2"))[0] for i in range(3)], 0)
        big = (base[rng.integers(0, base.shape[0], 768)] + 0.05 * rng.standard_normal((768, base.shape[1]))).astype(np.float16).astype(np.float32)
        idx = StubIndex(big)
        i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=111, designed_f0=dz, f0_method="pm", file_index=(idx, big), index_rate=0.75, protect=0.33)
        assert len(idx.calls) == 1
        i16n, _, _, _ = run_ref_pipeline(ns, hub, vcd, audio, seed=111, designed_f0=dz, f0_method="pm", file_index="", index_rate=0.75, protect=0.33)
        np.savez_compressed(os.path.join(OUT, "pipeline_2s_index.npz"), audio=audio, big_f16=big.astype(np.float16), ix=idx.calls[0].astype(np.int32), out_i16=i16,
                            out_i16_noindex=i16n, sr=np.int64(sr), pitch=cap["pitch"], pitchf=cap["pitchf"], noise_seed=np.int64(111), index_rate=np.float64(0.75),
                            protect=np.float64(0.33))
        print("pipeline_2s_index", i16.shape, "distinct rows hit", len(set(idx.calls[0].tolist())), "max |with - without| =", int(np.abs(i16.astype(np.int32) - i16n).max()))
    if "rmvpeplus" in which:
        i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=112, f0_method="rmvpe+", f0_up_key=-2)
        np.savez_compressed(os.path.join(OUT, "pipeline_2s_rmvpeplus.npz"), audio=audio, out_i16=i16, sr=np.int64(sr), pitch=cap["pitch"], pitchf=cap["pitchf"],
                            noise_seed=np.int64(112), f0_up_key=np.int64(-2))
        print("pipeline_2s_rmvpeplus", i16.shape, "min f0", float(cap["pitchf"].min()), "frames at the 50 Hz floor (x 2^(-2/12))", int((np.abs(cap["pitchf"] - 50 * 2 ** (-2 / 12)) < 1e-3).sum()))
    if "f0file" in which:
        # a hand-drawn contour over 0.3 .. 1.5 s: a glide, an unvoiced gap (0 Hz points) and a jump; 4 digits after the point like a text editor would save
        pts = [(0.30, 220.0), (0.55, 330.0), (0.70, 330.0), (0.71, 0.0), (0.90, 0.0), (0.91, 440.0), (1.20, 392.5), (1.50, 261.63)]
        text = "\n".join(f"{t:.4f},{f:.4f}" for t, f in pts) + "\n"
        path = os.path.join(ns.ws, "f0_curve.csv")
        with open(path, "w") as f:
            f.write(text)
        fobj = types.SimpleNamespace(name=path)
        i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=113, designed_f0=dz, f0_method="pm", f0_file=fobj, f0_up_key=2)
        i16n, _, capn, _ = run_ref_pipeline(ns, hub, vcd, audio, seed=113, designed_f0=dz, f0_method="pm", f0_up_key=2)
        np.savez_compressed(os.path.join(OUT, "pipeline_2s_f0file.npz"), audio=audio, f0_text=np.frombuffer(text.encode(), dtype=np.uint8), out_i16=i16, sr=np.int64(sr),
                            pitch=cap["pitch"], pitchf=cap["pitchf"], pitchf_nofile=capn["pitchf"], noise_seed=np.int64(113), f0_up_key=np.int64(2))
        print("pipeline_2s_f0file", i16.shape, "frames changed by the splice", int((cap["pitchf"] != capn["pitchf"]).sum()))
    if "featinput" in which:
        pu = ref_shim.load_preprocessing_utils()
        exp = tempfile.mkdtemp(prefix="rvc_featinput_")
        clips = {"0_0.wav": S.synth_audio(1.7, seed=61), "0_1.wav": S.synth_audio(2.3, seed=62)}
        pu.load_input_audio = lambda path, sr: (clips[os.path.basename(path)].copy(), sr)       # (the wav reader needs soundfile / ffmpeg)
        out = {}
        for version, D in (("v2", 768), ("v1", 256)):
            d = os.path.join(exp, version)
            for sub in ("2a_f0", "2b-f0nsf", f"3_feature{D}"):
                os.makedirs(os.path.join(d, sub))
            fi = pu.FeatureInput(hub, "rmvpe", d, samplerate=16000, hop_size=160, device="cpu", version=version, if_f0=True)
            paths = [(os.path.join(d, n), os.path.join(d, "2a_f0", n), os.path.join(d, "2b-f0nsf", n), os.path.join(d, f"3_feature{D}", n)) for n in clips]
            with ref_shim.chdir_ws(), torch.no_grad():
                fi.go(paths)
            for n in clips:
                key = n.replace(".wav", "")
                coarse, nsf, feat = (np.load(os.path.join(d, sub, n + ".npy")) for sub in ("2a_f0", "2b-f0nsf", f"3_feature{D}"))
                out[f"{version}_{key}_coarse"], out[f"{version}_{key}_nsf"], out[f"{version}_{key}_feat"] = coarse, nsf, feat
                print("featinput", version, key, coarse.dtype, coarse.shape, nsf.dtype, nsf.shape, feat.dtype, feat.shape, "coarse max", int(coarse.max()))
            # second go(): everything present -> nothing rewritten (skip rule, :167-172)
            before = {n: os.path.getmtime(os.path.join(d, f"3_feature{D}", n + ".npy")) for n in clips}
            with ref_shim.chdir_ws(), torch.no_grad():
                fi.go(paths)
            assert before == {n: os.path.getmtime(os.path.join(d, f"3_feature{D}", n + ".npy")) for n in clips}
            out[f"{version}_log"] = np.frombuffer(open(os.path.join(d, "extract_f0_feature.log")).read().encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "featinput.npz"), seconds=np.array([1.7, 2.3]), seeds=np.array([61, 62]), **out)
        shutil.rmtree(exp)


def _audio_digest(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def gen_heavy(ns, which):
    """Second weight family (synthetic.*_state_dict(seed=1, family="heavy"): log-normal channel gains + x10 - x30 outlier channels in HuBERT's FFN /
    LayerNorm and the generator's ResBlock pairs) through the REAL reference: the guard for the reduced-precision matrix arithmetic (bf16x3 everywhere,
    fp16x2 on the ResBlock pairs), which the plain Gaussian family exercises on one distribution only.  RMVPE keeps its seed-0 weights (its procedural
    family is rescaled for a sensible voiced / unvoiced mix; the pitch path is pinned elsewhere).
      heavy_hubert   hubert_1s_heavy           stage taps
      heavy_synth    synth_40k_v2_heavy        stage taps, T = 16
      heavy_2s       pipeline_2s_rmvpe_heavy   vc_single, 2 s
      heavy_30s      pipeline_30s_40k_v2_heavy BASELINE configs[2] size"""
    import time
    if "heavy_hubert" in which:
        gen_hubert(ns, seed=1, family="heavy", name="hubert_1s_heavy")
    if "heavy_synth" in which:
        gen_synth(ns, S.CONFIG_40K_V2, "v2", "40k_v2_heavy", T=16, full_taps=True, seed=1, family="heavy")
    if "heavy_2s" in which or "heavy_30s" in which:
        hub, vcd = build_models(ns, S.CONFIG_40K_V2, "v2", seed=1, family="heavy", rmvpe_seed=0)
        if "heavy_2s" in which:
            audio = S.synth_audio(2.0, seed=7)
            i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=99)
            np.savez_compressed(os.path.join(OUT, "pipeline_2s_rmvpe_heavy.npz"), audio=audio, out_i16=i16, sr=np.int64(sr),
                                pitch=cap["pitch"], pitchf=cap["pitchf"], noise_seed=np.int64(99))
            print("pipeline_2s_rmvpe_heavy", i16.shape, sr, shapes, "voiced", (cap["pitchf"] > 0).mean(), "rms", float(np.sqrt(np.mean(i16.astype(np.float64) ** 2))))
        if "heavy_30s" in which:
            audio = S.synth_audio(30.0, seed=100)
            t0 = time.time()
            i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=301)
            np.savez_compressed(os.path.join(OUT, "pipeline_30s_40k_v2_heavy.npz"), audio_seconds=np.float64(30.0), audio_seed=np.int64(100), audio_sha256=_audio_digest(audio),
                                out_i16=i16, sr=np.int64(sr), pitch=cap["pitch"].astype(np.int16), pitchf=cap["pitchf"], noise_seed=np.int64(301),
                                n_segments=np.int64(len(shapes) // 2), seg_T=np.array([s[2] for s in shapes[0::2]], dtype=np.int64))
            print("pipeline_30s_40k_v2_heavy", i16.shape, sr, "voiced", (cap["pitchf"] > 0).mean(), "rms", float(np.sqrt(np.mean(i16.astype(np.float64) ** 2))), f"{time.time() - t0:.1f} s")


def gen_fullsize(ns, which):
    """BASELINE.json's full-size configurations through the REAL reference with the real segmentation constants (x_pad 1, x_query 6,
    x_center 38, x_max 41).  The input clips are regenerated from comfy-rvc_amd/synthetic.py::synth_audio on both sides (a SHA-256 of the
    samples is stored instead of 1.9-2.9 MB of audio); stored: int16 output, coarse pitch, f0.
      pipeline_30s_40k_v2  configs[2] (C3): 30 s clip, 40k_v2, rmvpe f0
      pipeline_30s_48k_v2  one clip of configs[3] (C4): 30 s, 48k_v2
      pipeline_45s_40k_v2  > x_max = 41 s: the cut search of vc_infer_pipeline.py:123-135 runs with the real constants
      rmvpe_60s            configs[1] (C2): RMVPE alone on the padded 60 s clip: f0, per-frame salience maximum / arg-max, salience sums"""
    import time
    if "full40" in which or "full45" in which:
        hub, vcd = build_models(ns, S.CONFIG_40K_V2, "v2")
        for tag, secs, aseed, nseed in (("pipeline_30s_40k_v2", 30.0, 100, 301), ("pipeline_45s_40k_v2", 45.0, 102, 303)):
            if ("full40" if secs == 30.0 else "full45") not in which:
                continue
            audio = S.synth_audio(secs, seed=aseed)
            t0 = time.time()
            i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=nseed)
            np.savez_compressed(os.path.join(OUT, tag + ".npz"), audio_seconds=np.float64(secs), audio_seed=np.int64(aseed), audio_sha256=_audio_digest(audio),
                                out_i16=i16, sr=np.int64(sr), pitch=cap["pitch"].astype(np.int16), pitchf=cap["pitchf"], noise_seed=np.int64(nseed),
                                n_segments=np.int64(len(shapes) // 2), seg_T=np.array([s[2] for s in shapes[0::2]], dtype=np.int64))
            print(tag, i16.shape, sr, "segments", len(shapes) // 2, [s[2] for s in shapes[0::2]], "voiced", (cap["pitchf"] > 0).mean(), f"{time.time() - t0:.1f} s")
    if "full48" in which:
        hub, vcd = build_models(ns, S.CONFIG_48K_V2, "v2")
        audio = S.synth_audio(30.0, seed=200)
        t0 = time.time()
        i16, sr, cap, shapes = run_ref_pipeline(ns, hub, vcd, audio, seed=302)
        np.savez_compressed(os.path.join(OUT, "pipeline_30s_48k_v2.npz"), audio_seconds=np.float64(30.0), audio_seed=np.int64(200), audio_sha256=_audio_digest(audio),
                            out_i16=i16, sr=np.int64(sr), pitch=cap["pitch"].astype(np.int16), pitchf=cap["pitchf"], noise_seed=np.int64(302),
                            n_segments=np.int64(len(shapes) // 2))
        print("pipeline_30s_48k_v2", i16.shape, sr, f"{time.time() - t0:.1f} s")
    if "rmvpe60" in which:
        mdir = os.path.join(ns.ws, "models")
        torch.save(to_torch_sd(S.rmvpe_state_dict(0)), os.path.join(mdir, "rmvpe.pt"))
        m = ns.rmvpe.RMVPE(os.path.join(mdir, "rmvpe.pt"), is_half=False, device="cpu")
        clip = S.synth_audio(60.0, seed=9)
        audio = np.pad(clip, (16000, 16000), mode="reflect")             # as VC.pipeline pads it (vc_infer_pipeline.py:141)
        t0 = time.time()
        with torch.no_grad():
            mel = m.mel_extractor(torch.from_numpy(audio).float().unsqueeze(0), center=True)
            hidden = m.mel2hidden(mel).squeeze(0).numpy()
        f0 = m.decode(hidden.copy(), thred=0.03)
        assert np.array_equal(f0, m.infer_from_audio(audio, thred=0.03))
        np.savez_compressed(os.path.join(OUT, "rmvpe_60s.npz"), audio_seconds=np.float64(60.0), audio_seed=np.int64(9), audio_sha256=_audio_digest(audio),
                            f0=f0, sal_max=hidden.max(axis=1), sal_argmax=hidden.argmax(axis=1).astype(np.int16),
                            sal_rowsum=hidden.astype(np.float64).sum(axis=1), sal_colsum=hidden.astype(np.float64).sum(axis=0),
                            sal_sub=hidden[::50].copy())
        print("rmvpe_60s", hidden.shape, "voiced", (f0 > 0).mean(), f"{time.time() - t0:.1f} s")


def gen_mdx23c():
    """The reference's own TFC_TDF_net (lib/karafan/tfc_tdf.py is pure torch and imports stand-alone) and demix_mdxv3's arithmetic on a
    reduced configuration with procedural weights: spectrogram, first conv, first encoder scale, bottleneck, mask-head output, separated
    chunk, and a 3-chunk overlap-add of a longer clip."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_tfc_tdf", os.path.join(ref_shim.REF_ROOT, "lib", "karafan", "tfc_tdf.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    class NS(dict):
        __getattr__ = dict.__getitem__

    def ns(d):
        return NS({k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    cfg = S.mdx23c_config(**S.MDX23C_SMALL)
    net = m.TFC_TDF_net(ns(cfg)).eval()
    sd = S.mdx23c_state_dict(cfg, 0)
    assert [k for k in net.state_dict()] == [n for n, _, _ in S.mdx23c_spec(cfg)], "state-dict layout differs from the reference module"
    net.load_state_dict(to_torch_sd(sd), strict=True)
    rng = np.random.default_rng(31)
    C = cfg["audio"]["chunk_size"]
    x = (rng.standard_normal((2, C)) * 0.3).astype(np.float32)
    taps = {}
    hooks = [net.first_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("first_conv", o[0])),
             net.encoder_blocks[0].tfc_tdf.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc0", o[0])),
             net.bottleneck_block.register_forward_hook(lambda mod, i, o: taps.__setitem__("bottleneck", o[0])),
             net.final_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("mask_out", o[0]))]
    with torch.no_grad():
        specg = net.stft(torch.from_numpy(x)[None])[0]
        y = net(torch.from_numpy(x)[None])[0]
    for h in hooks:
        h.remove()
    # demix_mdxv3's arithmetic (inference.py:32-74) around the reference module, overlap 4, on a clip of 2.6 chunks
    overlap = 4
    clip = (rng.standard_normal((2, int(2.6 * C))) * 0.3).astype(np.float32)
    H = C // overlap
    L = clip.shape[1]
    pad_size = H - (L - C) % H
    mix = torch.cat([torch.zeros(2, C - H), torch.from_numpy(clip), torch.zeros(2, pad_size + C - H)], 1)
    chunks = mix.unfold(1, C, H).transpose(0, 1)
    X = torch.zeros(2, *mix.shape)
    with torch.no_grad():
        for cnt, ch in enumerate(chunks):
            X[..., cnt * H: cnt * H + C] += net(ch[None])[0]
    est = (X[..., C - H: -(pad_size + C - H)] / overlap).numpy()
    np.savez_compressed(os.path.join(OUT, "mdx23c_small.npz"), x=x, spec=np_(specg), out=np_(y), clip=clip, demix=est, overlap=np.int64(overlap),
                        **{k: np_(v) for k, v in taps.items()})
    print("mdx23c_small", y.shape, "rms", float(y.pow(2).mean().sqrt()), "demix", est.shape, {k: tuple(v.shape) for k, v in taps.items()})


def gen_mdx23c_full():
    """ONE 5.9 s chunk through the reference's TFC_TDF_net at the SHIPPED recipe (lib/karafan/Data/model_2_stem_full_band_8k.yaml: n_fft 8192,
    dim_f 4096, dim_t 256, 128 channels + 128 per scale, 5 scales, 112 M procedural parameters; minutes of CPU).  The fixture keeps what a
    value test needs and stays small: every 64th output sample, one dense window, per-(stem, channel) norms, and norm / sub-sampled taps of the
    first conv, the first encoder scale, the bottleneck and the mask head.  The input is regenerated from its seed (SHA-256 stored)."""
    import hashlib
    import importlib.util
    import time
    import yaml
    spec = importlib.util.spec_from_file_location("ref_tfc_tdf", os.path.join(ref_shim.REF_ROOT, "lib", "karafan", "tfc_tdf.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    class NS(dict):
        __getattr__ = dict.__getitem__

    def ns(d):
        return NS({k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG as cfg
    ref_yaml = yaml.safe_load(open(os.path.join(ref_shim.REF_ROOT, "lib", "karafan", "Data", "model_2_stem_full_band_8k.yaml")))
    for sec in ("audio", "model"):      # the recipe the product ships IS the reference's yaml
        for k, v in cfg[sec].items():
            assert ref_yaml[sec][k] == v, (sec, k, ref_yaml[sec][k], v)
    net = m.TFC_TDF_net(ns(cfg)).eval()
    sd = S.mdx23c_state_dict(cfg, 0)
    assert [k for k in net.state_dict()] == [n for n, _, _ in S.mdx23c_spec(cfg)], "state-dict layout differs from the reference module"
    net.load_state_dict(to_torch_sd(sd), strict=True)
    C = cfg["audio"]["chunk_size"]
    x = S.mdx23c_full_chunk()
    assert x.shape == (2, C)
    taps = {}
    hooks = [net.first_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("first_conv", o[0])),
             net.encoder_blocks[0].tfc_tdf.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc0", o[0])),
             net.bottleneck_block.register_forward_hook(lambda mod, i, o: taps.__setitem__("bottleneck", o[0])),
             net.final_conv.register_forward_hook(lambda mod, i, o: taps.__setitem__("mask_out", o[0]))]
    t0 = time.time()
    with torch.no_grad():
        y = net(torch.from_numpy(x)[None])[0]
    for h in hooks:
        h.remove()
    y = np_(y)
    out = dict(out_sub=y[..., ::64].copy(), out_win=y[..., 100000:104096].copy(),
               out_norm=np.sqrt((y.astype(np.float64) ** 2).sum(-1)), out_absmax=np.abs(y).max(-1),
               audio_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(x).tobytes()).digest(), dtype=np.uint8))
    for k, v in taps.items():
        v = np_(v)
        out[k + "_norm"] = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim))))       # per channel
        out[k + "_sub"] = v.reshape(v.shape[0], -1)[::max(1, v.shape[0] // 8), ::997].copy()
    np.savez_compressed(os.path.join(OUT, "mdx23c_full_chunk.npz"), **out)
    print("mdx23c_full_chunk", y.shape, "rms", float(np.sqrt((y ** 2).mean())), f"{time.time() - t0:.1f} s", {k: tuple(v.shape) for k, v in out.items()})


def gen_mdx23c_demix_full():
    """The reference's OWN demix_mdxv3 (lib/karafan/inference.py:32-74, imported through the shim with onnxruntime stubbed - the function never touches it)
    around the reference's TFC_TDF_net at the SHIPPED recipe: a 2.96 s stereo clip at overlap 2 = THREE overlapping full-size chunks, zero padding,
    NaN guard, overlap-add, division.  Stored: every 16th sample of both stems, a dense window across a chunk boundary, per-(stem, channel) norms.
    ~10 minutes of CPU."""
    import hashlib
    import importlib
    import time
    ns = ref_shim.load_reference()
    sys.modules.setdefault("onnxruntime", types.ModuleType("onnxruntime"))
    with ref_shim.chdir_ws():
        inf = importlib.import_module(f"{ref_shim.PKG}.lib.karafan.inference")

    class NS(dict):
        __getattr__ = dict.__getitem__

    def nsd(d):
        return NS({k: nsd(v) if isinstance(v, dict) else v for k, v in d.items()})
    from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG as cfg
    net = inf.tfc_tdf.TFC_TDF_net(nsd(cfg)).eval()
    net.load_state_dict(to_torch_sd(S.mdx23c_state_dict(cfg, 0)), strict=True)
    C = cfg["audio"]["hop_length"] * (cfg["inference"]["dim_t"] - 1)
    overlap = 2
    L = C // overlap
    mix = np.stack([S.synth_audio(L / 44100.0, seed=71, sr=44100)[:L], S.synth_audio(L / 44100.0, seed=72, sr=44100)[:L]]).astype(np.float32)
    assert mix.shape == (2, L)
    t0 = time.time()
    est = inf.demix_mdxv3(mix, net, "cpu", nsd(cfg), overlap)
    y = np.stack([est["Vocals"], est["Instrumental"]])
    assert y.shape == (2, 2, L)
    w0 = L // 2 - 2048
    np.savez_compressed(os.path.join(OUT, "mdx23c_demix_full.npz"), seeds=np.array([71, 72]), n=np.int64(L), overlap=np.int64(overlap),
                        audio_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(mix).tobytes()).digest(), dtype=np.uint8),
                        out_sub=y[..., ::16].copy(), out_win=y[..., w0:w0 + 4096].copy(), win0=np.int64(w0),
                        out_norm=np.sqrt((y.astype(np.float64) ** 2).sum(-1)), out_absmax=np.abs(y).max(-1))
    print("mdx23c_demix_full", y.shape, "rms", float(np.sqrt((y ** 2).mean())), f"{time.time() - t0:.1f} s")


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_shim.load_reference()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["hostdsp", "hubert", "rmvpe", "synth", "pipeline"]
    if "hostdsp" in which:
        gen_hostdsp(ns)
    if "hubert" in which:
        gen_hubert(ns)
    if "rmvpe" in which:
        gen_rmvpe(ns)
    if "synth" in which:
        gen_synth(ns, S.CONFIG_40K_V2, "v2", "40k_v2", T=16, full_taps=True)
        gen_synth(ns, S.CONFIG_48K_V2, "v2", "48k_v2", T=12)
        gen_synth(ns, S.CONFIG_40K_V1, "v1", "40k_v1", T=12)
    if "synth_shapes" in which:      # the other shipped generator shapes (added later; does not touch the vectors above)
        gen_synth(ns, S.CONFIG_32K_V1, "v1", "32k_v1", T=12)
        gen_synth(ns, S.CONFIG_48K_V1, "v1", "48k_v1", T=12)
        gen_synth(ns, S.CONFIG_32K_V2, "v2", "32k_v2", T=12)
    if "synth_nono" in which:        # the no-f0 model family
        gen_synth_nono(ns, S.CONFIG_40K_V2, "v2", "40k_v2_nono")
        gen_synth_nono(ns, S.CONFIG_40K_V1, "v1", "40k_v1_nono")
    if "pipeline" in which:
        gen_pipeline(ns)
    if {"index", "rmvpeplus", "f0file", "featinput"} & set(which):      # round 5: branches pinned through the real reference (see gen_pinned)
        gen_pinned(ns, which)
    if "mdx23c" in which:
        gen_mdx23c()
    if "mdx23c_full" in which:       # one chunk at the shipped recipe (minutes of CPU)
        gen_mdx23c_full()
    if "mdx23c_demix_full" in which: # three full-size chunks through the reference's own demix_mdxv3 (~10 minutes of CPU)
        gen_mdx23c_demix_full()
    if {"heavy_hubert", "heavy_synth", "heavy_2s", "heavy_30s"} & set(which):   # round 6: the second weight family
        gen_heavy(ns, which)
    if {"full40", "full45", "full48", "rmvpe60"} & set(which):   # BASELINE.json's full-size configurations (minutes of CPU time)
        gen_fullsize(ns, which)


if __name__ == "__main__":
    main()
