"""CPU: the product's host-side logic (no GPU compute) against the golden vectors captured from the reference."""
import numpy as np
import pytest

from conftest import golden, rel_err


def _fe():
    from comfy_rvc_amd.config import Config
    from comfy_rvc_amd.pitch_extraction import FeatureExtractor
    return FeatureExtractor(40000, Config())


def test_constants_match_reference_cpu_set():
    fe = _fe()
    assert (fe.t_pad, fe.t_pad_tgt, fe.t_pad2, fe.t_query, fe.t_center, fe.t_max) == (16000, 40000, 32000, 96000, 608000, 656000)
    assert (fe.sr, fe.window, fe.f0_bins) == (16000, 160, 256)


def test_get_f0_postprocessing_matches_reference():
    g = golden("hostdsp.npz")
    fe = _fe()
    fe.f0_method_dict["pm"] = lambda **k: g["f0_in"].copy()
    for key, tune in ((0, False), (-5, False), (7, True)):
        coarse, f0 = fe.get_f0(np.zeros(16000), key, "pm", f0_autotune=tune, f0_min=50, f0_max=1600)
        assert coarse.dtype == np.int16 and np.array_equal(coarse, g[f"coarse_k{key}_a{int(tune)}"])
        assert np.allclose(f0, g[f"f0_k{key}_a{int(tune)}"], rtol=1e-12, atol=0)
    coarse, _ = fe.get_f0(np.zeros(16000), 0, ["pm"], f0_min=50, f0_max=1600)       # single-element method list
    assert np.array_equal(coarse, g["coarse_k0_a0"])


def test_unsupported_f0_methods_fail_loudly():
    import pytest
    fe = _fe()
    with pytest.raises(NotImplementedError):
        fe.get_f0(np.zeros(16000), 0, "harvest")


def test_filter_coefficients_rms_and_remix():
    from scipy import signal
    from comfy_rvc_amd import vc_infer_pipeline as P
    from comfy_rvc_amd.lib.audio import remix_audio
    from comfy_rvc_amd.lib.model_utils import change_rms
    g = golden("hostdsp.npz")
    assert np.array_equal(signal.filtfilt(P.bh, P.ah, g["audio"]), g["filtfilt"])
    out = change_rms(g["rms_d1"], 16000, g["rms_d2"].copy(), 40000, 0.25)
    assert rel_err(out, g["rms_out"]) < 1e-6
    rem, sr = remix_audio((g["remix_in"], 16000), target_sr=16000)
    assert sr == 16000 and np.array_equal(rem, g["remix_out"])


def test_vc_single_returns_none_like_the_reference_on_bad_input():
    from comfy_rvc_amd.vc_infer_pipeline import vc_single
    assert vc_single(cpt=None, net_g=None, vc=None, hubert_model=object()) is None
    assert vc_single(cpt={"config": [40000]}, net_g=object(), vc=object(), hubert_model=object(), input_audio=None) is None


def test_node_surface_matches_reference():
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N
    assert set(N.NODE_CLASS_MAPPINGS) >= {"LoadRVCModelNode", "RVCNode", "LoadHubertModel", "LoadPitchExtractionParams"}
    it = N.LoadPitchExtractionParams.INPUT_TYPES()["required"]
    assert it["f0_method"][0] == ["crepe", "mangio-crepe", "rmvpe", "rmvpe+"] and it["f0_method"][1]["default"] == "rmvpe"
    assert N.LoadPitchExtractionParams.RETURN_TYPES == ("PITCH_EXTRACTION",)
    assert N.RVCNode.RETURN_TYPES == ("VHS_AUDIO", "AUDIO") and N.RVCNode.OUTPUT_NODE is True and N.RVCNode.FUNCTION == "convert"
    params = N.LoadPitchExtractionParams().load_params(f0_method="rmvpe", f0_autotune=False, index_rate=.75, resample_sr=0,
                                                       rms_mix_rate=.25, protect=.25, crepe_hop_length=160)[0]
    assert params["f0_method"] == "rmvpe" and params["protect"] == .25
    a = N.to_audio_dict(np.zeros(100, dtype=np.float32), 40000)
    assert tuple(a["waveform"].shape) == (1, 100, 1) and a["sample_rate"] == 40000


def test_clip_lanes_keep_order_overlap_and_propagate_errors():
    """ClipLanes (two clips in flight per GPU): results come back in clip order whichever lane converts them, the lanes really
    run concurrently, and a failing clip raises on the consumer side instead of hanging it."""
    import threading
    import time
    from comfy_rvc_amd.parallel import ClipLanes
    active, peak, lock, lane_of = [0], [0], threading.Lock(), {}

    def make(k):
        def fn(clip, i):
            with lock:
                active[0] += 1
                peak[0] = max(peak[0], active[0])
                lane_of[i] = k
            time.sleep(0.02 if i % 2 else 0.005)          # uneven durations: a free lane pulls the next clip
            with lock:
                active[0] -= 1
            return np.full(3, clip + i, dtype=np.int16)
        return fn
    pool = ClipLanes([make(0), make(1)])
    out = pool.map(list(range(10, 20)))
    assert [int(o[0]) for o in out] == [10 + 2 * i for i in range(10)]
    assert peak[0] == 2 and set(lane_of.values()) == {0, 1}
    assert pool.map([]) == []

    def bad(clip, i):
        if i == 3:
            raise ValueError("clip 3 is broken")
        return np.zeros(1, dtype=np.int16)
    with pytest.raises(ValueError, match="clip 3"):
        ClipLanes([bad, bad]).map(list(range(8)))


def test_clip_lanes_fail_instead_of_hanging_when_a_lane_cannot_start(monkeypatch):
    """A lane thread that dies outside a conversion (device setup) must fail the map, not leave the consumer waiting."""
    from comfy_rvc_amd import parallel as P
    pool = P.ClipLanes([lambda c, i: np.zeros(1, np.int16)] * 2)
    pool._streams = [None, None]                       # pretend to be a GPU pool whose device cannot be selected
    pool.device = "cuda:0"
    monkeypatch.setattr(P.torch.cuda, "set_device", lambda d: (_ for _ in ()).throw(RuntimeError("no such device")))
    with pytest.raises(RuntimeError, match="no such device"):
        pool.map([1, 2, 3])


@pytest.mark.parametrize("orig,target", [(44100, 16000), (48000, 16000), (40000, 48000), (32000, 44100)])
def test_resample_filter_meets_the_soxr_hq_specification(orig, target):
    """The stand-in for librosa.resample(res_type="soxr_hq") (absent here: parity unpinned) restates its specification: linear phase,
    flat up to 0.913 of the lower Nyquist frequency, > 120 dB down from Nyquist on, unit gain in every polyphase branch."""
    from comfy_rvc_amd.lib.audio import design_resample_filter
    h, half, up, down = design_resample_filter(orig, target)
    assert h.shape[0] == 2 * half + 1 and np.array_equal(h, h[::-1]) and up * orig == down * target
    assert max(abs(h[p::up].sum() - 1.0) for p in range(up)) < 1e-6
    H = np.abs(np.fft.rfft(h, 1 << 21)) / up
    f = np.fft.rfftfreq(1 << 21) * orig * up
    fn = min(orig, target) / 2
    assert 20 * np.log10(H[f >= fn].max()) < -120.0
    assert np.abs(20 * np.log10(H[f <= 0.913 * fn])).max() < 1e-4


def test_wav_bytes_round_trip_and_headers():
    """audio_to_bytes / bytes_to_audio (reference lib/audio.py:188-210): PCM_16 when the samples are int16-valued, float32 otherwise,
    [C, N] on the way back; checked against the stdlib wave reader and hand-built 8/24/32-bit and EXTENSIBLE streams."""
    import io
    import struct
    import wave
    from comfy_rvc_amd.lib.audio import audio_to_bytes, bytes_to_audio, get_audio
    rng = np.random.default_rng(0)
    x16 = (rng.standard_normal(1001) * 9000).astype(np.int16)      # odd length: payload padding
    b = audio_to_bytes(x16, 40000)
    w = wave.open(io.BytesIO(b))
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 40000, 1001)
    assert np.array_equal(np.frombuffer(w.readframes(1001), dtype="<i2"), x16)
    y, sr = bytes_to_audio(b)
    assert sr == 40000 and y.dtype == np.float64 and np.array_equal(y * 32768.0, x16.astype(np.float64))
    st = np.stack([np.sin(np.arange(700) / 9.0), np.cos(np.arange(700) / 7.0)]).astype(np.float32) * 0.6    # [C, N] float
    b2 = audio_to_bytes(st, 16000)
    assert struct.unpack("<H", b2[20:22])[0] == 3                   # IEEE float
    y2, sr2 = get_audio(lambda: b2)                                 # VHS_AUDIO thunk -> bytes -> (ndarray [C, N], sr)
    assert sr2 == 16000 and y2.shape == (2, 700) and np.array_equal(y2.astype(np.float32), st)
    b3 = audio_to_bytes(st[0], 16000, to_int16=True, to_stereo=True)
    y3, _ = bytes_to_audio(b3)
    assert y3.shape == (2, 700) and np.array_equal(y3[0], y3[1]) and np.abs(y3[0] - st[0]).max() < 1.0 / 32768 + 1e-9

    def riff(tag, bits, payload, nch=1, ext=False):
        align = nch * bits // 8
        fmt = struct.pack("<HHIIHH", 0xFFFE if ext else tag, nch, 8000, 8000 * align, align, bits)
        if ext:
            fmt += struct.pack("<HHI", 22, bits, 0) + struct.pack("<H", tag) + b"\0" * 14
        junk = b"LIST" + struct.pack("<I", 3) + b"abc\0"          # odd-sized chunk ahead of the data chunk
        body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + junk + b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body
    v = np.array([-8388608, -1, 0, 1, 8388607], dtype=np.int64)
    p24 = b"".join(struct.pack("<i", int(t))[:3] for t in v)
    assert np.array_equal(bytes_to_audio(riff(1, 24, p24))[0], v / 8388608.0)
    assert np.array_equal(bytes_to_audio(riff(1, 8, bytes([0, 128, 255])))[0], np.array([-1.0, 0.0, 127 / 128.0]))
    assert np.array_equal(bytes_to_audio(riff(1, 32, struct.pack("<2i", -2 ** 31, 2 ** 30), ext=True))[0], np.array([-1.0, 0.5]))
    assert np.array_equal(bytes_to_audio(riff(3, 64, struct.pack("<2d", 0.25, -0.5)))[0], np.array([0.25, -0.5]))
    with pytest.raises(NotImplementedError):
        bytes_to_audio(b"fLaC" + b"\0" * 64)
    with pytest.raises(NotImplementedError):
        audio_to_bytes(x16, 40000, format="FLAC")


def test_crepe_host_decoding_matches_oracle():
    """lib/crepe.py's host-side decoding (bin masking, softmax, banded Viterbi, cents + dither, periodicity, NaN-aware median / mean
    filters) against the oracle's literal restatement of torchcrepe, on flat (procedural network) and peaked (moving ridge) salience."""
    import torch
    from comfy_rvc_amd import synthetic as S
    from comfy_rvc_amd.lib import crepe as pc
    from oracle import crepe as oc
    x = S.synth_audio(2.0, seed=3)
    with torch.no_grad():
        flat = oc.infer(S.crepe_state_dict("tiny", 0), oc.preprocess(torch.from_numpy(x)[None], 160), "tiny").numpy()
    n = flat.shape[0]
    centre = (180 + 120 * np.sin(np.arange(n) / 9.0)).astype(int)
    ridge = np.stack([0.02 + 0.9 * np.exp(-0.5 * ((np.arange(360) - c) / 2.0) ** 2) for c in centre]).astype(np.float32)
    assert (pc.frequency_to_bins(50.0), pc.frequency_to_bins(1600.0, ceil=True)) == (oc.frequency_to_bins(50.0), oc.frequency_to_bins(1600.0, torch.ceil)) == (39, 340)
    for P in (flat, ridge):
        np.random.seed(5)
        a, pa = oc.postprocess(torch.from_numpy(P), 50, 1100, True)
        np.random.seed(5)
        b, pb = pc.postprocess(P.T, 50, 1100, True)
        assert np.allclose(a[0].numpy(), b, rtol=1e-6) and np.array_equal(pa[0].numpy(), pb)
        assert np.array_equal(oc.filter_median(pa, 3)[0].numpy(), pc.filter_median(pb, 3))
        assert np.allclose(oc.filter_mean(a, 3)[0].numpy(), pc.filter_mean(b, 3), rtol=1e-6, equal_nan=True)
    assert len(set(np.round(b))) > 50                               # the ridge case really moves across bins


def test_faiss_ivf_flat_reader_roundtrip(tmp_path):
    """lib/faiss_io.py: an IVF{n},Flat file written in faiss's on-disk layout (full and sparse list-size tables, empty lists, ids in list
    order) comes back as big_npy in id order - what faiss.read_index + reconstruct_n(0, ntotal) gives the reference (pitch_extraction.py:52-73)."""
    import struct
    from comfy_rvc_amd.lib import faiss_io as F
    rng = np.random.default_rng(3)
    x = rng.standard_normal((700, 48)).astype(np.float32)
    for nlist, sparse in ((16, False), (64, True), (5, None)):
        p = str(tmp_path / f"added_IVF{nlist}_Flat_nprobe_1_v2.index")
        cent = x[rng.choice(700, nlist, replace=False)] if sparse is not True else np.concatenate([x[:8], 100 + rng.standard_normal((nlist - 8, 48)).astype(np.float32)])
        assign = F.write_ivf_flat(p, x, nlist, centroids=cent, sparse=sparse)
        v, info = F.read_index_vectors(p)
        assert v.dtype == np.float32 and np.array_equal(v, x)
        assert info["kind"] == "ivf_flat" and info["nlist"] == nlist and info["ntotal"] == 700 and info["d"] == 48 and info["metric"] == 1
        assert np.array_equal(info["list_of"], assign) and np.array_equal(info["centroids"], cent)
        raw = open(p, "rb").read()
        assert raw[:4] == b"IwFl" and struct.unpack_from("<i", raw, 4)[0] == 48 and struct.unpack_from("<q", raw, 8)[0] == 700
        assert (b"sprs" in raw) == (sparse is True)
    # a flat file, a truncated file, an unsupported type
    flat = b"IxF2" + struct.pack("<iqqqBi", 48, 700, 1 << 20, 1 << 20, 1, 1) + struct.pack("<Q", 700 * 48) + x.tobytes()
    v, info = F.read_index_vectors(flat)
    assert np.array_equal(v, x) and info["kind"] == "flat"
    with pytest.raises(ValueError):
        F.read_index_vectors(raw[:len(raw) // 2])
    with pytest.raises(ValueError, match="IwPQ"):
        F.read_index_vectors(b"IwPQ" + raw[4:])


def test_device_f0_shortcut_honours_subclass_and_class_level_overrides():
    """the fused rmvpe + device post-processing path may only replace get_f0 / get_rmvpe / _rmvpe when they ARE the base
    implementations - an instance attribute, a subclass override or a class-level patch must send the clip through self.get_f0."""
    from comfy_rvc_amd.config import Config
    from comfy_rvc_amd import vc_infer_pipeline as V
    vc = V.VC(40000, Config())
    assert all(V._is_base_method(vc, n) for n in ("get_f0", "get_rmvpe", "_rmvpe"))

    class Sub(V.VC):
        def get_rmvpe(self, x, *a, **k):
            return np.zeros(4)
    sub = Sub(40000, Config())
    assert not V._is_base_method(sub, "get_rmvpe") and V._is_base_method(sub, "get_f0")

    class Sub2(V.VC):
        def get_f0(self, *a, **k):
            return None
    assert not V._is_base_method(Sub2(40000, Config()), "get_f0")

    vc.get_f0 = lambda *a, **k: None                     # instance attribute
    assert not V._is_base_method(vc, "get_f0")
    orig = V.VC.__dict__.get("get_f0")
    V.VC.get_f0 = lambda self, *a, **k: None             # class-level patch
    try:
        assert not V._is_base_method(V.VC(40000, Config()), "get_f0")
    finally:
        if orig is None:
            del V.VC.get_f0
        else:
            V.VC.get_f0 = orig
    assert V._is_base_method(V.VC(40000, Config()), "get_f0")


def test_oracle_ivf_probe_semantics():
    """oracle.pipeline.index_search_ivf restates faiss IndexIVFFlat.search (the reference's index type, custom_nodes/rvc_nodes.py:500-554, searched at
    vc_infer_pipeline.py:65): properties that pin the restatement - the answer lies in a probed cell, is the nearest vector of those cells, equals
    the exact search when every cell is probed, differs from it for a share of the queries at nprobe 1, and is (-1, FLT_MAX) for empty cells."""
    from oracle.pipeline import index_search, index_search_ivf
    rng = np.random.default_rng(0)
    big = rng.standard_normal((3000, 24)).astype(np.float32)
    cen = big[rng.choice(3000, 20, replace=False)]
    lo = (((big[:, None, :] - cen[None]) ** 2).sum(2)).argmin(1).astype(np.int32)
    q = rng.standard_normal((400, 24)).astype(np.float32)
    s0, i0 = index_search(q, big)
    for nprobe in (1, 4, 20):
        s, i = index_search_ivf(q, big, cen, lo, nprobe)
        dc = ((q[:, None, :].astype(np.float64) - cen[None].astype(np.float64)) ** 2).sum(2)
        cells = np.argsort(dc, axis=1, kind="stable")[:, :nprobe]
        assert all(lo[i[t, 0]] in cells[t] for t in range(q.shape[0]))
        for t in range(0, 400, 37):
            rows = np.nonzero(np.isin(lo, cells[t]))[0]
            d = ((big[rows].astype(np.float64) - q[t].astype(np.float64)) ** 2).sum(1)
            assert rows[d.argmin()] == i[t, 0] and abs(d.min() - s[t, 0]) < 1e-4
        assert np.all(s[:, 0] >= s0[:, 0] - 1e-5)
        if nprobe == 20:
            assert np.array_equal(i, i0)
        if nprobe == 1:
            assert 0.2 < float((i != i0).mean()) < 0.95
    cen2 = np.concatenate([cen, np.full((1, 24), 40.0, np.float32)])
    s, i = index_search_ivf(np.full((2, 24), 40.0, np.float32), big, cen2, lo, 1)
    assert list(i[:, 0]) == [-1, -1] and np.all(s == np.finfo(np.float32).max)


def test_rvc_node_result_cache_is_bounded_by_bytes():
    """RVCNode caches conversions by content hash (upstream caches preview files by name, reference custom_nodes/rvc_nodes.py:176-183); here the cache is an LRU
    bounded by BYTES, so a long-lived ComfyUI server cannot grow by 2.4 MB per distinct 30 s conversion for ever."""
    from comfy_rvc_amd.custom_nodes.rvc_nodes import RVCNode, _ByteLRU
    c = _ByteLRU(10_000)
    wav = lambda n: (np.zeros(n, dtype=np.int16), 40000)
    c.put("a", wav(2000)); c.put("b", wav(2000))                      # 4000 B each
    assert "a" in c and "b" in c and c.bytes == 8000
    c.get("a")                                                         # "a" is now the most recently used
    c.put("c", wav(2000))                                              # 12000 B > bound: the least recently used ("b") goes
    assert "a" in c and "c" in c and "b" not in c and c.bytes == 8000 and len(c) == 2
    c.put("a", wav(1000))                                              # replacing an entry accounts for its old size
    assert c.bytes == 6000
    c.put("huge", wav(6000))                                           # one entry larger than the bound is not kept, nothing else is evicted for it
    assert "huge" not in c and c.bytes == 6000
    assert isinstance(RVCNode._cache, _ByteLRU) and RVCNode._cache.max_bytes == RVCNode.CACHE_BYTES >= 64 << 20
