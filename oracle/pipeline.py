"""CPU oracle: host DSP and orchestration of the RVC inference path (TEST INFRASTRUCTURE ONLY).

Restates reference vc_infer_pipeline.py:25-196 (VC.vc, VC.pipeline), pitch_extraction.py:250-303
(get_f0 post-processing), lib/model_utils.py:39-57 (change_rms) and lib/audio.py:144-163,:274-304 on top
of oracle/nets.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
Noise is explicit: `noise_fn(shape) -> tensor` is called in the reference's draw order
(randn[1,192,T] then randn[1,T*upp,1] per segment).
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

from . import nets

MAX_INT16 = 32768
BH, AH = signal.butter(N=5, Wn=48, btype="high", fs=16000)   # reference vc_infer_pipeline.py:21

AUTOTUNE_NOTES = np.array([
    65.41, 69.30, 73.42, 77.78, 82.41, 87.31, 92.50, 98.00, 103.83, 110.00, 116.54, 123.47,
    130.81, 138.59, 146.83, 155.56, 164.81, 174.61, 185.00, 196.00, 207.65, 220.00, 233.08, 246.94,
    261.63, 277.18, 293.66, 311.13, 329.63, 349.23, 369.99, 392.00, 415.30, 440.00, 466.16, 493.88,
    523.25, 554.37, 587.33, 622.25, 659.25, 698.46, 739.99, 783.99, 830.61, 880.00, 932.33, 987.77,
    1046.50, 1108.73, 1174.66, 1244.51, 1318.51, 1396.91, 1479.98, 1567.98, 1661.22, 1760.00, 1864.66, 1975.53,
    2093.00, 2217.46, 2349.32, 2489.02, 2637.02, 2793.83, 2959.96, 3135.96, 3322.44, 3520.00, 3729.31, 3951.07])


class Constants:
    """FeatureExtractor.__init__ (reference pitch_extraction.py:14-31) with the CPU segmentation set (config.py:130-135)."""

    def __init__(self, tgt_sr, x_pad=1, x_query=6, x_center=38, x_max=41):
        self.x_pad, self.x_query, self.x_center, self.x_max = x_pad, x_query, x_center, x_max
        self.sr, self.window, self.f0_bins = 16000, 160, 256
        self.t_pad = self.sr * x_pad
        self.t_pad_tgt = tgt_sr * x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sr * x_query
        self.t_center = self.sr * x_center
        self.t_max = self.sr * x_max


def hz_to_mel(hz):
    return 2595 * np.log10(1 + hz / 700)


def remix_audio(audio, sr, target_sr=16000, max_volume=.95):
    """lib/audio.py:144-163 for the non-resampling branch: float32, mean over channels, peak-limit to 0.95."""
    audio = np.array(audio, dtype="float32")
    assert sr == target_sr, "resampling branch is parity-unpinned (librosa.resample absent)"
    if audio.ndim > 1:
        audio = np.nanmean(audio, axis=0)
    audio_max = np.abs(audio).max() / max_volume
    if audio_max > 1:
        audio = audio / audio_max
    return audio, target_sr


def autotune_f0(f0):
    """lib/audio.py:274-300 with threshold 0: snap every frame to the nearest note."""
    out = [AUTOTUNE_NOTES[np.argmin(np.abs(AUTOTUNE_NOTES - freq))] for freq in f0]
    return np.array(out, dtype="float32")


def f0_postprocess(f0, f0_up_key=0, f0_autotune=False, f0_min=50, f0_max=1600, f0_bins=256, inp_f0=None, x_pad=1):
    """get_f0 tail (reference pitch_extraction.py:276-303): autotune, transpose, the f0-file splice, coarse mel quantisation.
    inp_f0: float32 [m, 2] rows of (seconds, Hz) read from the user's f0 file (vc_infer_pipeline.py:146-151); the curve is resampled to 100 fps
    over its own time span and written over the extracted pitch from the first frame of the un-padded clip on (:281-291) - AFTER the
    transposition, so a spliced curve is not transposed."""
    f0 = np.array(f0, copy=True)
    if f0_autotune:
        f0 = autotune_f0(f0)
    f0 *= pow(2, f0_up_key / 12)
    if inp_f0 is not None:
        tf0 = 100
        delta_t = np.round((inp_f0[:, 0].max() - inp_f0[:, 0].min()) * tf0 + 1).astype("int16")
        replace_f0 = np.interp(list(range(delta_t)), inp_f0[:, 0] * 100, inp_f0[:, 1])
        n = f0[x_pad * tf0: x_pad * tf0 + len(replace_f0)].shape[0]
        f0[x_pad * tf0: x_pad * tf0 + len(replace_f0)] = replace_f0[:n]
    f0_mel_min, f0_mel_max = hz_to_mel(f0_min), hz_to_mel(f0_max)
    f0_mel = hz_to_mel(f0)
    f0_mel = (f0_mel - f0_mel_min) * (f0_bins - 2) / (f0_mel_max - f0_mel_min) + 1
    f0_mel = np.clip(f0_mel, a_min=1, a_max=f0_bins - 1)
    return np.rint(f0_mel).astype(np.int16), f0


def rms(y, frame_length, hop_length):
    """librosa.feature.rms (0.10.2): zero centre-pad, frame, sqrt(mean(x^2)) -> [1, n_frames]."""
    y = np.pad(np.asarray(y), int(frame_length // 2), mode="constant")
    n_frames = 1 + (y.shape[-1] - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.sqrt(np.mean(np.abs(y[idx]) ** 2, axis=-2, keepdims=True))


def change_rms(data1, sr1, data2, sr2, rate):
    """lib/model_utils.py:39-57."""
    rms1 = torch.from_numpy(rms(data1, sr1 // 2 * 2, sr1 // 2))
    rms2 = torch.from_numpy(rms(data2, sr2 // 2 * 2, sr2 // 2))
    rms1 = F.interpolate(rms1.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(rms2.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    data2 = data2 * (torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))).numpy()
    return data2


def segment_points(audio, c):
    """Cut points (reference vc_infer_pipeline.py:124-135): quietest sample of a 160-tap moving sum near every t_center."""
    audio_pad = np.pad(audio, (c.window // 2, c.window // 2), mode="reflect")
    opt_ts = []
    if audio_pad.shape[0] > c.t_max:
        audio_sum = np.zeros_like(audio)
        for i in range(c.window):
            audio_sum += audio_pad[i: i - c.window]
        for t in range(c.t_center, audio.shape[0], c.t_center):
            a = np.abs(audio_sum[t - c.t_query: t + c.t_query])
            opt_ts.append(t - c.t_query + np.where(a == a.min())[0][0])
    return opt_ts


def index_search(npy, big_npy):
    """Exact k = 1 nearest neighbour in squared L2, the quantity a faiss Flat/IVF-Flat L2 index reports (reference
    vc_infer_pipeline.py:65; faiss itself is absent offline, its IVF probe approximates this).  float64 distances so that the
    arg-min is the true one; returns (score float32 [T, 1], ix int64 [T, 1])."""
    a = np.asarray(npy, dtype=np.float64)
    b = np.asarray(big_npy, dtype=np.float64)
    d = (a * a).sum(1)[:, None] - 2.0 * (a @ b.T) + (b * b).sum(1)[None, :]
    ix = d.argmin(1)
    return d[np.arange(a.shape[0]), ix].astype(np.float32)[:, None], ix.astype(np.int64)[:, None]


def index_search_ivf(npy, big_npy, centroids, list_of, nprobe=1):
    """k = 1 search of a faiss `IndexIVFFlat` (L2), the index the reference builds and opens (custom_nodes/rvc_nodes.py:500-554:
    `index_factory(dim, "IVF{n},Flat")`, `nprobe = 1`; pitch_extraction.py:52-73; searched at vc_infer_pipeline.py:65).  faiss (third
    party, unpinned in requirements.txt, absent here) publishes the algorithm as: the coarse quantiser - an IndexFlatL2 over the
    `nlist` centroids - returns the `nprobe` centroids nearest to the query; only the inverted lists of those cells are scanned, the
    smallest squared L2 distance among THEIR vectors wins (IndexIVF::search -> search_preassigned -> IVFFlatScanner::scan_codes);
    when the probed lists are empty the label is -1 and the distance is float32 max (HeapArray init of a min-distance heap).
    `list_of[j]` = the list row j of big_npy was added to.  float64 distances, ties to the smaller index; returns
    (score float32 [T, 1], ix int64 [T, 1])."""
    a = np.asarray(npy, dtype=np.float64)
    b = np.asarray(big_npy, dtype=np.float64)
    cen = np.asarray(centroids, dtype=np.float64)
    lo = np.asarray(list_of).astype(np.int64)
    dc = (a * a).sum(1)[:, None] - 2.0 * (a @ cen.T) + (cen * cen).sum(1)[None, :]
    cells = np.argsort(dc, axis=1, kind="stable")[:, :int(nprobe)]
    d = (a * a).sum(1)[:, None] - 2.0 * (a @ b.T) + (b * b).sum(1)[None, :]
    ok = (lo[None, None, :] == cells[:, :, None]).any(1)
    d = np.where(ok, d, np.inf)
    ix = d.argmin(1)
    sc = d[np.arange(a.shape[0]), ix]
    none = ~ok.any(1)
    ix = np.where(none, -1, ix)
    sc = np.where(none, np.finfo(np.float32).max, sc)
    return sc.astype(np.float32)[:, None], ix.astype(np.int64)[:, None]


def vc_segment(hubert_sd, synth_sd, config, version, sid, audio0, pitch, pitchf, protect, noise_fn, c,
               n_hubert_layers=None, big_npy=None, index_rate=0.0, ivf=None):
    """VC.vc (reference vc_infer_pipeline.py:25-114); index retrieval over big_npy when given: the IVF probe of the reference's faiss index
    when `ivf = (centroids, list_of, nprobe)` describes one (index_search_ivf), an exact search otherwise."""
    feats = torch.from_numpy(audio0).float().view(1, -1)
    feats = nets.hubert_extract_features(hubert_sd, feats, version, n_layers=n_hubert_layers)
    feats0 = feats.clone()
    if big_npy is not None and index_rate > 0:
        npy = feats[0].numpy().astype("float32")
        score, ix = index_search(npy, big_npy) if ivf is None else index_search_ivf(npy, big_npy, *ivf)
        with np.errstate(divide="ignore", invalid="ignore"):
            weight = np.square(1 / score)
            weight /= weight.sum(axis=1, keepdims=True)
        npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
        feats = torch.from_numpy(npy.astype("float32")).unsqueeze(0) * index_rate + (1 - index_rate) * feats
    feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    p_len = min(audio0.shape[0] // c.window, feats.shape[1])
    upp = int(np.prod(config[12]))
    T = feats.shape[1]
    if pitch is None:                                       # no-f0 model (reference :84,:105-108): no protect blend, one noise draw
        o = nets.synth_infer(synth_sd, config, feats, None, None, sid, noise_fn((1, config[2], T)), None)
        return o[0, 0].float().numpy()
    pitch, pitchf = pitch[:, :p_len], pitchf[:, :p_len]
    if protect < 0.5:
        pitchff = pitchf.clone()
        pitchff[pitchf > 0] = 1
        pitchff[pitchf < 1] = protect
        pitchff = pitchff.unsqueeze(-1)
        feats = feats * pitchff + feats0 * (1 - pitchff)
    noise_z = noise_fn((1, config[2], T))
    noise_src = noise_fn((1, T * upp, 1))
    o = nets.synth_infer(synth_sd, config, feats, pitch, pitchf, sid, noise_z, noise_src)
    return o[0, 0].float().numpy()


def pipeline(hubert_sd, rmvpe_sd, synth_sd, config, version, audio, sid=0, f0_up_key=0, f0_method="rmvpe",
             rms_mix_rate=0.25, protect=0.33, f0_autotune=False, noise_fn=None, f0_override=None,
             x_pad=1, x_query=6, x_center=38, x_max=41, return_float=False, n_hubert_layers=None, big_npy=None, index_rate=0.0,
             if_f0=1, ivf=None, inp_f0=None):
    """VC.pipeline (reference vc_infer_pipeline.py:116-196) for resample_sr=0; inp_f0 = the parsed f0 file (see f0_postprocess); optional index retrieval (exact, or the IVF probe when `ivf` is given); if_f0=0 is the
    no-pitch model family (*_nono): no f0 front-end, pitch None all the way (:151-152,:172-179).

    audio: 16 kHz mono float32 (already remixed).  Returns int16 [N] at tgt_sr (and the float waveform
    before normalisation if return_float).  f0_override(audio_pad) -> f0 replaces the pitch front-end.
    """
    tgt_sr = config[-1]
    c = Constants(tgt_sr, x_pad, x_query, x_center, x_max)
    audio = signal.filtfilt(BH, AH, audio)
    opt_ts = segment_points(audio, c)
    audio_pad = np.pad(audio, (c.t_pad, c.t_pad), mode="reflect")
    pitch = pitchf = None
    if if_f0:
        if f0_override is not None:
            f0 = np.asarray(f0_override(audio_pad), dtype=np.float64)
        else:
            f0 = nets.rmvpe_infer_from_audio(rmvpe_sd, audio_pad, thred=0.03)
            if f0_method == "rmvpe+":
                f0 = np.clip(f0, a_min=50, a_max=1600)   # infer_from_audio_with_pitch (lib/rmvpe.py:649-659)
        pitch, pitchf = f0_postprocess(f0, f0_up_key, f0_autotune, inp_f0=inp_f0, x_pad=x_pad)
        p_len = min(pitch.shape[0], pitchf.shape[0])
        pitch = torch.from_numpy(pitch[:p_len].astype(np.int64)).unsqueeze(0)
        pitchf = torch.from_numpy(pitchf[:p_len].astype(np.float32)).unsqueeze(0)
    s, t, out = 0, None, []
    for t in opt_ts:
        t = t // c.window * c.window
        a = audio_pad[s: t + c.t_pad2 + c.window]
        ps = pitch[:, s // c.window: (t + c.t_pad2) // c.window + 1] if if_f0 else None
        pfs = pitchf[:, s // c.window: (t + c.t_pad2) // c.window + 1] if if_f0 else None
        out.append(vc_segment(hubert_sd, synth_sd, config, version, sid, a, ps, pfs, protect, noise_fn, c,
                              n_hubert_layers, big_npy, index_rate, ivf)[c.t_pad_tgt: -c.t_pad_tgt])
        s = t
    a = audio_pad[t:]
    ps = pitch[:, t // c.window:] if if_f0 and t is not None else pitch
    pfs = pitchf[:, t // c.window:] if if_f0 and t is not None else pitchf
    out.append(vc_segment(hubert_sd, synth_sd, config, version, sid, a, ps, pfs, protect, noise_fn, c,
                          n_hubert_layers, big_npy, index_rate, ivf)[c.t_pad_tgt: -c.t_pad_tgt])
    audio_opt = np.concatenate(out)
    if rms_mix_rate < 1:
        audio_opt = change_rms(audio, 16000, audio_opt, tgt_sr, rms_mix_rate)
    audio_f = audio_opt
    audio_max = np.abs(audio_opt).max() / 0.99
    audio_i16 = (audio_opt * MAX_INT16 / audio_max).astype(np.int16)
    return (audio_i16, audio_f) if return_float else audio_i16


def parse_f0_file(text):
    """The f0 file of `vc_single(f0_file=...)`: one "seconds,Hz" pair per line (reference vc_infer_pipeline.py:146-151)."""
    return np.array([list(map(float, line.split(","))) for line in text.strip("\n").split("\n")], dtype="float32")


def feature_input(hubert_sd, rmvpe_sd, x, version="v2"):
    """FeatureInput.go's three arrays for one 16 kHz clip (reference preprocessing_utils.py:129-193): HuBERT features float32 [T_h, 768 | 256]
    (compute_feats, all 12 layers run, layer 12 / final_proj(layer 9) taken), and compute_f0 = get_f0(x, 0, "rmvpe") with get_f0's OWN defaults
    f0_min 50 / f0_max 1100 (pitch_extraction.py:261-262; inference passes 1600) -> coarse int16 [n], f0 float64 [n].  The clip is NOT padded."""
    feats = nets.hubert_extract_features(hubert_sd, torch.from_numpy(np.asarray(x)).float().view(1, -1), version)
    feats = feats.squeeze(0).float().numpy()
    f0 = nets.rmvpe_infer_from_audio(rmvpe_sd, np.asarray(x), thred=0.03)
    coarse, nsf = f0_postprocess(f0, 0, False, f0_min=50, f0_max=1100)
    return coarse, nsf, feats
