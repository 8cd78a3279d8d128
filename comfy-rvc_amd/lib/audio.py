"""Host-side audio helpers on the inference path (mirror of reference lib/audio.py:14,:115-124,:144-163,:274-304)."""
import functools
import math
from collections.abc import Mapping

import numpy as np

MAX_INT16 = 32768

# Note table C2..B7 in Hz.  These are the exact constants of reference lib/audio.py:17-30 (hand-rounded equal temperament,
# e.g. E5 = 659.25 rather than 659.26), kept verbatim because autotune parity depends on every digit.
AUTOTUNE_NOTES = np.array([
    65.41, 69.30, 73.42, 77.78, 82.41, 87.31, 92.50, 98.00, 103.83, 110.00, 116.54, 123.47,
    130.81, 138.59, 146.83, 155.56, 164.81, 174.61, 185.00, 196.00, 207.65, 220.00, 233.08, 246.94,
    261.63, 277.18, 293.66, 311.13, 329.63, 349.23, 369.99, 392.00, 415.30, 440.00, 466.16, 493.88,
    523.25, 554.37, 587.33, 622.25, 659.25, 698.46, 739.99, 783.99, 830.61, 880.00, 932.33, 987.77,
    1046.50, 1108.73, 1174.66, 1244.51, 1318.51, 1396.91, 1479.98, 1567.98, 1661.22, 1760.00, 1864.66, 1975.53,
    2093.00, 2217.46, 2349.32, 2489.02, 2637.02, 2793.83, 2959.96, 3135.96, 3322.44, 3520.00, 3729.31, 3951.07])


def get_merge_func(merge_type):
    """reference lib/utils.py:104-108"""
    return {"min": np.nanmin, "max": np.nanmax, "median": np.nanmedian}.get(merge_type, np.nanmean)


def hz_to_mel(hz):
    return 2595 * np.log10(1 + hz / 700)


def get_audio(audio):
    """ComfyUI AUDIO dict ({'waveform': [1, N, C], 'sample_rate'}) or a VHS_AUDIO thunk -> (ndarray [C, N], sr)."""
    if callable(audio):
        audio = audio()
    if isinstance(audio, Mapping):
        return audio["waveform"].squeeze(0).transpose(0, 1).numpy(), audio["sample_rate"]
    if isinstance(audio, (bytes, bytearray, memoryview)):
        return bytes_to_audio(bytes(audio))
    return audio


# ---------------------------------------------------------------------------------------------- VHS_AUDIO byte streams
# The reference moves audio between nodes as encoded bytes (VHS_AUDIO = thunk returning bytes): audio_to_bytes writes a WAV
# through soundfile (lib/audio.py:188-204: PCM_16 when the samples exceed 1 in magnitude, i.e. int16-valued, IEEE float32
# otherwise) and bytes_to_audio reads any libsndfile format back as float64 [C, N] (lib/audio.py:206-210).  soundfile is not
# available to this build, so the RIFF/WAVE container is written and parsed here (PCM 8/16/24/32, IEEE float 32/64, plain and
# WAVE_FORMAT_EXTENSIBLE headers); compressed containers (flac / mp3 / ogg) need an external codec and raise.
def audio_to_bytes(audio, sr, target_sr=None, to_int16=False, to_stereo=False, format="WAV"):
    """(samples [N] | [C, N] | [N, C], sr) -> WAV bytes, laid out as the reference's audio_to_bytes produces them."""
    import struct
    if str(format).upper() != "WAV":
        raise NotImplementedError(f"audio_to_bytes: only the WAV container is built in (asked for {format!r}; flac/mp3 need libsndfile/ffmpeg)")
    audio = np.array(audio, dtype="float32")
    if to_int16:
        audio_max = np.abs(audio).max() / .99
        if audio_max > 1:
            audio = audio / audio_max
        audio = np.clip(audio * MAX_INT16, a_min=-MAX_INT16 + 1, a_max=MAX_INT16 - 1)
    if to_stereo and audio.ndim < 2:
        audio = np.stack([audio, audio], axis=-1)
    if audio.ndim > 1 and audio.shape[0] < audio.shape[1]:
        audio = audio.T                                             # frames x channels
    rate = int(sr if target_sr is None else target_sr)
    as_int = audio.size > 0 and np.abs(audio).max() > 1
    data = np.ascontiguousarray(audio.astype("<i2" if as_int else "<f4"))
    nch = 1 if data.ndim == 1 else data.shape[1]
    bps = data.dtype.itemsize
    payload = data.tobytes()
    fmt = struct.pack("<HHIIHH", 1 if as_int else 3, nch, rate, rate * nch * bps, nch * bps, 8 * bps)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if not as_int:                                                  # non-PCM formats carry a fact chunk (frames per channel)
        chunks += b"fact" + struct.pack("<II", 4, data.shape[0])
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def bytes_to_audio(data, **kwargs):
    """WAV bytes -> (float64 [N] or [C, N] scaled to [-1, 1), sr), as soundfile.read returns them (reference lib/audio.py:206-210)."""
    import struct
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        kind = {b"fLaC": "flac", b"OggS": "ogg", b"ID3": "mp3"}.get(bytes(data[:4]), {b"ID3": "mp3"}.get(bytes(data[:3]), "unknown"))
        raise NotImplementedError(f"bytes_to_audio: only RIFF/WAVE streams are built in (got a {kind} stream; flac/mp3 need libsndfile/ffmpeg)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise ValueError("bytes_to_audio: malformed WAV stream (fmt / data chunk missing)")
    tag, nch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                            # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    nframes = len(payload) // max(align, 1)
    raw = payload[:nframes * align]
    if tag == 3 and bits in (32, 64):
        x = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif tag == 1 and bits in (16, 32):
        x = np.frombuffer(raw, dtype="<i2" if bits == 16 else "<i4").astype(np.float64) / float(1 << (bits - 1))
    elif tag == 1 and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float64) / float(1 << 23)
    else:
        raise NotImplementedError(f"bytes_to_audio: WAV format tag {tag} with {bits} bits is not supported")
    if nch > 1:
        x = x.reshape(-1, nch)
        if x.shape[1] < x.shape[0]:
            x = x.T                                                 # channels x frames
    return x, int(rate)


# ---------------------------------------------------------------------------------------------- resampling
# The reference resamples with librosa.resample (default res_type "soxr_hq": linear phase, pass band up to 0.913 of the lower
# Nyquist frequency, stop band from Nyquist, 20-bit precision).  librosa / soxr are not available to this build, so that exact
# arithmetic cannot be reproduced ("parity unpinned"); what is restated is its specification: a linear-phase
# Kaiser-windowed sinc with the same band edges and ~122 dB stop band, evaluated exactly (float64 taps, float64 accumulation) by the
# polyphase kernel rvc_resample.  Output length = ceil(n * target / orig), as librosa computes it.
RESAMPLE_PASSBAND, RESAMPLE_ATTEN_DB = 0.913, 122.0


@functools.lru_cache(maxsize=16)
def design_resample_filter(orig_sr, target_sr):
    """(taps float64 [2*half+1], half, up, down): low-pass on the `up`-times up-sampled grid, DC gain `up`."""
    g = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // g, int(orig_sr) // g
    fs_up = float(orig_sr) * up
    fn = min(orig_sr, target_sr) / 2.0
    width = (1.0 - RESAMPLE_PASSBAND) * fn / fs_up                 # transition band, cycles / sample on the up-sampled grid
    fc = 0.5 * (1.0 + RESAMPLE_PASSBAND) * fn / fs_up              # -6 dB point in the middle of it
    beta = 0.1102 * (RESAMPLE_ATTEN_DB - 8.7)
    half = int(math.ceil((RESAMPLE_ATTEN_DB - 7.95) / (14.36 * width) / 2.0))
    k = np.arange(-half, half + 1, dtype=np.float64)
    taps = 2.0 * fc * np.sinc(2.0 * fc * k) * np.kaiser(2 * half + 1, beta)
    taps *= up / taps.sum()
    return taps, half, up, down


def resample_audio(audio, orig_sr, target_sr, device="cuda:0"):
    """float32 [..., N] at orig_sr -> float32 [..., ceil(N * target / orig)] at target_sr on the GPU (rvc_resample)."""
    import torch
    from .. import _lib
    audio = np.asarray(audio, dtype=np.float32)
    if int(orig_sr) == int(target_sr):
        return audio
    taps, half, up, down = design_resample_filter(int(orig_sr), int(target_sr))
    n_in = audio.shape[-1]
    n_out = int(math.ceil(n_in * float(target_sr) / float(orig_sr)))
    flat = np.ascontiguousarray(audio.reshape(-1, n_in))
    dev = torch.device(device)
    with torch.cuda.device(dev):
        h = torch.from_numpy(taps).to(dev)
        x = torch.from_numpy(flat).to(dev)
        y = torch.empty(flat.shape[0], n_out, dtype=torch.float32, device=dev)
        for c in range(flat.shape[0]):
            _lib.check(_lib.lib.rvc_resample(_lib.current_stream(), _lib.ptr(x[c]), n_in, _lib.ptr(h), half, up, down, _lib.ptr(y[c]), n_out))
        out = y.cpu().numpy()
    return out.reshape(audio.shape[:-1] + (n_out,))


resample = resample_audio      # (remix_audio has a boolean parameter of that name)


def remix_audio(input_audio, target_sr=None, norm=False, to_int16=False, resample=False, axis=0, merge_type=None, max_volume=.95, **kwargs):
    """float32 mono at target_sr, peak-limited to max_volume (reference lib/audio.py:144-163)."""
    audio = np.array(input_audio[0], dtype="float32")
    if target_sr is None:
        target_sr = input_audio[1]
    if resample or input_audio[1] != target_sr:
        audio = resample_audio(audio, input_audio[1], target_sr, **{k: v for k, v in kwargs.items() if k == "device"})
    if audio.ndim > 1:
        audio = get_merge_func(merge_type)(audio, axis=axis)
    if norm:
        peak = np.abs(audio).max(axis=axis, keepdims=True)
        audio = audio / np.where(peak < np.finfo(np.float32).tiny, 1.0, peak)
    audio_max = np.abs(audio).max() / max_volume
    if audio_max > 1:
        audio = audio / audio_max
    if to_int16:
        audio = np.clip(audio * MAX_INT16, a_min=1 - MAX_INT16, a_max=MAX_INT16 - 1).astype("int16")
    return audio, target_sr


def autotune_f0(f0, threshold=0.):
    """Snap each frame to the nearest note unless it is closer than `threshold` (reference lib/audio.py:274-300)."""
    f0 = np.asarray(f0)
    diff = np.abs(AUTOTUNE_NOTES[None, :] - f0[:, None])
    idx = np.argmin(diff, axis=1)
    near = diff[np.arange(f0.shape[0]), idx] < threshold
    return np.where(near, f0, AUTOTUNE_NOTES[idx]).astype("float32")


def pad_audio(*audios, axis=0):
    """Right-pad to the longest and stack (reference lib/audio.py:257-262)."""
    arrs = [a for a in audios if a is not None]
    maxlen = max((len(a) for a in arrs), default=0)
    return np.stack([np.pad(a, (0, maxlen - len(a))) for a in arrs], axis=axis)
