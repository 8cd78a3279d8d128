"""CPU oracle for the CREPE f0 front-ends ("crepe", "mangio-crepe": reference pitch_extraction.py:76-150).  TEST INFRASTRUCTURE ONLY
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package).

PARITY UNPINNED.  The arithmetic lives in the third-party package `torchcrepe` (reference requirements.txt:24, unpinned; call sites
pitch_extraction.py:86,:101-111,:133-147), which is neither vendored in /root/reference nor installable here, and the reference holds no
test or golden vector for it.  What follows restates torchcrepe 0.0.23's published algorithm, function by function, with plain torch
ops so that every step reads like the original:
  core.preprocess      frames of 1024 every hop over audio zero-padded by 512, (x - mean) / max(1e-10, std) per frame (std unbiased)
  model.Crepe          6 x [zero-pad, Conv2d(k x 1), ReLU, BatchNorm2d(eps 0.0010000000474974513), MaxPool(2 x 1)], Linear -> sigmoid
  core.postprocess     bins below fmin / from ceil(fmax) on set to -inf, decoder, optional periodicity (probability at the decoded bin)
  decode.viterbi       softmax over bins, librosa.sequence.viterbi with the triangular (width 12) transition matrix, uniform prior
  convert              bins -> cents (20 * bin + 1997.3794084376191) + triangular dither in +-20 cents (scipy.stats.triang) -> Hz
  filter.median / mean NaN-aware window filters (reflect / zero padded, masked counts)
and the two call sites of the reference on top of them.
"""
import numpy as np
import scipy.stats
import torch
import torch.nn.functional as F

SAMPLE_RATE, WINDOW_SIZE, PITCH_BINS, CENTS_PER_BIN = 16000, 1024, 360, 20
CHANNELS = {"full": [1024, 128, 128, 128, 256, 512], "tiny": [128, 16, 16, 16, 32, 64]}


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def preprocess(audio, hop_length, pad=True):
    """torchcrepe.core.preprocess for 16 kHz input, one batch: audio [1, L] -> normalised frames [n, 1024]."""
    audio = _t(audio).float().reshape(1, -1)
    if pad:
        total = 1 + int(audio.size(1) // hop_length)
        audio = F.pad(audio, (WINDOW_SIZE // 2, WINDOW_SIZE // 2))
    else:
        total = 1 + int((audio.size(1) - WINDOW_SIZE) // hop_length)
    frames = F.unfold(audio[:, None, None, :], kernel_size=(1, WINDOW_SIZE), stride=(1, hop_length))
    frames = frames.transpose(1, 2).reshape(-1, WINDOW_SIZE)[:total].clone()
    frames -= frames.mean(dim=1, keepdim=True)
    frames /= torch.max(torch.tensor(1e-10), frames.std(dim=1, keepdim=True))
    return frames


def infer(sd, frames, model="full", taps=None):
    """torchcrepe.model.Crepe.forward: frames [n, 1024] -> probabilities [n, 360]."""
    sd = {k: _t(v).float() for k, v in sd.items()}
    x = frames[:, None, :, None]

    def layer(x, i, padding=(0, 0, 31, 32), stride=(1, 1)):
        x = F.pad(x, padding)
        x = F.conv2d(x, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], stride=stride)
        x = F.relu(x)
        if taps is not None and i == 1:
            taps["conv1"] = x[0, :, :, 0].numpy().copy()
        x = F.batch_norm(x, sd[f"conv{i}_BN.running_mean"], sd[f"conv{i}_BN.running_var"], sd[f"conv{i}_BN.weight"], sd[f"conv{i}_BN.bias"],
                         False, 0.0, 0.0010000000474974513)
        return F.max_pool2d(x, (2, 1), (2, 1))
    x = layer(x, 1, (0, 0, 254, 254), (4, 1))
    for i in range(2, 7):
        x = layer(x, i)
    x = x.permute(0, 2, 1, 3).reshape(x.shape[0], -1)
    if taps is not None:
        taps["embed"] = x.numpy().copy()
    return torch.sigmoid(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]))


def frequency_to_bins(frequency, quantize_fn=torch.floor):
    cents = 1200 * torch.log2(torch.tensor(float(frequency)) / 10.)
    return int(quantize_fn((cents - 1997.3794084376191) / CENTS_PER_BIN).int())


def transition_matrix():
    xx, yy = np.meshgrid(range(PITCH_BINS), range(PITCH_BINS))
    tr = np.maximum(12 - abs(xx - yy), 0)
    return tr / tr.sum(axis=1, keepdims=True)


def librosa_viterbi(prob, transition):
    """librosa.sequence.viterbi (0.10): prob [n_states, n_steps], uniform initial distribution, log domain, first maximum wins."""
    n_states, n_steps = prob.shape
    eps = np.finfo(prob.dtype).tiny
    log_trans = np.log(transition + np.finfo(transition.dtype).tiny)
    log_prob = np.log(prob.T + eps)
    log_init = np.log(np.full(n_states, 1.0 / n_states) + np.finfo(np.float64).tiny)
    value = np.zeros((n_steps, n_states), dtype=np.float64)
    ptr = np.zeros((n_steps, n_states), dtype=np.int64)
    value[0] = log_prob[0] + log_init
    for t in range(1, n_steps):
        trans_out = value[t - 1] + log_trans.T                      # [to, from]
        ptr[t] = np.argmax(trans_out, axis=1)
        value[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    state = np.zeros(n_steps, dtype=np.int64)
    state[-1] = np.argmax(value[-1])
    for t in range(n_steps - 2, -1, -1):
        state[t] = ptr[t + 1, state[t + 1]]
    return state


def bins_to_frequency(bins):
    """convert.bins_to_frequency with its dither (scipy's global numpy RNG, like torchcrepe): float32 arithmetic."""
    cents = CENTS_PER_BIN * bins.to(torch.float32) + 1997.3794084376191
    noise = scipy.stats.triang.rvs(c=0.5, loc=-CENTS_PER_BIN, scale=2 * CENTS_PER_BIN, size=cents.size())
    cents = cents + cents.new_tensor(noise)
    return 10 * 2 ** (cents / 1200)


def postprocess(probabilities, fmin, fmax, return_periodicity=False):
    """core.postprocess with decode.viterbi: probabilities [n, 360] of one clip -> pitch [1, n] (, periodicity [1, n])."""
    p = probabilities.t()[None].detach().clone()                      # [1, 360, n]
    minidx, maxidx = frequency_to_bins(fmin), frequency_to_bins(fmax, torch.ceil)
    p[:, :minidx] = -float("inf")
    p[:, maxidx:] = -float("inf")
    seq = torch.nn.functional.softmax(p, dim=1)
    bins = torch.from_numpy(np.array([librosa_viterbi(s.numpy(), transition_matrix()) for s in seq]))
    pitch = bins_to_frequency(bins)
    if not return_periodicity:
        return pitch
    stacked = p.transpose(1, 2).reshape(-1, PITCH_BINS)
    per = stacked.gather(1, bins.reshape(-1, 1).to(torch.int64)).reshape(p.size(0), p.size(2))
    return pitch, per


def filter_median(signals, win_length):
    signals = signals.unsqueeze(1)
    mask = ~torch.isnan(signals)
    masked_x = torch.where(mask, signals, torch.zeros_like(signals))
    padding = win_length // 2
    x = F.pad(masked_x, (padding, padding), mode="reflect")
    m = F.pad(mask.float(), (padding, padding), mode="constant", value=0)
    x = x.unfold(2, win_length, 1)
    m = m.unfold(2, win_length, 1)
    x = x.contiguous().view(x.size()[:3] + (-1,))
    m = m.contiguous().view(m.size()[:3] + (-1,))
    x_masked = torch.where(m.bool(), x.float(), torch.tensor(float("inf"))).to(x)
    x_sorted, _ = torch.sort(x_masked, dim=-1)
    valid = m.sum(dim=-1)
    idx = ((valid - 1) // 2).clamp(min=0)
    out = x_sorted.gather(-1, idx.unsqueeze(-1).long()).squeeze(-1)
    out[torch.isinf(out)] = float("nan")
    return out.squeeze(1)


def filter_mean(signals, win_length=9):
    signals = signals.unsqueeze(1)
    mask = ~torch.isnan(signals)
    masked_x = torch.where(mask, signals, torch.zeros_like(signals))
    ones = torch.ones(signals.size(1), 1, win_length)
    s = F.conv1d(masked_x, ones, stride=1, padding=win_length // 2)
    cnt = F.conv1d(mask.float(), ones, stride=1, padding=win_length // 2).clamp(min=1)
    avg = s / cnt
    avg[avg == 0] = float("nan")
    return avg.squeeze(1)


def predict(sd, audio, hop_length, fmin, fmax, model="full", return_periodicity=False, pad=True, taps=None):
    frames = preprocess(audio, hop_length, pad)
    with torch.no_grad():
        probs = infer(sd, frames, model, taps)
    if taps is not None:
        taps["probabilities"] = probs.numpy().copy()
    return postprocess(probs, fmin, fmax, return_periodicity)


def get_f0_official_crepe(sd, x, f0_min, f0_max, model="full", window=160):
    """FeatureExtractor.get_f0_official_crepe_computation (reference pitch_extraction.py:122-150)."""
    audio = torch.tensor(np.copy(x))[None].float()
    f0, pd = predict(sd, audio, window, f0_min, f0_max, model, return_periodicity=True)
    pd = filter_median(pd, 3)
    f0 = filter_mean(f0, 3)
    f0[pd < 0.1] = 0
    return f0[0].cpu().numpy()


def get_f0_mangio_crepe(sd, x, f0_min, f0_max, hop_length=160, model="full"):
    """FeatureExtractor.get_f0_crepe_computation (reference pitch_extraction.py:76-120)."""
    x = x.astype(np.float32)
    x /= np.quantile(np.abs(x), 0.999)
    audio = torch.from_numpy(x).clone().unsqueeze(0)
    pitch = predict(sd, audio, hop_length, f0_min, f0_max, model, pad=True)
    p_len = x.shape[0] // hop_length
    source = np.array(pitch.squeeze(0).cpu().float().numpy())
    source[source < 0.001] = np.nan
    target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
    return np.nan_to_num(target)
