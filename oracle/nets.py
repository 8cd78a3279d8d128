"""CPU oracle: straight-line restatement of the three networks on the RVC inference path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product (comfy-rvc_amd/) never does.  Written with plain torch-CPU fp32
functional ops (F.conv1d, matmul, ...) so it runs on the GPU box's host cores without the reference.

Pinned against the reference: `oracle/gen_golden.py` imports the real reference in the build
container (oracle/ref_shim.py), runs it with the procedural weights of comfy-rvc_amd/synthetic.py and
explicit noise, and stores per-stage taps in tests/golden/*.npz; tests/test_oracle_golden.py checks
every function below against those taps.  HuBERT's arithmetic lives in third-party `transformers`
(unpinned by the reference; 5.15.0 here) - restated from
transformers/models/hubert/modeling_hubert.py:45-477,878-955.

All state dicts use the reference's own key names (weight_g / weight_v etc.); folding happens here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # reference lib/infer_pack/modules.py:13


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def tensors(sd):
    return {k: _t(v) for k, v in sd.items()}


def weight_norm_fold(v, g, dim=0):
    """torch.nn.utils.weight_norm: w = v * g / ||v|| with the norm over all dims but `dim`."""
    return torch._weight_norm(v, g, dim)


# =====================================================================================  HuBERT
def hubert_feature_encoder(sd, audio):
    """HubertFeatureEncoder (modeling_hubert.py:154-213): conv0+GroupNorm(512,512)+GELU, 6x conv+GELU. -> [1,512,T_h]"""
    x = audio[:, None]
    strides = (5, 2, 2, 2, 2, 2, 2)
    for i, s in enumerate(strides):
        x = F.conv1d(x, sd[f"feature_extractor.conv_layers.{i}.conv.weight"], None, stride=s)
        if i == 0:
            x = F.group_norm(x, 512, sd["feature_extractor.conv_layers.0.layer_norm.weight"],
                             sd["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
        x = F.gelu(x)
    return x


def hubert_pos_conv(sd, h):
    """HubertPositionalConvEmbedding (modeling_hubert.py:43-92): grouped k128 conv, weight_norm(dim=2), drop last, GELU. h:[1,T,768]"""
    w = weight_norm_fold(sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"],
                         sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"], 2)
    y = F.conv1d(h.transpose(1, 2), w, sd["encoder.pos_conv_embed.conv.bias"], padding=64, groups=16)
    y = F.gelu(y[:, :, :-1])
    return y.transpose(1, 2)


def hubert_layer(sd, l, h):
    """HubertEncoderLayer (post-LN, modeling_hubert.py:371-404) with sdpa attention (12 heads x 64)."""
    p = f"encoder.layers.{l}."
    B, T, _ = h.shape
    q = F.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]).view(B, T, 12, 64).transpose(1, 2)
    k = F.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"]).view(B, T, 12, 64).transpose(1, 2)
    v = F.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"]).view(B, T, 12, 64).transpose(1, 2)
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, scale=64 ** -0.5)
    a = a.transpose(1, 2).reshape(B, T, 768)
    a = F.linear(a, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
    h = F.layer_norm(h + a, (768,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    f = F.gelu(F.linear(h, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
    f = F.linear(f, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
    return F.layer_norm(h + f, (768,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)


def hubert_extract_features(sd, audio, version="v2", taps=None, n_layers=None):
    """HubertModelWithFinalProj.extract_features (reference lib/infer_pack/loaders.py:55-61).

    audio: float32 [1, L].  Returns [1, T_h, 768] (v2: hidden_states[11]) or [1, T_h, 256]
    (v1: final_proj(hidden_states[8])).  Only the layers that feed the tapped state are run
    (the reference runs all 12 and discards the rest); `n_layers` forces a count for timing.
    """
    sd = tensors(sd)
    audio = _t(audio).float()
    with torch.no_grad():
        x = hubert_feature_encoder(sd, audio)
        if taps is not None:
            taps["conv_stack"] = x
        h = x.transpose(1, 2)
        h = F.layer_norm(h, (512,), sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], 1e-5)
        h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
        pos = hubert_pos_conv(sd, h)
        if taps is not None:
            taps["pos_conv"] = pos
        h = F.layer_norm(h + pos, (768,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
        need = 8 if version == "v1" else 11
        if n_layers is not None:
            need = n_layers
        for l in range(need):
            if taps is not None and l in (0, 8):
                taps[f"hidden_{l}"] = h
            h = hubert_layer(sd, l, h)
        if version == "v1":
            h = F.linear(h, sd["final_proj.weight"], sd["final_proj.bias"])
    return h


# =====================================================================================  RMVPE
def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)   # scipy.signal.get_window("hann", n, fftbins=True)


def stft_forward_basis(n_fft=1024):
    """STFT.__init__ (reference lib/rmvpe.py:88-109): [real; imag] rows of fft(eye) x hann -> [1026,1,1024] f32."""
    fb = np.fft.fft(np.eye(n_fft))
    cutoff = n_fft // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])])
    basis = torch.FloatTensor(fb[:, None, :])
    basis *= torch.from_numpy(hann_periodic(n_fft)).float()
    return basis


def mel_filterbank(sr=16000, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0):
    """librosa.filters.mel(htk=True, norm='slaney') as used by MelSpectrogram (reference lib/rmvpe.py:492-499)."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)

    w = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


_MEL_CACHE = {}


def rmvpe_mel(audio):
    """MelSpectrogram.forward(center=True) (reference lib/rmvpe.py:510-556 + STFT.transform :114-150). audio [1,L] -> [1,128,n]."""
    if "b" not in _MEL_CACHE:
        _MEL_CACHE["b"] = (stft_forward_basis(1024), torch.from_numpy(mel_filterbank()).float())
    basis, melb = _MEL_CACHE["b"]
    audio = _t(audio).float()
    with torch.no_grad():
        x = F.pad(audio.view(audio.shape[0], 1, 1, -1), (512, 512, 0, 0, 0, 0), mode="reflect").squeeze(1)
        ft = F.conv1d(x, basis, stride=160, padding=0)
        mag = torch.sqrt(ft[:, :513, :] ** 2 + ft[:, 513:, :] ** 2)
        mel = torch.matmul(melb, mag)
        return torch.log(torch.clamp(mel, min=1e-5))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.01, 1e-5)


def _conv_block_res(sd, p, x):
    """ConvBlockRes (reference lib/rmvpe.py:233-268)."""
    y = F.conv2d(x, sd[p + "conv.0.weight"], None, padding=1)
    y = F.relu(_bn(sd, p + "conv.1.", y))
    y = F.conv2d(y, sd[p + "conv.3.weight"], None, padding=1)
    y = F.relu(_bn(sd, p + "conv.4.", y))
    if (p + "shortcut.weight") in sd:
        return y + F.conv2d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
    return y + x


def gru_scan(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of nn.GRU (gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn))). x:[T,I] -> [T,H]."""
    T = x.shape[0]
    H = w_hh.shape[1]
    gi = F.linear(x, w_ih, b_ih)
    h = torch.zeros(H)
    out = torch.empty(T, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    whh_t = w_hh.t().contiguous()
    for t in order:
        gh = h @ whh_t + b_hh
        r = torch.sigmoid(gi[t, :H] + gh[:H])
        z = torch.sigmoid(gi[t, H:2 * H] + gh[H:2 * H])
        n = torch.tanh(gi[t, 2 * H:] + r * gh[2 * H:])
        h = (1 - z) * n + z * h
        out[t] = h
    return out


def rmvpe_e2e(sd, mel, taps=None):
    """E2E.forward (reference lib/rmvpe.py:464-470) on mel [1,128,T_r] (T_r multiple of 32) -> salience [1,T_r,360]."""
    sd = tensors(sd)
    mel = _t(mel).float()
    with torch.no_grad():
        x = mel.transpose(-1, -2).unsqueeze(1)
        x = _bn(sd, "unet.encoder.bn.", x)
        skips = []
        for i in range(5):
            for b in range(4):
                x = _conv_block_res(sd, f"unet.encoder.layers.{i}.conv.{b}.", x)
            skips.append(x)
            x = F.avg_pool2d(x, (2, 2))
            if taps is not None and i == 0:
                taps["enc0_pool"] = x
        for i in range(4):
            for b in range(4):
                x = _conv_block_res(sd, f"unet.intermediate.layers.{i}.conv.{b}.", x)
        if taps is not None:
            taps["intermediate"] = x
        for i in range(5):
            p = f"unet.decoder.layers.{i}."
            x = F.conv_transpose2d(x, sd[p + "conv1.0.weight"], None, stride=(2, 2), padding=(1, 1), output_padding=(1, 1))
            x = F.relu(_bn(sd, p + "conv1.1.", x))
            x = torch.cat((x, skips[-1 - i]), dim=1)
            for b in range(4):
                x = _conv_block_res(sd, p + f"conv2.{b}.", x)
        if taps is not None:
            taps["unet_out"] = x
        x = F.conv2d(x, sd["cnn.weight"], sd["cnn.bias"], padding=1)
        x = x.transpose(1, 2).flatten(-2)[0]            # [T, 384], feature = c*128 + mel
        fwd = gru_scan(x, sd["fc.0.gru.weight_ih_l0"], sd["fc.0.gru.weight_hh_l0"], sd["fc.0.gru.bias_ih_l0"], sd["fc.0.gru.bias_hh_l0"])
        bwd = gru_scan(x, sd["fc.0.gru.weight_ih_l0_reverse"], sd["fc.0.gru.weight_hh_l0_reverse"],
                       sd["fc.0.gru.bias_ih_l0_reverse"], sd["fc.0.gru.bias_hh_l0_reverse"], reverse=True)
        g = torch.cat([fwd, bwd], dim=1)
        if taps is not None:
            taps["gru"] = g
        return torch.sigmoid(F.linear(g, sd["fc.1.weight"], sd["fc.1.bias"])).unsqueeze(0)


def rmvpe_mel2hidden(sd, mel, taps=None):
    """RMVPE.mel2hidden (reference lib/rmvpe.py:590-605): right reflect-pad frames to a multiple of 32, E2E, crop."""
    mel = _t(mel).float()
    n = mel.shape[-1]
    pad = min(32 * ((n - 1) // 32 + 1) - n, n)
    melp = F.pad(mel, (0, pad), mode="reflect")
    return rmvpe_e2e(sd, melp, taps)[:, :n]


CENTS_MAPPING = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))


def rmvpe_decode(salience, thred=0.03):
    """RMVPE.decode + to_local_average_cents (reference lib/rmvpe.py:607-612,:661-685). salience [n,360] f32 -> f0 [n] f64."""
    salience = np.asarray(salience)
    center = np.argmax(salience, axis=1) + 4
    sal = np.pad(salience, ((0, 0), (4, 4)))
    idx = center[:, None] + np.arange(-4, 5)[None, :]
    todo_s = np.take_along_axis(sal, idx, axis=1)
    todo_c = CENTS_MAPPING[idx]
    cents = np.sum(todo_s * todo_c, 1) / np.sum(todo_s, 1)
    cents[np.max(sal, axis=1) <= thred] = 0
    f0 = 10 * (2 ** (cents / 1200))
    f0[f0 == 10] = 0
    return f0


def rmvpe_infer_from_audio(sd, audio, thred=0.03, taps=None):
    """RMVPE.infer_from_audio (reference lib/rmvpe.py:614-623): audio float [L] -> f0 [L/160+1] (float64)."""
    a = torch.from_numpy(np.asarray(audio)).float().unsqueeze(0)
    mel = rmvpe_mel(a)
    if taps is not None:
        taps["mel"] = mel
    hidden = rmvpe_mel2hidden(sd, mel, taps).squeeze(0).numpy()
    if taps is not None:
        taps["salience"] = hidden
    return rmvpe_decode(hidden, thred)


# =====================================================================================  synthesizer
def _layer_norm_c(x, gamma, beta):
    """modules.LayerNorm: layer_norm over the channel axis of [B,C,T] (reference lib/infer_pack/modules.py:25-28)."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, 1e-5).transpose(1, -1)


def _rel_embeddings(emb, length, window=10):
    """MultiHeadAttention._get_relative_embeddings (reference attentions.py:291-307)."""
    pad_length = max(length - (window + 1), 0)
    start = max((window + 1) - length, 0)
    if pad_length > 0:
        emb = F.pad(emb, (0, 0, pad_length, pad_length, 0, 0))
    return emb[:, start:start + 2 * length - 1]


def _rel_to_abs(x):
    """_relative_position_to_absolute_position (reference attentions.py:309-328): [b,h,l,2l-1] -> [b,h,l,l]."""
    b, h, l, _ = x.size()
    x = F.pad(x, (0, 1))
    x = F.pad(x.view(b, h, l * 2 * l), (0, l - 1))
    return x.view(b, h, l + 1, 2 * l - 1)[:, :, :l, l - 1:]


def _abs_to_rel(x):
    """_absolute_position_to_relative_position (reference attentions.py:330-344): [b,h,l,l] -> [b,h,l,2l-1]."""
    b, h, l, _ = x.size()
    x = F.pad(x, (0, l - 1))
    x = F.pad(x.view(b, h, l * l + l * (l - 1)), (l, 0))
    return x.view(b, h, l, 2 * l)[:, :, :, 1:]


def enc_p_attention(sd, p, x, n_heads):
    """MultiHeadAttention.forward with window_size=10, heads_share (reference attentions.py:212-271). x:[1,C,T], full mask."""
    q = F.conv1d(x, sd[p + "conv_q.weight"], sd[p + "conv_q.bias"])
    k = F.conv1d(x, sd[p + "conv_k.weight"], sd[p + "conv_k.bias"])
    v = F.conv1d(x, sd[p + "conv_v.weight"], sd[p + "conv_v.bias"])
    b, d, t = q.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    qs = q / math.sqrt(kc)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    rel_k = _rel_embeddings(sd[p + "emb_rel_k"], t)
    scores = scores + _rel_to_abs(torch.matmul(qs, rel_k.unsqueeze(0).transpose(-2, -1)))
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    rel_v = _rel_embeddings(sd[p + "emb_rel_v"], t)
    out = out + torch.matmul(_abs_to_rel(p_attn), rel_v.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(out, sd[p + "conv_o.weight"], sd[p + "conv_o.bias"])


def enc_p_forward(sd, config, phone, pitch, taps=None):
    """TextEncoder{256,768}.forward with lengths == T (reference lib/infer_pack/models.py:43-58,:90-105;
    attentions.Encoder.forward :57-69; FFN.forward :387-395).  phone [1,T,D], pitch int64 [1,T] -> m_p, logs_p [1,192,T]."""
    hidden, n_heads, n_layers, ksz = config[3], config[5], config[6], config[7]
    x = F.linear(phone, sd["enc_p.emb_phone.weight"], sd["enc_p.emb_phone.bias"])
    if pitch is not None:                                   # TextEncoder*(f0=False) of the *_nono synthesizers has no emb_pitch (models.py:47-50)
        x = x + F.embedding(pitch, sd["enc_p.emb_pitch.weight"])
    x = x * math.sqrt(hidden)
    x = F.leaky_relu(x, 0.1)
    x = torch.transpose(x, 1, -1)
    if taps is not None:
        taps["enc_p_in"] = x
    pl, pr = (ksz - 1) // 2, ksz // 2
    for l in range(n_layers):
        y = enc_p_attention(sd, f"enc_p.encoder.attn_layers.{l}.", x, n_heads)
        x = _layer_norm_c(x + y, sd[f"enc_p.encoder.norm_layers_1.{l}.gamma"], sd[f"enc_p.encoder.norm_layers_1.{l}.beta"])
        p = f"enc_p.encoder.ffn_layers.{l}."
        y = F.conv1d(F.pad(x, (pl, pr)), sd[p + "conv_1.weight"], sd[p + "conv_1.bias"])
        y = torch.relu(y)
        y = F.conv1d(F.pad(y, (pl, pr)), sd[p + "conv_2.weight"], sd[p + "conv_2.bias"])
        x = _layer_norm_c(x + y, sd[f"enc_p.encoder.norm_layers_2.{l}.gamma"], sd[f"enc_p.encoder.norm_layers_2.{l}.beta"])
        if taps is not None and l == 0:
            taps["enc_p_layer0"] = x
    stats = F.conv1d(x, sd["enc_p.proj.weight"], sd["enc_p.proj.bias"])
    inter = config[2]
    return stats[:, :inter], stats[:, inter:]


def _wn(sd, p, x, g, hidden, n_layers=3, ksz=5):
    """modules.WN.forward (reference lib/infer_pack/modules.py:184-209) with dilation_rate 1, full mask."""
    output = torch.zeros_like(x)
    gc = F.conv1d(g, weight_norm_fold(sd[p + "cond_layer.weight_v"], sd[p + "cond_layer.weight_g"]), sd[p + "cond_layer.bias"])
    for i in range(n_layers):
        w = weight_norm_fold(sd[p + f"in_layers.{i}.weight_v"], sd[p + f"in_layers.{i}.weight_g"])
        x_in = F.conv1d(x, w, sd[p + f"in_layers.{i}.bias"], padding=(ksz - 1) // 2)
        in_act = x_in + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        w = weight_norm_fold(sd[p + f"res_skip_layers.{i}.weight_v"], sd[p + f"res_skip_layers.{i}.weight_g"])
        rs = F.conv1d(acts, w, sd[p + f"res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = x + rs[:, :hidden]
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output


def flow_reverse(sd, config, z_p, g, taps=None):
    """ResidualCouplingBlock.forward(reverse=True) (reference models.py:185-192; ResidualCouplingLayer modules.py:436-455; Flip :373-380)."""
    hidden, half = config[3], config[2] // 2
    x = z_p
    for f in (3, 2, 1, 0):
        x = torch.flip(x, [1])
        p = f"flow.flows.{2 * f}."
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, sd[p + "pre.weight"], sd[p + "pre.bias"])
        h = _wn(sd, p + "enc.", h, g, hidden)
        m = F.conv1d(h, sd[p + "post.weight"], sd[p + "post.bias"])
        x = torch.cat([x0, x1 - m], 1)
        if taps is not None and f == 3:
            taps["flow_first"] = x
    return x


def sine_source(sd, f0, upp, sr, noise_src, taps=None):
    """SourceModuleHnNSF.forward + SineGen.forward (reference models.py:361-411,:455-467), harmonic_num=0.

    f0 [1,T] f32 (Hz, 0 = unvoiced); noise_src [1,T*upp,1] = the torch.randn_like draw.  Returns [1,T*upp,1].
    torch.cumsum on CPU keeps an fp64 running sum for fp32 input, which this inherits.
    """
    f0 = f0[:, None].transpose(1, 2)                       # [1,T,1]
    rad_values = (f0 / sr) % 1
    tmp_over_one = torch.cumsum(rad_values, 1)
    tmp_over_one *= upp
    tmp_over_one = F.interpolate(tmp_over_one.transpose(2, 1), scale_factor=float(upp), mode="linear", align_corners=True).transpose(2, 1)
    rad_values = F.interpolate(rad_values.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    tmp_over_one %= 1
    idx = (tmp_over_one[:, 1:, :] - tmp_over_one[:, :-1, :]) < 0
    cumsum_shift = torch.zeros_like(rad_values)
    cumsum_shift[:, 1:, :] = idx * -1.0
    sine = torch.sin(torch.cumsum(rad_values + cumsum_shift, dim=1) * 2 * np.pi) * 0.1
    uv = (f0 > 0).float()
    uv = F.interpolate(uv.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    noise_amp = uv * 0.003 + (1 - uv) * 0.1 / 3
    sine = sine * uv + noise_amp * noise_src
    if taps is not None:
        taps["sine_waves"] = sine
    return torch.tanh(F.linear(sine, sd["dec.m_source.l_linear.weight"], sd["dec.m_source.l_linear.bias"]))


def generator_forward(sd, config, z, f0, g, noise_src, taps=None):
    """GeneratorNSF.forward (reference models.py:542-564) + ResBlock1.forward (modules.py:295-308); f0 None = the plain
    Generator of the *_nono synthesizers (models.py:292-311): no harmonic source, no noise convs."""
    rb_k, rb_d, up_rates, up_init, up_k, sr = config[10], config[11], config[12], config[13], config[14], config[17]
    upp = int(np.prod(up_rates))
    har = None
    if f0 is not None:
        har = sine_source(sd, f0, upp, sr, noise_src, taps).transpose(1, 2)
        if taps is not None:
            taps["har_source"] = har
    x = F.conv1d(z, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, sd["dec.cond.weight"], sd["dec.cond.bias"])
    nk = len(rb_k)
    for i, (u, k) in enumerate(zip(up_rates, up_k)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        w = weight_norm_fold(sd[f"dec.ups.{i}.weight_v"], sd[f"dec.ups.{i}.weight_g"])
        x = F.conv_transpose1d(x, w, sd[f"dec.ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if har is not None:
            if i + 1 < len(up_rates):
                sf0 = int(np.prod(up_rates[i + 1:]))
                xs_ = F.conv1d(har, sd[f"dec.noise_convs.{i}.weight"], sd[f"dec.noise_convs.{i}.bias"], stride=sf0, padding=sf0 // 2)
            else:
                xs_ = F.conv1d(har, sd[f"dec.noise_convs.{i}.weight"], sd[f"dec.noise_convs.{i}.bias"])
            x = x + xs_
        if taps is not None:
            taps[f"gen_ups{i}"] = x
        xs = None
        for j in range(nk):
            p = f"dec.resblocks.{i * nk + j}."
            y = x
            for m in range(3):
                d = rb_d[j][m]
                w1 = weight_norm_fold(sd[p + f"convs1.{m}.weight_v"], sd[p + f"convs1.{m}.weight_g"])
                w2 = weight_norm_fold(sd[p + f"convs2.{m}.weight_v"], sd[p + f"convs2.{m}.weight_g"])
                xt = F.leaky_relu(y, LRELU_SLOPE)
                xt = F.conv1d(xt, w1, sd[p + f"convs1.{m}.bias"], padding=(rb_k[j] * d - d) // 2, dilation=d)
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, w2, sd[p + f"convs2.{m}.bias"], padding=(rb_k[j] - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
        if taps is not None:
            taps[f"gen_stage{i}"] = x
    x = F.leaky_relu(x)                                     # default slope 0.01 (reference models.py:561)
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


def synth_infer(sd, config, phone, pitch, nsff0, sid, noise_z, noise_src, taps=None):
    """SynthesizerTrnMs{256,768}NSFsid.infer (reference models.py:682-693,:798-809) with explicit noise.

    phone [1,T,D] f32, pitch int64 [1,T], nsff0 f32 [1,T], sid int; noise_z [1,192,T] and
    noise_src [1,T*upp,1] are the two torch.randn_like draws in the reference's order.  -> [1,1,T*upp].
    pitch = nsff0 = noise_src = None: SynthesizerTrnMs{256,768}NSFsid_nono.infer (models.py:905-916,:1011-1022), one draw.
    """
    sd = tensors(sd)
    phone, noise_z = _t(phone).float(), _t(noise_z).float()
    if pitch is not None:
        pitch, nsff0, noise_src = _t(pitch).long(), _t(nsff0).float(), _t(noise_src).float()
    with torch.no_grad():
        g = sd["emb_g.weight"][int(sid)].view(1, -1, 1)
        m_p, logs_p = enc_p_forward(sd, config, phone, pitch, taps)
        z_p = m_p + torch.exp(logs_p) * noise_z * 0.66666
        z = flow_reverse(sd, config, z_p, g, taps)
        if taps is not None:
            taps.update(m_p=m_p, logs_p=logs_p, z_p=z_p, z=z)
        return generator_forward(sd, config, z, nsff0, g, noise_src, taps)
