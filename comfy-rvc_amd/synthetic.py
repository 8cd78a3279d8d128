"""Procedural ("synthetic") checkpoints for the three networks on the RVC inference path.

No real checkpoint (content-vec-best.safetensors, rmvpe.pt, RVC *.pth) is reachable offline, so parity
tests and bench.py run on deterministic pseudo-random weights w = f(seed, tensor_name, shape).
The state-dict *names and shapes* follow the reference so the very same dict loads into the
reference modules with `load_state_dict(strict=True)`:
  * HuBERT   : transformers HubertModel + final_proj   (reference lib/infer_pack/loaders.py:10-17)
  * RMVPE    : E2E(4, 1, (2, 2))                       (reference lib/rmvpe.py:431-470, :579)
  * synthesizer `cpt` dict {weight, config, f0, version, sr, info}
                                                        (reference training_cli.py:38-74,
                                                         vc_infer_pipeline.py:198-221)
Values depend only on (seed, name, shape) through numpy's PCG64, so the build container and the
GPU box regenerate identical tensors; nothing is stored on disk.
"""
import hashlib
from collections import OrderedDict

import numpy as np

CONFIG_40K_V2 = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [10, 10, 2, 2], 512, [16, 16, 4, 4], 109, 256, 40000]
CONFIG_48K_V2 = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [12, 10, 2, 2], 512, [24, 20, 4, 4], 109, 256, 48000]
# the remaining shipped generator shapes (reference configs/32k.json, 48k.json: five upsampling stages down to 16 channels, transposed
# convolutions whose kernel is not a multiple of the stride; configs/32k_v2.json)
CONFIG_32K_V1 = [513, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [10, 4, 2, 2, 2], 512, [16, 16, 4, 4, 4], 109, 256, 32000]
CONFIG_48K_V1 = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [10, 6, 2, 2, 2], 512, [16, 16, 4, 4, 4], 109, 256, 48000]
CONFIG_32K_V2 = [513, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [10, 8, 2, 2], 512, [20, 16, 4, 4], 109, 256, 32000]
CONFIG_40K_V1 = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 [10, 10, 2, 2], 512, [16, 16, 4, 4], 109, 256, 40000]


def _rng(seed, name):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:16], "little")))


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(shape) * std).astype(np.float32)


def _uniform(seed, name, shape, lo, hi):
    return _rng(seed, name).uniform(lo, hi, shape).astype(np.float32)


def _fan_in(shape):
    return int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])


# ----------------------------------------------------------------------------------- HuBERT
def hubert_spec(num_layers=12):
    """name -> shape for HubertModelWithFinalProj(HubertConfig()) (transformers 5.x parametrized names)."""
    s = OrderedDict()
    s["masked_spec_embed"] = (768,)
    kern = (10, 3, 3, 3, 3, 2, 2)
    for i, k in enumerate(kern):
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (512, 1 if i == 0 else 512, k)
        if i == 0:
            s["feature_extractor.conv_layers.0.layer_norm.weight"] = (512,)
            s["feature_extractor.conv_layers.0.layer_norm.bias"] = (512,)
    s["feature_projection.layer_norm.weight"] = (512,)
    s["feature_projection.layer_norm.bias"] = (512,)
    s["feature_projection.projection.weight"] = (768, 512)
    s["feature_projection.projection.bias"] = (768,)
    s["encoder.pos_conv_embed.conv.bias"] = (768,)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, 128)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (768, 48, 128)
    s["encoder.layer_norm.weight"] = (768,)
    s["encoder.layer_norm.bias"] = (768,)
    for l in range(num_layers):
        p = f"encoder.layers.{l}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{proj}.weight"] = (768, 768)
            s[p + f"attention.{proj}.bias"] = (768,)
        s[p + "layer_norm.weight"] = (768,)
        s[p + "layer_norm.bias"] = (768,)
        s[p + "feed_forward.intermediate_dense.weight"] = (3072, 768)
        s[p + "feed_forward.intermediate_dense.bias"] = (3072,)
        s[p + "feed_forward.output_dense.weight"] = (768, 3072)
        s[p + "feed_forward.output_dense.bias"] = (768,)
        s[p + "final_layer_norm.weight"] = (768,)
        s[p + "final_layer_norm.bias"] = (768,)
    s["final_proj.weight"] = (256, 768)
    s["final_proj.bias"] = (256,)
    return s


def _lognormal(seed, name, n, sigma):
    """n gains exp(N(0, sigma)), rescaled to unit mean square (a layer's output power stays what the plain family gives it)."""
    g = np.exp(_rng(seed, name).standard_normal(n) * sigma)
    return (g / np.sqrt(np.mean(g * g))).astype(np.float32)


def _outliers(seed, name, n, count, lo, hi):
    """`count` channel indices of `n` and their gains in [lo, hi) - the few x10 - x30 channels trained checkpoints carry."""
    rng = _rng(seed, name)
    idx = rng.choice(n, size=count, replace=False)
    return idx, rng.uniform(lo, hi, count).astype(np.float32)


def _heavy_hubert(sd, seed):
    """family="heavy": what a TRAINED ContentVec looks like to the kernels and the plain Gaussian family does not - per-channel gains spread log-normally
    and a few outlier channels: six FFN hidden units per layer at x10 - x30 (their output_dense columns scaled back, so the residual stream keeps
    its scale while the 3072-row intermediate carries the large values through GELU and the bf16 hi / lo split), and three residual-stream channels
    per layer whose LayerNorm gain is x8 - x20 (the "massive activations" of transformer checkpoints; the next layer's projections read them through
    columns scaled back).  Layer 10's output LayerNorm - the v2 feature the synthesizer and the index consume - keeps plain gains."""
    for l in range(12):
        p = f"encoder.layers.{l}."
        g = _lognormal(seed, p + "ffn.gain", 3072, 0.4)
        idx, big = _outliers(seed, p + "ffn.outliers", 3072, 6, 10.0, 30.0)
        g[idx] *= big
        sd[p + "feed_forward.intermediate_dense.weight"] = sd[p + "feed_forward.intermediate_dense.weight"] * g[:, None]
        sd[p + "feed_forward.intermediate_dense.bias"] = sd[p + "feed_forward.intermediate_dense.bias"] * g
        sd[p + "feed_forward.output_dense.weight"] = sd[p + "feed_forward.output_dense.weight"] / g[None, :]
        if l in (10, 11):
            continue
        h = _lognormal(seed, p + "ln.gain", 768, 0.3)
        idx, big = _outliers(seed, p + "ln.outliers", 768, 3, 8.0, 20.0)
        h[idx] *= big
        sd[p + "final_layer_norm.weight"] = sd[p + "final_layer_norm.weight"] * h
        sd[p + "final_layer_norm.bias"] = sd[p + "final_layer_norm.bias"] * h
        q = f"encoder.layers.{l + 1}."
        for name in ("attention.q_proj", "attention.k_proj", "attention.v_proj", "feed_forward.intermediate_dense"):
            sd[q + name + ".weight"] = sd[q + name + ".weight"] / h[None, :]
    return OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items())


def hubert_state_dict(seed=0, family="plain"):
    """family: "plain" (Gaussian, fan-in scaled) or "heavy" (_heavy_hubert: log-normal channel gains + outlier channels on top of the plain draw)."""
    assert family in ("plain", "heavy")
    if family == "heavy":
        return _heavy_hubert(hubert_state_dict(seed), seed)
    sd = OrderedDict()
    for name, shape in hubert_spec().items():
        if name.endswith("layer_norm.weight"):
            v = _uniform(seed, name, shape, 0.8, 1.2)
        elif name.endswith("layer_norm.bias"):
            v = _normal(seed, name, shape, 0.05)
        elif name.endswith("original0"):          # weight-norm g, dim=2 -> one gain per kernel tap
            v = _uniform(seed, name, shape, 0.8, 1.2) * np.float32(np.sqrt(768 * 48) * 0.5 / np.sqrt(48 * 128))
        elif name.endswith("original1"):
            v = _normal(seed, name, shape, 1.0)
        elif name.endswith(".bias"):
            v = _normal(seed, name, shape, 0.02)
        elif name == "masked_spec_embed":
            v = _uniform(seed, name, shape, 0.0, 1.0)
        elif "feature_extractor" in name:         # conv + GELU stack: keep the variance roughly constant
            v = _normal(seed, name, shape, 1.4 / np.sqrt(_fan_in(shape)))
        else:
            v = _normal(seed, name, shape, 1.0 / np.sqrt(_fan_in(shape)))
        sd[name] = v
    return sd


HUBERT_CONFIG = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                     conv_dim=[512] * 7, conv_stride=[5, 2, 2, 2, 2, 2, 2], conv_kernel=[10, 3, 3, 3, 3, 2, 2],
                     num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5,
                     classifier_proj_size=256, feat_extract_norm="group", conv_bias=False,
                     do_stable_layer_norm=False, feat_proj_layer_norm=True)


# ----------------------------------------------------------------------------------- RMVPE
def _convblockres_spec(s, p, cin, cout):
    for conv, bn, ci in (("conv.0", "conv.1", cin), ("conv.3", "conv.4", cout)):
        s[p + f"{conv}.weight"] = (cout, ci, 3, 3)
        for q, shp in (("weight", (cout,)), ("bias", (cout,)), ("running_mean", (cout,)),
                       ("running_var", (cout,)), ("num_batches_tracked", ())):
            s[p + f"{bn}.{q}"] = shp
    if cin != cout:
        s[p + "shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "shortcut.bias"] = (cout,)


def rmvpe_spec():
    """name -> shape for E2E(n_blocks=4, n_gru=1, kernel_size=(2,2)) in state_dict order."""
    s = OrderedDict()
    for q, shp in (("weight", (1,)), ("bias", (1,)), ("running_mean", (1,)), ("running_var", (1,)),
                   ("num_batches_tracked", ())):
        s[f"unet.encoder.bn.{q}"] = shp
    cin, cout = 1, 16
    for i in range(5):
        for b in range(4):
            _convblockres_spec(s, f"unet.encoder.layers.{i}.conv.{b}.", cin if b == 0 else cout, cout)
        cin, cout = cout, cout * 2
    cin, cout = 256, 512
    for i in range(4):
        for b in range(4):
            _convblockres_spec(s, f"unet.intermediate.layers.{i}.conv.{b}.", cin if b == 0 else cout, cout)
        cin = cout
    cin = 512
    for i in range(5):
        cout = cin // 2
        p = f"unet.decoder.layers.{i}."
        s[p + "conv1.0.weight"] = (cin, cout, 3, 3)
        for q, shp in (("weight", (cout,)), ("bias", (cout,)), ("running_mean", (cout,)),
                       ("running_var", (cout,)), ("num_batches_tracked", ())):
            s[p + f"conv1.1.{q}"] = shp
        for b in range(4):
            _convblockres_spec(s, p + f"conv2.{b}.", cout * 2 if b == 0 else cout, cout)
        cin = cout
    s["cnn.weight"] = (3, 16, 3, 3)
    s["cnn.bias"] = (3,)
    for sfx in ("", "_reverse"):
        s[f"fc.0.gru.weight_ih_l0{sfx}"] = (768, 384)
        s[f"fc.0.gru.weight_hh_l0{sfx}"] = (768, 256)
        s[f"fc.0.gru.bias_ih_l0{sfx}"] = (768,)
        s[f"fc.0.gru.bias_hh_l0{sfx}"] = (768,)
    s["fc.1.weight"] = (360, 512)
    s["fc.1.bias"] = (360,)
    return s


def rmvpe_state_dict(seed=0):
    sd = OrderedDict()
    for name, shape in rmvpe_spec().items():
        if name.endswith("num_batches_tracked"):
            v = np.array(1000, dtype=np.int64)
        elif name.endswith("running_mean"):
            v = _normal(seed, name, shape, 0.1)
        elif name.endswith("running_var"):
            v = _uniform(seed, name, shape, 0.5, 1.5)
        elif name == "unet.encoder.bn.weight":
            v = np.full(shape, 0.25, dtype=np.float32)   # log-mel spans roughly [-11.5, 3] -> O(1)
        elif name == "unet.encoder.bn.running_mean":
            v = np.full(shape, -4.0, dtype=np.float32)
        elif name.endswith(".conv.4.weight"):           # BN gain of the residual branch: keeps the 52-block U-Net O(1)
            v = _uniform(seed, name, shape, 0.15, 0.3)
        elif len(shape) == 1 and (".conv.1." in name or ".conv1.1." in name) and name.endswith("weight"):
            v = _uniform(seed, name, shape, 0.8, 1.2)
        elif "gru" in name:
            v = _uniform(seed, name, shape, -1.0 / 16.0, 1.0 / 16.0)
        elif name == "fc.1.bias":
            v = _normal(seed, name, shape, 0.3) - np.float32(6.0)
        elif name == "fc.1.weight":                     # per-bin part + a part shared by all bins (gives unvoiced frames)
            v = _normal(seed, name, shape, 2.5 / np.sqrt(512)) + _normal(seed, name + ".shared", (1, 512), 8.0 / np.sqrt(512))
        elif name == "cnn.weight":
            v = _normal(seed, name, shape, 0.15 / np.sqrt(_fan_in(shape)))
        elif name.endswith(".bias"):
            v = _normal(seed, name, shape, 0.05)
        elif "conv1.0.weight" in name:                  # ConvTranspose2d [Cin, Cout, 3, 3]; ~9/4 taps hit per output
            v = _normal(seed, name, shape, np.sqrt(2.0 / (shape[0] * 9 / 4)))
        elif name.endswith("conv.0.weight") or name.endswith("conv.3.weight"):   # He scaling in front of the ReLUs
            v = _normal(seed, name, shape, np.sqrt(2.0 / _fan_in(shape)))
        else:
            v = _normal(seed, name, shape, 1.0 / np.sqrt(_fan_in(shape)))
        sd[name] = v
    return sd


# ----------------------------------------------------------------------------------- synthesizer
def synth_spec(config, version="v2"):
    """name -> shape for SynthesizerTrnMs{256,768}NSFsid(*config) after `del net_g.enc_q`."""
    (_spec, _seg, inter, hidden, filt, n_heads, n_layers, ksz, _pd, _rb, rb_k, rb_d, up_rates, up_init, up_k,
     n_spk, gin, _sr) = config
    s = OrderedDict()
    s["enc_p.emb_phone.weight"] = (hidden, 768 if version == "v2" else 256)
    s["enc_p.emb_phone.bias"] = (hidden,)
    s["enc_p.emb_pitch.weight"] = (256, hidden)
    kc = hidden // n_heads
    for l in range(n_layers):
        p = f"enc_p.encoder.attn_layers.{l}."
        s[p + "emb_rel_k"] = (1, 21, kc)
        s[p + "emb_rel_v"] = (1, 21, kc)
        for c in ("conv_q", "conv_k", "conv_v", "conv_o"):
            s[p + c + ".weight"] = (hidden, hidden, 1)
            s[p + c + ".bias"] = (hidden,)
    for l in range(n_layers):
        s[f"enc_p.encoder.norm_layers_1.{l}.gamma"] = (hidden,)
        s[f"enc_p.encoder.norm_layers_1.{l}.beta"] = (hidden,)
    for l in range(n_layers):
        p = f"enc_p.encoder.ffn_layers.{l}."
        s[p + "conv_1.weight"] = (filt, hidden, ksz)
        s[p + "conv_1.bias"] = (filt,)
        s[p + "conv_2.weight"] = (hidden, filt, ksz)
        s[p + "conv_2.bias"] = (hidden,)
    for l in range(n_layers):
        s[f"enc_p.encoder.norm_layers_2.{l}.gamma"] = (hidden,)
        s[f"enc_p.encoder.norm_layers_2.{l}.beta"] = (hidden,)
    s["enc_p.proj.weight"] = (inter * 2, hidden, 1)
    s["enc_p.proj.bias"] = (inter * 2,)
    s["dec.m_source.l_linear.weight"] = (1, 1)
    s["dec.m_source.l_linear.bias"] = (1,)
    nu = len(up_rates)
    for i in range(nu):
        c_cur = up_init // (2 ** (i + 1))
        if i + 1 < nu:
            sf0 = int(np.prod(up_rates[i + 1:]))
            s[f"dec.noise_convs.{i}.weight"] = (c_cur, 1, sf0 * 2)
        else:
            s[f"dec.noise_convs.{i}.weight"] = (c_cur, 1, 1)
        s[f"dec.noise_convs.{i}.bias"] = (c_cur,)
    s["dec.conv_pre.weight"] = (up_init, inter, 7)
    s["dec.conv_pre.bias"] = (up_init,)
    for i in range(nu):
        cin, cout = up_init // (2 ** i), up_init // (2 ** (i + 1))
        s[f"dec.ups.{i}.bias"] = (cout,)
        s[f"dec.ups.{i}.weight_g"] = (cin, 1, 1)
        s[f"dec.ups.{i}.weight_v"] = (cin, cout, up_k[i])
    for i in range(nu):
        ch = up_init // (2 ** (i + 1))
        for j, k in enumerate(rb_k):
            p = f"dec.resblocks.{i * len(rb_k) + j}."
            for cs in ("convs1", "convs2"):
                for m in range(3):
                    s[p + f"{cs}.{m}.bias"] = (ch,)
                    s[p + f"{cs}.{m}.weight_g"] = (ch, 1, 1)
                    s[p + f"{cs}.{m}.weight_v"] = (ch, ch, k)
    s["dec.conv_post.weight"] = (1, up_init // (2 ** nu), 7)
    s["dec.cond.weight"] = (up_init, gin, 1)
    s["dec.cond.bias"] = (up_init,)
    for f in range(4):
        p = f"flow.flows.{2 * f}."
        s[p + "pre.weight"] = (hidden, inter // 2, 1)
        s[p + "pre.bias"] = (hidden,)
        for l in range(3):
            s[p + f"enc.in_layers.{l}.bias"] = (2 * hidden,)
            s[p + f"enc.in_layers.{l}.weight_g"] = (2 * hidden, 1, 1)
            s[p + f"enc.in_layers.{l}.weight_v"] = (2 * hidden, hidden, 5)
        for l in range(3):
            rs = 2 * hidden if l < 2 else hidden
            s[p + f"enc.res_skip_layers.{l}.bias"] = (rs,)
            s[p + f"enc.res_skip_layers.{l}.weight_g"] = (rs, 1, 1)
            s[p + f"enc.res_skip_layers.{l}.weight_v"] = (rs, hidden, 1)
        s[p + "enc.cond_layer.bias"] = (2 * hidden * 3,)
        s[p + "enc.cond_layer.weight_g"] = (2 * hidden * 3, 1, 1)
        s[p + "enc.cond_layer.weight_v"] = (2 * hidden * 3, gin, 1)
        s[p + "post.weight"] = (inter // 2, hidden, 1)
        s[p + "post.bias"] = (inter // 2,)
    s["emb_g.weight"] = (n_spk, gin)
    return s


F0_ONLY_KEYS = ("enc_p.emb_pitch.", "dec.m_source.", "dec.noise_convs.")   # absent from the *_nono (no-f0) synthesizers


def _heavy_synth(sd, config, seed):
    """family="heavy" for the synthesizer (before the fp16 rounding of the checkpoint): log-normal weight_g everywhere it exists (generator, flow), and in
    every ResBlock pair three output channels of convs1 at x10 - x30 with the matching input columns of convs2's weight_v scaled back - the pair's
    intermediate (the tensor the split-resident image carries) then holds a few channels an order of magnitude above the rest, as trained
    vocoders do, while the stage tensor keeps its scale.  LayerNorm gains of the text encoder spread log-normally too."""
    nk = len(config[10])
    for name in list(sd.keys()):
        if name.endswith("weight_g"):
            sd[name] = sd[name] * _lognormal(seed, name + ".gain", sd[name].shape[0], 0.4).reshape(sd[name].shape)
        elif name.endswith("gamma"):
            sd[name] = sd[name] * _lognormal(seed, name + ".gain", sd[name].shape[0], 0.3)
    for i in range(len(config[12])):
        for j in range(nk):
            for m in range(3):
                p = f"dec.resblocks.{i * nk + j}."
                ch = sd[p + f"convs1.{m}.weight_g"].shape[0]
                idx, big = _outliers(seed, p + f"{m}.outliers", ch, 3, 10.0, 30.0)
                g = sd[p + f"convs1.{m}.weight_g"].copy()
                g[idx, 0, 0] *= big
                sd[p + f"convs1.{m}.weight_g"] = g
                b = sd[p + f"convs1.{m}.bias"].copy()
                b[idx] *= big
                sd[p + f"convs1.{m}.bias"] = b
                v = sd[p + f"convs2.{m}.weight_v"].copy()
                v[:, idx, :] /= big[None, :, None]
                sd[p + f"convs2.{m}.weight_v"] = v
    return sd


def synth_state_dict(config, version="v2", seed=0, fp16_round=True, f0=True, family="plain"):
    """family: "plain" or "heavy" (_heavy_synth)."""
    assert family in ("plain", "heavy")
    if family == "heavy":
        sd = _heavy_synth(synth_state_dict(config, version, seed, fp16_round=False, f0=f0), config, seed)
        if fp16_round:
            sd = OrderedDict((k, v.astype(np.float16).astype(np.float32)) for k, v in sd.items())
        return OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items())
    spec = synth_spec(config, version)
    sd = OrderedDict()
    for name, shape in spec.items():
        if not f0 and name.startswith(F0_ONLY_KEYS):
            continue
        if name.endswith("weight_g"):
            vshape = spec[name[:-1] + "v"]
            fan = int(np.prod(vshape[1:]))
            gain = 1.0
            if ".ups." in name:                    # ConvTranspose1d: only k/stride taps hit per output
                gain = 1.2
            # |v| per dim-0 slice is ~ sqrt(fan) for unit-normal v; g sets the effective row norm
            v = _uniform(seed, name, shape, 0.8, 1.2) * np.float32(gain)
            if ".ups." in name:
                k = vshape[2]
                v = v * np.float32(np.sqrt(vshape[1] * k) / np.sqrt(vshape[0] * 2.0))
            elif "res_skip" in name or "cond_layer" in name:
                v = v * np.float32(0.5)
        elif name.endswith("weight_v"):
            v = _normal(seed, name, shape, 1.0)
        elif name.endswith("gamma"):
            v = _uniform(seed, name, shape, 0.8, 1.2)
        elif name.endswith("beta"):
            v = _normal(seed, name, shape, 0.05)
        elif "emb_rel" in name:
            v = _normal(seed, name, shape, shape[-1] ** -0.5)
        elif name == "enc_p.emb_pitch.weight":
            v = _normal(seed, name, shape, 0.05)
        elif name == "emb_g.weight":
            v = _normal(seed, name, shape, 0.5)
        elif name == "dec.m_source.l_linear.weight":
            v = np.full(shape, 0.9, dtype=np.float32)
        elif name == "dec.m_source.l_linear.bias":
            v = np.full(shape, 0.01, dtype=np.float32)
        elif name == "dec.conv_post.weight":
            v = _normal(seed, name, shape, 0.4 / np.sqrt(_fan_in(shape)))
        elif "noise_convs" in name and name.endswith("weight"):
            v = _normal(seed, name, shape, 2.0 / np.sqrt(_fan_in(shape)))
        elif "flow" in name and "post.weight" in name:
            v = _normal(seed, name, shape, 0.3 / np.sqrt(_fan_in(shape)))
        elif name == "enc_p.proj.weight":
            v = _normal(seed, name, shape, 0.5 / np.sqrt(_fan_in(shape)))
        elif name.endswith(".bias"):
            v = _normal(seed, name, shape, 0.02)
        else:
            v = _normal(seed, name, shape, 1.0 / np.sqrt(_fan_in(shape)))
        if fp16_round:                             # real RVC checkpoints store fp16 (training_cli.py:38-74)
            v = v.astype(np.float16).astype(np.float32)
        sd[name] = v
    return sd


def synth_checkpoint(config=None, version="v2", seed=0, f0=1, family="plain"):
    """The `cpt` dict layout `get_vc` reads (reference vc_infer_pipeline.py:199-221), as numpy arrays."""
    config = list(CONFIG_40K_V2 if config is None else config)
    return {"weight": synth_state_dict(config, version, seed, f0=bool(f0), family=family), "config": config, "f0": int(f0), "version": version,
            "sr": {32000: "32k", 40000: "40k", 48000: "48k"}[config[-1]], "info": "synthetic"}


# ----------------------------------------------------------------------------------- inputs
def synth_audio(seconds, seed=0, sr=16000):
    """SURVEY 8(d) synthetic clip: AM'd glide 110-440 Hz with 20 % silent gaps plus a little noise."""
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    rng = _rng(seed, "audio")
    f = 110.0 * 2.0 ** (2.0 * (0.5 - 0.5 * np.cos(2 * np.pi * t / 7.0)))
    phase = 2 * np.pi * np.cumsum(f) / sr
    x = 0.3 * np.sin(phase) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
    gap = ((t % 2.5) > 2.0)
    x = np.where(gap, 0.0, x)
    x = x + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)


def designed_f0(n_frames, seed=0):
    """A voiced/unvoiced contour at 100 fps (Hz, 0 = unvoiced) for driving the synthesizer directly."""
    t = np.arange(n_frames, dtype=np.float64) / 100.0
    f0 = 220.0 * 2.0 ** (0.5 * np.sin(2 * np.pi * t / 1.7) + 0.3 * np.sin(2 * np.pi * t / 0.31))
    uv = ((t % 0.9) > 0.7)
    f0 = np.where(uv, 0.0, f0)
    return f0.astype(np.float32)


# ----------------------------------------------------------------------------------- CREPE (torchcrepe state-dict names)
CREPE_CHANNELS = {"full": [1024, 128, 128, 128, 256, 512], "tiny": [128, 16, 16, 16, 32, 64]}


def crepe_state_dict(model="full", seed=0):
    """Procedural weights in torchcrepe's layout (conv{i}.weight [Co, Ci, k, 1], conv{i}_BN.*, classifier.*): unit-gain convolutions,
    BatchNorm statistics near the identity, a classifier whose logits spread over a few units (pitch salience with clear maxima)."""
    ch = CREPE_CHANNELS[model]
    sd, cin = {}, 1
    for i, co in enumerate(ch, 1):
        k = 512 if i == 1 else 64
        sd[f"conv{i}.weight"] = _normal(seed, f"crepe.{model}.conv{i}.w", (co, cin, k, 1), np.sqrt(2.0 / (cin * k)))
        sd[f"conv{i}.bias"] = _normal(seed, f"crepe.{model}.conv{i}.b", (co,), 0.05)
        sd[f"conv{i}_BN.weight"] = _uniform(seed, f"crepe.{model}.bn{i}.g", (co,), 0.8, 1.2)
        sd[f"conv{i}_BN.bias"] = _normal(seed, f"crepe.{model}.bn{i}.b", (co,), 0.1)
        sd[f"conv{i}_BN.running_mean"] = _uniform(seed, f"crepe.{model}.bn{i}.m", (co,), 0.2, 0.6)
        sd[f"conv{i}_BN.running_var"] = _uniform(seed, f"crepe.{model}.bn{i}.v", (co,), 0.3, 0.7)
        cin = co
    F = 4 * ch[-1]
    sd["classifier.weight"] = _normal(seed, f"crepe.{model}.fc.w", (PITCH_BINS_CREPE, F), 2.0 / np.sqrt(F))
    sd["classifier.bias"] = _normal(seed, f"crepe.{model}.fc.b", (PITCH_BINS_CREPE,), 0.5) - 1.0
    return sd


PITCH_BINS_CREPE = 360


# ----------------------------------------------------------------------------------- MDX23C (karafan TFC_TDF_net state-dict names)
def mdx23c_config(n_fft=8192, hop=1024, dim_f=4096, dim_t=256, num_channels=128, growth=128, num_scales=5, num_subbands=4, blocks=2, bottleneck=4,
                  overlap=8):
    """The fields of reference lib/karafan/Data/model_2_stem_full_band_8k.yaml that the network and demix_mdxv3 read (defaults = that file)."""
    return {"audio": {"chunk_size": hop * (dim_t - 1), "dim_f": dim_f, "dim_t": dim_t, "hop_length": hop, "n_fft": n_fft, "num_channels": 2,
                      "sample_rate": 44100},
            "model": {"act": "gelu", "bottleneck_factor": bottleneck, "growth": growth, "norm": "InstanceNorm", "num_blocks_per_scale": blocks,
                      "num_channels": num_channels, "num_scales": num_scales, "num_subbands": num_subbands, "scale": [2, 2]},
            "training": {"instruments": ["Vocals", "Instrumental"], "target_instrument": None},
            "inference": {"batch_size": 1, "dim_t": dim_t, "num_overlap": overlap}}


MDX23C_SMALL = dict(n_fft=512, hop=64, dim_f=256, dim_t=16, num_channels=32, growth=16, num_scales=2, num_subbands=4)


def mdx23c_spec(cfg):
    """(name, shape, kind) for every tensor of TFC_TDF_net(cfg).state_dict(), in the reference's construction order (tfc_tdf.py:147-195)."""
    m, a = cfg["model"], cfg["audio"]
    k, n, l, g, bn = m["num_subbands"], m["num_scales"], m["num_blocks_per_scale"], m["growth"], m["bottleneck_factor"]
    dim_c = k * a["num_channels"] * 2
    S = len(cfg["training"]["instruments"])
    c, f = m["num_channels"], a["dim_f"] // k
    out = [("first_conv.weight", (c, dim_c, 1, 1), "conv")]

    def norm(p, ch):
        out.append((p + ".weight", (ch,), "gamma")); out.append((p + ".bias", (ch,), "beta"))

    def tfc(prefix, in_c, ch, ff):
        for i in range(l):
            p = f"{prefix}.blocks.{i}."
            norm(p + "tfc1.0", in_c); out.append((p + "tfc1.2.weight", (ch, in_c, 3, 3), "conv"))
            norm(p + "tdf.0", ch); out.append((p + "tdf.2.weight", (ff // bn, ff), "conv"))
            norm(p + "tdf.3", ch); out.append((p + "tdf.5.weight", (ff, ff // bn), "conv"))
            norm(p + "tfc2.0", ch); out.append((p + "tfc2.2.weight", (ch, ch, 3, 3), "conv"))
            out.append((p + "shortcut.weight", (ch, in_c, 1, 1), "conv"))
            in_c = ch
    for i in range(n):
        tfc(f"encoder_blocks.{i}.tfc_tdf", c, c, f)
        norm(f"encoder_blocks.{i}.downscale.conv.0", c); out.append((f"encoder_blocks.{i}.downscale.conv.2.weight", (c + g, c, 2, 2), "conv"))
        f //= 2; c += g
    tfc("bottleneck_block", c, c, f)
    for i in range(n):
        norm(f"decoder_blocks.{i}.upscale.conv.0", c); out.append((f"decoder_blocks.{i}.upscale.conv.2.weight", (c, c - g, 2, 2), "tconv"))
        f *= 2; c -= g
        tfc(f"decoder_blocks.{i}.tfc_tdf", 2 * c, c, f)
    out.append(("final_conv.0.weight", (c, c + dim_c, 1, 1), "conv"))
    out.append(("final_conv.2.weight", (S * dim_c, c, 1, 1), "conv"))
    return out


def mdx23c_state_dict(cfg, seed=0):
    sd = {}
    for name, shape, kind in mdx23c_spec(cfg):
        if kind == "gamma":
            sd[name] = _uniform(seed, "mdx." + name, shape, 0.8, 1.2)
        elif kind == "beta":
            sd[name] = _normal(seed, "mdx." + name, shape, 0.1)
        else:
            fan = int(np.prod(shape[1:])) if kind == "conv" else int(shape[0] * shape[2] * shape[3])
            sd[name] = _normal(seed, "mdx." + name, shape, 1.0 / np.sqrt(fan))
    return sd


def mdx23c_full_chunk(seed=5):
    """The stereo 44.1 kHz chunk (261120 samples = 5.9 s) of the full-recipe MDX23C fixture (tests/golden/mdx23c_full_chunk.npz): the
    voice-like test signal on both channels with different seeds plus a noise bed, so that the mask network sees structured input."""
    a = np.stack([synth_audio(261120 / 44100.0 + 0.01, seed=seed, sr=44100)[:261120], synth_audio(261120 / 44100.0 + 0.01, seed=seed + 1, sr=44100)[:261120]])
    rng = np.random.default_rng(seed)
    return (a + 0.02 * rng.standard_normal(a.shape)).astype(np.float32)
