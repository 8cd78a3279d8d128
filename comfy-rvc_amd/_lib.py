"""ctypes binding of librvc_hip.so (C ABI: include/rvc_hip.h).

The product has no CPU path: importing this module without the built library raises, and creating a
context without a gfx950 device raises.  torch is used only for device memory / streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVC_HIP_LIB") or os.path.join(_HERE, "csrc", "librvc_hip.so")   # env override: A/B kernel builds


class RvcHipError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise RvcHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C comfy-rvc_amd/csrc`).  There is no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

c_void_p, c_int, c_int64, c_float, c_char_p = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_char_p
P = C.POINTER


class HubertTaps(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("conv_stack", "pos_conv", "hidden_0", "hidden_8")]


class RmvpeTaps(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("unet_out", "gru")]


class CrepeTaps(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("conv1", "embed")]


class Mdx23Config(C.Structure):
    _fields_ = [(n, c_int) for n in ("n_fft", "hop", "dim_f", "dim_t", "num_channels", "growth", "num_scales", "num_subbands", "blocks_per_scale",
                                     "bottleneck", "num_targets", "audio_channels")]


class SynthConfig(C.Structure):
    _fields_ = [("inter_channels", c_int), ("hidden_channels", c_int), ("filter_channels", c_int), ("n_heads", c_int),
                ("n_layers", c_int), ("kernel_size", c_int), ("n_resblock_kernels", c_int),
                ("resblock_kernel_sizes", c_int * 3), ("resblock_dilations", (c_int * 3) * 3), ("n_upsamples", c_int),
                ("upsample_rates", c_int * 8), ("upsample_kernel_sizes", c_int * 8), ("upsample_initial_channel", c_int),
                ("spk_embed_dim", c_int), ("gin_channels", c_int), ("sr", c_int), ("feat_dim", c_int)]


class SynthTaps(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("enc_p_layer0", "m_p", "logs_p", "z_p", "z", "sine_waves", "har_source", "gen_ups0",
                                        "gen_last")]


# every symbol include/rvc_hip.h declares (tests check that the library exports all of them)
SIGNATURES = {
    "rvc_last_error": (c_char_p, []),
    "rvc_version": (c_char_p, []),
    "rvc_ctx_create": (c_int, [c_int, P(c_void_p)]),
    "rvc_ctx_destroy": (c_int, [c_void_p]),
    "rvc_ctx_workspace_bytes": (c_int64, [c_void_p]),
    "rvc_ctx_set_conv_precision": (c_int, [c_void_p, c_int]),
    "rvc_hubert_create": (c_int, [c_void_p, P(c_void_p)]),
    "rvc_hubert_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int]),
    "rvc_hubert_finalize": (c_int, [c_void_p]),
    "rvc_hubert_destroy": (c_int, [c_void_p]),
    "rvc_hubert_num_frames": (c_int64, [c_int64]),
    "rvc_hubert_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, P(HubertTaps)]),
    "rvc_rmvpe_create": (c_int, [c_void_p, P(c_void_p)]),
    "rvc_rmvpe_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int]),
    "rvc_rmvpe_finalize": (c_int, [c_void_p]),
    "rvc_rmvpe_destroy": (c_int, [c_void_p]),
    "rvc_rmvpe_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, P(RmvpeTaps)]),
    "rvc_f0_post": (c_int, [c_void_p, c_void_p, c_int64, C.c_double, C.c_double, C.c_double, c_int, c_void_p, c_void_p]),
    "rvc_rmvpe_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "rvc_rmvpe_status": (c_int, [c_void_p, c_void_p]),
    "rvc_rmvpe_debug_fault": (c_int, [c_void_p, c_int, C.c_uint]),
    "rvc_rmvpe_repaired": (c_int, [c_void_p, c_void_p]),
    "rvc_crepe_create": (c_int, [c_void_p, c_int, P(c_void_p)]),
    "rvc_crepe_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int]),
    "rvc_crepe_finalize": (c_int, [c_void_p]),
    "rvc_crepe_destroy": (c_int, [c_void_p]),
    "rvc_crepe_num_frames": (c_int64, [c_int64, c_int, c_int]),
    "rvc_crepe_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, P(CrepeTaps)]),
    "rvc_crepe_viterbi": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "rvc_mdx23_create": (c_int, [c_void_p, P(Mdx23Config), P(c_void_p)]),
    "rvc_mdx23_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int]),
    "rvc_mdx23_finalize": (c_int, [c_void_p]),
    "rvc_mdx23_destroy": (c_int, [c_void_p]),
    "rvc_mdx23_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rvc_mdx23_set_streams": (c_int, [c_void_p, c_int]),
    "rvc_mdx23_demix": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "rvc_synth_create": (c_int, [c_void_p, P(SynthConfig), P(c_void_p)]),
    "rvc_synth_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int]),
    "rvc_synth_finalize": (c_int, [c_void_p]),
    "rvc_synth_destroy": (c_int, [c_void_p]),
    "rvc_synth_upp": (c_int, [c_void_p]),
    "rvc_synth_has_f0": (c_int, [c_void_p]),
    "rvc_synth_infer": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                c_void_p, P(SynthTaps)]),
    "rvc_vc_segment": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float, c_int,
                               c_void_p, c_void_p, c_void_p]),
    "rvc_vc_segment_feats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "rvc_index_create": (c_int, [c_void_p, c_void_p, c_int64, c_int, P(c_void_p)]),
    "rvc_index_create_ivf": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, P(c_void_p)]),
    "rvc_index_nprobe": (c_int, [c_void_p]),
    "rvc_index_destroy": (c_int, [c_void_p]),
    "rvc_index_ntotal": (c_int64, [c_void_p]),
    "rvc_index_search": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rvc_index_blend": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "rvc_preprocess": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                             c_void_p, c_int, c_void_p, c_void_p]),
    "rvc_postprocess": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_float, c_void_p]),
    "rvc_op_gemm_split": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_float] + [c_int] * 5),
    "rvc_op_conv2d_split": (c_int, [c_void_p] * 7 + [c_int] * 9),
    "rvc_op_wn_in_gate_split": (c_int, [c_void_p] * 6 + [c_int] * 4),
    "rvc_op_gemm_split_qkv": (c_int, [c_void_p] * 6 + [c_int] * 4),
    "rvc_op_conv1d_s2_split": (c_int, [c_void_p] * 6 + [c_int] * 5),
    "rvc_op_conv1d_split": (c_int, [c_void_p] * 6 + [c_int] * 9),
    "rvc_op_conv1d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 +
                      [c_int, c_float, c_int, c_float, c_int, c_float, c_int]),
    "rvc_op_conv_transpose1d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_int, c_float, c_int]),
    "rvc_op_conv2d3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5),
    "rvc_op_conv_transpose2d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5),
    "rvc_op_gemm_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 4),
    "rvc_conv1d_plan_create": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [P(c_void_p)]),
    "rvc_conv1d_plan_run": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_float]),
    "rvc_conv1d_plan_pair_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int]),
    "rvc_conv1d_plan_resblock_run": (c_int, [P(c_void_p), c_void_p, c_void_p, c_int, c_void_p, c_float, c_int, P(c_int), c_void_p, c_void_p, c_void_p]),
    "rvc_conv1d_plan_pair_split_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int]),
    "rvc_conv1d_plan_destroy": (c_int, [c_void_p]),
    "rvc_op_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "rvc_op_attention_split": (c_int, [c_void_p] * 7 + [c_int, c_int]),
    "rvc_op_attention_split_rel": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int]),
    "rvc_op_cbr2_small": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int]),
    "rvc_op_layernorm_c_split": (c_int, [c_void_p] * 6 + [c_int, c_int]),
    "rvc_op_conv3_small": (c_int, [c_void_p] * 7 + [c_int] * 6),
    "rvc_op_gemm_split_swapped": (c_int, [c_void_p] * 4 + [c_int] * 5),
    "rvc_op_conv2d3x3_plus_1x1": (c_int, [c_void_p] * 7 + [c_int] * 6),
    "rvc_op_gemm_split_swapped_res": (c_int, [c_void_p] * 5 + [c_int] * 5),
    "rvc_op_attention_rel": (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p, c_void_p]),
    "rvc_op_layernorm_c": (c_int, [c_void_p] * 6 + [c_int, c_int]),
    "rvc_resample": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_int64]),
    "rvc_set_conv_precision": (c_int, [c_int]),
    "rvc_set_pair_arithmetic": (c_int, [c_int]),
    "rvc_get_pair_arithmetic": (c_int, []),
    "rvc_conv1d_plan_pair_arithmetic": (c_int, [c_void_p, c_void_p, c_int]),
    "rvc_prof_dump_csv": (c_int, [c_char_p]),
    "rvc_prof_enable": (c_int, [c_int]),
    "rvc_prof_collect": (c_int, [P(C.c_double), P(C.c_double), P(c_int64)]),
    "rvc_prof_collect_ex": (c_int, [P(C.c_double), C.c_double, C.c_double]),
    "rvc_prof_cfg_name": (c_char_p, [c_int]),
    "rvc_op_sine_source": (c_int, [c_void_p] * 5 + [c_int, c_int, c_float, c_float, c_float] + [c_void_p] * 3),
}

# instrumentation hooks of -DRVC_EXPERIMENTS builds (include/rvc_hip.h, last section): bound when the loaded library has them (RVC_HIP_LIB=<variant build>),
# absent from the product library - callers test `_lib.has_experiments`
EXPERIMENT_SIGNATURES = {
    "rvc_debug_conv_timing": (c_int, [P(C.c_uint64), c_int]),
    "rvc_debug_x3p_check": (c_int, []),
    "rvc_debug_set_x3s_mode": (c_int, [c_int]),
    "rvc_debug_gemm_split_bench": (c_int, [c_void_p] + [c_int] * 8 + [P(c_float), c_int, c_int]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args
has_experiments = all(hasattr(lib, _n) for _n in EXPERIMENT_SIGNATURES)
if has_experiments:
    for _name, (_res, _args) in EXPERIMENT_SIGNATURES.items():
        _fn = getattr(lib, _name)
        _fn.restype = _res
        _fn.argtypes = _args


def require_experiments():
    """Tools that read the instrumentation hooks call this first: they need a -DRVC_EXPERIMENTS build (tools/build_variant.sh NAME ..., RVC_HIP_LIB=<path>)."""
    if not has_experiments:
        raise RuntimeError(f"{LIB_PATH} is the product library: the rvc_debug_* hooks exist in -DRVC_EXPERIMENTS builds only - "
                           "bash tools/build_variant.sh exp && RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_exp.so python <tool>")


def check(status):
    if status != 0:
        raise RvcHipError(lib.rvc_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return c_void_p(t.data_ptr())
    return c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_ctx = {}


def get_ctx(device_index=0):
    """One rvc_ctx per GPU (lazily created)."""
    if device_index not in _ctx:
        h = c_void_p()
        check(lib.rvc_ctx_create(int(device_index), C.byref(h)))
        _ctx[device_index] = h
    return _ctx[device_index]


def set_tensors(set_fn, handle, state_dict):
    """Feeds every float tensor of a state dict (torch tensors or numpy arrays) to rvc_*_set_tensor."""
    import numpy as np
    for name, v in state_dict.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu()
            if not v.dtype.is_floating_point:
                continue
            v = v.float().contiguous().numpy()
        else:
            v = np.asarray(v)
            if not np.issubdtype(v.dtype, np.floating):
                continue
            v = np.ascontiguousarray(v, dtype=np.float32)
        shape = (c_int64 * max(v.ndim, 1))(*(v.shape if v.ndim else (1,)))
        check(set_fn(handle, name.encode(), c_void_p(v.ctypes.data), shape, max(v.ndim, 1)))
