"""MI355X-native RVC voice-conversion inference path (HuBERT -> RMVPE -> NSF-HiFiGAN synthesizer).

Python here is host plumbing only (argument handling, host DSP that the reference also does in
numpy/scipy, weight folding at load time); all network compute runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in include/rvc_hip.h (built into comfy-rvc_amd/csrc/librvc_hip.so).
"""
import os as _os

# Every clip lane drives two HIP streams (RMVPE / synthesizer and the HuBERT side stream).  The HIP runtime multiplexes streams onto
# 4 hardware queues by default; two busy streams that land on the same queue serialise (measured: 2 lanes 890 xRT instead of 1115).
# Read by the runtime when it initialises, so it has to be in the environment before the first HIP call; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"


# ComfyUI discovers a node pack through these two module attributes (reference __init__.py:12-31).  They resolve on first access
# (PEP 562) so that `import comfy_rvc_amd.synthetic` and the like stay free of torch / the HIP library; accessing them without the
# built library raises (there is no CPU path).
WEB_DIRECTORY = None


def __getattr__(name):
    if name in ("NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"):
        from .custom_nodes import rvc_nodes as _n, uvr as _u
        globals()["NODE_CLASS_MAPPINGS"] = {**_u.NODE_CLASS_MAPPINGS, **_n.NODE_CLASS_MAPPINGS}
        globals()["NODE_DISPLAY_NAME_MAPPINGS"] = {**_u.NODE_DISPLAY_NAME_MAPPINGS, **_n.NODE_DISPLAY_NAME_MAPPINGS}
        return globals()[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
