"""MI355X-native RVC voice-conversion inference path (HuBERT -> RMVPE -> NSF-HiFiGAN synthesizer).

Python here is host plumbing only (argument handling, host DSP that the reference also does in
numpy/scipy, weight folding at load time); all network compute runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in include/rvc_hip.h (built into comfy-rvc_amd/csrc/librvc_hip.so).
"""
__version__ = "0.1.0"
