import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session", autouse=True)
def _x3p_wait_count_check():
    """On a -DRVC_X3P_CHECK build (RVC_HIP_LIB=.../librvc_hip_x3pcheck.so) the pipelined bf16x3 kernels count every wait whose compile-time
    vmcnt exceeded the exact run-time count; a whole `-m gpu` session must end with none.  Ordinary builds report -1."""
    yield
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from comfy_rvc_amd import _lib
        if not _lib.has_experiments:          # product library: no instrumentation hooks (include/rvc_hip.h, last section)
            return
        bad = _lib.lib.rvc_debug_x3p_check()
    except Exception:
        return
    assert bad <= 0, f"{bad} waits of the pipelined kernels had a too large compile-time count"


@pytest.fixture
def pair_arith():
    """Sets the ResBlock-pair arithmetic (rvc_set_pair_arithmetic: 0 bf16x3, 1 fp16x2) for one test and restores the previous mode afterwards."""
    from comfy_rvc_amd import _lib
    prev = _lib.lib.rvc_get_pair_arithmetic()

    def set_mode(mode):
        _lib.check(_lib.lib.rvc_set_pair_arithmetic(int(mode)))
    yield set_mode
    _lib.check(_lib.lib.rvc_set_pair_arithmetic(prev))


def x3p_check_count(L):
    """Waits of the pipelined / persistent kernels whose compile-time vmcnt exceeded the exact run-time count since the last call: a number on a
    -DRVC_X3P_CHECK -DRVC_EXPERIMENTS build (RVC_HIP_LIB=.../variants/librvc_hip_x3pcheck.so), -1 on every other build."""
    return L.lib.rvc_debug_x3p_check() if L.has_experiments else -1


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    """max |a-b| / max |b| : the 1e-3 'rel-tol fp32' gate of BASELINE.json is applied on this scale-normalised error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="session")
def noise_tape():
    import torch

    class Tape:
        def __init__(self, seed):
            self.g = torch.Generator().manual_seed(int(seed))

        def __call__(self, shape):
            return torch.randn(tuple(shape), generator=self.g)

    return Tape


def golden_clip(g):
    """The input clip of a full-size golden: regenerated from synthetic.synth_audio (seconds, seed) and checked against the stored SHA-256."""
    import hashlib
    from comfy_rvc_amd import synthetic as S
    audio = S.synth_audio(float(g["audio_seconds"]), seed=int(g["audio_seed"]))
    return audio


def check_clip_digest(audio, g):
    import hashlib
    d = np.frombuffer(hashlib.sha256(np.ascontiguousarray(audio).tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(d, g["audio_sha256"]), "synthetic.synth_audio no longer reproduces the clip this golden was generated from"


def parity_stats(out_i16, ref_i16, lsb=33):
    """int16 waveform vs the reference's: share of samples within `lsb`, the largest and the 99.99th-percentile deviation."""
    d = np.abs(np.asarray(out_i16, dtype=np.int32) - np.asarray(ref_i16, dtype=np.int32))
    return {"n": int(d.size), "within": float(np.mean(d <= lsb)), "max": int(d.max()), "p9999": float(np.percentile(d, 99.99)),
            "mean": float(d.mean())}


def record_parity(name, stats):
    """Keeps the measured parity figures of the full-size cases (gpurun_out/ travels back from the GPU box; copied to profiles/)."""
    import json
    path = os.path.join(ROOT, "gpurun_out", "fullsize_parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.isfile(path):
            with open(path) as f:
                data = json.load(f)
        data[name] = stats
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass


def f0_frames_ok(f0, f0_ref, salience_ref, salience_err, rtol=1e-3):
    """RMVPE's f0 passes through an arg-max over 360 salience bins (lib/rmvpe.py:661-685): a frame may legitimately land on another bin ONLY where
    the reference's own two best bins are closer than the numerical noise between the two implementations.  Returns (n_differing, n_unexplained):
    frames outside rtol, and those among them whose reference salience has a clear winner (top-2 gap > 4 x the measured salience error) - the
    second number must be 0.  No percentile: every frame is either equal or an explained tie."""
    import numpy as np
    f0, f0_ref = np.asarray(f0, dtype=np.float64), np.asarray(f0_ref, dtype=np.float64)
    bad = ~np.isclose(f0, f0_ref, rtol=rtol, atol=1e-6)
    n_bad = int(bad.sum())
    if n_bad == 0:
        return 0, 0
    sal = np.asarray(salience_ref)[: f0_ref.shape[0]]
    top2 = np.sort(sal[bad], axis=1)[:, -2:]
    gap = top2[:, 1] - top2[:, 0]
    # a frame that sits at the voicing threshold (max salience ~ 0.03) may also switch between 0 Hz and voiced
    near_thr = np.abs(top2[:, 1] - 0.03) <= 4 * salience_err
    unexplained = int(((gap > 4 * salience_err) & ~near_thr).sum())
    return n_bad, unexplained
