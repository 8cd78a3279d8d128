"""GPU: the collectives of the N > 1 path on RCCL itself.  A 1-GPU box cannot hold two RCCL ranks (one rank per device), so the process
group has ONE rank - what is exercised is everything except the wire: `init_process_group("nccl")`, communicator creation, all_gather /
gather of the uint8 view of the int16 waveforms, all_reduce(MAX) of the float64 step time, barrier, and bench.py's own N = 1 flow with
the collectives forced on (`--force-collective`).  The N = 2 control flow is covered on CPU over gloo (tests/test_parallel_gloo.py) and, with REAL
conversions, by two gloo ranks sharing device 0 (last test): two processes with their own HIP contexts, split CPU masks, a ragged gather - what is
left untested on a 1-GPU box is the xGMI wire itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RVC_BENCH_BACKEND"):
        e.pop(k, None)
    return e


def test_rccl_collectives_of_the_gather_path_world1(tmp_path):
    out = str(tmp_path / "rccl.json")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py"), out], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.load(open(out))
    assert r["ok"] and r["backend"] == "nccl" and r["max"] == 1.25 and r["len"] == 1199200 * 3 + 1
    if r["bound_cpus"] is not None:                       # bound next to the GPU: the mask is exactly what was asked for
        assert r["affinity"] == r["bound_cpus"]


def test_bench_n1_with_forced_rccl_collectives():
    """bench.py --gpus 1 --force-collective: the rank initialises "nccl" with world_size 1 and every step's waveforms go through the gather."""
    e = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--seconds", "4", "--clips", "3",
                        "--force-collective", "--no-cpu-baseline", "--no-roofline"], env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["backend"] == "nccl" and line["nccl_ranks"] == 1 and line["value"] > 0
    assert line["config"]["gathers_per_step"] == 1


def test_two_ranks_share_device_0_over_gloo_with_real_conversions():
    """bench.py --gpus 2 with RVC_BENCH_BACKEND=gloo: the parent starts two rank processes that both use device 0 - each with its own HIP context, model
    replicas (three lanes) and half of the device's CPUs - which convert real clips of rank-dependent length (--ragged) and hand every step's int16
    waveforms to the padded gather; max-over-ranks timing and the single JSON line come from rank 0.  The N > 1 path minus the RCCL wire."""
    e = dict(_env(), RVC_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--seconds", "4", "--clips", "3", "--ragged",
                        "--no-cpu-baseline", "--no-roofline"], env=e, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                                       # rank 0 prints, rank 1 is silent
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo" and line["nccl_ranks"] is None and line["value"] > 0
    assert line["config"]["gathers_per_step"] == 1 and line["config"]["clips_per_step"] == 6 and line["self_launched"] is True
    assert line["data"] == "synthetic"
