"""CPU: the rank -> NUMA-node CPU binding of parallel.bind_rank_to_numa against a fabricated sysfs tree (KFD topology + PCI devices)."""
import os

import pytest

from comfy_rvc_amd import parallel as P


def _fake_sysfs(root, gpus):
    """gpus: list of (domain, bus, device, function, cpulist).  Node 0 is a CPU node (simd_count 0), as on a real box."""
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, fn, cpus) in enumerate(gpus, start=1):
        (nodes / str(i)).mkdir()
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {(bus << 8) | (dev << 3) | fn}\ndomain {dom}\n")
        d = root / "bus/pci/devices" / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n")


def test_gpu_numa_cpus_reads_kfd_order_and_visible_devices(tmp_path, monkeypatch):
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    _fake_sysfs(tmp_path, [(0, 0x05, 0, 0, "0-3,8-11"), (0, 0x15, 0, 0, "0-3,8-11"), (1, 0x85, 0, 0, "4-7,12-15")])
    assert P.gpu_numa_cpus(0, str(tmp_path)) == {0, 1, 2, 3, 8, 9, 10, 11}
    assert P.gpu_numa_cpus(2, str(tmp_path)) == {4, 5, 6, 7, 12, 13, 14, 15}
    assert P.gpu_numa_cpus(3, str(tmp_path)) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert P.gpu_numa_cpus(0, str(tmp_path)) == {4, 5, 6, 7, 12, 13, 14, 15}
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")
    assert P.gpu_numa_cpus(0, str(tmp_path)) is None
    assert P.gpu_numa_cpus(0, str(tmp_path / "nothing")) is None            # no KFD at all: unknown, nothing is bound
    # a machine with more GPUs than the container may touch: their nodes are listed but unreadable and do not count
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    os.remove(tmp_path / "class/kfd/kfd/topology/nodes/1/properties")
    os.mkdir(tmp_path / "class/kfd/kfd/topology/nodes/1/properties")       # open() fails with an OSError, like EPERM does
    assert P.gpu_numa_cpus(0, str(tmp_path)) == {0, 1, 2, 3, 8, 9, 10, 11} and P.gpu_numa_cpus(1, str(tmp_path)) == {4, 5, 6, 7, 12, 13, 14, 15}


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="Linux only")
def test_bind_rank_splits_a_shared_node_and_restores(tmp_path, monkeypatch):
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    if len(cpus) < 4:
        pytest.skip("needs 4 CPUs")
    lst = ",".join(str(c) for c in cpus)
    _fake_sysfs(tmp_path, [(0, 5, 0, 0, lst), (0, 6, 0, 0, lst)])            # two GPUs on one node: the ranks split its CPUs
    try:
        a = P.bind_rank_to_numa(0, 2, str(tmp_path))
        assert a == set(cpus[: len(cpus) // 2]) and os.sched_getaffinity(0) == a
        os.sched_setaffinity(0, before)
        b = P.bind_rank_to_numa(1, 2, str(tmp_path))
        assert b == set(cpus[len(cpus) // 2: 2 * (len(cpus) // 2)]) and not (a & b)
        os.sched_setaffinity(0, before)
        assert P.bind_rank_to_numa(0, 1, str(tmp_path / "nothing")) is None and os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)
