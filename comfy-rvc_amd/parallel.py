"""Multi-GPU sharding of independent clips: one process per GPU, clip i -> rank i mod N, one gather at the end.

The reference is batch-1 / single-device throughout (vc_infer_pipeline.py:48, lib/rmvpe.py:616); clips share no state, so
the only exchange step is collecting the int16 waveforms on rank 0 (<= 2.9 MB per 30 s clip at 48 kHz: latency-bound, one
collective).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

Inside one GPU the same independence is used a second time (ClipLanes): a batch-1 clip cannot fill 256 CUs during its narrow
stages (text encoder and flow at 100 fps, the transformer layers, the deep U-Net levels, the GRU scan), so several clips are kept in
flight, each on its own host thread, HIP streams and model replica (weights 0.85 GB + as much again in weight images + arenas: ~7.5 GB per
lane; 288 GB of HBM make the replica free).  Measured on MI355X, 30 s clips (round 6, profiles/r6zz_*): ~1580 xRT one at a time, ~2440 xRT with
three lanes; a fourth adds nothing (the chip is at its power limit under the generator's persistent kernel).  RMVPE alone keeps eight in flight.
"""
import glob
import os
import threading

import numpy as np
import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------- CPU affinity of a rank
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(local_rank, sysfs="/sys"):
    """CPUs of the NUMA node the `local_rank`-th GPU hangs off, read from sysfs WITHOUT touching HIP: the KFD topology lists the compute
    nodes in the order the HIP runtime enumerates them (nodes with simd_count > 0 are GPUs); a node's PCI address is domain : location_id
    (bus << 8 | device << 3 | function), and the PCI device directory names its NUMA node's CPUs (`local_cpulist`).  {HIP,ROCR}_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES lists of plain indices are honoured.  Returns None when anything is missing (no KFD, single-node box ...)."""
    try:
        gpus = []
        nodes = sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")), key=lambda d: int(os.path.basename(d)))
        for d in nodes:
            props = {}
            try:
                with open(os.path.join(d, "properties")) as f:
                    for line in f:
                        k, _, v = line.strip().partition(" ")
                        props[k] = v
            except OSError:
                continue                 # a GPU of the machine this container was not given (its node is listed but unreadable): HIP does not see it either
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
                gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis:
                idx = [int(t) for t in vis.split(",") if t.strip().lstrip("-").isdigit()]
                if len(idx) != len([t for t in vis.split(",") if t.strip()]):
                    return None                      # UUID syntax: not resolvable from sysfs alone
                gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
        if not (0 <= local_rank < len(gpus)):
            return None
        with open(os.path.join(sysfs, "bus/pci/devices", gpus[local_rank], "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError, IndexError):
        return None


def bind_rank_to_numa(local_rank, local_world=1, sysfs="/sys", slot=None, nslots=None):
    """Pins the calling process (and the lane / side-stream threads it starts later) to the CPUs next to its GPU.  MUST run before the first
    HIP call of the process: the runtime's own helper threads inherit the mask, and nothing is re-exec'ed (a process that has touched the
    GPU is never replaced; bench.py's ranks are started fresh by its launcher or by torch.distributed.run).  The ranks that share a NUMA
    node split its CPUs evenly so that N ranks x (lanes + side streams + HIP helper threads) do not sit on each other.  Returns the set of
    CPUs bound to, or None when the topology is unknown (nothing is changed then).  slot / nslots: ranks that share one DEVICE (same
    local_rank) split its CPUs by `slot` instead."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    cpus = gpu_numa_cpus(local_rank, sysfs)
    if not cpus:
        return None
    allowed = os.sched_getaffinity(0)
    mine = sorted(cpus & allowed)
    if not mine:
        return None
    sharers = [r for r in range(max(local_world, 1)) if gpu_numa_cpus(r, sysfs) == cpus] or [local_rank]
    if slot is not None and nslots:
        # several ranks on ONE device (the gloo debug mode of bench.py: every rank has LOCAL_RANK 0): split that device's CPUs by rank
        if nslots > 1 and len(mine) >= 2 * nslots:
            per = len(mine) // nslots
            mine = mine[(slot % nslots) * per:(slot % nslots + 1) * per]
    elif len(sharers) > 1 and len(mine) >= 2 * len(sharers):
        k = sharers.index(local_rank) if local_rank in sharers else 0
        per = len(mine) // len(sharers)
        mine = mine[k * per:(k + 1) * per]
    os.sched_setaffinity(0, mine)
    return set(mine)


def shard_indices(n_items, rank=None, world=None):
    """Indices of the clips this rank converts (round-robin: clip i -> rank i mod N)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, n_items, world))


def gather_waveforms(wav, device="cpu", dst=0, to_host=True, force_collective=False):
    """Collects one variable-length int16 waveform per rank on `dst`.  Returns the list there and None elsewhere.

    `wav` is a host array or an int16 tensor already on `device` (no upload then).  to_host=False leaves the gathered waveforms where the
    collective delivered them (int16 tensors on `device`: with RCCL the receive buffers in rank `dst`'s HBM) - every rank has already
    downloaded its OWN clips when it converted them, so rank `dst` need not download everybody's a second time.  force_collective=True
    runs the collectives even in a process group of one (the single-GPU RCCL test)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return [np.asarray(wav.cpu() if isinstance(wav, torch.Tensor) else wav)]
    world, rank = dist.get_world_size(), dist.get_rank()
    if isinstance(wav, torch.Tensor):
        w = wav.to(device=device, dtype=torch.int16).contiguous()
    else:
        w = torch.as_tensor(np.ascontiguousarray(wav), dtype=torch.int16).to(device)
    n = torch.tensor([w.numel()], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    buf = torch.zeros(max(max(lens), 1), dtype=torch.int16, device=device)
    buf[: w.numel()] = w
    raw = buf.view(torch.uint8)                      # bytes on the wire: every backend (RCCL, gloo) carries uint8
    out = [torch.empty_like(raw) for _ in range(world)] if rank == dst else None
    dist.gather(raw, out, dst=dst)
    if rank != dst:
        return None
    if not to_host:
        return [o.view(torch.int16)[:ln] for o, ln in zip(out, lens)]
    return [o.view(torch.int16)[:ln].cpu().numpy() for o, ln in zip(out, lens)]


def convert_clips(clips, convert_fn, device="cpu", dst=0):
    """Round-robin shards `clips` over the ranks, converts the local ones with convert_fn(clip) -> int16 array, and
    returns on `dst` the outputs in the original clip order (None elsewhere).  convert_fn may be a ClipLanes: the rank's clips
    then go through its lanes concurrently, each lane function receiving (clip, global clip index)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_indices(len(clips), rank, world)
    if isinstance(convert_fn, ClipLanes):
        pool = ClipLanes([(lambda c, j, fn=fn: fn(c, mine[j])) for fn in convert_fn.lanes], convert_fn.device)
        outs = pool.map([clips[i] for i in mine])
    else:
        outs = [convert_fn(clips[i]) for i in mine]
    rounds = (len(clips) + world - 1) // world
    result = [None] * len(clips)
    for r in range(rounds):
        local = outs[r] if r < len(outs) else np.zeros(0, dtype=np.int16)
        got = gather_waveforms(local, device, dst)
        if got is not None:
            for src, wv in enumerate(got):
                idx = r * world + src
                if idx < len(clips):
                    result[idx] = wv
    return result if rank == dst else None


class ClipLanes:
    """W conversion lanes on ONE GPU.  `lanes` is a list of callables fn(clip, index) -> int16 array; each must own its models
    (net_g / HuBERT / RMVPE handles carry per-handle workspaces and are not shared between threads).  Every lane runs on its own
    host thread with its own torch stream as the current stream, and pulls the next unconverted clip when it is free.

    Determinism: outputs do not depend on which lane converts a clip as long as fn derives the synthesizer noise from `index`
    (VC.noise_fn); with the global generator the draw order between concurrent clips is undefined."""

    def __init__(self, lanes, device=None):
        self.lanes = list(lanes)
        assert self.lanes, "at least one lane"
        self.device = torch.device(device) if device is not None else None
        self._streams = None
        if self.device is not None and self.device.type == "cuda":
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            self._streams = [torch.cuda.Stream(self.device) for _ in self.lanes]

    def imap(self, clips):
        """Yields the outputs in clip order, each as soon as it (and all earlier ones) is ready."""
        n = len(clips)
        results, done = [None] * n, [threading.Event() for _ in range(n)]
        lock, errors, counter = threading.Lock(), [], iter(range(n))

        def work(k):
            try:
                if self._streams is not None:
                    torch.cuda.set_device(self.device)
                    with torch.cuda.stream(self._streams[k]):
                        loop(k)
                        self._streams[k].synchronize()
                else:
                    loop(k)
            except BaseException as e:              # noqa: BLE001 - anything outside a conversion (device setup ...) fails the whole map
                errors.append(e)
            finally:
                if errors:
                    for d in done:
                        d.set()

        def loop(k):
            while True:
                with lock:
                    i = next(counter, None)
                if i is None or errors:
                    return
                try:
                    results[i] = self.lanes[k](clips[i], i)
                except BaseException as e:          # noqa: BLE001 - re-raised on the consumer side
                    errors.append(e)
                finally:
                    done[i].set()

        threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(min(len(self.lanes), max(n, 1)))]
        for t in threads:
            t.start()
        try:
            for i in range(n):
                done[i].wait()
                if errors:
                    raise errors[0]
                yield results[i]
                results[i] = None
        finally:
            for t in threads:
                t.join()

    def map(self, clips):
        return list(self.imap(clips))
