"""Multi-GPU sharding of independent clips: one process per GPU, clip i -> rank i mod N, one gather at the end.

The reference is batch-1 / single-device throughout (vc_infer_pipeline.py:48, lib/rmvpe.py:616); clips share no state, so
the only exchange step is collecting the int16 waveforms on rank 0 (<= 2.9 MB per 30 s clip at 48 kHz: latency-bound, one
collective).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_items, rank=None, world=None):
    """Indices of the clips this rank converts (round-robin: clip i -> rank i mod N)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, n_items, world))


def gather_waveforms(wav, device="cpu", dst=0):
    """Collects one variable-length int16 waveform per rank on `dst`.  Returns the list there and None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [np.asarray(wav)]
    world, rank = dist.get_world_size(), dist.get_rank()
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
    return [o.view(torch.int16)[:ln].cpu().numpy() for o, ln in zip(out, lens)]


def convert_clips(clips, convert_fn, device="cpu", dst=0):
    """Round-robin shards `clips` over the ranks, converts the local ones with convert_fn(clip) -> int16 array, and
    returns on `dst` the outputs in the original clip order (None elsewhere)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_indices(len(clips), rank, world)
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
