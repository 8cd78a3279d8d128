"""Worker of tests/test_parallel_gloo.py::test_c4_shape_world8: BASELINE.json configs[3]'s sharding (64 clips, 8 per rank, 8 ranks) with ragged
stub outputs - convert_clips through three lanes per rank, and bench.py's per-step gather of a rank's concatenated clips."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def out_len(i):
    return 1439 - (i * 37) % 211          # a miniature of 1 439 040 samples, clip-dependent tail


def main():
    import torch
    import torch.distributed as dist
    from comfy_rvc_amd import parallel as P
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_clips = 64
        clips = [np.full(16, i, dtype=np.float32) for i in range(n_clips)]
        mine = P.shard_indices(n_clips)
        assert mine == list(range(rank, n_clips, world)) and len(mine) == 8
        seen = []

        def lane(c, i):
            seen.append(i)
            assert int(c[0]) == i                                     # the lane function sees the GLOBAL clip index of the clip it was handed
            return np.full(out_len(i), (i * 257 + rank) % 32768, dtype=np.int16)
        res = P.convert_clips(clips, P.ClipLanes([lane, lane, lane]), device="cpu")
        assert sorted(seen) == mine
        # bench.py's exchange: the rank's 8 clips of a step as ONE int16 vector, one gather, rank 0 keeps tensors
        step = torch.cat([torch.full((out_len(i),), i, dtype=torch.int16) for i in mine])
        got = P.gather_waveforms(step, "cpu", to_host=False, force_collective=True)
        if rank == 0:
            assert len(res) == n_clips
            ok = all(r.dtype == np.int16 and r.shape == (out_len(i),) and np.all(r == (i * 257 + i % world) % 32768) for i, r in enumerate(res))
            lens = [int(g.numel()) for g in got]
            heads = [[int(g[0]), int(g[-1])] for g in got]
            with open(sys.argv[1], "w") as f:
                json.dump({"ok": bool(ok), "lens": lens, "heads": heads}, f)
        else:
            assert res is None and got is None
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
