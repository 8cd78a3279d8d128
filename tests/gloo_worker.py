"""Worker process of tests/test_parallel_gloo.py: RANK / WORLD_SIZE / MASTER_* come from the environment, result goes to argv[1]."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from comfy_rvc_amd import parallel as P
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips = [np.full(100 + 37 * i, i, dtype=np.float32) for i in range(5)]       # ragged lengths, 5 clips over 2 ranks
        assert P.shard_indices(5) == list(range(rank, 5, world))
        res = P.convert_clips(clips, lambda c: (c * 3).astype(np.int16)[: c.shape[0] - rank], device="cpu")
        single = P.gather_waveforms(np.arange(10 + rank, dtype=np.int16), "cpu")
        if rank == 0:
            with open(sys.argv[1], "w") as f:
                json.dump({"res": [r.tolist() for r in res], "single": [s.tolist() for s in single]}, f)
        else:
            assert res is None and single is None
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
