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
        # the same through two conversion lanes per rank (threads); lane functions see the GLOBAL clip index
        lanes = P.ClipLanes([lambda c, i: (c * 3).astype(np.int16)[: c.shape[0] - rank] + 0 * i for _ in range(2)])
        res_l = P.convert_clips(clips, lanes, device="cpu")
        assert (res_l is None) == (res is None) and (res is None or all(np.array_equal(a, b) for a, b in zip(res, res_l)))
        seen = []
        P.convert_clips(clips, P.ClipLanes([lambda c, i: (seen.append(i), np.zeros(1, np.int16))[1]]), device="cpu")
        assert sorted(seen) == list(range(rank, 5, world))
        single = P.gather_waveforms(np.arange(10 + rank, dtype=np.int16), "cpu")
        # tensors in, tensors out (bench.py: the gathered copy stays where the collective delivered it; every rank keeps its own clips)
        import torch as _t
        kept = P.gather_waveforms(_t.arange(10 + rank, dtype=_t.int16), "cpu", to_host=False)
        assert (kept is None) == (rank != 0)
        if rank == 0:
            assert all(isinstance(k, _t.Tensor) and k.dtype == _t.int16 for k in kept) and [k.tolist() for k in kept] == [s_.tolist() for s_ in single]
        # training-prep feature dump: files are sharded rank::world with no collective (stub network: host logic only)
        import torch
        from scipy.io import wavfile
        from comfy_rvc_amd.preprocessing_utils import FeatureInput
        d = os.path.dirname(sys.argv[1])

        class Stub:
            def extract_features(self, version, source, **kw):
                return torch.full((1, source.shape[1] // 320, 768), float(rank))
        if rank == 0:
            for i in range(5):
                wavfile.write(os.path.join(d, f"c{i}.wav"), 16000, np.zeros(3200 * (i + 1), dtype=np.int16))
        dist.barrier()
        paths = [(os.path.join(d, f"c{i}.wav"), os.path.join(d, f"a{i}"), os.path.join(d, f"b{i}"), os.path.join(d, f"f{i}")) for i in range(5)]
        fi = FeatureInput(Stub(), "rmvpe", None, device="cpu", version="v2", if_f0=False)
        n_done = fi.go(paths)
        assert n_done == len(range(rank, 5, world))
        dist.barrier()
        if rank == 0:
            owners = [int(np.load(os.path.join(d, f"f{i}.npy"))[0, 0]) for i in range(5)]
            shapes = [list(np.load(os.path.join(d, f"f{i}.npy")).shape) for i in range(5)]
            with open(sys.argv[1], "w") as f:
                json.dump({"res": [r.tolist() for r in res], "single": [s.tolist() for s in single], "owners": owners, "shapes": shapes}, f)
        else:
            assert res is None and single is None
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
