"""Worker of tests/test_hip_rccl.py: ONE rank, backend "nccl" (= RCCL on ROCm), on a real GPU.  Pushes the buffers of the N > 1 path - the
int16 waveforms as bytes, the length vector, the float64 timing scalar - through the collectives bench.py and parallel.py use (all_gather,
gather, all_reduce(MAX, float64), barrier) and writes what came back to argv[1]."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from comfy_rvc_amd import parallel as P
    bound = P.bind_rank_to_numa(0, 1)                       # before the first HIP call, as bench.py does
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        rng = np.random.default_rng(0)
        wav = rng.integers(-32768, 32767, size=1199200 * 3 + 1, dtype=np.int16)       # three 30 s clips at 40 kHz, odd length
        got = P.gather_waveforms(wav, "cuda:0", force_collective=True)
        assert len(got) == 1 and got[0].dtype == np.int16 and np.array_equal(got[0], wav)
        dev = P.gather_waveforms(torch.from_numpy(wav).cuda(), "cuda:0", to_host=False, force_collective=True)   # device in, device out
        assert dev[0].is_cuda and dev[0].dtype == torch.int16 and np.array_equal(dev[0].cpu().numpy(), wav)
        empty = P.gather_waveforms(np.zeros(0, np.int16), "cuda:0", force_collective=True)
        assert empty[0].shape == (0,)
        t = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([wav.shape[0]], dtype=torch.int64, device="cuda:0")
        lens = [torch.zeros_like(n)]
        dist.all_gather(lens, n)
        dist.barrier()
        torch.cuda.synchronize()
        with open(sys.argv[1], "w") as f:
            json.dump({"ok": True, "max": float(t.item()), "len": int(lens[0].item()), "backend": dist.get_backend(),
                       "bound_cpus": sorted(bound) if bound else None, "affinity": sorted(os.sched_getaffinity(0))}, f)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
