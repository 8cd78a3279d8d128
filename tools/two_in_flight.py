"""Experiment: W lanes (host thread + model replicas + torch stream each) converting 30 s clips concurrently on one GPU.
   mode "threads": fixed K clips per raw thread; mode "pool": parallel.ClipLanes."""
import os, sys, threading, time
sys.path.insert(0, '.')
import torch
from comfy_rvc_amd import _lib, synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.parallel import ClipLanes
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single

dev = "cuda:0"
torch.cuda.set_device(0)
cfg = Config(device=dev)
params = dict(sid=0, f0_up_key=0, f0_method="rmvpe", index_rate=0.0, rms_mix_rate=0.25, protect=0.33, resample_sr=0)
mode = sys.argv[1] if len(sys.argv) > 1 else "threads"
same_audio = len(sys.argv) > 2 and sys.argv[2] == "same"


def make_worker(i):
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG, device=dev)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg, device=dev)
    vc = VC(40000, cfg)
    vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0), device=dev)
    vc.noise_on_device = True
    audio = S.synth_audio(30.0, seed=100 if same_audio else 100 + i)
    stream = torch.cuda.Stream(dev)

    def one(clip=None, idx=0):
        out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio if clip is None else clip, 16000), config=cfg, **params)
        assert out is not None
        return out[0]

    def run(k):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            return [one() for _ in range(k)]
    return run, one, audio


for W in (1, 2):
    workers = [make_worker(i) for i in range(W)]
    K = 8
    if mode == "threads":
        for w in workers:
            w[0](2)                                   # warm-up, sequential, on the worker's stream
        torch.cuda.synchronize()
        res = [None] * W
        def tgt(i):
            res[i] = workers[i][0](K)
        th = [threading.Thread(target=tgt, args=(i,)) for i in range(W)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        secs = sum(len(o) for r in res for o in r) / 40000.0
    else:
        pool = ClipLanes([w[1] for w in workers], device=dev)
        clips = [workers[0][2]] * (W * K)
        if mode == "pool_warm_main":
            for w in workers:
                w[1]()
        pool.map(clips[: 3 * W])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pool.map(clips)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        secs = sum(len(o) for o in out) / 40000.0
    print(f"{mode} workers {W}: {secs / dt:7.1f} xRT   {dt / (W * K) * 1e3:6.2f} ms per clip", flush=True)
    del workers
