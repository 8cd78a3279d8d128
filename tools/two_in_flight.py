"""Experiment: W host threads, each with its own model replicas and torch stream, converting 30 s clips concurrently on one GPU."""
import os, sys, threading, time
sys.path.insert(0, '.')
import torch
from comfy_rvc_amd import _lib, synthetic as S
from comfy_rvc_amd.config import Config
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single

dev = "cuda:0"
torch.cuda.set_device(0)
cfg = Config(device=dev)
params = dict(sid=0, f0_up_key=0, f0_method="rmvpe", index_rate=0.0, rms_mix_rate=0.25, protect=0.33, resample_sr=0)


def make_worker(i):
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG, device=dev)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0), config=cfg, device=dev)
    vc = VC(40000, cfg)
    vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0), device=dev)
    vc.noise_on_device = True
    audio = S.synth_audio(30.0, seed=100 + i)
    stream = torch.cuda.Stream(dev)

    def run(k):
        torch.cuda.set_device(0)
        outs = []
        with torch.cuda.stream(stream):
            for _ in range(k):
                out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(audio, 16000), config=cfg, **params)
                assert out is not None
                outs.append(out[0])
        return outs
    return run


for W in (1, 2, 3):
    workers = [make_worker(i) for i in range(W)]
    for w in workers:
        w(2)                                   # warm-up, sequential
    torch.cuda.synchronize()
    K = 8
    res = [None] * W
    def tgt(i):
        res[i] = workers[i](K)
    th = [threading.Thread(target=tgt, args=(i,)) for i in range(W)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    secs = sum(len(o) for r in res for o in r) / 40000.0
    print(f"workers {W}: {secs / dt:7.1f} xRT   {dt / (W * K) * 1e3:6.2f} ms per clip", flush=True)
    del workers
