import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
from comfy_rvc_amd.lib.rmvpe import RMVPE
hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
rm = RMVPE(S.rmvpe_state_dict(0))
a = torch.from_numpy(S.synth_audio(32.0, seed=1)).cuda()
side = torch.cuda.Stream()
def run(name, fn, stream=None):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if stream is not None:
            with torch.cuda.stream(stream): r = fn()
        else: r = fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:32s} enqueue {1e3*(t1-t0):6.2f} ms   total {1e3*(t2-t0):6.2f} ms")
run("hubert main stream", lambda: hub.extract_features(a[None], version="v2", channel_major=True))
run("hubert side stream", lambda: hub.extract_features(a[None], version="v2", channel_major=True), side)
run("rmvpe main stream", lambda: rm.infer(a))
