#!/usr/bin/env python3
"""Headline benchmark: end-to-end 40k_v2 voice conversion throughput in audio-seconds per wall-second (xRT).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = every rank converts LANES (default 3) synthetic 30 s / 16 kHz clips end to end through `vc_single` (host float32
array in, int16 host array out: zero-phase high-pass, RMVPE pitch, HuBERT features, SynthesizerTrnMs768NSFsid at 40 kHz), the
clips of a rank in flight concurrently on its GPU (parallel.ClipLanes: one host thread, stream set and model replica per lane -
a batch-1 clip cannot fill 256 CUs in its narrow stages), and the int16 waveforms are gathered on rank 0 over RCCL (the path's
only exchange step).  `--lanes 1` is the strictly sequential one-clip-at-a-time rate.  Clips are independent, so the work shards
clip-per-GPU with no data-path collective besides that gather: weak scaling.  Weights are procedural
(comfy-rvc_amd/synthetic.py) - no checkpoint is reachable offline - and compute is fp32 (fp32 MFMA, or the bf16x3 split with fp32 accumulation).
Rank 0 prints ONE JSON line (metric/value/roofline/cpu_baseline ...).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per busy stream (see comfy-rvc_amd/__init__.py); before HIP initialises

import numpy as np   # noqa: E402
import torch         # noqa: E402

CLIP_SECONDS = 30.0
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same table, "Peak BF16/FP16 MFMA" dense
HBM_PEAK_BPS = 8.0e12              # same guide, HBM3E 8 TB/s (6.3 TB/s achievable)


def cpu_baseline(seconds=3.0, config=None):
    """The CPU oracle (validated restatement of the reference) timed on this box's host cores on a bounded sample."""
    from comfy_rvc_amd import synthetic as S
    from oracle import pipeline as opl
    threads = torch.get_num_threads()
    audio = S.synth_audio(seconds, seed=1)
    config = config or S.CONFIG_40K_V2
    sds = (S.hubert_state_dict(0), S.rmvpe_state_dict(0), S.synth_state_dict(config, "v2", 0))
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    out = opl.pipeline(sds[0], sds[1], sds[2], config, "v2", audio, noise_fn=lambda shp: torch.randn(shp, generator=g),
                       n_hubert_layers=12)   # the reference runs all 12 HuBERT layers (and discards the last)
    dt = time.perf_counter() - t0
    return {"value": round(out.shape[0] / float(config[-1]) / dt, 4), "unit": "audio-sec/wall-sec", "cores": int(threads), "kind": "port",
            "sample": f"1 x {seconds:g} s clip, same procedural weights, oracle.pipeline (torch-CPU fp32 restatement of the "
                      f"reference, validated against reference goldens), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=CLIP_SECONDS)
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("RVC_BENCH_LANES", "3")), help="clips in flight per GPU")
    ap.add_argument("--variant", choices=["40k_v2", "48k_v2"], default="40k_v2",
                    help="40k_v2 = the configuration the metric is quoted on (BASELINE.json configs[2]); 48k_v2 = configs[3]'s model")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dist = None
    backend = os.environ.get("RVC_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:      # debugging the N > 1 control flow on a 1-GPU box: RVC_BENCH_BACKEND=gloo with LOCAL_RANK=0 on every rank
            dist.init_process_group(backend=backend)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if backend == "nccl" else "cpu"      # where the collectives' tensors live
    torch.cuda.set_device(local_rank)

    from comfy_rvc_amd import _lib, synthetic as S
    from comfy_rvc_amd.config import Config
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.lib.rmvpe import RMVPE
    from comfy_rvc_amd.parallel import gather_waveforms
    from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single

    cfg = Config(device=dev)
    from comfy_rvc_amd.parallel import ClipLanes
    audio = S.synth_audio(args.seconds, seed=100 + rank)
    params = dict(sid=0, f0_up_key=0, f0_method="rmvpe", index_rate=0.0, rms_mix_rate=0.25, protect=0.33, resample_sr=0)
    n_lanes = max(1, args.lanes)
    SYN_CFG = S.CONFIG_40K_V2 if args.variant == "40k_v2" else S.CONFIG_48K_V2

    def make_lane():
        hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG, device=dev)
        vcd = get_vc(S.synth_checkpoint(SYN_CFG, "v2", 0), config=cfg, device=dev)
        vc = VC(SYN_CFG[-1], cfg)
        vc.model_rmvpe = RMVPE(S.rmvpe_state_dict(0), device=dev)
        vc.noise_on_device = True          # the reference draws its noise with the compute device's generator as well

        def convert(clip, i=0):
            out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=vc, hubert_model=hub, input_audio=(clip, 16000), config=cfg, **params)
            assert out is not None, "vc_single failed"
            return out[0]
        return convert, vc
    lanes = [make_lane() for _ in range(n_lanes)]
    vc = lanes[0][1]
    pool = ClipLanes([fn for fn, _ in lanes], device=dev)

    def step():                            # single clip on lane 0, caller's thread and stream (warm-up, roofline pass)
        return lanes[0][0](audio)

    def run_steps(k):
        """k steps = k * LANES clips of this rank through the lanes (a free lane pulls the next clip); the waveforms are handed to the
        gather in clip order as they complete."""
        wav = None
        for wav in pool.imap([audio] * (k * n_lanes)):
            if world > 1:
                gather_waveforms(wav, coll_dev)
        return wav

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for fn, _ in lanes:                    # every lane sizes its workspaces once, sequentially
        fn(audio)
    wav = run_steps(args.warmup) if args.warmup > 0 else step()
    sync()
    t0 = time.perf_counter()
    wav = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    delivered = wav.shape[0] / float(SYN_CFG[-1])                       # audio seconds of one converted clip
    value = delivered * n_lanes * world * args.steps / dt

    # informational: one clip alone on the GPU (what a single ComfyUI graph execution sees), lane 0, a few untimed-for-the-headline passes
    sync()
    t1 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    alone_ms = (time.perf_counter() - t1) / 3 * 1e3
    sync()

    roofline = None
    if rank == 0 and not args.no_roofline:
        # one extra, untimed-for-the-headline pass with every conv-kernel launch bracketed by HIP events on its own stream
        _lib.check(_lib.lib.rvc_prof_enable(1))
        vc.overlap_streams = False      # serialise the two front-ends so that event-bracketed kernel times are not inflated by overlap
        step()
        torch.cuda.synchronize()
        NCFG = 24   # RVC_PROF_CFGS
        ms = (C.c_double * NCFG)(); fl = (C.c_double * NCFG)(); ln = (C.c_int64 * NCFG)()
        ex = (C.c_double * (NCFG * 8))()
        ridge_f32 = FP32_MFMA_PEAK_TFLOPS * 1e12 / HBM_PEAK_BPS
        ridge_x3 = BF16_MFMA_PEAK_TFLOPS / 3.0 * 1e12 / HBM_PEAK_BPS
        _lib.check(_lib.lib.rvc_prof_collect_ex(ex, ridge_f32, ridge_x3))
        _lib.check(_lib.lib.rvc_prof_collect(ms, fl, ln))
        _lib.check(_lib.lib.rvc_prof_enable(0))
        vc.overlap_streams = True
        per_cfg = {_lib.lib.rvc_prof_cfg_name(i).decode(): {"launches": int(ln[i]), "ms": round(ms[i], 3),
                                                            "tflops": round(fl[i] / ms[i] / 1e9, 2) if ms[i] > 0 else 0.0}
                   for i in range(NCFG) if ln[i]}
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            pass

        def family(idx, regime, kernel, desc, peak_tf, note):
            """One roofline entry: the launches of a kernel family in one regime (0: MFMA-bound, 4: HBM-bound by arithmetic intensity)."""
            t = sum(ex[i * 8 + regime] for i in idx); f_ = sum(ex[i * 8 + regime + 1] for i in idx)
            by = sum(ex[i * 8 + regime + 2] for i in idx); l_ = sum(ex[i * 8 + regime + 3] for i in idx)
            if l_ == 0:
                return None
            tr = traffic.get(kernel)
            e = {"bound": "mfma" if regime == 0 else "hbm"}
            if regime == 0:
                ach = f_ / (t * 1e-3) / 1e12
                e.update({"achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4)})
            else:
                ach = by / (t * 1e-3) / 1e9
                e.update({"achieved": round(ach, 1), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": round(ach / (HBM_PEAK_BPS / 1e9), 4)})
            e.update({"traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
                      "traffic_note": None if tr is None else "HBM bytes per launch averaged over ALL launches of this kernel: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + "
                                                              "WRITE_SIZE, separate passes (profiles/r1_pmc_traffic.json, tools/pmc_traffic.py)",
                      "kernel": desc, "peak_note": note, "launches_per_clip": int(l_), "avg_launch_us": round(t * 1e3 / l_, 2),
                      "algorithmic_gflop_per_launch": round(f_ / l_ / 1e9, 3), "algorithmic_mbytes_per_launch": round(by / l_ / 1e6, 2),
                      "kernel_ms_per_clip": round(t, 2), "algorithmic_tflop_per_clip": round(f_ / 1e12, 3)})
            return e
        X3 = "rvc::conv_x3_kernel<WM,WN,AM,AN> (bf16x3 split: 3 v_mfma_f32_32x32x16_bf16 per fp32 product block, fp32 accumulate)"
        F32 = "rvc::conv_mfma_kernel<WM,WN,AM,AN,MODE> (fp32 v_mfma_f32_32x32x2_f32)"
        x3_peak = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)
        split_note = ("launches are split by arithmetic intensity (algorithmic FLOP / algorithmic HBM byte) against the ridge peak FLOP/s / 8 TB/s: "
                      "this entry holds the %s-bound ones")
        fams = [family(range(14, NCFG), 0, "rvc::conv_x3_kernel", X3, x3_peak, "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per algorithmic product; " + split_note % "MFMA"),
                family(range(14, NCFG), 4, "rvc::conv_x3_kernel", X3, x3_peak, "HBM 8 TB/s; " + split_note % "HBM"),
                family(range(0, 14), 0, "rvc::conv_mfma_kernel", F32, FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA peak; " + split_note % "MFMA"),
                family(range(0, 14), 4, "rvc::conv_mfma_kernel", F32, FP32_MFMA_PEAK_TFLOPS, "HBM 8 TB/s; " + split_note % "HBM")]
        fams = [r for r in fams if r]
        fams.sort(key=lambda r: -r["kernel_ms_per_clip"])          # the dominant entry = the one with the most kernel time per clip
        roofline = fams[0]
        roofline["per_tile_config"] = per_cfg
        roofline["other_kernels"] = fams[1:]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(config=SYN_CFG)

    if rank == 0:
        line = {
            "metric": f"audio-sec/wall-sec (xRT), {args.variant} end-to-end VC", "value": round(value, 2), "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 tensors end to end; eligible convolutions multiply on the bf16 matrix cores as a 3-term hi/lo split with fp32 accumulation (error ~1e-5, parity tolerance 1e-3), the rest on the fp32 matrix cores",
            "config": {"workload": f"Full VC {args.variant} (HuBERT -> RMVPE -> SynthesizerTrnMs768NSFsid), {args.seconds:g} s 16 kHz clips, {n_lanes} per GPU "
                                   "per step (in flight concurrently), vc_single host array in -> int16 host array out (BASELINE.json configs[2])",
                       "clips_per_step": world * n_lanes, "clips_in_flight_per_gpu": n_lanes, "audio_seconds_delivered_per_clip": round(delivered, 3),
                       "one_clip_alone_ms": round(alone_ms, 2), "one_clip_alone_xrt": round(delivered / alone_ms * 1e3, 1),
                       "weights": "procedural (comfy-rvc_amd/synthetic.py)", "noise": "device generator",
                       "parallelism": f"clip-per-GPU x{world} ({n_lanes} lanes each), RCCL gather of int16 waveforms"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
