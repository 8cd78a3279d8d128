#!/usr/bin/env python3
"""Headline benchmark: end-to-end 40k_v2 voice conversion throughput in audio-seconds per wall-second (xRT).

    python bench.py --gpus N --steps K --warmup W        # N > 1 and no WORLD_SIZE in the environment: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = every rank converts CLIPS (default 12; 8 for --variant 48k_v2 = BASELINE.json configs[3]) synthetic 30 s / 16 kHz
clips end to end through `vc_single` (host float32 array in, int16 host array out: zero-phase high-pass, RMVPE pitch, HuBERT
features, SynthesizerTrnMs768NSFsid at 40 kHz), LANES (default 3) of them in flight concurrently on its GPU (parallel.ClipLanes:
one host thread, stream set and model replica per lane - a batch-1 clip cannot fill 256 CUs in its narrow stages), and the int16
waveforms of the step are gathered on rank 0 over RCCL (the path's only exchange step).  `--lanes 1` is the strictly sequential
one-clip-at-a-time rate.  Clips are independent, so the work shards clip-per-GPU with no data-path collective besides that gather:
weak scaling.  Weights are procedural (comfy-rvc_amd/synthetic.py) - no checkpoint is reachable offline - and compute is fp32
(fp32 MFMA, or the bf16x3 split with fp32 accumulation).  Rank 0 prints ONE JSON line (metric/value/roofline/cpu_baseline ...).

`--dry-run` replaces the conversion with a stub (no HIP call at all) and exists to exercise the launcher, the process group, the
per-step gather and the timing protocol on a box without GPUs (tests/test_parallel_gloo.py); its line says so in `data`.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per busy stream (see comfy-rvc_amd/__init__.py); before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

CLIP_SECONDS = 30.0
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same table, "Peak BF16/FP16 MFMA" dense
HBM_PEAK_BPS = 8.0e12              # same guide, HBM3E 8 TB/s (6.3 TB/s achievable)
PMC_TRAFFIC_FILE = os.path.join("profiles", "r6zz_pmc_traffic.json")      # (fallback only: --no-traffic, N > 1 or no rocprofv3)
MAX_LINE_BYTES = 6144              # the driver parses ONE JSON line from stdout; round 4's 17.9 KB line was not parsed


class Delivered:
    """One converted clip: the host array vc_single returned and the device tensor the same samples still occupy (for the gather)."""
    __slots__ = ("host", "dev")

    def __init__(self, host, dev):
        self.host, self.dev = host, dev


def launch_ranks(n):
    """Parent of a self-launched N-rank run: starts one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and waits.
    Runs before anything in this process has touched HIP (importing torch does not); the children are ordinary subprocesses, nothing
    is exec'ed.  Rank 0's stdout is this process's stdout, so the JSON line appears exactly once."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    one_gpu = os.environ.get("RVC_BENCH_BACKEND", "nccl") != "nccl"      # gloo debugging of the control flow: every rank on device 0
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RVC_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:      # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _set_affinity_all_threads(mask):
    """sched_setaffinity for every thread of this process (the call with pid 0 only moves the calling thread; OpenMP / torch pool threads that
    already exist keep the mask they were created under)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for t in tids:
        try:
            os.sched_setaffinity(t, mask)
        except OSError:
            pass


def cpu_baseline(config=None, seconds=CLIP_SECONDS, seed=100, budget_s=40.0):
    """The CPU oracle (validated restatement of the reference) timed on this box's host cores as BASELINE.md section 4 describes, on the
    SAME clip the GPU line converts (same length, same seed: rank 0's clip): 1 s warm-up clip + up to 3 timed runs inside budget_s of CPU
    time (a 30 s clip takes ~14 s on 32 threads, so normally two), torch.set_num_threads(k) with k stated (bounded by the rank's CPU
    affinity), per-stage seconds.  n_runs says how many were timed; the median (upper of two) is reported."""
    import numpy as np
    import torch
    from comfy_rvc_amd import synthetic as S
    from oracle import nets, pipeline as opl
    ncpu = os.cpu_count() or 1
    navail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else ncpu      # (the caller has restored the process-wide mask)
    k = max(1, min(navail, 32))                # torch-CPU conv/GEMM on these sizes stops scaling well before 32 threads
    prev = torch.get_num_threads()
    torch.set_num_threads(k)
    config = config or S.CONFIG_40K_V2
    sds = (S.hubert_state_dict(0), S.rmvpe_state_dict(0), S.synth_state_dict(config, "v2", 0))
    stage = {}

    def timed(name, fn):
        def w(*a, **kw):
            t = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                stage[name] = stage.get(name, 0.0) + time.perf_counter() - t
        return w
    orig = (nets.hubert_extract_features, nets.rmvpe_infer_from_audio, nets.synth_infer)
    nets.hubert_extract_features, nets.rmvpe_infer_from_audio, nets.synth_infer = (timed("hubert", orig[0]), timed("rmvpe", orig[1]),
                                                                                      timed("synthesizer", orig[2]))

    def run(secs, seed):
        g = torch.Generator().manual_seed(0)
        audio = S.synth_audio(secs, seed=seed)
        stage.clear()
        t0 = time.perf_counter()
        out = opl.pipeline(sds[0], sds[1], sds[2], config, "v2", audio, noise_fn=lambda shp: torch.randn(shp, generator=g),
                           n_hubert_layers=12)   # the reference runs all 12 HuBERT layers (and discards the last)
        dt = time.perf_counter() - t0
        st = dict(stage)
        st["host_dsp"] = max(dt - sum(st.values()), 0.0)
        return out.shape[0] / float(config[-1]), dt, st
    try:
        run(1.0, 2)                               # warm-up: thread pool, lazy constants, allocator
        runs = []
        t_all = time.perf_counter()
        for _ in range(3):
            runs.append(run(seconds, seed))
            if time.perf_counter() - t_all + runs[-1][1] > budget_s:
                break
    finally:
        nets.hubert_extract_features, nets.rmvpe_infer_from_audio, nets.synth_infer = orig
        torch.set_num_threads(prev)
    runs.sort(key=lambda r: r[1])
    delivered, dt, st = runs[len(runs) // 2]
    return {"value": round(delivered / dt, 4), "unit": "audio-sec/wall-sec", "cores": int(k), "kind": "port", "cpu": cpu_model(),
            "host_logical_cpus": int(ncpu), "cpus_in_affinity_mask": int(navail), "n_runs": len(runs), "wall_s_median": round(dt, 2),
            "stage_seconds": {n: round(v, 2) for n, v in sorted(st.items())},
            "sample": f"1 x {seconds:g} s clip (the GPU line's clip, seed {seed}), 1 s warm-up + median of {len(runs)} run(s), oracle.pipeline (torch-CPU fp32 "
                      f"restatement pinned to reference goldens), {k} threads"}


def cpu_baseline_rmvpe(seconds=60.0, seed=100, budget_s=30.0):
    """oracle.nets.rmvpe_infer_from_audio (the validated CPU restatement of lib/rmvpe.RMVPE) on the GPU line's clip: 1 s warm-up, up to 3 runs."""
    import torch
    from comfy_rvc_amd import synthetic as S
    from oracle import nets
    navail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    k = max(1, min(navail, 32))
    prev = torch.get_num_threads()
    torch.set_num_threads(k)
    sd = S.rmvpe_state_dict(0)
    try:
        nets.rmvpe_infer_from_audio(sd, S.synth_audio(1.0, seed=2))
        audio = S.synth_audio(seconds, seed=seed)
        runs, t_all = [], time.perf_counter()
        for _ in range(3):
            t0 = time.perf_counter()
            nets.rmvpe_infer_from_audio(sd, audio)
            runs.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all + runs[-1] > budget_s:
                break
    finally:
        torch.set_num_threads(prev)
    runs.sort()
    dt = runs[len(runs) // 2]
    return {"value": round(seconds / dt, 4), "unit": "audio-sec/wall-sec", "cores": int(k), "kind": "port", "cpu": cpu_model(), "n_runs": len(runs),
            "wall_s_median": round(dt, 2),
            "sample": f"1 x {seconds:g} s clip (the GPU line's clip, seed {seed}), 1 s warm-up + median of {len(runs)} run(s), oracle.nets.rmvpe_infer_from_audio "
                      f"(torch-CPU fp32 restatement pinned to reference goldens), {k} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=0.0, help="clip length (default 30 s; 60 s for --variant rmvpe_60s)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("RVC_BENCH_LANES", "0")),
                    help="clips in flight per GPU (default 3; 8 for --variant rmvpe_60s, whose GRU scan holds 16 CUs for 60 %% of a clip: profiles/r5_rmvpe_lanes.txt)")
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU per step (default 12; 8 for --variant 48k_v2 as BASELINE.json configs[3] states; 24 for rmvpe_60s; 3 for uvr_48k_v2)")
    ap.add_argument("--variant", choices=["40k_v2", "48k_v2", "uvr_48k_v2", "rmvpe_60s"], default="40k_v2",
                    help="40k_v2 = the configuration the metric is quoted on (BASELINE.json configs[2]); 48k_v2 = configs[3]'s model; uvr_48k_v2 = "
                         "configs[4]'s chain: MDX23C vocal split of a stereo 44.1 kHz clip (overlap 8) -> VC of the vocal stem with the 48k_v2 model; "
                         "rmvpe_60s = configs[1]: lib/rmvpe.RMVPE pitch extraction alone on 60 s clips (host audio in -> float64 f0 out)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes (roofline.traffic then comes from the committed file)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # the child of a --pmc pass: convert the clips, print nothing else
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: stub conversion; checks launcher / process group / gather / timing")
    ap.add_argument("--force-collective", action="store_true", help="N = 1: still create the process group and run every step's gather (exercises RCCL on a 1-GPU box)")
    ap.add_argument("--ragged", action="store_true", help="ranks deliver different lengths (the gather's padded path): with --dry-run stub outputs of clip- and rank-dependent lengths, on the GPU rank r converts clips 0.25 r s longer (tests only: value is computed from rank 0's clip)")
    ap.add_argument("--no-bind", action="store_true", help="do not pin the rank to the CPUs of its GPU's NUMA node")
    args = ap.parse_args()
    if args.seconds <= 0:
        args.seconds = 60.0 if args.variant == "rmvpe_60s" else CLIP_SECONDS
    if args.lanes <= 0:
        args.lanes = 8 if args.variant == "rmvpe_60s" else 3

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # nothing in this process has initialised HIP

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    # CPU affinity first: nothing in this process has touched HIP yet (importing torch / counting devices does not), so the runtime's helper
    # threads, the lane threads and the side-stream callbacks all inherit the mask.  One NUMA node's worth of CPUs per rank.
    bound = None
    orig_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if not args.no_bind and not args.dry_run:
        from comfy_rvc_amd.parallel import bind_rank_to_numa
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        # gloo debug mode (RVC_BENCH_BACKEND != nccl: every rank runs on device 0 with LOCAL_RANK 0): the device's CPUs are split by RANK, not
        # handed to all ranks at once (advisor, round 3)
        shared_dev = world > 1 and os.environ.get("RVC_BENCH_BACKEND", "nccl") != "nccl"
        bound = bind_rank_to_numa(local_rank, local_world, slot=rank if shared_dev else None, nslots=world if shared_dev else None)

    import numpy as np
    import torch
    dist = None
    backend = os.environ.get("RVC_BENCH_BACKEND", "nccl")
    use_gpu = not args.dry_run
    if use_gpu:
        torch.cuda.set_device(local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:      # the N > 1 control flow on a 1-GPU (or no-GPU, with --dry-run) box: RVC_BENCH_BACKEND=gloo
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if backend == "nccl" else "cpu"      # where the collectives' tensors live

    from comfy_rvc_amd import synthetic as S
    from comfy_rvc_amd.parallel import ClipLanes, gather_waveforms

    n_lanes = max(1, args.lanes)
    chain = args.variant == "uvr_48k_v2"
    pitch_only = args.variant == "rmvpe_60s"
    n_clips = args.clips if args.clips > 0 else (8 if args.variant == "48k_v2" else (3 if chain else (24 if pitch_only else 12)))
    SYN_CFG = S.CONFIG_40K_V2 if args.variant == "40k_v2" else S.CONFIG_48K_V2
    if chain:      # stereo 44.1 kHz "song": the voice-like synthetic signal on both channels plus different noise beds
        audio = np.stack([S.synth_audio(args.seconds, seed=100 + rank, sr=44100), S.synth_audio(args.seconds, seed=300 + rank, sr=44100)])
    else:
        audio = S.synth_audio(args.seconds + (0.25 * rank if (args.ragged and not args.dry_run) else 0.0), seed=100 + rank)
    params = dict(sid=0, f0_up_key=0, f0_method="rmvpe", index_rate=0.0, rms_mix_rate=0.25, protect=0.33, resample_sr=0)
    vc = None

    if use_gpu:
        from comfy_rvc_amd import _lib
        from comfy_rvc_amd.config import Config
        from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
        from comfy_rvc_amd.lib.rmvpe import RMVPE
        from comfy_rvc_amd.vc_infer_pipeline import VC, get_vc, vc_single
        cfg = Config(device=dev)
        # procedural weights: generated ONCE per rank and shared by its lanes (each lane still owns its device replica: handles carry workspaces)
        rmvpe_sd = S.rmvpe_state_dict(0)
        hub_sd, syn_ckpt = (None, None) if pitch_only else (S.hubert_state_dict(0), S.synth_checkpoint(SYN_CFG, "v2", 0))
        mdx_sd = None
        if chain:
            from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG as _MC
            mdx_sd = S.mdx23c_state_dict(_MC, 0)

        def make_pitch_lane():
            rm = RMVPE(rmvpe_sd, device=dev)

            def convert(clip, i=0):      # RMVPE.infer_from_audio (reference lib/rmvpe.py:614-659): 16 kHz audio -> f0 per 10 ms frame, float64, on the host
                f0 = rm.infer_from_audio(clip, thred=0.03)
                assert f0.dtype == np.float64 and f0.shape[0] == clip.shape[0] // 160 + 1
                return f0
            return convert, None

        convert_mdx = []
        def make_lane():   # (one model set per lane; the MDX23C networks are kept for the single-clip measurement below)
            if pitch_only:
                return make_pitch_lane()
            hub = HubertModelWithFinalProj(hub_sd, S.HUBERT_CONFIG, device=dev)
            vcd = get_vc(syn_ckpt, config=cfg, device=dev)
            lvc = VC(SYN_CFG[-1], cfg)
            lvc.model_rmvpe = RMVPE(rmvpe_sd, device=dev)
            lvc.noise_on_device = True          # the reference draws its noise with the compute device's generator as well

            mdx = None
            if chain:
                from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG
                from comfy_rvc_amd.lib.karafan.inference import demix_mdxv3
                from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
                mdx = TFC_TDF_net(MDX23C_CONFIG, device=dev)
                mdx.load_state_dict(mdx_sd)
                convert_mdx.append(mdx)

            def convert(clip, i=0):
                sr_in = 16000
                if chain:      # UVR5Node.split -> vocal stem -> RVCNode.convert (reference custom_nodes/uvr.py:56-100, rvc_nodes.py:186-206)
                    clip, sr_in = demix_mdxv3(clip, mdx, dev, MDX23C_CONFIG, MDX23C_CONFIG["inference"]["num_overlap"])["Vocals"], 44100
                out = vc_single(cpt=vcd["cpt"], net_g=vcd["net_g"], vc=lvc, hubert_model=hub, input_audio=(clip, sr_in), config=cfg, **params)
                assert out is not None, "vc_single failed"
                if collective and backend == "nccl":      # the gather takes the int16 result where rvc_postprocess left it (HBM): no host re-upload
                    return Delivered(out[0], lvc.last_i16)
                return out[0]
            return convert, lvc
        lanes = [make_lane() for _ in range(n_lanes)]
        vc = lanes[0][1]
        pool = ClipLanes([fn for fn, _ in lanes], device=dev)
    else:
        n_out = int((2 * ((int(args.seconds * 16000) + 32000 - 400) // 320 + 1)) * (SYN_CFG[-1] // 100) - 2 * SYN_CFG[-1])

        def stub(clip, i=0):                    # --dry-run: the documented output length (minus a clip-dependent tail with --ragged), no compute
            time.sleep(0.002)
            cut = (7 * rank + i) % 13 * 48 if args.ragged else 0
            return np.zeros(clip.shape[0] // 160 + 1, dtype=np.float64) if pitch_only else np.full(n_out - cut, rank, dtype=np.int16)
        convert_mdx = []
        lanes = [(stub, None) for _ in range(n_lanes)]
        pool = ClipLanes([fn for fn, _ in lanes], device=None)

    def step():                            # single clip on lane 0, caller's thread and stream (warm-up, roofline pass)
        return lanes[0][0](audio)

    def run_steps(k):
        """k steps = k * CLIPS clips of this rank through the lanes (a free lane pulls the next clip while the previous step is gathered);
        the waveforms of a step are handed to ONE gather (the path's only exchange) in clip order."""
        wav, batch = None, []
        for res in pool.imap([audio] * (k * n_clips)):
            wav = res.host if isinstance(res, Delivered) else res
            batch.append(res)
            if len(batch) == n_clips:
                if collective:   # every rank already holds ITS clips on the host (vc_single delivered them); rank 0 keeps the gathered copy in HBM
                    if isinstance(batch[0], Delivered):
                        cur = torch.cuda.current_stream()
                        for b in batch:      # produced (and completed) on the lanes' streams, consumed on this one: keep the caching allocator from re-using them early
                            b.dev.record_stream(cur)
                        step_wav = torch.cat([b.dev for b in batch])                 # device-resident int16, concatenated on the device
                    else:
                        step_wav = np.concatenate(batch).view(np.int16)            # gloo / dry run / f0 vectors (which travel as their bytes)
                    gather_waveforms(step_wav, coll_dev, to_host=False, force_collective=True)
                batch = []
        return wav

    def sync():
        if dist is not None:
            dist.barrier()
        if use_gpu:
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
    delivered = args.seconds if pitch_only else wav.shape[0] / float(SYN_CFG[-1])       # audio seconds of one converted clip (pitch: of one analysed clip)
    value = delivered * n_clips * world * args.steps / dt

    # informational: one clip alone on the GPU (what a single ComfyUI graph execution sees), lane 0, a few untimed-for-the-headline passes
    sync()
    alone = []
    for _ in range(1 if args.pmc_child else 9):
        t1 = time.perf_counter()
        step()
        if use_gpu:
            torch.cuda.synchronize()
        alone.append(time.perf_counter() - t1)
    alone_ms = sorted(alone)[len(alone) // 2] * 1e3          # median of 9
    alone3_ms = None
    if use_gpu and convert_mdx:            # the UVR node's setting for a single conversion: a clip's chunks over three streams (UVR5Node: load_mdx23c)
        convert_mdx[0].set_streams(3)
        a3 = []
        for _ in range(6):
            t1 = time.perf_counter(); step(); torch.cuda.synchronize(); a3.append(time.perf_counter() - t1)
        alone3_ms = sorted(a3[1:])[2] * 1e3          # median of 5 after one pass that allocates the extra arenas
        if os.environ.get("RVC_BENCH_DEBUG"): print("alone", [round(x * 1e3) for x in alone], "alone3", [round(x * 1e3) for x in a3], file=sys.stderr)
        convert_mdx[0].set_streams(1)
    sync()

    roofline, detail = None, None
    if rank == 0 and use_gpu and not args.no_roofline and not args.pmc_child:
        live = None
        if world == 1 and not args.no_traffic and os.environ.get("RVC_BENCH_TRAFFIC", "1") != "0":
            live = pmc_traffic_live(args.variant, args.seconds)
        roofline, detail = roofline_pass(_lib, vc, step, torch, live)
        if roofline is not None and pitch_only:
            roofline["note"] = "conv kernels only; rvc::gru_scan_kernel (serial, latency-bound) is the largest single kernel of this variant"
    cpu = None
    if rank == 0 and world == 1 and use_gpu and not args.no_cpu_baseline and not args.pmc_child:
        # the CPU baseline is "the reference's path on this box's host cores" (BASELINE.md section 4), not on the one NUMA node the rank was pinned
        # to for the GPU run: every thread of the process (torch's intra-op pool included) gets the original mask back for this leg
        if bound is not None and orig_affinity is not None:
            _set_affinity_all_threads(orig_affinity)
        cpu = None if chain else (cpu_baseline_rmvpe(seconds=args.seconds) if pitch_only else cpu_baseline(config=SYN_CFG, seconds=args.seconds))      # (the CPU oracle of the separation net at full size takes minutes per chunk)

    if rank == 0:
        cfg_idx = {"40k_v2": 2, "48k_v2": 3, "uvr_48k_v2": 4, "rmvpe_60s": 1}[args.variant]
        line = {
            "metric": ("audio-sec/wall-sec (xRT), RMVPE pitch extraction alone" if pitch_only else f"audio-sec/wall-sec (xRT), {args.variant} end-to-end VC"),
            "value": round(value, 2), "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if use_gpu else "dry-run (stub conversion, NO GPU work: launcher / collective check only)",
            "dtype_note": ("fp32 tensors, fp32 accumulation everywhere; matrix products as a 3-term bf16 hi/lo split (fp32-equivalent), the generator's ResBlock pairs as "
                           "fp16x2 (weight = one fp16 term, activation fp16 hi+lo; full-size goldens <= 33 LSB) unless RVC_H2=0; rest fp32 MFMA") if _pair_arith(use_gpu) else
                          "fp32 tensors; convolutions as a 3-term bf16 hi/lo split on the MFMA units with fp32 accumulation (fp32-equivalent), rest fp32 MFMA",
            "ranks": world, "backend": (backend if collective else None), "nccl_ranks": (world if (collective and backend == "nccl") else None),
            "cpu_affinity": (f"{len(bound)} CPUs of the GPU's NUMA node" if bound else "unbound (topology unknown or --no-bind)"),
            "self_launched": bool(os.environ.get("RVC_BENCH_SELF_LAUNCHED")), "timed_region_s": round(dt, 3),
            "config": {"workload": (f"RMVPE pitch extraction alone, {args.seconds:g} s 16 kHz clips, {n_clips} per GPU per step ({n_lanes} in flight), host audio in -> "
                                    f"float64 f0 out (BASELINE.json configs[1])") if pitch_only else
                                   ("MDX23C vocal split (stereo 44.1 kHz, overlap 8) -> " if chain else "") +
                                   f"Full VC {args.variant} (HuBERT -> RMVPE -> SynthesizerTrnMs768NSFsid), {args.seconds:g} s {'44.1 kHz stereo' if chain else '16 kHz'} clips, {n_clips} per GPU "
                                   f"per step ({n_lanes} in flight), vc_single host array in -> int16 host array out (BASELINE.json configs[{cfg_idx}])",
                       "clips_per_step": world * n_clips, "clips_per_gpu_per_step": n_clips, "clips_in_flight_per_gpu": n_lanes,
                       "gathers_per_step": 1 if collective else 0,
                       "audio_seconds_delivered_per_clip": round(delivered, 3),
                       "one_clip_alone_ms": round(alone_ms, 2), "one_clip_alone_xrt": round(delivered / alone_ms * 1e3, 1),
                       **({"one_clip_alone_ms_3_chunk_streams": round(alone3_ms, 2)} if alone3_ms is not None else {}),
                       "weights": "procedural (comfy-rvc_amd/synthetic.py)", "noise": "device generator",
                       "parallelism": f"clip-per-GPU x{world} ({n_lanes} lanes each), one RCCL gather of the step's int16 waveforms"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        text = json.dumps(line, separators=(",", ":"))
        if len(text) >= MAX_LINE_BYTES:          # degrade instead of losing the whole result: drop the optional parts, largest first (the detail file keeps them)
            for drop in (("roofline", "others"), ("cpu_baseline", "sample"), ("config", "parallelism"), ("dtype_note",), ("cpu_affinity",)):
                obj = line
                for k in drop[:-1]:
                    obj = obj.get(k) if isinstance(obj, dict) else None
                if isinstance(obj, dict) and drop[-1] in obj:
                    obj.pop(drop[-1]); line["truncated"] = True
                text = json.dumps(line, separators=(",", ":"))
                if len(text) < MAX_LINE_BYTES:
                    break
            print(f"bench line exceeded {MAX_LINE_BYTES} bytes: optional fields dropped (see gpurun_out/bench_detail.json)", file=sys.stderr)
        if detail is not None:
            detail["line"] = line
            write_detail(detail)
        print(text, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


KERNEL_DESC = {
    "conv_x3q_kernel": "rvc::conv_x3q_kernel<AM,AN,KT,R,XSPLIT,YSPLIT,RADD,H2> - PERSISTENT software-pipelined split-MFMA implicit-GEMM Conv1d (the split-resident ResBlock pairs of "
                       "the three wide generator stages): resident workgroups walk over their tiles in one stream of (tile, chunk, tap) units, residual added block "
                       "by block inside the tile; H2 (default): 2 v_mfma_f32_32x32x16_f16 per product block (weight one fp16 term, activation fp16 hi + lo), else 3 "
                       "v_mfma_f32_32x32x16_bf16; fp32 accumulate",
    "conv_x3p_kernel": "rvc::conv_x3p_kernel<AM,AN,KT,XSPLIT,YSPLIT,S2> - software-pipelined bf16x3 implicit-GEMM Conv1d (generator ResBlocks / up-samplers, "
                       "HuBERT stride-2 layers): 3 v_mfma_f32_32x32x16_bf16 per fp32 product block, fp32 accumulate",
    "conv_x3pf_kernel": "rvc::conv_x3pf_kernel<KT,WM> - fused ResBlock pair (32- / 64-channel generator stages), bf16x3",
    "conv_rbh_kernel": "rvc::conv_rbh_kernel<KT,ACC> - persistent fused ResBlock pair of the 32-channel stage, fp16x2, both weight sets resident in LDS (sequences too short for conv_rb3_kernel)",
    "conv_rb3_kernel": "rvc::conv_rb3_kernel<CH,KT,ACC,RESIDENT> - a whole ResBlock1 (three pairs) of the 32-channel stage / the 64-channel stage's 3-tap ResBlock per launch, fp16x2: "
                       "x read once, residual stream in registers, six convolutions over an LDS-resident fp16 hi / lo image",
    "conv_x3g_kernel": "rvc::conv_x3g_kernel - pipelined bf16x3 GEMM (k = 1 projections, taps / 3x3 modes on short sequences)",
    "conv_x3s_kernel": "rvc::conv_x3s_kernel - bf16x3 GEMM on split-resident operands (both tiles by LDS-DMA, in-kernel split-K)",
    "conv_x3_kernel": "rvc::conv_x3_kernel<WM,WN,AM,AN> - staged bf16x3 kernel (2-D 3x3, other strides)",
    "conv_mfma_kernel": "rvc::conv_mfma_kernel<WM,WN,AM,AN,MODE> - fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM",
}


def _pair_arith(use_gpu):
    try:
        from comfy_rvc_amd import _lib
        return bool(use_gpu) and _lib.lib.rvc_get_pair_arithmetic() == 1
    except Exception:   # noqa: BLE001 - dry runs without the library
        return False


def pmc_traffic_live(variant, seconds, timeout_s=75):
    """HBM bytes per launch of every kernel, measured NOW: two child processes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters
    only - no trace domains - as /opt/skills/guides/MI355X_MICROARCH.md's HBM section prescribes) around `python3 bench.py --pmc-child` (one lane, one clip
    at a time), summarised like tools/pmc_traffic.py: KiB -> bytes, FETCH_SIZE x 2 on gfx950 (128-byte requests tallied at 64), WRITE_SIZE as reported.
    The children are ordinary subprocesses started after this process's own measurement is over (nothing is exec'ed; the GPU is otherwise idle).  Returns
    ({kernel family: {hbm_bytes_per_launch, launches}}, note) or None when rocprofv3 is missing, times out or writes nothing (the committed file is used then)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        return None
    t0 = time.perf_counter()
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rvc_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--lanes", "1",
                   "--steps", "2", "--warmup", "1", "--clips", "1", "--variant", variant, "--seconds", str(seconds), "--no-cpu-baseline", "--no-roofline", "--no-traffic", "--no-bind"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RVC_PROF_CSV"):
                env.pop(k, None)
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") != counter:
                        continue
                    m = re.match(r"(rvc::\w+)", re.sub(r"^void ", "", row["Kernel_Name"]))
                    if not m:
                        continue
                    e = per.setdefault(m.group(1), {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
                    e[counter][0] += 1
                    e[counter][1] += float(row["Counter_Value"])
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, e in per.items():
        n = max(e["FETCH_SIZE"][0], e["WRITE_SIZE"][0])
        if n:
            out[k] = {"launches": n, "hbm_bytes_per_launch": (e["FETCH_SIZE"][1] * 1024.0 * 2.0 + e["WRITE_SIZE"][1] * 1024.0) / n}
    if not out:
        return None
    return out, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE(x2) + WRITE_SIZE, 2 child passes of 4 clips each (the first cold), "
                 f"{time.perf_counter() - t0:.0f} s")


def roofline_pass(_lib, vc, step, torch, live=None):
    """One extra, untimed-for-the-headline clip with every conv-kernel launch bracketed by HIP events on the stream it is launched on
    (front-ends serialised for that pass) and tagged with its algorithmic FLOPs / HBM bytes.  The per-launch table is grouped by KERNEL.
    Returns (roofline, detail): `roofline` is the COMPACT object of the kernel with the most time per clip (the same kernel tops
    profiles/*_kernel_stats_lanes1.csv), priced against its own bound (arithmetic intensity of ITS launches against the ridge) with ITS PMC
    traffic - it goes on the JSON line, which the driver parses and which therefore stays small; `detail` (every kernel, launch classes, the
    long notes) goes to gpurun_out/bench_detail.json and stderr."""
    import csv
    import tempfile
    _lib.check(_lib.lib.rvc_prof_enable(1))
    if vc is not None:
        vc.overlap_streams = False      # serialise the two front-ends so that event-bracketed kernel times are not inflated by overlap
    step()
    torch.cuda.synchronize()
    dump = os.environ.get("RVC_PROF_CSV")
    tmp = None
    if not dump:
        tmp = tempfile.NamedTemporaryFile(suffix=".csv", delete=False)
        tmp.close()
        dump = tmp.name
    _lib.check(_lib.lib.rvc_prof_dump_csv(dump.encode()))      # per-launch table (shape, tile, us, FLOPs, algorithmic bytes); kept when RVC_PROF_CSV names it
    _lib.check(_lib.lib.rvc_prof_enable(0))
    if vc is not None:
        vc.overlap_streams = True
    with open(dump) as f:
        rows = list(csv.DictReader(f))
    if tmp is not None:
        os.unlink(tmp.name)
    traffic, traffic_src = {}, None
    if live is not None:
        traffic, traffic_src = live
    else:
        try:
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as f:
                doc = json.load(f)
            traffic, traffic_src = doc["kernels"], f"file {PMC_TRAFFIC_FILE} (tree {str(doc.get('commit', '?')).split(' ')[0]}), not measured in this run"
        except (OSError, ValueError, KeyError):
            pass
    return roofline_from_rows(rows, traffic, traffic_src)


def roofline_from_rows(rows, traffic, traffic_src):
    """Formatter shared by the live pass and tests/test_parallel_gloo.py (which feeds it a synthetic full-size launch table and checks the
    size of the resulting line)."""
    x3_peak = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)
    kernels = {}
    for r in rows:
        kernels.setdefault(r["kernel"], []).append(r)

    def entry(name, rs):
        us = sum(float(r["us"]) for r in rs)
        gf = sum(float(r["alg_gflop"]) for r in rs)
        mb = sum(float(r["alg_mbytes"]) for r in rs)
        n = len(rs)
        x3 = name != "conv_mfma_kernel"
        # matrix instructions per algorithmic product of THIS kernel's launches, FLOP-weighted: 3 (bf16x3), 2 (fp16x2: the ResBlock pairs on conv_x3q_kernel
        # since round 6) - the peak a launch is priced against is the dense 16-bit MFMA peak divided by that
        terms = (sum(float(r.get("mfma_per_product", 3) or 3) * float(r["alg_gflop"]) for r in rs) / gf) if (x3 and gf > 0) else 3.0
        peak_tf = round(BF16_MFMA_PEAK_TFLOPS / terms, 1) if x3 else FP32_MFMA_PEAK_TFLOPS
        ridge = peak_tf * 1e12 / HBM_PEAK_BPS
        mfma = mb <= 0 or (gf * 1e9) / (mb * 1e6) >= ridge
        tf = gf / us * 1e3 if us > 0 else 0.0                    # 1 GFLOP / us = 1e15 FLOP/s = 1000 TFLOP/s
        gbs = mb / us * 1e3 if us > 0 else 0.0                   # 1 MB / us = 1e12 B/s = 1000 GB/s
        e = {"bound": "mfma" if mfma else "hbm"}
        if mfma:
            e.update({"achieved": round(tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4)})
        else:
            e.update({"achieved": round(gbs, 1), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_BPS / 1e9), 4)})
        tr = traffic.get("rvc::" + name)
        if x3 and abs(terms - 3.0) > 1e-6:
            e.update({"mfma_per_product": round(terms, 3), "frac_if_priced_as_bf16x3": round(tf / x3_peak, 4) if mfma else None})
        e.update({"traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
                  "traffic_src": None if tr is None else traffic_src,
                  "kernel": "rvc::" + name,
                  "launches_per_clip": n, "avg_launch_us": round(us / n, 2),
                  "algorithmic_gflop_per_launch": round(gf / n, 3), "algorithmic_mbytes_per_launch": round(mb / n, 2),
                  "kernel_ms_per_clip": round(us / 1e3, 3), "algorithmic_tflop_per_clip": round(gf / 1e3, 3),
                  "algorithmic_tflops": round(tf, 2), "algorithmic_gbps": round(gbs, 1),
                  "intensity_flop_per_byte": round(gf * 1e3 / mb, 1) if mb > 0 else None, "ridge_flop_per_byte": round(ridge, 1)})
        # detail only: the launch classes of this kernel (shape -> launches, us, TFLOP/s), largest first, and the prose
        cls = {}
        for r in rs:
            k = (r["tile"], r["Ci"], r["Co"], r["k"], r["dil"], r["stride"], r["Tout"], r["Wd"], r["fused_pair"], r["ksplit"])
            c = cls.setdefault(k, [0, 0.0, 0.0, 0.0])
            c[0] += 1; c[1] += float(r["us"]); c[2] += float(r["alg_gflop"]); c[3] += float(r["alg_mbytes"])
        top = sorted(cls.items(), key=lambda kv: -kv[1][1])[:8]
        d = dict(e)
        d.update({"description": KERNEL_DESC.get(name, "rvc::" + name),
                  "peak_note": (f"dense 16-bit MFMA peak 2500 TFLOP/s / {terms:.3g} MFMAs per algorithmic product (3 = bf16x3, 2 = fp16x2)" if x3 else "fp32 MFMA peak") + " at the nominal 2.4 GHz clock",
                  "traffic_note": "HBM bytes per launch of THIS kernel (mean over its launches): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes "
                                  "(tools/pmc_traffic.py); " + ("no figure" if tr is None else str(traffic_src)),
                  "frac_of_peaks": {"fp32_mfma_157.3": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "bf16x3_833.3": round(tf / x3_peak, 4),
                                    "bf16_dense_2500": round(tf / BF16_MFMA_PEAK_TFLOPS, 4), "hbm_8TBps": round(gbs / (HBM_PEAK_BPS / 1e9), 4)},
                  "top_classes": [{"tile": k[0], "Ci": int(k[1]), "Co": int(k[2]), "k": int(k[3]), "dil": int(k[4]), "stride": int(k[5]), "Tout": int(k[6]), "Wd": int(k[7]),
                                   "pair": int(k[8]), "ksplit": int(k[9]), "launches": c[0], "us": round(c[1] / c[0], 1),
                                   "tflops": round(c[2] / c[1] / 1e-3, 1) if c[1] > 0 else 0.0, "alg_gbps": round(c[3] / c[1] * 1e3, 0) if c[1] > 0 else 0.0}
                                  for k, c in top]})
        return e, d
    ents = sorted((entry(n, rs) for n, rs in kernels.items()), key=lambda ed: -ed[0]["kernel_ms_per_clip"])
    if not ents:
        return None, None
    roofline = ents[0][0]
    # a one-line table of the rest (name, ms per clip, fraction of its own bound) so that the line still shows where the other time goes
    roofline["others"] = [[e["kernel"].replace("rvc::", "").replace("_kernel", ""), e["kernel_ms_per_clip"], e["bound"], e["frac"]] for e, _ in ents[1:8]]
    # the generator's Conv1d kernels taken together (BASELINE.json's "fraction of the conv1d roofline"): their algorithmic FLOPs over their summed launch time, priced against
    # the dense 16-bit MFMA peak divided by the FLOP-weighted matrix instructions per product of their launches (2 fp16x2, 3 bf16x3)
    fam = [(n, rs) for n, rs in kernels.items() if n in ("conv_x3q_kernel", "conv_rb3_kernel", "conv_rbh_kernel", "conv_x3pf_kernel", "conv_x3p_kernel")]
    if fam:
        f_us = sum(float(r["us"]) for _, rs in fam for r in rs)
        f_gf = sum(float(r["alg_gflop"]) for _, rs in fam for r in rs)
        f_terms = sum(float(r.get("mfma_per_product", 3) or 3) * float(r["alg_gflop"]) for _, rs in fam for r in rs) / max(f_gf, 1e-9)
        f_peak = BF16_MFMA_PEAK_TFLOPS / f_terms
        roofline["conv1d_family"] = {"kernels": sorted(n.replace("_kernel", "") for n, _ in fam), "launches_per_clip": sum(len(rs) for _, rs in fam), "ms_per_clip": round(f_us / 1e3, 3),
                                     "algorithmic_tflop_per_clip": round(f_gf / 1e3, 3), "tflops": round(f_gf / f_us * 1e3, 1) if f_us > 0 else 0.0,
                                     "mfma_per_product": round(f_terms, 3), "peak": round(f_peak, 1), "frac": round(f_gf / f_us * 1e3 / f_peak, 4) if f_us > 0 else 0.0,
                                     "frac_if_priced_as_bf16x3": round(f_gf / f_us * 1e3 / x3_peak, 4) if f_us > 0 else 0.0}
    roofline["all_conv_kernels_ms_per_clip"] = round(sum(e["kernel_ms_per_clip"] for e, _ in ents), 3)
    roofline["all_conv_kernels_tflops"] = round(sum(e["algorithmic_tflop_per_clip"] for e, _ in ents) / max(roofline["all_conv_kernels_ms_per_clip"], 1e-9) * 1e3, 1)
    return roofline, {"kernels": [d for _, d in ents]}


def write_detail(detail):
    """gpurun_out/bench_detail.json (merged back from the GPU box) + one stderr line; never stdout."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as f:
            json.dump(detail, f, indent=1)
    except OSError:
        pass
    print("bench detail: " + json.dumps(detail), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
