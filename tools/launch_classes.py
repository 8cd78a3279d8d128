"""Per-launch-class table of the convolution kernels from bench.py's event-timed pass (RVC_PROF_CSV=<file> python bench.py ...), optionally
joined with per-dispatch HBM counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the SAME command, CSV output).

    python tools/launch_classes.py gpurun_out/r2_launches.csv [fetch_pmc.csv write_pmc.csv] > profiles/r2_launch_classes.md

A class = (kernel, tile, Ci, Co, k, dilation, stride, Tout, width, fused pair, split-K).  Columns: launches per clip, mean us, algorithmic
TFLOP/s and its fraction of the three peaks (fp32 MFMA 157.3, bf16x3 833.3 = 2500 / 3, dense bf16 2500), algorithmic MB and GB/s
(fraction of 8 TB/s), and - when counters are given - measured HBM MB per launch and the ratio to the algorithmic bytes.  The PMC
join is by launch order within a kernel family: both runs execute the same deterministic launch sequence.
"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def pmc_per_dispatch(path, counter, scale):
    """kernel family -> list of bytes per dispatch in dispatch order"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0 * scale))
    rows.sort()
    out = defaultdict(list)
    for _, name, v in rows:
        m = re.search(r"rvc::(\w+)", name)
        out[m.group(1) if m else name].append(v)
    return out


def main(argv):
    launches = list(csv.DictReader(open(argv[0])))
    fetch = write = None
    if len(argv) >= 3:
        fetch = pmc_per_dispatch(argv[1], "FETCH_SIZE", 2.0)      # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM)
        write = pmc_per_dispatch(argv[2], "WRITE_SIZE", 1.0)
    seq = defaultdict(int)
    classes = OrderedDict()
    for r in launches:
        key = (r["kernel"], r["tile"], r["Ci"], r["Co"], r["k"], r["dil"], r["stride"], r["Tout"], r["Wd"], r["fused_pair"], r["ksplit"])
        c = classes.setdefault(key, {"n": 0, "us": 0.0, "gf": 0.0, "mb": 0.0, "wg": r["workgroups"], "hbm": 0.0, "hbm_n": 0})
        c["n"] += 1; c["us"] += float(r["us"]); c["gf"] += float(r["alg_gflop"]); c["mb"] += float(r["alg_mbytes"])
        if fetch is not None:
            i = seq[r["kernel"]]; seq[r["kernel"]] += 1
            fl, wl = fetch.get(r["kernel"], []), write.get(r["kernel"], [])
            # the profiled command may run several clips: the launch sequence repeats, take the LAST repetition (steady state)
            per = len([x for x in launches if x["kernel"] == r["kernel"]])
            if len(fl) >= per and len(wl) >= per:
                c["hbm"] += (fl[len(fl) - per + i] + wl[len(wl) - per + i]) / 1e6; c["hbm_n"] += 1
    print("| kernel | tile | Ci | Co | k | dil | stride | Tout | W | pair | splitK | WGs | n/clip | us | sum ms | TFLOP/s | of 157.3 | of 833 | of 2500 | alg MB | GB/s | of 8 TB/s | HBM MB (PMC) | PMC/alg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot = defaultdict(float)
    for key, c in sorted(classes.items(), key=lambda kv: -kv[1]["us"]):
        us = c["us"] / c["n"]; tf = c["gf"] / c["us"] * 1e3 if c["us"] else 0; gbs = c["mb"] / c["us"] * 1e3 if c["us"] else 0
        hbm = c["hbm"] / c["hbm_n"] if c["hbm_n"] else None
        alg = c["mb"] / c["n"]
        print("| " + " | ".join(list(key[:11]) + [c["wg"], str(c["n"]), f"{us:.1f}", f"{c['us'] / 1e3:.2f}", f"{tf:.1f}", f"{tf / 157.3:.3f}", f"{tf / 833.3:.3f}",
                                                  f"{tf / 2500:.3f}", f"{alg:.1f}", f"{gbs:.0f}", f"{gbs / 8000:.3f}",
                                                  "-" if hbm is None else f"{hbm:.1f}", "-" if hbm is None else f"{hbm / alg:.2f}"]) + " |")
        tot[key[0]] += c["us"]
    print()
    for k, v in tot.items():
        print(f"sum {k}: {v / 1e3:.2f} ms per clip")


if __name__ == "__main__":
    main(sys.argv[1:])
