"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into per-kernel HBM traffic.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv \
        profiles/r1_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is used as reported (uncalibrated).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(rvc::\w+)", name)
    return m.group(1) if m else re.sub(r"\(.*$", "", name)[:60]


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = agg[family(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
        fb = fe.get(k, [0, 0.0])[1] * 1024.0 * 2.0          # KiB -> bytes, gfx950 half-count correction
        wb = wr.get(k, [0, 0.0])[1] * 1024.0
        res[k] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1),
                  "hbm_bytes_per_launch": (fb + wb) / max(n, 1), "hbm_bytes_total": fb + wb}
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-roofline (2 clips per trace)",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B); WRITE_SIZE as reported",
           "kernels": res}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_total"])[:12]:
        print(f"{k:40s} launches {v['launches']:5d}  fetch/launch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write/launch {v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
