"""Averages the counters of tools/pmc_conv.sh over the dispatches of the convolution kernel (the longest-running kernel family of the run)."""
import csv, glob, os, sys
from collections import defaultdict
pre = sys.argv[1]
tot = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(pre + "*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        pats = [x for x in os.environ.get("PMC_KERNELS", "conv_x3,conv_mfma_kernel").split(",") if x]      # kernel name substrings
        if not any(x in k for x in pats): continue
        a = tot[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in tot.items():
    print(k)
    for c, (n, v) in cs.items():
        print(f"  {c:42s} {v / n:16.1f}   (n={n})")
