"""Per-class table of the split-resident GEMM launches of one clip (bench.py's event-timed pass):  RVC_PROF_CSV=f.csv python bench.py ... ; python tools/x3s_classes.py f.csv [kernel ...]"""
import csv, sys
kernels = sys.argv[2:] or ["conv_x3s_kernel"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["kernel"] in kernels]
cls = {}
for r in rows:
    k = (r["kernel"], r["tile"], r["Ci"], r["Co"], r["k"], r["Tout"], r["Wd"], r["ksplit"], r["workgroups"])
    c = cls.setdefault(k, [0, 0.0, 0.0]); c[0] += 1; c[1] += float(r["us"]); c[2] += float(r["alg_gflop"])
tot = 0.0
for k, c in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    tot += c[1]
    if c[1] > 40: print(" ".join(k), f"{c[0]} x {c[1] / c[0]:.1f} us {c[2] / c[1] * 1e3:.0f} TF sum {c[1] / 1e3:.2f} ms")
print(f"total {tot / 1e3:.2f} ms over {len(rows)} launches")
