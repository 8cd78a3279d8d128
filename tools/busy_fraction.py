"""GPU busy fraction from a rocprofv3 kernel trace (CSV): union of the kernel intervals / wall, over the longest stretch of the trace without a gap of more than GAP ms
(the benchmark's steady state), plus the mean number of kernels in flight.   python tools/busy_fraction.py kernel_trace.csv [GAP_ms=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# split into stretches at idle gaps > gap
stretches, cur, cur_end = [], [iv[0]], iv[0][1]
for s, e in iv[1:]:
    if s - cur_end > gap:
        stretches.append(cur); cur = []
    cur.append((s, e)); cur_end = max(cur_end, e)
stretches.append(cur)
best = max(stretches, key=lambda st: max(e for _, e in st) - st[0][0])
t0, t1 = best[0][0], max(e for _, e in best)
busy, end, summed = 0, t0, 0
for s, e in best:
    summed += e - s
    if e > end:
        busy += e - max(s, end); end = e
print(f"stretch {(t1 - t0) / 1e6:.1f} ms, {len(best)} kernels: busy {busy / (t1 - t0):.3f} of wall, summed kernel time / wall = {summed / (t1 - t0):.2f} kernels in flight on average")
