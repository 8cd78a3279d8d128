"""Turns a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --stats`) into a compact per-kernel CSV for profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    if "distribution_elementwise_grid_stride_kernel" in name:
        return "at::native::normal_kernel (torch.randn on device)"
    return re.sub(r"\(.*$", "", name)[:110]


def main(db, out, passes):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write(f"# source: rocprofv3 --kernel-trace --stats ; durations in microseconds ; {passes} pipeline passes in the trace\n")
        f.write("kernel,calls,total_us,avg_us,percent,us_per_pass\n")
        for n, c, t, a, p in rows:
            f.write(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{p:.2f},{t / passes:.1f}\n")
        f.write(f"# total kernel time per pass: {sum(r[2] for r in rows) / passes / 1000:.2f} ms\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
