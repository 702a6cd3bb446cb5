"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`)
as the per-kernel table `--stats` prints: calls, total / average / min / max duration, share.

    python profiles/summarize.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{n:6d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}  {name[:140]}")


if __name__ == "__main__":
    main(sys.argv[1])
