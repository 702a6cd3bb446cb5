"""Per-frame timeline from a rocprofv3 rocpd database of tools/stream_trace.py: kernels sorted by start time, frames split
at the patchify kernel; reports per kernel-name mean duration, mean gap BEFORE it, and the busy / idle split of a frame."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
frames, cur = [], []
first = "sf_stream_params_kernel" if any("sf_stream_params_kernel" in r[0] for r in rows) else "sf_patchify"   # first kernel of a frame
for name, s, e in rows:
    if first in name and cur:
        frames.append(cur); cur = []
    cur.append((name, s, e))
frames.append(cur)
frames = [f for f in frames if len(f) > 50][64:]          # skip the first (untimed) repeat
agg = collections.OrderedDict()
busy = idle = span = 0.0
for f in frames:
    prev_end = None
    for name, s, e in f:
        short = name.split("(")[0].replace("void ", "")[:60]
        d = agg.setdefault(short, [0, 0.0, 0.0, []])
        d[0] += 1; d[1] += (e - s)
        if prev_end is not None:
            d[2] += max(0, s - prev_end); idle += max(0, s - prev_end); d[3].append(max(0, s - prev_end))
        busy += e - s
        prev_end = e
    span += f[-1][2] - f[0][1]
n = len(frames)
print(f"{n} frames; per frame: kernels {sum(len(f) for f in frames)/n:.1f}, busy {busy/n/1e3:.1f} us, idle between kernels {idle/n/1e3:.1f} us, span {span/n/1e3:.1f} us")
print(f"{'per_frame':>9} {'dur_us':>8} {'gap_before_us':>13} {'gap_p50/p90':>11} {'total_us/frame':>14}  kernel")
for k, (c, d, g, gl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    gl.sort()
    q = f"{gl[len(gl)//2]/1e3:.2f}/{gl[int(len(gl)*0.9)]/1e3:.2f}" if gl else "-"
    print(f"{c/n:9.1f} {d/c/1e3:8.2f} {g/c/1e3:13.2f} {q:>11} {(d+g)/n/1e3:14.1f}  {k}")
