"""Durations of the sf_gemm256 launches of tools/g256_intercept.py in a rocprofv3 rocpd database, in launch order
(21 launches per K: 256, 768, 1536, 3072)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r[0] for r in db.execute("select end-start from kernels where name like '%sf_gemm256%' order by start")]
n = len(rows) // 4
for i, K in enumerate((256, 768, 1536, 3072)):
    seg = sorted(rows[i * n + 1:(i + 1) * n])
    print(f"K={K}: median {seg[len(seg)//2]/1e3:.1f} us  min {seg[0]/1e3:.1f}")
