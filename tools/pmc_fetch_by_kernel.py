import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if row.get("Counter_Name") == "FETCH_SIZE":
            acc[row["Kernel_Name"]][0] += float(row["Counter_Value"]); acc[row["Kernel_Name"]][1] += 1
for k, (v, n) in sorted(acc.items()):
    if "gemm" in k: print(f"{k[:70]:70s} n={n:4d} FETCH avg {v/n/1024:8.1f} MB (x2 = {2*v/n/1024:.1f} MB)")
