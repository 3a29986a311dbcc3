"""Per-kernel mean of every counter in rocprofv3 counter_collection csvs under the given dirs."""
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "gs::" not in k: continue
                k = k.split("(")[0].replace("void gs::", "")
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:36s} {tot[k][c]/cnt[k][c]:16.0f}  (n={cnt[k][c]})")
