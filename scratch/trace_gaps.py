"""Kernel-trace csv -> per-step GPU busy time, kernel count and idle gaps (launch-bound shapes)."""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps = intervals between consecutive projection_fwd launches
starts = [i for i, r in enumerate(rows) if "projection_fwd_kernel" in r[2]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    steps.append((wall, busy, len(seg)))
steps = steps[len(steps) // 2:]  # the later (timed) half
n = len(steps)
print(f"{n} steps: wall {sum(s[0] for s in steps)/n/1e3:.1f} us, GPU busy {sum(s[1] for s in steps)/n/1e3:.1f} us, "
      f"{sum(s[2] for s in steps)/n:.1f} kernels per step")
a, b = starts[-2], starts[-1]
prev_end = None
for s, e, name in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"  gap {gap:7.1f} us  run {(e - s)/1e3:7.1f} us  {name[:70]}")
    prev_end = e
