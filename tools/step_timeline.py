#!/usr/bin/env python3
"""One train step as a timeline: kernel, queue, start offset, duration, gap to the previous kernel's end -- from a rocprofv3 --kernel-trace CSV.
Usage: step_timeline.py <dir with *kernel_trace.csv> [anchor-kernel-substring (default prep_kernel)] [which occurrence from the end (default 2)]"""
import csv, glob, sys, os
d = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else "prep_kernel"; back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = None
print(f"step of {i1 - i0} kernels, {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us from {anchor} to {anchor}")
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
    prev_end = e if prev_end is None else max(prev_end, e)
