#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: python profiles/gap_analysis.py <t_kernel_trace.csv> [anchor substring]
Prints, per (previous kernel -> next kernel) pair, the mean gap in us, and the busy / wall time between the first and last anchor kernel."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "patchnn"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda k: k[0])
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
lo, hi = idx[len(idx) // 3], idx[-1]          # steady state: skip the first third
seg = ks[lo:hi]
busy = sum(e - s for s, e, _ in seg)
wall = seg[-1][1] - seg[0][0]
n = sum(1 for k in seg if anchor in k[2])
print(f"{n} iterations: wall {wall / n / 1e3:.1f} us / iteration, kernels busy {busy / n / 1e3:.1f} us, idle {(wall - busy) / n / 1e3:.1f} us")
gaps = defaultdict(list)
for a, b in zip(seg, seg[1:]):
    gaps[(a[2][:60], b[2][:60])].append(max(0, b[0] - a[1]))
for (a, b), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{sum(g) / n / 1e3:8.1f} us/iter  n={len(g):4d} mean {sum(g) / len(g) / 1e3:7.1f} us   {a}  ->  {b}")
