#!/usr/bin/env python3
"""Kernel sequence of ONE steady-state iteration from a rocprofv3 --kernel-trace CSV: python profiles/iter_sequence.py <t_kernel_trace.csv> <anchor substring> [which]
Prints every kernel between two consecutive anchor kernels (start-relative us, duration us, name) -- what an iteration really launches."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda k: k[0])
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
w = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) * 3 // 4
lo, hi = idx[w], idx[w + 1]
t0 = ks[lo][0]
tot = 0
for s, e, n in ks[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n[:150]}")
    tot += e - s
print(f"iteration: wall {(ks[hi][0] - t0) / 1e3:.1f} us, kernels {tot / 1e3:.1f} us, {hi - lo} launches")
