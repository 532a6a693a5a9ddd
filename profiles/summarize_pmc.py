#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs (one dir per --pmc pass) per kernel: mean counter value per dispatch."""
import csv, glob, os, re, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
pat = sys.argv[2] if len(sys.argv) > 2 else "render_"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        m = re.search(r"::([A-Za-z0-9_]+<[^>]*>|[A-Za-z0-9_]+)\(", k.replace("(anonymous namespace)", "anon"))
        short = m.group(1) if m else k[:48]
        rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
tr = glob.glob(os.path.join(root, "trace", "*kernel_stats.csv"))
if tr:
    print("--- kernel_stats (no counters) ---")
    for r in csv.DictReader(open(tr[0])):
        if pat in r["Name"] or "bwd_" in r["Name"]:
            m = re.search(r"::([A-Za-z0-9_]+<[^>]*>|[A-Za-z0-9_]+)\(", r["Name"].replace("(anonymous namespace)", "anon"))
            print("  ", m.group(1) if m else r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"])
