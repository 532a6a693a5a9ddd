#!/usr/bin/env python3
"""Rewrite the machine-kept parts of DESIGN.md's current-state table from the NEWEST tracked bench line (profiles/rNN_bench.json): the
headline-of-record marker and the Time / frac / traffic cells of the two headline rows.  tests/test_design_table_cpu.py checks the result."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d_bench\.json", f))[-1]
line = json.loads(open(os.path.join(ROOT, "profiles", newest)).read().strip().splitlines()[-1])
s = line["summary"]
fwd, bwd = f'{s["fwd_ms"]:.2f}', f'{s["bwd_ms"]:.2f}'
ffrac, bfrac = f'{s["fwd_frac"]:.3f}', f'{line["roofline"]["frac"]:.3f}'
ftr, btr = s["fwd_traffic_GB"] / 24.146, s["bwd_traffic_GB"] / 47.739          # counter bytes / algorithmic bytes of a cfg3 launch
p = os.path.join(ROOT, "DESIGN.md")
t = open(p).read()
marker = f"<!-- headline-of-record: file=profiles/{newest} mpix_s={line['value']:.0f} fwd_ms={fwd} bwd_ms={bwd} bwd_frac={bfrac} -->"
t = re.sub(r"<!-- headline-of-record:.*?-->\n", "", t)
out = []
for l in t.splitlines():
    if l.startswith("| `render_fwd2x_k` (`K1"):
        c = l.split(" | ")
        c[2], c[3], c[4] = f"{fwd} ms", f"**{ffrac}**", f"{ftr:.2f}x"
        l = " | ".join(c)
    elif l.startswith("| `render_bwd_pair_k` (`K2"):
        c = l.split(" | ")
        c[2], c[3], c[4] = f"{bwd} ms (with its pre-pass, HIP events of the bench)", f"**{bfrac}**", f"{btr:.2f}x"
        l = " | ".join(c)
    elif l.startswith("| Kernel (file) | What it is"):
        out.append(marker)
        out.append("")
    out.append(l)
open(p, "w").write("\n".join(out) + "\n")
print(marker)
