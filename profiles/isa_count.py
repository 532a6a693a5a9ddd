#!/usr/bin/env python3
"""Instruction histogram of one kernel's main loop from hipcc -S output:  python profiles/isa_count.py <file.s> <symbol substring> [loop]
With `loop`, only the largest backward-branch loop body is counted (the per-plane loop of the render kernels)."""
import re
import sys
from collections import Counter

src, pat = open(sys.argv[1]).read().splitlines(), sys.argv[2]
start = next(i for i, l in enumerate(src) if l.endswith(":") is False and re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start + 1:end]
if len(sys.argv) > 3:
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\S+:", l)}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\S*\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    body = body[best[1]:best[2]]
ins = [l.split()[0] for l in body if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.strip().endswith(":")]
c = Counter(ins)
print("instructions", len(ins), "valu", sum(v for k, v in c.items() if k.startswith("v_")), "vmem", sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))),
      "lds", sum(v for k, v in c.items() if k.startswith("ds_")), "salu", sum(v for k, v in c.items() if k.startswith("s_")))
for k, v in c.most_common(45):
    print(f"  {k:28s} {v}")
