#!/usr/bin/env python3
"""stage-1 iteration at the reference's native shape only (examples/stage1_step.py run()), for `rocprofv3 --kernel-trace --stats`: GPU time per
iteration against the wall time per iteration = how launch / host bound the iteration is."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
g.build()
import stage1_step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print(json.dumps({"iters": n + 3, **stage1_step.run(n)}))
