#!/usr/bin/env python3
"""The stage-1 iteration on a whole 720p frame only (examples/stage1_step.py cfg2 leg): target for rocprofv3 --stats (GPU time per iteration vs wall)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import stage1_step
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.1
print(json.dumps(stage1_step.run(40, frame=(720, 1280), crop=(720, 1280), scale=scale)))
