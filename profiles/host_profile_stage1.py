#!/usr/bin/env python3
"""Host-side cost of a 720p stage-1 iteration: cProfile over examples/stage1_step.py (sorted by own time)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import __graft_entry__ as g; g.build()
import stage1_step
stage1_step.run(5) if len(sys.argv) > 1 else stage1_step.run(5, frame=(720, 1280), crop=(720, 1280), scale=1.1)
pr = cProfile.Profile(); pr.enable()
stage1_step.run(200) if len(sys.argv) > 1 else stage1_step.run(60, frame=(720, 1280), crop=(720, 1280), scale=1.1)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "tottime").print_stats(45); print(s.getvalue()[:6000])
