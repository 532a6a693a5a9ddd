#!/usr/bin/env python3
"""Host-side cost of a tile-culled stage-2 iteration on the reference's schedule: cProfile over examples/stage2_schedule.py (sorted by own time)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import __graft_entry__ as g
g.build()
import stage2_schedule
pr = cProfile.Profile()
pr.enable()
out = stage2_schedule.run(8, 2, sparsify="--dense" not in sys.argv)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print(s.getvalue()[:7000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_callers("method 'to' of|method 'cpu' of|method 'item' of")
print(s.getvalue()[:5000])
print(out["iters_per_s"], out["iters"])
