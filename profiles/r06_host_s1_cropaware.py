#!/usr/bin/env python3
"""Host-side cost of a stage-1 iteration with the crop-aware optimiser (optim.Stage1Adam): cProfile over examples/stage1_train.py --crop-aware-adam on,
and the kernel stats of the same run -- is the path host bound, and on what?"""
import cProfile, pstats, sys, os, io, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import __graft_entry__ as g; g.build()
import stage1_train
for mode in (False, True):
    r = stage1_train.run(8, 6, crop_aware_adam=mode)
    print(json.dumps({"crop_aware": mode, "it_s": r["iters_per_s"], "dense_epochs": r.get("iters_per_s_dense_epochs")}))
pr = cProfile.Profile(); pr.enable()
stage1_train.run(8, 6, crop_aware_adam=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:7000])
