import sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import __graft_entry__ as g; g.build()
import stage1_step as S
for name, kw in (("native", {}), ("720p_1p1", dict(frame=(720, 1280), crop=(720, 1280), scale=1.1)), ("720p_1p6", dict(frame=(720, 1280), crop=(720, 1280), scale=1.6)),
                 ("360p_crop_1p6_big", dict(frame=(720, 1280), crop=(360, 640), scale=1.6))):
    for ca in (False, True):
        r = S.run(iters=30, crop_aware_adam=ca, **kw)
        print(name, "crop_aware" if ca else "whole_stack", "%.1f it/s" % r["iters_per_s"], r["shape"][-60:])
