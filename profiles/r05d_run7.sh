#!/bin/bash
# session 4 of round 5, GPU call 7: the stage-2 weighted total in one launch (MPMeshVid.objective) -- tests, A/B of the schedules
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mpv.py tests/test_gpu_optim.py -x -q > $O/tests_s2.txt 2>&1; tail -3 $O/tests_s2.txt
last() { python - "$1" "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[2], round(d["iters_per_s"], 1), [round(x["iters_per_s"]) for x in d.get("levels", [])])
PY
}
for r in 1 2 3; do
  python examples/stage2_schedule.py --sparsify > $O/s2c_obj_$r.json 2>/dev/null; last $O/s2c_obj_$r.json "tile-culled objective"
  python examples/stage2_schedule.py --sparsify --generic-objective > $O/s2c_gen_$r.json 2>/dev/null; last $O/s2c_gen_$r.json "tile-culled generic  "
done
python examples/stage2_schedule.py > $O/s2d_obj.json 2>/dev/null; last $O/s2d_obj.json "dense objective"
python examples/stage2_schedule.py --generic-objective > $O/s2d_gen.json 2>/dev/null; last $O/s2d_gen.json "dense generic  "
