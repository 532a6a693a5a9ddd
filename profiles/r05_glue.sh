#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/glue; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stage1_driver.py tests/test_gpu_mpv.py tests/test_gpu_optim.py tests/test_gpu_reference_modules.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
last() { python - "$1" "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[2], round(d["iters_per_s"], 1), [round(x["iters_per_s"]) for x in d.get("levels", [])])
PY
}
for r in 1 2; do
  timeout 300 python examples/stage2_schedule.py --sparsify > $O/schedc_$r.json 2> $O/schedc_$r.err; last $O/schedc_$r.json "tile-culled run $r"
done
timeout 300 python examples/stage2_schedule.py > $O/sched.json 2> $O/sched.err; last $O/sched.json "dense"
timeout 300 python examples/stage1_train.py > $O/s1train.json 2> $O/s1train.err; tail -c 420 $O/s1train.json
