#!/bin/bash
# Round 5, session 3: the tile-culled fused step -- 32-wide one-frame regions (desc->variant 5), per-texel records + moments requested in
# front of the barrier: tests, tile-culled schedule A/B, kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/rw32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_optim.py -x -q > $O/test_optim.log 2>&1; tail -3 $O/test_optim.log
last() { python - "$1" "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[2], round(d["iters_per_s"], 1), [round(x["iters_per_s"]) for x in d["levels"]])
PY
}
for r in 1 2; do for v in 0 5; do
  timeout 300 python examples/stage2_schedule.py --sparsify --bwd-variant $v > $O/schedc_v${v}_$r.json 2> $O/schedc_v${v}_$r.err
  last $O/schedc_v${v}_$r.json "variant $v run $r"
done; done
for v in 0 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v$v -o t -- python examples/stage2_schedule.py --sparsify --bwd-variant $v > $O/prof_v$v.log 2>&1
  cp $O/prof_v$v/t_kernel_stats.csv $O/kernel_stats_schedc_v$v.csv; rm -rf $O/prof_v$v
  head -6 $O/kernel_stats_schedc_v$v.csv | cut -c1-150
done
