#!/bin/bash
# Round 5, session 3: tile-culled backward -- tests, the full-frame legs (profiles/cull_lean.py), the tile-culled schedule with kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/cull; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_optim.py tests/test_gpu_mpv.py tests/test_gpu_reference_modules.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python profiles/cull_lean.py > $O/cull_lean.txt 2>&1; tail -6 $O/cull_lean.txt
for r in 1 2; do
  timeout 300 python examples/stage2_schedule.py --sparsify > $O/schedc_$r.json 2> $O/schedc_$r.err
  python - $O/schedc_$r.json <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
d=json.loads(l)
print("schedc", round(d["iters_per_s"], 1), [round(x["iters_per_s"]) for x in d["levels"]])
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python examples/stage2_schedule.py --sparsify > $O/prof.log 2>&1
cp $O/prof/t_kernel_stats.csv $O/kernel_stats_schedc.csv; rm -rf $O/prof; head -6 $O/kernel_stats_schedc.csv | cut -c1-200
