#!/bin/bash
# kernel-level breakdown of the end-to-end stage-2 iteration (examples/stage2_step.py) with rocprofv3 --stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2prof -o t -- python examples/stage2_step.py --iters 20 > gpurun_out/s2prof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s2prof/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%5.1f%%  calls %5s  avg %9.1f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
rm -rf gpurun_out/s2prof
