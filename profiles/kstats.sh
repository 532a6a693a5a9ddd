#!/bin/bash
# rocprofv3 kernel statistics of one command (run on the GPU box through gpurun):  profiles/kstats.sh <name> <command...>
# -> gpurun_out/<name>/t_kernel_stats.csv and the top kernels on stdout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=$1; shift
O=gpurun_out/$N
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- "$@" > $O/log.txt 2>&1
tail -8 $O/log.txt | grep -v "^W2026\|^E2026"
rm -f $O/t_kernel_trace.csv $O/t_agent_info.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/t_kernel_stats.csv")))
for r in rows[:${TOPN:-16}]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:9.4f} ms  {float(r["Percentage"]):6.2f} %')
PY
