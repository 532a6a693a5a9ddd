#!/bin/bash
# copy the summaries of gpurun_out/r04 (profiles/run_profiles_r04.sh) into the tracked profiles/ files
O=gpurun_out/${1:-r04}
cp $O/bench.json profiles/r04_bench.json
cp $O/kernel_stats_bench.csv profiles/r04_kernel_stats.csv
for l in target sched sched2k schedc schedc2k loss s1; do cp $O/kernel_stats_$l.csv profiles/r04_kernel_stats_$l.csv; done
cp $O/pmc_summary.txt profiles/r04_pmc_summary.txt
for l in sched schedc; do cp $O/pmc_summary_$l.txt profiles/r04_pmc_summary_$l.txt; done
