#!/bin/bash
# copy the summaries of gpurun_out/r05 (profiles/run_profiles_r05.sh) into the tracked profiles/ files
O=gpurun_out/${1:-r05}
cp $O/bench.json profiles/r05_bench.json
cp $O/bench_detail.json profiles/r05_bench_detail.json
cp $O/kernel_stats_bench.csv profiles/r05_kernel_stats.csv
for l in target sched sched2k schedc schedc2k loss s1 s1train; do cp $O/kernel_stats_$l.csv profiles/r05_kernel_stats_$l.csv; done
cp $O/pmc_summary.txt profiles/r05_pmc_summary.txt
for l in sched schedc; do cp $O/pmc_summary_$l.txt profiles/r05_pmc_summary_$l.txt; done
python profiles/pmc_to_traffic.py profiles/r05_pmc_summary.txt r05
