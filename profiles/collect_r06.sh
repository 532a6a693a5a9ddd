#!/bin/bash
# copy the summaries of gpurun_out/r06 (profiles/run_profiles_r06.sh) into the tracked profiles/ files
O=gpurun_out/${1:-r06}
cp $O/bench.json profiles/r06_bench.json
cp $O/bench_detail.json profiles/r06_bench_detail.json
cp $O/kernel_stats_bench.csv profiles/r06_kernel_stats.csv
for l in target sched sched2k schedc schedc2k schedx loss s1 s1train; do cp $O/kernel_stats_$l.csv profiles/r06_kernel_stats_$l.csv; done
cp $O/pmc_summary.txt profiles/r06_pmc_summary.txt
for l in sched schedc; do cp $O/pmc_summary_$l.txt profiles/r06_pmc_summary_$l.txt; done
python profiles/pmc_to_traffic.py profiles/r06_pmc_summary.txt r06
cp $O/n4_gloo.json profiles/r06_n4_gloo.json; cp $O/n2_gloo.json profiles/r06_n2_gloo.json; [ -f $O/n8_gloo.json ] && cp $O/n8_gloo.json profiles/r06_n8_gloo.json
python profiles/update_design_table.py
