#!/bin/bash
# Round-1 profile recipe (run on the GPU box through gpurun): bench line + rocprofv3 kernel stats + PMC passes.
# Outputs land in gpurun_out/ (scratch); the summaries are copied into profiles/ by hand and committed.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage2 > $O/trace.log 2>&1
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-loss --no-stage2"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
# (the TCC_HIT/MISS/REQ pass hung rocprofv3 for 40 min on the T=50 launch in round 1 -- every pass is now bounded)
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
python profiles/summarize_pmc.py $O > $O/pmc_summary.txt
# keep the scratch small enough to travel back (gpurun merges <= 64 MiB): drop raw per-dispatch traces, keep our kernels' counters
for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "render_|bwd_|patchnn|vote_fold|robust_|video_to_pixel" $f >> $f.tmp; mv $f.tmp $f; done
rm -f $O/*/p_kernel_trace.csv $O/trace/t_kernel_trace.csv $O/*/p_agent_info.csv
cp $O/trace/t_kernel_stats.csv $O/kernel_stats.csv
tail -c 2500 $O/bench.json
