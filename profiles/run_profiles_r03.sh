#!/bin/bash
# Round-3 profile recipe (run on the GPU box through gpurun): bench line + rocprofv3 kernel stats of the SAME command + PMC passes
# (separate runs per counter group, --kernel-trace only: gpurun refuses --pmc with the sys / hip / hsa trace domains).
# Outputs land in gpurun_out/$R (scratch); the summaries are copied into profiles/ and committed.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
if [ "$2" != "pmc-only" ]; then
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
fi
B="python profiles/pmc_target.py 50"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ttrace -o t -- $B > $O/ttrace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/mfma -o p -- $B > $O/mfma.log 2>&1
for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "render_|bwd_|reg_|patchnn|vote_fold|robust_|video_to|adam_" $f >> $f.tmp; mv $f.tmp $f; done
python profiles/summarize_pmc.py $O "" > $O/pmc_summary.txt
rm -f $O/*/p_kernel_trace.csv $O/trace/t_kernel_trace.csv $O/ttrace/t_kernel_trace.csv $O/*/p_agent_info.csv $O/*/t_agent_info.csv
[ -f $O/trace/t_kernel_stats.csv ] && cp $O/trace/t_kernel_stats.csv $O/kernel_stats_bench.csv
cp $O/ttrace/t_kernel_stats.csv $O/kernel_stats_target.csv
ls $O/*/; tail -30 $O/pmc_summary.txt
