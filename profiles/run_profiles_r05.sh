#!/bin/bash
# Round-5 profile recipe -- ONE run (on the GPU box through gpurun) regenerates every tracked summary from the same tree:
#   bench line (bench.json) + rocprofv3 kernel stats of the SAME command          -> r05_bench.json, r05_kernel_stats.csv
#   pmc_target.py (cfg3 render, reference geometry, looping loss; + fp16 stack)   -> r05_kernel_stats_target.csv, r05_pmc_summary.txt
#   the stage-2 schedule (dense / tile-culled, fused step / two kernels), the loss iteration, the stage-1 iteration -> r05_kernel_stats_{sched,sched2k,schedc,schedc2k,loss,s1}.csv
# PMC passes are separate runs per counter group with --kernel-trace only (gpurun refuses --pmc with the sys / hip / hsa trace domains).
# Outputs land in gpurun_out/$R (scratch); profiles/collect_r05.sh copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r05}
O=gpurun_out/$R
mkdir -p $O
VL3D_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
B="python profiles/pmc_target.py 50 fp16"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ttrace -o t -- $B > $O/ttrace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "render_|bwd_|reg_|patchnn|vote_fold|robust_|video_to|adam_|loop_" $f >> $f.tmp; mv $f.tmp $f; done
python profiles/summarize_pmc.py $O "" > $O/pmc_summary.txt
# sched: the dense schedule with the optimiser step inside the backward (the default); sched2k: vl3d_render_bwd + the step kernel;
# schedc / schedc2k: the same pair for the tile-culled model
for leg in "sched examples/stage2_schedule.py" "sched2k examples/stage2_schedule.py --two-kernels" "schedc examples/stage2_schedule.py --sparsify" \
           "schedc2k examples/stage2_schedule.py --sparsify --two-kernels" "loss profiles/loss_iter_prof.py" "s1 examples/stage1_step.py" "s1train examples/stage1_train.py"; do
  set -- $leg
  L=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L -o t -- python "$@" > $O/$L.log 2>&1
  cp $O/$L/t_kernel_stats.csv $O/kernel_stats_$L.csv
done
rm -f $O/*/p_kernel_trace.csv $O/*/t_kernel_trace.csv $O/*/p_agent_info.csv $O/*/t_agent_info.csv
[ -f $O/trace/t_kernel_stats.csv ] && cp $O/trace/t_kernel_stats.csv $O/kernel_stats_bench.csv
cp $O/ttrace/t_kernel_stats.csv $O/kernel_stats_target.csv
ls $O; tail -5 $O/pmc_summary.txt; head -c 1500 $O/bench.json
# HBM traffic of the fused backward + step on the schedule (means over the schedule's mix of levels and crops, like the kernel stats' averages)
for leg in "sched " "schedc --sparsify"; do
  set -- $leg
  L=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$L/$c -o p -- python examples/stage2_schedule.py --epochs 1 "$@" > $O/pmc_$L.$c.log 2>&1
    f=$O/pmc_$L/$c/p_counter_collection.csv; head -1 $f > $f.tmp; grep -E "render_|bwd_|adam_" $f >> $f.tmp; mv $f.tmp $f
    rm -f $O/pmc_$L/$c/p_kernel_trace.csv $O/pmc_$L/$c/p_agent_info.csv
  done
  python profiles/summarize_pmc.py $O/pmc_$L "" > $O/pmc_summary_$L.txt
done
