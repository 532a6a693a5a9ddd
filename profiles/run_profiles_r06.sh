#!/bin/bash
# Round-6 profile recipe -- ONE run (on the GPU box through gpurun) regenerates every tracked summary from the same tree:
#   bench line (bench.json) + rocprofv3 kernel stats of the SAME command          -> r06_bench.json, r06_kernel_stats.csv
#   pmc_target.py (cfg3 render, reference geometry, looping loss; + fp16 stack)   -> r06_kernel_stats_target.csv, r06_pmc_summary.txt
#   the stage-2 schedule (dense / tile-culled, fused step / two kernels), the loss iteration, the stage-1 iteration -> r06_kernel_stats_{sched,sched2k,schedc,schedc2k,loss,s1}.csv
# PMC passes are separate runs per counter group with --kernel-trace only (gpurun refuses --pmc with the sys / hip / hsa trace domains).
# Outputs land in gpurun_out/$R (scratch); profiles/collect_r06.sh copies the summaries into profiles/.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r06}
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
# schedc / schedc2k: the same pair for the tile-culled model; schedx: the tile-culled model in the TILE-EXACT layout (round 6: every quad owns its border texels)
for leg in "sched examples/stage2_schedule.py" "sched2k examples/stage2_schedule.py --two-kernels" "schedc examples/stage2_schedule.py --sparsify" \
           "schedc2k examples/stage2_schedule.py --sparsify --two-kernels" "schedx examples/stage2_schedule.py --sparsify --tile-exact" "loss profiles/loss_iter_prof.py" "s1 examples/stage1_step.py" "s1train examples/stage1_train.py"; do
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

# N = 4 dry run of the multi-GPU path on ONE MI355X (device kernels per rank, host collectives over gloo: a check of the N > 1 code, not a timing):
# four row bands whose parallax halos span two ranks, direct gather with one message per peer, halo-gradient exchange, band loss
VL3D_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 3 --warmup 1 --gather-algo direct > $O/n4_gloo.json 2> $O/n4_gloo.err
VL3D_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 3 --warmup 1 > $O/n2_gloo.json 2> $O/n2_gloo.err
# ... and at the node size north_star names: eight processes, eight bands of 90 rows
VL3D_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 8 --steps 2 --warmup 1 > $O/n8_gloo.json 2> $O/n8_gloo.err
tail -c 600 $O/n4_gloo.json
