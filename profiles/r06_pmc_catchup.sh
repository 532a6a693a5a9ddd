#!/bin/bash
# round 6: what bounds adam_window_catchup_k on a tile-culled window?  SQ counters of the kernel in the tile-culled schedule ($1 = extra flags of the example)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_cu; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python examples/stage2_schedule.py --sparsify $1 > $O/p$i.log 2>&1
  f=$O/p$i/p_counter_collection.csv; head -1 $f > $f.tmp; grep -E "catchup|adam_window_step" $f >> $f.tmp; mv $f.tmp $f
  python profiles/summarize_pmc.py $O/p$i "" 2>/dev/null | cut -c1-160
  rm -f $O/p$i/p_kernel_trace.csv $O/p$i/p_agent_info.csv
done
