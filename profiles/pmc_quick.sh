#!/bin/bash
# quick PMC passes on a T=10 launch (bounded by timeouts; TCC_HIT/MISS pass excluded: it hung rocprofv3 at T=50)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcq; rm -rf $O; mkdir -p $O
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-loss --no-stage2 --T 10 $EXTRA"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_INSTS_WAVE32_LDS --kernel-trace --output-format csv -d $O/sq3 -o p -- $B > $O/sq3.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
python profiles/summarize_pmc.py $O > $O/summary.txt; grep -v "^   .* mean=0$" $O/summary.txt | grep -A30 "render_bwd_tile_k\|render_fwd2" | head -70
