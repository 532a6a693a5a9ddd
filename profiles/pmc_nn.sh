#!/bin/bash
# PMC passes of the NN-search kernels alone (target: profiles/pmc_nn.py; separate runs per counter group, --kernel-trace only).
# Output: gpurun_out/$1/pmc_summary.txt  (copy to profiles/rNN_pmc_summary_nn.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_nn}; mkdir -p $O
B="python profiles/pmc_nn.py"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/mfma -o p -- $B > $O/mfma.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "patchnn|gram_major|pixel_major" $f >> $f.tmp; mv $f.tmp $f; done
python profiles/summarize_pmc.py $O "" > $O/pmc_summary.txt
rm -f $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv
grep -A24 "patchnn5_k" $O/pmc_summary.txt | head -30
