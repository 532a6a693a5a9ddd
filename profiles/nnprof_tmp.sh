cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
B="python profiles/pmc_nn.py"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/mfma -o p -- $B > $O/mfma.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/mem -o p -- $B > $O/mem.log 2>&1
for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "patchnn|gram_major|pixel_major" $f >> $f.tmp; mv $f.tmp $f; done
python profiles/summarize_pmc.py $O "" > $O/pmc_summary.txt
rm -f $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv
cat $O/pmc_summary.txt | head -120; tail -3 $O/mfma.log
