set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-loss --T 10"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/trace -o t -- $B > gpurun_out/pmc/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -o p -- $B > gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -o p -- $B > gpurun_out/pmc/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc/sq -o p -- $B > gpurun_out/pmc/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc/tcc -o p -- $B > gpurun_out/pmc/tcc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc/sq2 -o p -- $B > gpurun_out/pmc/sq2.log 2>&1
find gpurun_out/pmc -name "*.csv" | head -40
