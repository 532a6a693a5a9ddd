#!/bin/bash
# round 6: DYNAMIC instruction counts (SQ_INSTS_VALU per launch) and times of the regulariser kernels with their terms compiled out one by one
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_reg; mkdir -p $O
for v in shipped regab1 regab3 regab7 regab8 regab24; do
  unset VL3D_LIB_PATH; [ $v != shipped ] && export VL3D_LIB_PATH=$GRAFT_REPO_ROOT/videoloop3d_amd/lib/ab/$v.so
  python profiles/r06_reg_terms.py 6 2>/dev/null | tail -1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/$v -o p -- python profiles/r06_reg_terms.py 1 > $O/$v.log 2>&1
  f=$O/$v/p_counter_collection.csv; head -1 $f > $f.tmp; grep -E "render_|reg_" $f >> $f.tmp; mv $f.tmp $f
  python profiles/summarize_pmc.py $O/$v "" 2>/dev/null | grep -E "^render_|^reg_|SQ_INSTS_VALU|SQ_INSTS_LDS " | paste - - - | cut -c1-200 | sed "s/^/  $v: /"
  rm -f $O/$v/p_kernel_trace.csv $O/$v/p_agent_info.csv
done
