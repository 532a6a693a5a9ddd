#!/bin/bash
# round 6: which LDS reads of patchnn6_k conflict?  A/B of the shipped kernel against a measurement build whose norm reads all hit ONE word
# (-DVL3D_NN6_FLAT_NORMS: wrong norms, same instruction stream) -- the LDS counters of the search kernel only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in shipped flatnorms; do
  O=gpurun_out/r06_nn_$v; mkdir -p $O
  [ $v = flatnorms ] && export VL3D_LIB_PATH=$GRAFT_REPO_ROOT/videoloop3d_amd/lib/ab/nn_flatnorms.so
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- python profiles/pmc_nn.py > $O/sq2.log 2>&1
  for f in $O/*/p_counter_collection.csv; do head -1 $f > $f.tmp; grep -E "patchnn" $f >> $f.tmp; mv $f.tmp $f; done
  python profiles/summarize_pmc.py $O "" > $O/pmc_summary.txt
  rm -f $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv
  echo "== $v"; grep -E "patchnn|SQ_|GRBM" $O/pmc_summary.txt | cut -c1-120
done
