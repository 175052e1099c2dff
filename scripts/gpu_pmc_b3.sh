#!/bin/bash
# usage: scripts/gpu_pmc_b3.sh <tag>  -- PMC passes (separate runs, kernel-trace only) over tools/b3_run.py
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_b3_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/b3_run.py 3 > $out/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc"; tail -1 $out/$name.log
  if [ $rc -ge 124 ]; then exit $rc; fi
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
run grbm GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out 100 > gpurun_out/pmc_b3_summary_$tag.txt 2>&1
grep -A 30 "resblock_b3" gpurun_out/pmc_b3_summary_$tag.txt | head -120
