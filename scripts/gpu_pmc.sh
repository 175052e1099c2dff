#!/bin/bash
# usage: scripts_gpu_pmc.sh <tag>  -- PMC counter passes (separate runs, kernel-trace only) on a short bench
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-items 0 --no-roofline --no-graph --no-extra --no-train-step > $out/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc"; tail -2 $out/$name.log
  if [ $rc -ge 124 ]; then exit $rc; fi
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
find $out -name "*counter_collection.csv" | head
