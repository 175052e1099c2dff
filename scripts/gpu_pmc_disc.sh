#!/bin/bash
# usage: scripts/gpu_pmc_disc.sh <tag> [arithmetic: bf16x3 | bf16x3_ring] -- MFMA utilisation of the Conv2d kernels of one STFT discriminator (win 1024, batch 32):
# forward / backward-data / weight-gradient per layer through tools/disc_layer_times.py, two PMC passes (kernel trace only)
tag=${1:-x}
mode=${2:-}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_disc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "grbm GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python $GRAFT_REPO_ROOT/tools/disc_layer_times.py 1024 32 $mode > $out/$name.log 2>&1
  rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out 300 > $GRAFT_REPO_ROOT/gpurun_out/pmc_disc_summary_$tag.txt 2>&1
grep -E "^== |MFMA utilisation" $GRAFT_REPO_ROOT/gpurun_out/pmc_disc_summary_$tag.txt | head -90
