#!/bin/bash
# usage: scripts/gpu_pmc_c3.sh <tag> -- config 3 (attention bottleneck): bench lines (fp32 + bf16 attention) and the PMC pass that
# gives "MFMA utilisation on attention" (north_star): SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE per dispatch
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_c3_$tag
mkdir -p $out
python tools/config_bench.py all > gpurun_out/config_bench_$tag.jsonl 2> gpurun_out/config_bench_$tag.err || { tail -5 gpurun_out/config_bench_$tag.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C3bf16; do
  for pass in "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "grbm GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$cfg/$name -- python $GRAFT_REPO_ROOT/tools/config_bench.py $cfg > $out/${cfg}_$name.log 2>&1
    rc=$?; echo "$cfg $name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$cfg 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_c3_summary_${cfg}_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
grep -A14 "attention" gpurun_out/pmc_c3_summary_C3_$tag.txt | head -40
grep -A14 "attention" gpurun_out/pmc_c3_summary_C3bf16_$tag.txt | head -40
