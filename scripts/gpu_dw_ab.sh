#!/bin/bash
# A/B of the 1-D weight-gradient paths on the generator training step (kernel-trace totals per step)
for cfg in "dw_direct=0" "dw_direct=1" "dw_direct=2"; do
  export AGX_TUNING=$cfg
  ./scripts/gpu_prof_one.sh ab_$cfg $GRAFT_REPO_ROOT/tools/train_step_bench.py 32 3 > /dev/null || exit 1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_ab_$cfg/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 5e6
dw = sum(float(r["TotalDurationNs"]) for r in rows if "bwd_weight" in r["Name"]) / 5e6
red = sum(float(r["TotalDurationNs"]) for r in rows if "slice_reduce" in r["Name"]) / 5e6
print("$cfg: step %.2f ms, dW kernels %.2f ms, slice reduce %.2f ms" % (tot, dw, red))
PY
done
