#!/bin/bash
mkdir -p gpurun_out
for a in 0 1 2 3 7; do
  AGX_ABLATE=$a timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-items 0 > gpurun_out/abl_$a.json 2>/dev/null
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
  python - <<PY
import json
r=json.load(open("gpurun_out/abl_$a.json"))
k=r["roofline"]["kernels"]
print("ABL=$a", "step %.2f ms"%r["ms_per_step"], {n:(v["avg_us"],v["tflops"]) for n,v in k.items() if n.startswith("resblock")})
PY
done
