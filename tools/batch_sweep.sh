#!/bin/bash
# per-GPU batch sweep of the default workload (bench.py --batch B), with the dominant kernel's roofline: bash tools/batch_sweep.sh <B...>
for b in "$@"; do
  echo "batch $b: $(python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["kernel"], r["achieved"], r["frac"], [(k["name"], round(k["tflops"])) for k in j["kernels"][:6]])')"
done
