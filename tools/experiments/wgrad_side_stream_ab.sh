#!/bin/bash
# same-box A/B of the weight gradients on their own stream (HS_WGRAD_STREAM=1) at small batches
for w in clip_s1 vit256 dual256 clip_c4a; do
  for b in 1 3; do
    for rep in 1 2; do
      for on in 0 1; do
        r=$(HS_WGRAD_STREAM=$on python bench.py --workload $w --batch $b --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --no-batch1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['outputs_finite'])")
        echo "$w batch $b wgrad_stream=$on: $r"
      done
    done
  done
done
