#!/bin/bash
# same-box A/B of the round-6 step (fused optimizer sweep, weight copies outside the step) against the round-5 step (HS_LEGACY_STEP=1)
# usage: tools/r6_ab_step.sh  -> gpurun_out/r06_step_ab.txt
out=gpurun_out/r06_step_ab.txt
: > $out
python tools/bench_adamw.py >> $out 2>&1
for b in 1 3; do
  for rep in 1 2; do
    for legacy in 1 0; do
      echo "== batch $b legacy=$legacy rep $rep" >> $out
      HS_LEGACY_STEP=$legacy python bench.py --batch $b --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --no-batch1 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'volumes/s', d['ms_per_step'], 'ms')" >> $out
    done
  done
done
