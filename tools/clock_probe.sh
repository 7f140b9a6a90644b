#!/bin/bash
# shader clock / package power while a command runs (diagnostic): bash tools/clock_probe.sh <max samples shown> <command...>
# samples once per second until the command ends; prints the samples taken under load (package power > 400 W)
S=$1; shift
( "$@" > /dev/null 2>&1 ) &
PID=$!
n=0
while kill -0 $PID 2>/dev/null; do
  L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): /W=/' | tr '\n' ' ')
  W=$(echo "$L" | sed -e 's/.*W=//' -e 's/\..*//')
  if [ "${W:-0}" -gt 400 ] && [ $n -lt $S ]; then echo "$L"; n=$((n+1)); fi
  sleep 1
done
wait $PID
