#!/bin/bash
# round-4 validation run: the whole GPU test suite, then the default bench line
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r4_pytest.log | tail -25 &&
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/r4_bench.json; tail -5 gpurun_out/r4_bench.err
