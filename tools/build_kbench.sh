#!/bin/bash
# builds tools/kbench (host-only C++; links nothing but libdl + the HIP runtime)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -o tools/kbench -ldl
echo built tools/kbench
