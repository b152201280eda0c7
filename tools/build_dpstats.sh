#!/bin/bash
# Development aid: libpyani_gpu.so with the DP / cluster statistics compiled in (printed to stderr after every ANIm batch).
# Rebuild the product library afterwards:  python -c "from pyani_amd import build; build.build_gpu(force=True)"
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DPGA_DP_STATS -Iinclude -Ipyani_amd/csrc \
  -o pyani_amd/libpyani_gpu.so pyani_amd/csrc/pg_*.hip pyani_amd/csrc/pg_*.cpp -lpthread
