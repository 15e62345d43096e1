#!/bin/bash
# The GPU parity suite UNDER CONCURRENT MEMORY LOAD (a background kernel on another stream streams 768 MB in a loop; conftest fixture
# `_background_memory_load`).  usage: gpurun -- 'bash tools/gpu_under_load.sh <tag> <mode 0|1|2> [pytest args]'
tag=${1:-x}; mode=${2:-2}; shift; shift
mkdir -p gpurun_out
PG_BG_LOAD=$mode python -m pytest tests -m gpu -q "$@" > gpurun_out/tests_load_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_load_$tag.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests_load_$tag.log | tail -60
