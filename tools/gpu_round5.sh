#!/bin/bash
# Round-5 GPU pass: the full-configuration parity tests first (their own log, -s for the measured statistics), then the rest of the GPU
# suite (no -x: every failure in one pass), then the default bench line.   usage: gpurun --timeout 3300 -- 'bash tools/gpu_round5.sh <tag>'
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullconfig.py -q -s --timeout 1500 > gpurun_out/fullconfig_$tag.log 2>&1; echo "fullconfig rc=$?" | tee -a gpurun_out/fullconfig_$tag.log
grep -v "^$" gpurun_out/fullconfig_$tag.log | tail -25 | cut -c1-600
python -m pytest tests -m gpu -q --timeout 1500 --deselect tests/test_gpu_fullconfig.py "$@" > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
tail -25 gpurun_out/tests_$tag.log | cut -c1-400
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
cat gpurun_out/bench_$tag.json; tail -5 gpurun_out/bench_$tag.err
