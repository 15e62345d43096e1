#!/bin/bash
# One GPU-box pass: parity tests, default bench line, 2-rank plumbing on one GPU (gloo), logs under gpurun_out/.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_round.sh <tag> [pytest args]'
tag=${1:-x}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/tests_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/tests_$tag.log
tail -15 gpurun_out/tests_$tag.log
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
cat gpurun_out/bench_$tag.json
PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench2_$tag.json 2> gpurun_out/bench2_$tag.err; echo "bench2 rc=$?"
cat gpurun_out/bench2_$tag.json; tail -3 gpurun_out/bench2_$tag.err
