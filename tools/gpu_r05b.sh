#!/bin/bash
# r05b: (1) gemm256 tile-height sweep on the prefill shapes, (2) A/B of a -amdgpu-kernarg-preload-count=16 build (lib/libplangen_hip_old.so.bin) against the
# default build at bs=64 and bs=8, (3) the cross-batch pipelined pass, (4) the bench-contract test that failed in r05a.
mkdir -p gpurun_out
bash tools/tile_height_sweep.sh > gpurun_out/r05b_tile_height.log 2>&1; cat gpurun_out/r05b_tile_height.log
bash tools/ab_lib.sh 2 > gpurun_out/r05b_ab_kp64.log 2>&1; cat gpurun_out/r05b_ab_kp64.log
BATCH=8 bash tools/ab_lib.sh 2 > gpurun_out/r05b_ab_kp8.log 2>&1; cat gpurun_out/r05b_ab_kp8.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-rccl-selftest --pipeline 4 > gpurun_out/r05b_pipeline.json 2> gpurun_out/r05b_pipeline.err; python -c "
import json; j=json.load(open('gpurun_out/r05b_pipeline.json')); print(j['value'], j.get('pipelined'))"; tail -3 gpurun_out/r05b_pipeline.err
python -m pytest tests/test_gpu_edges.py -q -k "bench_line_contract" 2>&1 | tail -3
