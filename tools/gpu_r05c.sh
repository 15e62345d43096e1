#!/bin/bash
# r05c: A/B of the kernarg-ordered attention / 56-byte rmsnorm build (new) against the plain kernarg-preload build (old = lib/libplangen_hip_old.so.bin), then the whole
# GPU suite + the default bench line on the new build.
mkdir -p gpurun_out
bash tools/ab_lib.sh 2 > gpurun_out/r05c_ab64.log 2>&1; cat gpurun_out/r05c_ab64.log
BATCH=8 bash tools/ab_lib.sh 2 > gpurun_out/r05c_ab8.log 2>&1; cat gpurun_out/r05c_ab8.log
bash tools/gpu_round5.sh r05c
