#!/bin/bash
# r05d: new op tests, the round-5 evidence pass, the GPU suite under concurrent memory load
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "tile_heights or fused_norm" 2>&1 | tail -4
bash tools/profile_r05.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -60 gpurun_out/profile_r05.log | cut -c1-300
bash tools/gpu_under_load.sh r05 2 --deselect tests/test_gpu_fullconfig.py
