#!/bin/bash
# Round 5: 256 / 224 / 192-row tiles of gemm256_kernel on the prefill shapes of the bench batch (M = packed prompt tokens); mode 4 / 5 / 6 pin the
# height, 1 = pick_tile_height().  Every line verifies bit-identity against the 128x128 kernel (maxdiff 0).  usage: gpurun -- 'bash tools/tile_height_sweep.sh [M]'
M=${1:-13285}
for shape in "6144 2048" "2048 2048" "11264 2048" "2048 5632"; do
  for mode in 4 5 6 1; do python3 tools/big_gemm.py $M $shape $mode 30 2; done
done
