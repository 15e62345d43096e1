#!/bin/bash
# gemm256_kernel on the prefill shapes of the bench batch (M = packed prompt tokens): tile height 256 / 224 / 192 rows (mode 4 / 5 / 6 pin it, 1 = pick_tile_height())
# in the default two-phase schedule, and the four-phase schedule of rounds 1-4 (mode + 8: 12 / 13 pin 256 / 224 rows, 9 = auto).  Every line verifies bit-identity against the 128x128 kernel (maxdiff 0).
# usage: gpurun -- 'bash tools/tile_height_sweep.sh [M] [modes]'
M=${1:-13285}; MODES=${2:-"4 5 6 1 12 13 9"}
for shape in "6144 2048" "2048 2048" "11264 2048" "2048 5632"; do
  for mode in $MODES; do python3 tools/big_gemm.py $M $shape $mode 30 2; done
done
