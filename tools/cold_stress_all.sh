#!/bin/bash
# cold-launch stress (one launch at a time on an idle GPU, fresh operands, vs fp64) over every production decode-GEMM
# class at 16 / 64 / 128 rows + the other LDS-DMA kernels + the prefill path.  gpurun --timeout 3000 -- 'bash tools/cold_stress_all.sh 250'
n=${1:-250}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
{
for M in 16 64 128; do
  for shape in "6144 2048" "2048 2048" "2048 5632" "16384 2048"; do python3 $ROOT/tools/op_gemm_stress.py $n $M $shape | tail -3; done
done
python3 $ROOT/tools/op_cold_stress.py 150 | tail -6
python3 $ROOT/tools/prefill_cold_stress.py 150 0.15 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $OUT/r03_cold_stress.txt
