#!/bin/bash
# What the epilogue's loads cost the 256x256 GEMM: the same shape without / with an fp32 residual / with a bias (PG_BENCH_RES, PG_BENCH_BIAS on pg_bench_gemm).
# usage: gpurun -- 'bash tools/epi_cost.sh'
for shape in "38400 2048 2048" "38400 2048 5632" "20000 2048 2048" "20000 2048 5632" "conv 64 96 96 256 256 0" "conv 64 48 48 512 512 0"; do
  for env in "" "PG_BENCH_RES=1" "PG_BENCH_BIAS=1" "PG_BENCH_RES=1 PG_BENCH_BIAS=1"; do
    echo "[$env] $(env $env python3 tools/big_gemm.py $shape 1 20 0 2>/dev/null | tail -1)"
  done
done
