for shape in "38400 2048 2048" "38400 2048 5632" "conv 64 96 96 256 256 0" "conv 64 48 48 512 512 0"; do
  for env in "" "PG_BENCH_RES=1" "PG_BENCH_BIAS=1" "PG_BENCH_RES=1 PG_BENCH_BIAS=1"; do
    echo "[$env] $(env $env python3 tools/big_gemm.py $shape 1 20 0 2>/dev/null | tail -1)"
  done
done
