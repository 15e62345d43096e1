for m in 1 3 1 3; do
timeout 120 python3 tools/big_gemm.py 8192 8192 8192 $m 10
timeout 120 python3 tools/big_gemm.py 16384 6144 2048 $m
timeout 120 python3 tools/big_gemm.py 16384 2048 5632 $m
done
