#!/bin/bash
# HBM-traffic passes of the dominant kernel (decode attention) -> gpurun_out/pmc_attn_<tag>.json (copy to profiles/pmc_attn.json):
# separate rocprofv3 --pmc runs for FETCH_SIZE and WRITE_SIZE (--kernel-trace only), gfx950 FETCH_SIZE correction in tools/pmc_traffic.py.
# usage: gpurun --timeout 900 -- 'bash tools/pmc_attn_pass.sh r04b'
tag=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --batch 64 --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${tag}_$c -o p -- $BENCH --tokens 24 > $OUT/pmc_${tag}_$c.log 2>&1
done
F=$(find $OUT/pmc_${tag}_FETCH_SIZE -name '*results.db' | head -1); W=$(find $OUT/pmc_${tag}_WRITE_SIZE -name '*results.db' | head -1)
python3 $ROOT/tools/pmc_traffic.py $F $W --tokens 24 --json $OUT/pmc_attn_$tag.json \
  --cmd "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $BENCH --tokens 24" | tee $OUT/${tag}_pmc_attn_traffic.txt
rm -rf $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
