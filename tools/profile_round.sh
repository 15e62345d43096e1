#!/bin/bash
# Round profile on the GPU box: kernel trace of one default bench step + PMC HBM-traffic passes for the dominant kernel.
# usage: gpurun --timeout 1800 -- 'bash tools/profile_round.sh r02'
# Counters are collected in their OWN passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
tag=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --batch 64 --steps 1 --warmup 0"
echo "== kernel trace"; rm -rf $OUT/prof_$tag
rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o k -- $BENCH > $OUT/prof_$tag.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find $OUT/prof_$tag -name '*results.db' | head -1) > $OUT/prof_${tag}_summary.md
head -30 $OUT/prof_${tag}_summary.md
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"; rm -rf $OUT/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${tag}_$c -o p -- $BENCH --tokens 24 > $OUT/pmc_${tag}_$c.log 2>&1
done
F=$(find $OUT/pmc_${tag}_FETCH_SIZE -name '*results.db' | head -1); W=$(find $OUT/pmc_${tag}_WRITE_SIZE -name '*results.db' | head -1)
python3 $ROOT/tools/pmc_traffic.py $F $W --tokens 24 --json $OUT/pmc_attn_$tag.json \
  --cmd "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $BENCH --tokens 24" | tee $OUT/pmc_${tag}_traffic.txt
for pat in gemm_skinny3 gemm_sk4 rmsnorm_kernel; do
  echo "-- $pat"; python3 $ROOT/tools/pmc_dump.py $pat $F $W
done | tee $OUT/pmc_${tag}_gemm.txt
# the raw databases exceed gpurun_out's 64 MiB return limit when the loop is stream-launched: keep the summaries only
rm -rf $OUT/prof_$tag $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
