#!/bin/bash
# Kernel trace of one bench step at a given per-GPU batch: gpurun -- 'bash tools/trace_batch.sh 8 tag'
b=${1:-8}; tag=${2:-b$b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_$tag
rocprofv3 --kernel-trace --stats -d $OUT/trace_$tag -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --no-secondary --batch $b --steps 1 --warmup 0 > $OUT/trace_$tag.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find $OUT/trace_$tag -name '*results.db' | head -1) > $OUT/trace_${tag}_summary.md
rm -rf $OUT/trace_$tag
head -24 $OUT/trace_${tag}_summary.md
