#!/bin/bash
# gpurun -- 'bash tools/trace_gaps.sh <batch> <tag>': kernel trace of a 96-token decode at the given batch + launch-gap table (tools/trace_gaps.py)
b=${1:-64}; tag=${2:-gaps$b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_$tag
rocprofv3 --kernel-trace -d $OUT/trace_$tag -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --batch $b --steps 1 --warmup 0 --tokens 96 > $OUT/trace_$tag.log 2>&1
python3 $ROOT/tools/trace_gaps.py $(find $OUT/trace_$tag -name '*results.db' | head -1) > $OUT/${tag}_gaps.md 2>&1
rm -rf $OUT/trace_$tag
cat $OUT/${tag}_gaps.md
