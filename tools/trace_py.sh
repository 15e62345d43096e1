#!/bin/bash
# Kernel trace of an arbitrary python tool: gpurun -- 'bash tools/trace_py.sh tag tools/x.py args...'
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_$tag
rocprofv3 --kernel-trace --stats -d $OUT/trace_$tag -o k -- python3 $ROOT/$@ > $OUT/trace_$tag.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find $OUT/trace_$tag -name '*results.db' | head -1) > $OUT/trace_${tag}_summary.md
rm -rf $OUT/trace_$tag
head -30 $OUT/trace_${tag}_summary.md
