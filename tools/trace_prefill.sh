#!/bin/bash
# Kernel trace of the prefill phase alone (bench step with 1 image token): gpurun -- 'bash tools/trace_prefill.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_prefill
rocprofv3 --kernel-trace --stats -d $OUT/trace_prefill -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --batch 64 --steps 1 --warmup 0 --tokens 1 > $OUT/trace_prefill.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find $OUT/trace_prefill -name '*results.db' | head -1) > $OUT/trace_prefill_summary.md
rm -rf $OUT/trace_prefill
tail -2 $OUT/trace_prefill.log | cut -c1-600
head -40 $OUT/trace_prefill_summary.md
