#!/bin/bash
# Kernel trace of a secondary workload: gpurun -- 'bash tools/trace_cfg.sh mmu'
wl=${1:-mmu}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_$wl
rocprofv3 --kernel-trace --stats -d $OUT/trace_$wl -o k -- python3 $ROOT/tools/bench_configs.py $wl --steps 1 > $OUT/trace_$wl.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $(find $OUT/trace_$wl -name '*results.db' | head -1) > $OUT/trace_${wl}_summary.md
rm -rf $OUT/trace_$wl
tail -3 $OUT/trace_$wl.log
head -30 $OUT/trace_${wl}_summary.md
