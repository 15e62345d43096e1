#!/bin/bash
# Instruction-fetch side of the decode kernels: do the straight-line (fully unrolled, 14-27 KB) kernels wait on the instruction cache?
# usage: gpurun -- 'bash tools/pmc_icache.sh <tag> [bench args]'
tag=${1:-ic}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/${tag}_counters.txt; cat $OUT/${tag}_counters.txt; echo
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-shard-check --no-rccl-selftest --steps 1 --warmup 0 --tokens 24 $@"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/pmcic_${tag}_$n
  rocprofv3 --pmc $set --kernel-trace -d $OUT/pmcic_${tag}_$n -o p -- $BENCH > $OUT/pmcic_${tag}_$n.log 2>&1
  python3 - "$(find $OUT/pmcic_${tag}_$n -name '*results.db' | head -1)" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name").fetchall()
by = collections.defaultdict(dict); meta = {}
for name, c, v, n, dur in rows:
    by[name][c] = v; meta[name] = (n, dur)
for name in sorted(by, key=lambda k: -meta[k][0] * meta[k][1])[:9]:
    n, dur = meta[name]
    print(f"{name[:70]:70s} n={n:5d} {dur / 1e3:8.1f} us  " + "  ".join(f"{c}={v:,.0f}" for c, v in sorted(by[name].items())))
PY
  rm -rf $OUT/pmcic_${tag}_$n
done
