#!/bin/bash
# SQ wave-state counters of one kernel (where do its wave cycles go): gpurun -- 'bash tools/pmc_sq.sh tag kernel_pattern tools/x.py args...'
tag=$1; pat=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmcsq_$tag
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmcsq_$tag -o p -- python3 $ROOT/$@ > $OUT/pmcsq_$tag.log 2>&1
python3 - "$(find $OUT/pmcsq_$tag -name '*results.db' | head -1)" "$pat" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select counter_name, avg(counter_value), count(*), avg(duration) from pmc_events where name like ? group by counter_name", (f"%{sys.argv[2]}%",)).fetchall()
d = {r[0]: r[1] for r in rows}
if not d: print("kernel not found"); sys.exit(0)
wc = d.get("SQ_WAVE_CYCLES", 1.0)
print(f"launches {rows[0][2]}  avg us {rows[0][3] / 1e3:.1f}")
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
    if k in d: print(f"{k:24s} {d[k]:16,.0f}  {d[k] / wc:6.3f} of SQ_WAVE_CYCLES")
print(f"{'SQ_WAVE_CYCLES':24s} {wc:16,.0f}")
PY
rm -rf $OUT/pmcsq_$tag
