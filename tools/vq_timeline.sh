#!/bin/bash
# Per-dispatch timeline of ONE VQ-16 decode of 64 images (the last decode of tools/vq_only.py): every kernel in launch order with its duration.
# usage: gpurun -- 'bash tools/vq_timeline.sh [tag]'
tag=${1:-vqtl}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_$tag
rocprofv3 --kernel-trace -d $OUT/trace_$tag -o k -- python3 $ROOT/tools/vq_only.py 64 2 > $OUT/trace_$tag.log 2>&1
python3 - "$(find $OUT/trace_$tag -name '*results.db' | head -1)" > $OUT/${tag}_timeline.md <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
v = "kernels" if "kernels" in views else next(x for x in views if "kernel" in x.lower() and "dispatch" in x.lower())
cols = [r[1] for r in db.execute(f"pragma table_info({v})")]
namec = "name" if "name" in cols else "kernel_name"
rows = list(db.execute(f"select {namec}, start, end from {v} order by start"))
# the last decode = from the last vq_gather kernel to the last conv3x3_out_halo
gi = max(i for i, r in enumerate(rows) if "vq_gather" in r[0])
oi = max(i for i, r in enumerate(rows) if "conv3x3_out" in r[0] or "conv3x3_small" in r[0])
seg = rows[gi:oi + 1]
def short(n):
    n = re.sub(r"^_Z\d+", "", n); return n[:64]
print(f"one VQ-16 decode of 64 images: {len(seg)} dispatches, {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms wall, {sum(e - s for _, s, e in seg) / 1e6:.2f} ms of kernel time\n")
print("| # | kernel | us | gap before us |\n|---|---|---|---|")
prev = None
for i, (n, s, e) in enumerate(seg):
    print(f"| {i} | `{short(n)}` | {(e - s) / 1e3:.1f} | {((s - prev) / 1e3 if prev else 0):.1f} |")
    prev = e
PY
rm -rf $OUT/trace_$tag
head -150 $OUT/${tag}_timeline.md
