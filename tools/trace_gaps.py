#!/usr/bin/env python3
"""Per-layer timeline of the decode loop from a rocprofv3 --kernel-trace database: for every pair of CONSECUTIVE dispatches in the steady-state
loop, the previous kernel's duration and the gap (next start - previous end), grouped by kernel class.  What a dependent launch really costs
on this chip = gap + the part of the next kernel's duration before its first useful byte; this table is the gap half.
usage: trace_gaps.py results.db"""
import collections
import re
import sqlite3
import sys

CLASSES = [("attn", r"attn_decode_fused"), ("rmsnorm", r"rmsnorm"), ("gemm_sk4", r"gemm_sk4"), ("gemm_sk3", r"gemm_skinny3"), ("cfg", r"cfg_"), ("bias_act", r"bias_act")]


def cls(name):
    for c, rx in CLASSES:
        if re.search(rx, name):
            return c
    return "other"


def main():
    db = sqlite3.connect(sys.argv[1])
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    v = "kernels" if "kernels" in views else next(x for x in views if "kernel" in x.lower() and "dispatch" in x.lower())
    cols = [r[1] for r in db.execute(f"pragma table_info({v})")]
    namec = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute(f"select {namec}, start, end from {v} order by start"))
    # steady-state decode loop = the longest run of dispatches whose classes are decode classes
    seq = [(cls(n), s, e, n) for n, s, e in rows]
    gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
    for (c0, s0, e0, n0), (c1, s1, e1, n1) in zip(seq, seq[1:]):
        if c0 == "other" or c1 == "other":
            continue
        g = (s1 - e0) / 1e3
        if g > 50:          # a host stall, not a launch gap
            continue
        gaps[(c0, c1)].append(g); durs[n0[:60]].append((e0 - s0) / 1e3)
    print("| previous -> next | pairs | gap us mean | p10 | p50 | p90 |\n|---|---|---|---|---|---|")
    tot = 0.0; cnt = 0
    for k, g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        g.sort(); n = len(g)
        print(f"| {k[0]} -> {k[1]} | {n} | {sum(g) / n:.2f} | {g[n // 10]:.2f} | {g[n // 2]:.2f} | {g[9 * n // 10]:.2f} |")
        tot += sum(g); cnt += n
    print(f"\nall decode pairs: {cnt} gaps, mean {tot / max(cnt, 1):.2f} us, total {tot / 1e3:.1f} ms")
    print("\n| kernel | launches | avg us |\n|---|---|---|")
    for k, d in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"| `{k}` | {len(d)} | {sum(d) / len(d):.2f} |")


if __name__ == "__main__":
    main()
