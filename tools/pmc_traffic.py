#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
per launch, next to the algorithmic K/V bytes of the same launches.

usage: pmc_traffic.py fetch.db write.db --batch 64 --prompt-len 256 --tokens 24
The bench's synthetic prompts are deterministic (seed 0), so the algorithmic bytes of the profiled
launches are recomputed here.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read -> doubled.
"""
import argparse
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def avg_counter(db, counter, pattern):
    con = sqlite3.connect(db)
    rows = list(con.execute("select counter_value from pmc_events where counter_name=? and name like ?", (counter, f"%{pattern}%")))
    vals = [r[0] for r in rows]
    return (sum(vals) / len(vals) if vals else float("nan")), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=24)
    ap.add_argument("--kernel", default="attn_decode_fused_kernel")
    ap.add_argument("--json", default=None, help="write the ratio + provenance (kernel symbol, source hash, commit, command) here")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    from bench import synth_prompts
    ids, mask = synth_prompts(a.batch, a.prompt_len, 102400, 100002, 0)
    lens = mask.sum(-1).tolist()
    shared = lens[1] if all(l == lens[1] for l in lens[1::2]) else 0
    keys = []
    for n in range(a.tokens - 1):
        k = shared
        for r, l in enumerate(lens):
            k += (l + n + 1) - (shared if (shared and r % 2) else 0)
        keys.append(k)
    algo = sum(keys) / len(keys) * 16 * 128 * 2 * 2
    f_kb, nf = avg_counter(a.fetch_db, "FETCH_SIZE", a.kernel)
    w_kb, nw = avg_counter(a.write_db, "WRITE_SIZE", a.kernel)
    print(f"kernel {a.kernel}: {nf} launches (FETCH pass), {nw} (WRITE pass)")
    print(f"algorithmic K/V bytes per launch      : {algo / 1e6:9.2f} MB")
    print(f"FETCH_SIZE raw (KB*1024) per launch    : {f_kb * 1024 / 1e6:9.2f} MB")
    print(f"FETCH_SIZE x2 (gfx950 correction)      : {2 * f_kb * 1024 / 1e6:9.2f} MB   ratio to algorithmic {2 * f_kb * 1024 / algo:.3f}")
    print(f"WRITE_SIZE raw per launch (uncalibrated): {w_kb * 1024 / 1e6:9.2f} MB")
    print(f"traffic (2*FETCH + WRITE)              : {(2 * f_kb + w_kb) * 1024 / 1e6:9.2f} MB")
    if a.json:
        import json
        import subprocess
        from bench import kernel_src_sha
        con = sqlite3.connect(a.fetch_db)
        sym = con.execute("select name from pmc_events where name like ? limit 1", (f"%{a.kernel}%",)).fetchone()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = ""
        json.dump({"kernel": a.kernel, "kernel_symbol": sym[0] if sym else a.kernel, "kernel_src_sha": kernel_src_sha(), "commit": commit,
                   "traffic_per_algorithmic_byte": (2 * f_kb + w_kb) * 1024 / algo, "algorithmic_mb_per_launch": algo / 1e6,
                   "fetch_x2_mb": 2 * f_kb * 1024 / 1e6, "write_mb": w_kb * 1024 / 1e6, "launches": nf, "command": a.cmd,
                   "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated"},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
