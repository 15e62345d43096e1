#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) --kernel-trace --stats database as a per-kernel table: calls, total ms, average us, share.

The traced process also creates the engine: weight upload (`__amd_rocclr_copyBuffer`, ~280 ms), torch's RNG / elementwise kernels of
``init_synthetic`` (`at::native::*`), the one-off layout conversions of ``pg_finalize_weights`` (tile_weights / interleave_qk / l2norm / fp32 table
GEMMs).  None of that is the timed step, and left in the table it owned 13-27 % of the "%" column.  By default those rows are
listed SEPARATELY under the table and the percentages are over the step's kernels only; --all keeps the raw view.
usage: rocpd_summary.py results.db [out.md] [--all]"""
import re
import sqlite3
import sys

SETUP = re.compile(r"__amd_rocclr_|at::native|tile_weights_kernel|interleave_qk_kernel|l2norm_rows_kernel|to_f32_kernel|gemm_f32_kernel|fill_bf16_kernel|fill_const_kernel")


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    raw = "--all" in sys.argv[1:]
    db = sqlite3.connect(args[0])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    step = [r for r in rows if raw or not SETUP.search(r[0])]
    setup = [r for r in rows if not raw and SETUP.search(r[0])]
    tot = sum(r[2] for r in step) or 1.0
    lines = ["| kernel | calls | total ms | avg us | % of the step's kernel time |" if not raw else "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in step:
        share = pct if raw else 100.0 * total / tot
        if share < 0.005 and calls < 10:
            continue
        lines.append(f"| `{name[:110]}` | {calls} | {total / 1e3:.2f} | {avg:.2f} | {share:.2f} |")
    if setup:
        lines += ["", f"Setup kernels of the traced process (engine creation / weight upload / synthetic init; NOT part of the step, excluded from the percentages above): "
                  f"{sum(r[1] for r in setup)} launches, {sum(r[2] for r in setup) / 1e3:.1f} ms -- " + ", ".join(f"`{r[0][:40]}` x{r[1]} {r[2] / 1e3:.1f} ms" for r in setup[:6])]
    out = "\n".join(lines) + "\n"
    if len(args) > 1:
        with open(args[1], "a") as f:
            f.write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()
