#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` database.

SQ_VALU_MFMA_BUSY_CYCLES is reported summed over the 32 SQ instances (8 XCDs x 4 shader engines); one instance's value is the
busy cycles of a SIMD's MFMA pipe when the SIMDs are evenly loaded (calibrated in round 1 on the 256x256 GEMM: value / 32 =
MFMAs per SIMD x 16 cycles exactly, profiles/r01_e_gemm256_counters.md).  Utilisation = (value / 32) / GRBM_GUI_ACTIVE, i.e. the
share of the cycles the chip actually ran; x effective clock / 2.4 GHz = share of the nominal 2.5 PFLOP/s peak.
usage: pmc_mfma.py results.db pattern [pattern ...]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
print("| kernel | launches | avg us | MFMA busy cycles / SIMD | GRBM_GUI_ACTIVE | MFMA util (of cycles run) | eff. clock GHz | of 2.4 GHz nominal peak |")
print("|---|---|---|---|---|---|---|---|")
for pat in sys.argv[2:]:
    rows = con.execute("select counter_name, avg(counter_value), count(*), avg(duration) from pmc_events where name like ? group by counter_name", (f"%{pat}%",)).fetchall()
    d = {r[0]: r for r in rows}
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        print(f"| `{pat}` | not found | | | | | | |"); continue
    mf, n, dur = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 32.0, d["SQ_VALU_MFMA_BUSY_CYCLES"][2], d["SQ_VALU_MFMA_BUSY_CYCLES"][3]
    gui = d["GRBM_GUI_ACTIVE"][1]
    clk = gui / dur if dur else 0.0
    print(f"| `{pat}` | {n} | {dur / 1e3:.1f} | {mf:,.0f} | {gui:,.0f} | {mf / gui:.3f} | {clk:.2f} | {mf / gui * clk / 2.4:.3f} |")
