#!/usr/bin/env python3
"""Average every collected counter per kernel-name pattern from rocprofv3 rocpd databases.
usage: pmc_dump.py pattern db1 [db2 ...]"""
import sqlite3, sys
pat = sys.argv[1]
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    rows = con.execute("select counter_name, avg(counter_value), count(*), avg(duration) from pmc_events where name like ? group by counter_name", (f"%{pat}%",)).fetchall()
    for name, val, n, dur in rows:
        print(f"{name:28s} {val:16.1f}   (n={n}, avg kernel ns {dur:.0f})")
