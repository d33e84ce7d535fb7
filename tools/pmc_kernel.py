#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database:  python tools/pmc_kernel.py results.db [kernel substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = "%" + (sys.argv[2] if len(sys.argv) > 2 else "k_integrate") + "%"
rows = db.execute("select counter_name, count(*), avg(value), sum(value) from counters_collection where kernel_name like ? group by 1 order by 1", (pat,))
for name, n, avg, tot in rows:
    print("%-32s dispatches %6d  avg/dispatch %16.1f  sum %18.1f" % (name, n, avg, tot))
