#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel time statistics and, when present, PMC counter sums.

  python tools/rocpd_summary.py [--tail KERNEL N] gpurun_out/prof/kt/kt_results.db [more.db ...] > profiles/rNN_summary.txt
"""
import sqlite3
import sys


def kernel_stats(db):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
    q = "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("%-72s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, c, s, a, mn, mx in rows:
        print("%-72s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (n[:72], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


def pmc_stats(db):
    try:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    except sqlite3.Error:
        return
    if not cols:
        return
    kn = "kernel_name" if "kernel_name" in cols else "name"
    cn = "counter_name" if "counter_name" in cols else "pmc_name"
    vn = "value" if "value" in cols else "counter_value"
    try:
        rows = list(db.execute("select %s, %s, count(*), sum(%s), avg(%s) from counters_collection group by 1,2 order by 1,2" % (kn, cn, vn, vn)))
    except sqlite3.Error as e:
        print("pmc query failed:", e, cols)
        return
    if rows:
        print("\n%-72s %-14s %8s %16s %14s" % ("kernel", "counter", "samples", "sum", "avg/dispatch"))
        for k, c, n, s, a in rows:
            print("%-72s %-14s %8d %16.1f %14.2f" % (k[:72], c, n, s, a))


def tail_stats(db, kernel, n):
    """Average duration of the LAST n dispatches of a kernel (the timed region of bench.py: warm-up launches come first)."""
    rows = [d for (d,) in db.execute("select end-start from kernels where name like ? order by start", ("%" + kernel + "%",))]
    if len(rows) >= n > 0:
        t = rows[-n:]
        print("last %d dispatches of %s: avg %.2f us, total %.1f us (all %d dispatches: avg %.2f us)" %
              (n, kernel, sum(t) / n / 1e3, sum(t) / 1e3, len(rows), sum(rows) / len(rows) / 1e3))


def slice_stats(db, kernel, start, count):
    """Average duration of dispatches [start, start + count) of a kernel, in launch order."""
    rows = [d for (d,) in db.execute("select end-start from kernels where name like ? order by start", ("%" + kernel + "%",))]
    t = rows[start:start + count]
    if t:
        print("dispatches %d..%d of %s: avg %.2f us, total %.1f us" % (start, start + len(t) - 1, kernel, sum(t) / len(t) / 1e3, sum(t) / 1e3))


args = sys.argv[1:]
slices = []
while "--slice" in args:
    i = args.index("--slice")
    slices.append((args[i + 1], int(args[i + 2]), int(args[i + 3])))
    del args[i:i + 4]
tails = []
while "--tail" in args:
    i = args.index("--tail")
    tails.append((args[i + 1], int(args[i + 2])))
    del args[i:i + 3]
for path in args:
    print("== %s" % path)
    db = sqlite3.connect(path)
    kernel_stats(db)
    for k, n in tails:
        tail_stats(db, k, n)
    for k, a, c in slices:
        slice_stats(db, k, a, c)
    pmc_stats(db)
    print()
