#!/usr/bin/env python3
"""One line per rocprofv3 --kernel-trace run (rocpd .db) of a one-frame-per-launch stream: mean duration of the front kernels and of the integrate kernel over
the last N frames, the integrate kernel's start-to-start period, and how long after the integrate kernel's start the compaction of the NEXT frame ended."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(db.execute("select %s, start, end from kernels order by start" % name))
pick = lambda key: [(s, e) for k, s, e in rows if key in k]
integ, alloc, comp, pre = pick("k_integrate"), pick("k_alloc"), pick("k_compactify"), pick("k_prepass")
mean = lambda v: sum(v) / max(1, len(v))
dur = lambda xs: mean([(e - s) / 1e3 for s, e in xs[-n:]])
period = mean([(b[0] - a[0]) / 1e3 for a, b in zip(integ[-n - 1:], integ[-n:])])
# for each of the last integrate launches: when did the last compaction that started during it end, relative to its start
lag = []
for s, e in integ[-n:-1]:
    c = [ce for cs, ce in comp if s <= cs <= e + 500e3 and ce > s]
    if c: lag.append((c[0] - s) / 1e3)
print("integrate %.0f us | alloc %.0f | compactify %.0f | prepass %.0f | period %.0f us = %.1f frames/s | next frame's compaction done %.0f us after the integrate kernel's start"
      % (dur(integ), dur(alloc), dur(comp), dur(pre), period, 1e6 / period, mean(lag)))
