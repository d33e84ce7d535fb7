#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run (rocpd .db): for a window of dispatches, start / end / duration per kernel and queue, relative to the
first one -- who overlaps whom, how long the gaps between consecutive launches of the same kernel are.

  python tools/timeline.py kt_results.db [first_dispatch_of_window] [count] [--skip NAME ...]
(a negative first counts from the end; --skip drops kernels whose name contains NAME before the window is cut, e.g. the synthetic renderer)
"""
import sqlite3
import sys


def main():
    argv = list(sys.argv)
    skip = []
    while "--skip" in argv:
        i = argv.index("--skip")
        skip.append(argv[i + 1])
        del argv[i:i + 2]
    db = sqlite3.connect(argv[1])
    first = int(argv[2]) if len(argv) > 2 else None
    count = int(argv[3]) if len(argv) > 3 else 60
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    rows = list(db.execute("select %s, start, end%s from kernels order by start" % (name, (", " + qcol) if qcol else "")))
    rows = [r for r in rows if not any(k in r[0] for k in skip)]
    if first is not None and first < 0:
        first = max(0, len(rows) + first)
    if first is not None and first >= len(rows):
        first = max(0, len(rows) - count)
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    integ = [i for i, r in enumerate(rows) if "k_integrate<1," in r[0]]
    if first is None:
        first = integ[len(integ) // 2] - 4 if integ else 0
    t0 = rows[first][1]
    print("%-36s %6s %10s %10s %9s" % ("kernel", "queue", "start_us", "end_us", "dur_us"))
    for r in rows[first:first + count]:
        print("%-36s %6s %10.1f %10.1f %9.1f" % (short(r[0]), r[3] if qcol else "-", (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3))
    # gaps between consecutive integrate launches over the whole run
    g = [(rows[b][1] - rows[a][2]) / 1e3 for a, b in zip(integ, integ[1:])]
    d = [(rows[i][2] - rows[i][1]) / 1e3 for i in integ]
    if g:
        g2 = sorted(g)
        print("\nk_integrate<1,..>: %d launches, duration mean %.1f us; gap to the next launch: median %.1f us, mean %.1f us, p90 %.1f us"
              % (len(d), sum(d) / len(d), g2[len(g2) // 2], sum(g) / len(g), g2[int(len(g2) * 0.9)]))


if __name__ == "__main__":
    main()
