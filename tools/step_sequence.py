#!/usr/bin/env python3
"""The launch sequence of one step from a rocprofv3 --kernel-trace database: kernel, duration, gap to the previous kernel's end (us).
usage: tools/step_sequence.py <rocprofv3 output dir> <first kernel of a step (substring)> [which step, default the last complete one]"""
import sqlite3, glob, sys
d, first = sys.argv[1], sys.argv[2]
c = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
rows = [(n.split('(')[0].replace('alignnet::', '').replace('void ', ''), s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
starts = [i for i, r in enumerate(rows) if first in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -2
lo, hi = starts[k], starts[k + 1]
prev = None
tot = 0.0
for i, (n, s, e) in enumerate(rows[lo:hi]):
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%3d %-58s %8.2f us   gap %6.2f" % (i, n[:58], (e - s) / 1e3, gap))
    prev = e; tot += (e - s) / 1e3
print("launches %d, kernel time %.1f us, span %.1f us" % (hi - lo, tot, (rows[hi - 1][2] - rows[lo][1]) / 1e3))
