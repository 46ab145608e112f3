# every launch of the kernels matching <pattern> in the last training step of a rocprofv3 rocpd database (microseconds, in launch order)
# usage: kstat_each.py <dir> <pattern>
import sqlite3, glob, sys
d, pat = sys.argv[1], sys.argv[2]
c = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
rows = list(c.execute("select name,(end-start) from kernels order by start"))
ends = [i for i, (n, _) in enumerate(rows) if 'adam_kernel' in n]
lo, hi = ends[-2] + 1, ends[-1] + 1
for n, dur in rows[lo:hi]:
    if pat in n:
        print("%-70s %9.1f us" % (n.split('(')[0].replace('alignnet::', '').replace('void ', '')[:70], dur / 1e3))
