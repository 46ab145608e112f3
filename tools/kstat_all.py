# per-kernel microseconds per training step from a rocprofv3 rocpd database (last N steps); usage: kstat_all.py <dir> [steps]
import sqlite3, glob, sys, collections
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
rows = list(c.execute("select name,(end-start) from kernels order by start"))
names = [n for n, _ in rows]
# a step = the span between consecutive adam_kernel launches
ends = [i for i, n in enumerate(names) if 'adam_kernel' in n]
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
per = collections.defaultdict(lambda: [0.0, 0])
for n, dur in rows[lo:hi]:
    k = n.split('(')[0].replace('alignnet::', '').replace('void ', '')
    per[k][0] += dur / 1e3; per[k][1] += 1
tot = sum(v[0] for v in per.values())
print("%-40s %9s %7s %8s" % ("kernel", "us/step", "n/step", "avg us"))
for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("%-40s %9.1f %7.1f %8.1f" % (k[:40], t / steps, n / steps, t / n))
print("total kernel us/step %.1f, launches/step %.1f" % (tot / steps, sum(v[1] for v in per.values()) / steps))
