# per-launch durations of one kernel in launch order (rocpd database): usage kstat_stage.py <dir> <substr> [period]
import sqlite3, glob, sys
d, sub = sys.argv[1], sys.argv[2]; period = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
v = [dur / 1e3 for n, dur in c.execute("select name,(end-start) from kernels order by start") if sub in n]
v = v[len(v) % period:]
for s in range(period):
    x = v[s::period][-10:]
    print("launch %d of each step: mean %.1f us  min %.1f  (last %d)" % (s, sum(x) / len(x), min(x), len(x)))
