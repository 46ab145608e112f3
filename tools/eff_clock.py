#!/usr/bin/env python3
"""Effective shader clock per kernel from one rocprofv3 run with `--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`
(MI355X_MICROARCH.md, DVFS: effective clock = GRBM_GUI_ACTIVE / kernel wall time; the counter is summed over the 8 XCDs).
usage: tools/eff_clock.py <rocprofv3 output dir> <kernel substring>[,<substring>...]"""
import sqlite3, glob, sys, collections
d, subs = sys.argv[1], sys.argv[2].split(',')
db = glob.glob(d + '/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("counters_collection columns:", cols, file=sys.stderr)
have_t = "start" in cols and "end" in cols
q = "select kernel_name, counter_name, value, dispatch_id" + (", start, end" if have_t else "") + " from counters_collection"
per = collections.defaultdict(dict)
for row in c.execute(q):
    name = row[0].split('(')[0].replace('alignnet::', '').replace('void ', '')
    if not any(s in name for s in subs): continue
    e = per[(name, row[3])]
    e[row[1]] = e.get(row[1], 0.0) + row[2]
    if have_t: e["_dur"] = (row[5] - row[4]) / 1e3
if not have_t:   # durations from the kernel trace of the same run, by dispatch id
    for did, name, s, e in c.execute("select dispatch_id, name, start, end from kernels"):
        k = (name.split('(')[0].replace('alignnet::', '').replace('void ', ''), did)
        if k in per: per[k]["_dur"] = (e - s) / 1e3
agg = collections.defaultdict(list)
for (name, did), e in per.items():
    if "_dur" in e and "GRBM_GUI_ACTIVE" in e: agg[name].append(e)
for name, lst in sorted(agg.items()):
    lst = lst[len(lst) // 2:]
    n = len(lst)
    dur = sum(e["_dur"] for e in lst) / n
    gui = sum(e["GRBM_GUI_ACTIVE"] for e in lst) / n
    mf = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for e in lst) / n
    sq = sum(e.get("SQ_BUSY_CYCLES", 0.0) for e in lst) / n
    clk = gui / 8 / dur / 1e3
    print("%-44s n %4d  %8.1f us  GRBM_GUI_ACTIVE/8 %9.0f -> %.3f GHz  MFMA busy cycles / (1024 SIMDs x cycles) %.3f  (SQ_BUSY_CYCLES %.0f)"
          % (name, n, dur, gui / 8, clk, mf / 1024 / max(gui / 8, 1), sq))
