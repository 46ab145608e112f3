#!/bin/bash
# ON THE GPU BOX: like trace_step.sh, then every launch of the last step in order (name, grid, us)
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format rocpd -d $OUT -o t -- python $REPO/tools/ab_step.py --rounds 1 --steps 30 "$@" > $OUT/run.log 2>&1
cd $REPO
python - $OUT <<'PY' > gpurun_out/trace_${TAG}_each.txt
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = list(c.execute("select name, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z, start, end from kernels order by start"))
import os
mark = os.environ.get('STEP_MARK', 'adam_kernel')
ends = [i for i, r in enumerate(rows) if mark in r[0]]
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = rows[lo][3]
for n, g, wg, s, e in rows[lo:hi]:
    print("%9.1f %8.1f  wgs %6d x %4d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g // max(wg, 1), wg, n.split('(')[0].replace('alignnet::', '').replace('void ', '')[:70]))
print("step span %.1f us, kernel sum %.1f us" % ((rows[hi - 1][4] - t0) / 1e3, sum(e - s for _, _, _, s, e in rows[lo:hi]) / 1e3))
PY
find $OUT -name "*.db" -delete
cat gpurun_out/trace_${TAG}_each.txt
