#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel trace of tools/ab_step.py (one option set) and the per-kernel table of the last steps.
# usage: tools/trace_step.sh <tag> [ab_step args...]     -> gpurun_out/trace_<tag>/ + gpurun_out/trace_<tag>.txt
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format rocpd -d $OUT -o t -- python $REPO/tools/ab_step.py --rounds 1 --steps 30 "$@" > $OUT/run.log 2>&1
cd $REPO
python tools/kstat_all.py $OUT 20 > gpurun_out/trace_$TAG.txt 2>&1
find $OUT -name "*.db" -delete
cat gpurun_out/trace_$TAG.txt
