#!/bin/bash
# Run a command on the MI355X box through gpurun, stamping the tree's commit into .build_commit first (.git does not travel;
# tools/summarize_prof.py records the stamp in every profile summary so that bench.py can say which tree a quoted PMC figure is from).
# usage: tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
c=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' || c="$c+dirty"
echo "$c" > .build_commit
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
