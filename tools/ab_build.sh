#!/bin/bash
# Build the kernels of a git revision (default HEAD) into alignnet-3d_amd/ab/<name>.so for same-box A/B runs:
#   tools/ab_build.sh [rev] [name];   ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/ab/<name>.so python bench.py ...
set -e
REV=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" alignnet-3d_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/alignnet-3d_amd/csrc" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/alignnet-3d_amd/ab"
cp "$TMP/alignnet-3d_amd/libalignnet_hip.so" "$ROOT/alignnet-3d_amd/ab/$NAME.so"
rm -rf "$TMP"
echo "built $ROOT/alignnet-3d_amd/ab/$NAME.so from $REV"
