#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --no-split-leg --no-pcie-leg --no-extra-legs}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format rocpd -d $OUT/pmc_$N -o pmc -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_$N.log 2>&1
  echo "pmc $C rc=$?"
done
cd $REPO
python tools/summarize_prof.py $OUT $TAG
