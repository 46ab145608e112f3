#!/bin/bash
# On the GPU box: alternate bench.py between the in-tree library and alignnet-3d_amd/ab/<name>.so (default base), N rounds.
NAME=${1:-base}; N=${2:-3}; shift 2 || true
for i in $(seq $N); do
  for lib in "" "$PWD/alignnet-3d_amd/ab/$NAME.so"; do
    ALIGNNET_HIP_LIB=$lib python bench.py --steps 30 --no-cpu-baseline --no-train-leg "$@" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('${lib:+ab/$NAME}${lib:-tree      }', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['frac'])"
  done
done
