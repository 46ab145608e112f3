#!/bin/bash
# On the GPU box: alternate the in-tree library and alignnet-3d_amd/ab/<name>.so on three bench lines (inference headline, split-bf16 leg off; bf16 and
# fp32 training), N rounds:  tools/ab_bench3.sh <name> [N]
NAME=${1:-base}; N=${2:-2}
for i in $(seq $N); do
  for lib in "" "$PWD/alignnet-3d_amd/ab/$NAME.so"; do
    tag="${lib:+ab/$NAME}${lib:-tree}"
    ALIGNNET_HIP_LIB=$lib python bench.py --steps 30 --no-cpu-baseline --no-train-leg --no-split-leg --no-pcie-leg --no-extra-legs --sustained-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tag infer', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['frac'])"
    for dt in bf16 f32; do
      ALIGNNET_HIP_LIB=$lib python bench.py --mode train --train-dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tag train $dt', j['value'], j['ms_per_step'])"
    done
  done
done
