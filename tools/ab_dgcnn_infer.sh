#!/bin/bash
# On the GPU box: alternate the in-tree library and alignnet-3d_amd/ab/<name>.so on the DGCNN inference lines (N = 4096, 512 pairs/step; exact fp32 and split-bf16), N rounds
NAME=${1:-base}; N=${2:-2}
for i in $(seq $N); do
  for lib in "" "$PWD/alignnet-3d_amd/ab/$NAME.so"; do
    tag="${lib:+ab/$NAME}${lib:-tree}"
    for dt in f32 bf16x3; do
      ALIGNNET_HIP_LIB=$lib python bench.py --workload dgcnn --batch 512 --steps 4 --warmup 1 --infer-dtype $dt --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$tag dgcnn infer $dt', j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_per_step'], r['frac'])"
    done
  done
done
