# Run ON THE GPU BOX (via gpurun): every bench line and rocprofv3 summary that profiles/<round>_* holds, in one call.
# Usage: bash tools/refresh_profiles.sh r02 [lines]
set -u
R=${1:-r04}
mkdir -p gpurun_out/fin
python bench.py > gpurun_out/fin/${R}_bench.json 2>gpurun_out/fin/bench.err
python bench.py --mode train --train-dtype f32 --steps 50 --warmup 5 > gpurun_out/fin/${R}_bench_train_f32.json 2>/dev/null
python bench.py --mode train --train-dtype bf16 --steps 50 --warmup 5 > gpurun_out/fin/${R}_bench_train_bf16.json 2>/dev/null
python bench.py --workload dgcnn --batch 512 --steps 4 --warmup 1 > gpurun_out/fin/${R}_bench_dgcnn.json 2>/dev/null
python bench.py --workload dgcnn --batch 512 --steps 4 --warmup 1 --infer-dtype bf16x3 > gpurun_out/fin/${R}_bench_dgcnn_split.json 2>/dev/null
python bench.py --mode train --workload dgcnn --points 1024 --steps 10 --warmup 2 > gpurun_out/fin/${R}_bench_train_dgcnn_n1024.json 2>/dev/null
python bench.py --mode train --workload dgcnn --batch 64 --steps 5 --warmup 1 > gpurun_out/fin/${R}_bench_train_dgcnn_n4096_b64.json 2>/dev/null
python bench.py --mode train --workload dgcnn --batch 512 --steps 2 --warmup 1 --sustained-seconds 0 > gpurun_out/fin/${R}_bench_train_dgcnn_n4096_b512.json 2>/dev/null   # BASELINE configs[4]'s per-GPU batch
python bench.py --mode train --workload dgcnn --train-dtype bf16 --points 1024 --steps 10 --warmup 2 > gpurun_out/fin/${R}_bench_train_dgcnn_bf16_n1024.json 2>/dev/null
python bench.py --mode train --workload dgcnn --train-dtype bf16 --batch 64 --steps 5 --warmup 1 > gpurun_out/fin/${R}_bench_train_dgcnn_bf16_n4096_b64.json 2>/dev/null
python bench.py --mode train --workload dgcnn --train-dtype bf16 --batch 512 --steps 3 --warmup 1 --sustained-seconds 0 > gpurun_out/fin/${R}_bench_train_dgcnn_bf16_n4096_b512.json 2>/dev/null
if [ "${2:-}" = "lines" ]; then   # bench lines only (no rocprofv3 passes): after a change that leaves the profiled kernels alone
  mkdir -p gpurun_out/profiles_${R}; cp gpurun_out/fin/${R}_bench*.json gpurun_out/profiles_${R}/; ls gpurun_out/profiles_${R}; cut -c1-180 gpurun_out/fin/${R}_bench.json; exit 0
fi
Q="--no-cpu-baseline --no-train-leg --no-split-leg --no-pcie-leg --no-extra-legs --sustained-seconds 0"
S="--sustained-seconds 0"
bash tools/profile.sh ${R} --steps 20 --warmup 3 $Q > gpurun_out/fin/p_${R}.log 2>&1
bash tools/profile.sh ${R}_train --mode train --train-dtype f32 --steps 20 --warmup 3 $S > gpurun_out/fin/p_train.log 2>&1
bash tools/profile.sh ${R}_train_bf16 --mode train --train-dtype bf16 --steps 20 --warmup 3 $S > gpurun_out/fin/p_train_bf16.log 2>&1
bash tools/profile.sh ${R}_split --steps 20 --warmup 3 $Q --infer-dtype bf16x3 > gpurun_out/fin/p_split.log 2>&1
bash tools/profile.sh ${R}_dgcnn --workload dgcnn --batch 512 --steps 4 --warmup 1 $S > gpurun_out/fin/p_dgcnn.log 2>&1
bash tools/profile.sh ${R}_dgcnn_split --workload dgcnn --batch 512 --steps 4 --warmup 1 --infer-dtype bf16x3 $S > gpurun_out/fin/p_dgcnn_split.log 2>&1
bash tools/profile.sh ${R}_train_dgcnn --mode train --workload dgcnn --points 1024 --steps 6 --warmup 2 $S > gpurun_out/fin/p_train_dgcnn.log 2>&1
bash tools/profile.sh ${R}_train_dgcnn_bf16 --mode train --workload dgcnn --train-dtype bf16 --points 1024 --steps 6 --warmup 2 $S > gpurun_out/fin/p_train_dgcnn_bf16.log 2>&1
mkdir -p gpurun_out/profiles_${R}
cp gpurun_out/prof_${R}*/summary/* gpurun_out/profiles_${R}/ 2>/dev/null
cp gpurun_out/fin/${R}_bench*.json gpurun_out/profiles_${R}/
for t in "" _train _train_bf16 _split _dgcnn _dgcnn_split _train_dgcnn _train_dgcnn_bf16; do
  grep '^{' gpurun_out/prof_${R}${t}/bench_trace.log > gpurun_out/profiles_${R}/${R}${t}_bench_under_rocprof.json 2>/dev/null
done
find gpurun_out -name "*.db" -delete
ls gpurun_out/profiles_${R}
cut -c1-180 gpurun_out/fin/${R}_bench.json
