set -u
R=r03
mkdir -p gpurun_out/fin
python bench.py > gpurun_out/fin/${R}_bench.json 2>gpurun_out/fin/bench.err
python bench.py --mode train --train-dtype f32 --steps 50 --warmup 5 > gpurun_out/fin/${R}_bench_train_f32.json 2>/dev/null
python bench.py --mode train --workload dgcnn --points 1024 --steps 10 --warmup 2 > gpurun_out/fin/${R}_bench_train_dgcnn_n1024.json 2>/dev/null
python bench.py --mode train --workload dgcnn --batch 64 --steps 5 --warmup 1 > gpurun_out/fin/${R}_bench_train_dgcnn_n4096_b64.json 2>/dev/null
python bench.py --mode train --workload dgcnn --batch 512 --steps 2 --warmup 1 --sustained-seconds 0 > gpurun_out/fin/${R}_bench_train_dgcnn_n4096_b512.json 2>/dev/null
S="--sustained-seconds 0"
bash tools/profile.sh ${R}_train --mode train --train-dtype f32 --steps 20 --warmup 3 $S > gpurun_out/fin/p_train.log 2>&1
bash tools/profile.sh ${R}_train_dgcnn --mode train --workload dgcnn --points 1024 --steps 6 --warmup 2 $S > gpurun_out/fin/p_train_dgcnn.log 2>&1
mkdir -p gpurun_out/profiles_${R}b
cp gpurun_out/prof_${R}_train/summary/* gpurun_out/prof_${R}_train_dgcnn/summary/* gpurun_out/profiles_${R}b/ 2>/dev/null
cp gpurun_out/fin/${R}_bench*.json gpurun_out/profiles_${R}b/
for t in _train _train_dgcnn; do
  grep '^{' gpurun_out/prof_${R}${t}/bench_trace.log > gpurun_out/profiles_${R}b/${R}${t}_bench_under_rocprof.json 2>/dev/null
done
find gpurun_out -name "*.db" -delete
ls gpurun_out/profiles_${R}b
