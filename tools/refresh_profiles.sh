set -u
mkdir -p gpurun_out/fin
python bench.py > gpurun_out/fin/bench.json 2>gpurun_out/fin/bench.err
python bench.py --mode train --train-dtype f32 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fin/bench_train_f32.json 2>/dev/null
python bench.py --mode train --train-dtype bf16 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fin/bench_train_bf16.json 2>/dev/null
python bench.py --mode train --workload dgcnn --points 1024 --steps 10 --warmup 2 > gpurun_out/fin/bench_train_dgcnn_n1024.json 2>/dev/null
python bench.py --mode train --workload dgcnn --batch 64 --steps 5 --warmup 1 > gpurun_out/fin/bench_train_dgcnn_n4096_b64.json 2>/dev/null
bash tools/profile.sh r01 > gpurun_out/fin/p_r01.log 2>&1
bash tools/profile.sh r01_train --mode train --train-dtype f32 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/fin/p_train.log 2>&1
bash tools/profile.sh r01_train_bf16 --mode train --train-dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/fin/p_train_bf16.log 2>&1
bash tools/profile.sh r01_split --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --infer-dtype bf16x3 > gpurun_out/fin/p_split.log 2>&1
bash tools/profile.sh r01_dgcnn --workload dgcnn --batch 512 --steps 4 --warmup 1 > gpurun_out/fin/p_dgcnn.log 2>&1
bash tools/profile.sh r01_dgcnn_split --workload dgcnn --batch 512 --steps 4 --warmup 1 --infer-dtype bf16x3 > gpurun_out/fin/p_dgcnn_split.log 2>&1
bash tools/profile.sh r01_train_dgcnn --mode train --workload dgcnn --points 1024 --steps 6 --warmup 2 > gpurun_out/fin/p_train_dgcnn.log 2>&1
find gpurun_out -name "*.db" -delete
ls gpurun_out
cut -c1-180 gpurun_out/fin/bench.json
