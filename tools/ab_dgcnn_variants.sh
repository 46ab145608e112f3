for lib in base pc0 lp1 lp3 base pc0 lp1 lp3; do
ALIGNNET_DBG=0 ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/ab/$lib.so python bench.py --workload dgcnn --batch 512 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['frac'])"
done
ALIGNNET_DBG=64 ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/ab/lp3.so python bench.py --workload dgcnn --batch 512 --steps 1 --warmup 0 2>&1 | grep slot-5 | head -3
