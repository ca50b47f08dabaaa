set -x
export PYTHONUNBUFFERED=1 SBM_NAME_TILES=1
mkdir -p gpurun_out
timeout -k 10 500 python tools/profile_ops.py --config 6 --batch 64 --reps 3 > gpurun_out/e20_ops_refiner.txt 2>&1
tail -3 gpurun_out/e20_ops_refiner.txt
