set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python tools/refiner_check.py --batch 64 > gpurun_out/e22_refiner.txt 2>&1
cat gpurun_out/e22_refiner.txt | tail -4
