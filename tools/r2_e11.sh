set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 300 python tools/profile_ops.py --config 4 > gpurun_out/e11_ops_c4.txt 2>&1 &&
timeout -k 10 300 python tools/profile_ops.py --config 5 > gpurun_out/e11_ops_c5.txt 2>&1
tail -n 3 gpurun_out/e11_ops_c5.txt
