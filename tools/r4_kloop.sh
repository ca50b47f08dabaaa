set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
SBM_PIX_VEC=0 PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_novec.txt 2>&1; grep total $O/stamps_novec.txt
timeout -k 10 300 python tools/ab_bench.py --reps 2 vec=default novec=env:SBM_PIX_VEC=0 > $O/ab.txt 2>&1; cat $O/ab.txt
