set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/${OUT:-r04f}; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps.txt 2>&1; grep total $O/stamps.txt
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default novec=env:SBM_PIX_VEC=0 nofuse=env:SBM_NO_FUSED_CVT=1 > $O/ab.txt 2>&1; cat $O/ab.txt
