set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/tests_parity.log
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default notouch=env:SBM_PIX_TOUCH=0 > $O/ab.txt 2>&1; tail -5 $O/ab.txt
PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps.txt 2>&1; grep -n "total\| dw " $O/stamps.txt | head -24
