set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04g; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps.txt 2>&1; grep total $O/stamps.txt
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default prev=build/prev.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 500 python tools/ab_bench.py --config 4 --reps 1 c4=default c4bal40=env:SBM_SPLITK_BALANCE=40 c4bal60=env:SBM_SPLITK_BALANCE=60 > $O/ab4.txt 2>&1; cat $O/ab4.txt
timeout -k 10 400 python tools/ab_bench.py --config 5 --reps 1 c5=default c5bal40=env:SBM_SPLITK_BALANCE=40 > $O/ab5.txt 2>&1; cat $O/ab5.txt
