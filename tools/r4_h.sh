set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04h; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps.txt 2>&1; grep total $O/stamps.txt
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default prev=build/prev.so > $O/ab.txt 2>&1; cat $O/ab.txt
