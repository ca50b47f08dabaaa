set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/tests_parity.log
SBM_LIB=build/stamp_vec.so timeout -k 10 300 python tools/vec_stamps.py 2>&1 | tail -10
timeout -k 10 300 python tools/ab_bench.py --reps 2 new=default > $O/ab.txt 2>&1; tail -3 $O/ab.txt
