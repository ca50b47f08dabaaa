set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
SBM_LIB=build/stamp_vec.so timeout -k 10 200 python tools/vec_stamps.py > $O/vec_stamps.txt 2>&1; tail -14 $O/vec_stamps.txt
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default prev=build/prev.so > $O/ab.txt 2>&1; cat $O/ab.txt
