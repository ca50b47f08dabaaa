set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "em_update or first_last or advance_and_convert" > $O/tests_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/tests_new.log
timeout -k 10 300 python tools/ab_bench.py --reps 2 new=default nofuse=env:SBM_NO_FUSED_EM=1 > $O/ab.txt 2>&1; tail -4 $O/ab.txt
