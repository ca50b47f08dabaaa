set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04m; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/dbg_attnblock.py 2>&1 | grep attnblock | tail -2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/tests_parity.log
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default noblock=env:SBM_ATTN_BLOCK=0 > $O/ab.txt 2>&1; tail -6 $O/ab.txt
