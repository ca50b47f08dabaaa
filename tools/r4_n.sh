set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/dbg_attnblock.py 2>&1 | grep attnblock | tail -2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/tests_gpu.log
timeout -k 10 300 python tools/ab_bench.py --reps 1 new=default noblock=env:SBM_ATTN_BLOCK=0 > $O/ab.txt 2>&1; tail -3 $O/ab.txt
