set -x
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_cond.py -x -q -m gpu > gpurun_out/t7_cond.log 2>&1; tail -12 gpurun_out/t7_cond.log
timeout -k 10 300 python -m pytest tests/test_gpu_alt.py -x -q -m gpu -k "cond or evaluate" > gpurun_out/t7_alt.log 2>&1; tail -4 gpurun_out/t7_alt.log
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "not chain_barrier" > gpurun_out/t7_r2.log 2>&1; tail -12 gpurun_out/t7_r2.log
timeout -k 10 120 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "chain_barrier" > gpurun_out/t7_abort.log 2>&1; tail -12 gpurun_out/t7_abort.log
