set -x
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/t3_pytest.log 2>&1; tail -15 gpurun_out/t3_pytest.log
