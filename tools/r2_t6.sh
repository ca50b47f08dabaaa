set -x
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_guidance.py -x -q -m gpu > gpurun_out/t6_guid.log 2>&1; tail -15 gpurun_out/t6_guid.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_alt.py tests/test_ddpm_refiner.py tests/test_fid.py -x -q -m gpu > gpurun_out/t6_rest.log 2>&1; tail -5 gpurun_out/t6_rest.log
