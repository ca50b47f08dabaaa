set -x
export PYTHONUNBUFFERED=1
timeout -k 10 120 python tools/chain_check.py 67 16 8 > gpurun_out/e2_check.log 2>&1 || echo "chain_check failed"
tail -3 gpurun_out/e2_check.log
timeout -k 10 120 python tools/chain_opcheck.py 67 8 > gpurun_out/e2_opcheck.log 2>&1 || echo "opcheck failed"
tail -3 gpurun_out/e2_opcheck.log
timeout -k 10 180 python tools/chain_stamps.py 256 > gpurun_out/e2_stamps.log 2>&1 || echo "stamps failed"
tail -2 gpurun_out/e2_stamps.log
timeout -k 10 180 python tools/microbatch_probe.py --groups 1 --n 300 > gpurun_out/e2_mb.log 2>&1 || echo "mb failed"
tail -2 gpurun_out/e2_mb.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/e2_pytest.log 2>&1; tail -5 gpurun_out/e2_pytest.log
