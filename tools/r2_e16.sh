set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/e16_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e16_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/e16_c4.json 2> gpurun_out/e16_c4.err || exit 1
SBM_NAME_TILES=1 timeout -k 10 300 python tools/profile_ops.py --config 4 > gpurun_out/e16_ops_c4.txt 2>&1
