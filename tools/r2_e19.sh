set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_cond.py tests/test_gpu_guidance.py -x -q -m gpu > gpurun_out/e19_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e19_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/e19_c2.json 2> gpurun_out/e19_c2.err || exit 1
SBM_NO_FUSED_ADVANCE=1 timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/e19_c2_noadv.json 2>> gpurun_out/e19_c2.err || exit 1
