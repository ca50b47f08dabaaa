set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_alt.py tests/test_ddpm_refiner.py tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/e21_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e21_tests.log; [ $rc -eq 0 ] || exit $rc
SBM_NAME_TILES=1 timeout -k 10 500 python tools/profile_ops.py --config 6 --batch 64 --reps 3 > gpurun_out/e21_ops_refiner.txt 2>&1
tail -2 gpurun_out/e21_ops_refiner.txt
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/e21_c5.json 2> gpurun_out/e21_c5.err || exit 1
