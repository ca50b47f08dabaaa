set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py tests/test_gpu_alt.py -x -q -m gpu > gpurun_out/e24_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e24_tests.log; [ $rc -eq 0 ] || exit $rc
for c in 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e24_c$c.json 2> gpurun_out/e24_c$c.err || exit 1
done
timeout -k 10 300 python tools/profile_ops.py --config 4 > gpurun_out/e24_ops_c4.txt 2>&1
