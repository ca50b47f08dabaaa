set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py tests/test_gpu_alt.py tests/test_ddpm_refiner.py -x -q -m gpu > gpurun_out/e12_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e12_tests.log; [ $rc -eq 0 ] || exit $rc
for c in 2 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e12_c$c.json 2> gpurun_out/e12_c$c.err || exit 1
done
SBM_NO_SPLITK=1 timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/e12_c2_nosplit.json 2>> gpurun_out/e12_c2.err || exit 1
timeout -k 10 300 python tools/profile_ops.py --config 5 > gpurun_out/e12_ops_c5.txt 2>&1
timeout -k 10 300 python tools/profile_ops.py --config 4 > gpurun_out/e12_ops_c4.txt 2>&1
timeout -k 10 300 python tools/profile_ops.py > gpurun_out/e12_ops_c2.txt 2>&1
