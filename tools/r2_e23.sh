set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/e23_tests.log 2>&1
rc=$?; tail -5 gpurun_out/e23_tests.log; [ $rc -eq 0 ] || exit $rc
for c in 2 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e23_c$c.json 2> gpurun_out/e23_c$c.err || exit 1
done
