# Round-end run on one GPU box: the whole GPU suite, then the measurement pipeline (tools/r3_profiles.sh).
set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/final_tests.log 2>&1
rc=$?; tail -4 gpurun_out/final_tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/r3_profiles.sh > gpurun_out/final_profiles.log 2>&1
tail -5 gpurun_out/final_profiles.log
