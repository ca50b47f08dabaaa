# one call: parity tests of the pixel path, then a same-box A/B of the in-tree library against build/base409.so (tools/build_variant.sh)
set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/tests_parity.log
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default base=build/base409.so > $O/ab.txt 2>&1; tail -5 $O/ab.txt
