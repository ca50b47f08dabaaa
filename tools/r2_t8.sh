set -x
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/t8_r2.log 2>&1; tail -25 gpurun_out/t8_r2.log | cut -c1-300
for C in 2 4 5; do
  timeout -k 10 400 python bench.py --config $C --steps 2 --warmup 1 > gpurun_out/t8_bench_c$C.json 2> gpurun_out/t8_bench_c$C.err || echo "bench config $C failed"
  tail -3 gpurun_out/t8_bench_c$C.err; cut -c1-600 gpurun_out/t8_bench_c$C.json
done
