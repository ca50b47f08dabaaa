# Round-4 opening run: sanity of the housekeeping build, the P10 recipe on the legacy / pixel paths, configs 4/5 with the corrector.
set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > $O/tests_round2.log 2>&1; tail -3 $O/tests_round2.log
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; tail -2 $O/c2.err
# P10 (Unet(128, ch=10)): legacy path, then the pixel kernel for <= 4x4 with the 8x8 level on the per-layer kernels, then everything <= 8x8
timeout -k 10 300 python bench.py --config 10 --steps 2 --warmup 1 --no-cpu-baseline > $O/p10_legacy.json 2> $O/p10_legacy.err; tail -2 $O/p10_legacy.err
SBM_PIX_VERBOSE=1 SBM_CHAIN_PIX=16 SBM_PIX_MAX_MB=1000 timeout -k 10 300 python bench.py --config 10 --steps 2 --warmup 1 --no-cpu-baseline > $O/p10_pix16.json 2> $O/p10_pix16.err; tail -2 $O/p10_pix16.err
SBM_PIX_VERBOSE=1 SBM_PIX_MAX_MB=1000 timeout -k 10 300 python bench.py --config 10 --steps 2 --warmup 1 --no-cpu-baseline > $O/p10_pix64.json 2> $O/p10_pix64.err; tail -2 $O/p10_pix64.err
timeout -k 10 300 python bench.py --config 10 --pc --steps 1 --warmup 1 --no-cpu-baseline > $O/p10_legacy_pc.json 2> $O/p10_legacy_pc.err; tail -2 $O/p10_legacy_pc.err
for C in 4 5; do
  timeout -k 10 400 python bench.py --config $C --extra --steps 1 --warmup 1 --no-cpu-baseline > $O/c${C}_extra.json 2> $O/c${C}_extra.err; tail -2 $O/c${C}_extra.err
done
ls -la $O
