set -x
export PYTHONUNBUFFERED=1
for G in 2 1; do
  export SBM_CHAIN_GROUPS=$G
  echo "=== GROUPS=$G ===" 
  timeout -k 10 120 python tools/chain_check.py 67 16 8 > gpurun_out/e1_check_g$G.log 2>&1 || echo "chain_check failed G=$G"
  tail -3 gpurun_out/e1_check_g$G.log
  timeout -k 10 180 python tools/chain_stamps.py 256 > gpurun_out/e1_stamps_g$G.log 2>&1 || echo "stamps failed"
  tail -2 gpurun_out/e1_stamps_g$G.log
  timeout -k 10 180 python tools/microbatch_probe.py --groups 1 --n 300 > gpurun_out/e1_mb_g$G.log 2>&1 || echo "mb failed"
  tail -2 gpurun_out/e1_mb_g$G.log
done
export SBM_CHAIN_GROUPS=2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or full_size or determinism or trajectory" > gpurun_out/e1_pytest.log 2>&1; tail -5 gpurun_out/e1_pytest.log
