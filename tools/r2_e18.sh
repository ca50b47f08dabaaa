export PYTHONUNBUFFERED=1 SBM_STAMPS=1 SBM_NAME_TILES=1
mkdir -p gpurun_out
timeout -k 10 300 python tools/profile_ops.py > gpurun_out/e18_stamps_c2.txt 2>&1
