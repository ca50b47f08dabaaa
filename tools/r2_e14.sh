set -x
export PYTHONUNBUFFERED=1 SBM_NAME_TILES=1
mkdir -p gpurun_out
for c in 4 5; do
timeout -k 10 300 python tools/profile_ops.py --config $c > gpurun_out/e14_ops_c${c}_wide.txt 2>&1 || exit 1
SBM_NO_TILE_64x128=1 timeout -k 10 300 python tools/profile_ops.py --config $c > gpurun_out/e14_ops_c${c}_narrow.txt 2>&1 || exit 1
done
