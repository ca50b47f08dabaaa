set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
run() { tag=$1; shift; for c in $CFGS; do env "$@" timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/e13_${tag}_c$c.json 2>> gpurun_out/e13.err || exit 1; done; }
CFGS="4 5 2" run base X=1
CFGS="4 5" run t1024 SBM_SPLITK_TARGET=1024
CFGS="4 5 2" run min4 SBM_SPLITK_MINCH=4
CFGS="4 5" run min8 SBM_SPLITK_MINCH=8
CFGS="4 5" run no64x128 SBM_NO_TILE_64x128=1
CFGS="4 5 2" run tiles768 SBM_SPLITK_TILES=768 SBM_SPLITK_TARGET=1024
CFGS="4 5" run t768 SBM_SPLITK_TARGET=768
