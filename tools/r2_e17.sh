set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/e17_${tag}.json 2>> gpurun_out/e17.err || exit 1; }
run base X=1
run nst2 SBM_GEMM_NST64=2
run nst3 SBM_GEMM_NST64=3
run nst2_split SBM_GEMM_NST64=2 SBM_SPLITK_TILES=768 SBM_SPLITK_TARGET=1024
run nst3_split SBM_GEMM_NST64=3 SBM_SPLITK_TILES=768 SBM_SPLITK_TARGET=1024
