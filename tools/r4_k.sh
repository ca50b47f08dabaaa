set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04k; rm -rf $O; mkdir -p $O
SBM_LIB=build/stamp_vec.so timeout -k 10 200 python tools/vec_stamps.py > $O/vec_stamps.txt 2>&1; cat $O/vec_stamps.txt | tail -32
