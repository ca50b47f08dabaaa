set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04i; rm -rf $O; mkdir -p $O
SBM_LIB=build/stamp_epi.so PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_epi.txt 2>&1; grep total $O/stamps_epi.txt
SBM_LIB=build/stamp_pre.so PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_pre.txt 2>&1; grep total $O/stamps_pre.txt
timeout -k 10 400 python tools/ab_bench.py --reps 1 new=default prev=build/prev.so > $O/ab.txt 2>&1; cat $O/ab.txt
