set -x
export PYTHONUNBUFFERED=1
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_default.txt 2>&1
SBM_LIB=build/noload.so PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_noload.txt 2>&1
SBM_LIB=build/nomfma.so PIX_PHASES=1 timeout -k 10 200 python tools/pix_stamps.py > $O/stamps_nomfma.txt 2>&1
timeout -k 10 600 python tools/ab_bench.py --reps 1 default=default noload=build/noload.so nomfma=build/nomfma.so > $O/ab.txt 2>&1
cat $O/ab.txt
