set -x
export PYTHONUNBUFFERED=1
export SBM_CHAIN_WG2=1
for args in "67 16 8" "256 16 64"; do
  timeout -k 10 120 python tools/chain_check.py $args 2>&1 | grep "e-0" | tail -1 | tr ' ' '\n' | awk '{ if ($1+0 > 1e-4) bad++; n++ } END{print "chain_check WG2: " bad+0 " bad of " n}'
done
timeout -k 10 300 python tools/kern_mix.py 2>&1 | grep -v amdgpu.ids | grep "chain\|row\|total"
timeout -k 10 180 python tools/chain_stamps.py 256 > gpurun_out/e9_stamps_wg2.log 2>&1; tail -1 gpurun_out/e9_stamps_wg2.log
