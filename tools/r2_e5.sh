set -x
export PYTHONUNBUFFERED=1
timeout -k 10 200 python tools/row_stamps.py 2>&1 | grep -v amdgpu.ids | tail -30
for args in "67 16 8" "256 16 64"; do
  timeout -k 10 120 python tools/chain_check.py $args 2>&1 | grep "e-0" | tail -1 | tr ' ' '\n' | awk '{ if ($1+0 > 1e-4) bad++; n++ } END{print "chain_check: " bad+0 " bad of " n}'
done
timeout -k 10 180 python tools/microbatch_probe.py --groups 1 --n 300 2>&1 | tail -1
timeout -k 10 300 python tools/kern_mix.py 2>&1 | grep -v amdgpu.ids
