set -x
export PYTHONUNBUFFERED=1
for args in "67 16 8" "256 16 64" "200 16 32"; do
  timeout -k 10 120 python tools/chain_check.py $args 2>&1 | grep "e-0" | tail -1 | tr ' ' '\n' | awk '{ if ($1+0 > 1e-4) bad++; n++ } END{print "chain_check: " bad+0 " bad of " n}'
done
timeout -k 10 300 python tools/kern_mix.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/row_stamps.py 2>&1 | grep "total"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
