set -x
export PYTHONUNBUFFERED=1
SBM_CHAIN_NO_PREFETCH=1 timeout -k 10 300 python tools/kern_mix.py 2>&1 | grep -v amdgpu.ids | grep "chain\|row\|total"
timeout -k 10 300 python tools/kern_mix.py 2>&1 | grep -v amdgpu.ids | grep "chain\|row\|total"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
