set -x
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
