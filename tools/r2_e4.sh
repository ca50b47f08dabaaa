set -x
export PYTHONUNBUFFERED=1
timeout -k 10 200 python tools/row_stamps.py 2>&1 | grep -v amdgpu.ids | tail -60
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "openai or oa or UNetModel or cel_rep or config5 or refiner" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu -k "config5" 2>&1 | tail -3
timeout -k 10 400 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['per_kernel_ms'])"
