set -x
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04j; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 400 python tools/ab_bench.py --reps 2 new=default nophase=env:SBM_PIX_PHASE=0 prev=build/prev.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_lds -o t --output-format csv -- python3 tools/lds_conflicts.py > $O/lds_run.log 2>&1; tail -2 $O/lds_run.log
python3 tools/lds_conflicts.py --summarise $O/pmc_lds > $O/lds_conflicts.txt 2>&1; cat $O/lds_conflicts.txt
rm -rf $O/pmc_lds
