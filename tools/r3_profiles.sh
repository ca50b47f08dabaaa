# Round-3 measurement run (one GPU box): bench lines of configs 2/4/5, rocprofv3 kernel stats of the same bench commands, and the PMC passes
# (traffic: FETCH_SIZE / WRITE_SIZE; SQ counters) of eager forwards, each counter set in a run of its own with --kernel-trace only.
set -x
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r03
rm -rf $O; mkdir -p $O
for C in ${CONFIGS:-2 4 5}; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_fetch_c$C.log 2>&1 || echo "fetch pass failed c$C"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_write_c$C.log 2>&1 || echo "write pass failed c$C"
  python3 tools/traffic_probe.py --summarise $O/pmc_fetch_c$C $O/pmc_write_c$C > $O/r03_c${C}_traffic.json
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_sq1_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_sq1_c$C.log 2>&1 || echo "sq1 pass failed c$C"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $O/pmc_sq2_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_sq2_c$C.log 2>&1 || echo "sq2 pass failed c$C"
  python3 tools/sq_summary.py $O/pmc_sq1_c$C $O/pmc_sq2_c$C > $O/r03_c${C}_sq_counters.json
  mkdir -p profiles && cp $O/r03_c${C}_traffic.json $O/r03_c${C}_sq_counters.json profiles/
  rm -rf $O/pmc_fetch_c$C $O/pmc_write_c$C $O/pmc_sq1_c$C $O/pmc_sq2_c$C
done
for C in ${CONFIGS:-2 4 5}; do
  rocprofv3 --kernel-trace --stats -d $O/prof_bench_c$C -o t --output-format csv -- python3 bench.py --config $C --steps $([ $C = 2 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline > $O/bench_prof_c$C.json 2> $O/bench_prof_c$C.err || echo "bench profile failed c$C"
  cp $(find $O/prof_bench_c$C -name "*kernel_stats.csv" | head -1) $O/r03_bench_c${C}_kernel_stats.csv
  rm -rf $O/prof_bench_c$C
done
for C in ${CONFIGS:-2 4 5}; do
  python3 bench.py --config $C --steps $([ $C = 2 ] && echo 20 || echo 3) --warmup 1 > $O/r03_bench_c$C.json 2> $O/r03_bench_c$C.err || echo "bench c$C failed"
  tail -2 $O/r03_bench_c$C.err
done
ls -la $O
