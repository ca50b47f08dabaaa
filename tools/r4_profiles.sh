# Round-4 measurement run (one GPU box): PMC passes (traffic: FETCH_SIZE / WRITE_SIZE; SQ counters) of eager forwards, each counter set in a run
# of its own with --kernel-trace only; rocprofv3 kernel stats of the bench commands (+ the per-launch split of the pixel kernel); the bench lines
# of configs 2 / 4 / 5 and of the P10 recipe, with and without the corrector; the `extra` blocks.
set -x
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04
rm -rf $O; mkdir -p $O profiles
for C in ${CONFIGS:-2 4 5}; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_fetch_c$C.log 2>&1 || echo "fetch pass failed c$C"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_write_c$C.log 2>&1 || echo "write pass failed c$C"
  python3 tools/traffic_probe.py --summarise $O/pmc_fetch_c$C $O/pmc_write_c$C > $O/r04_c${C}_traffic.json
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_sq1_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_sq1_c$C.log 2>&1 || echo "sq1 pass failed c$C"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $O/pmc_sq2_c$C -o t --output-format csv -- python3 tools/traffic_probe.py --config $C > $O/pmc_sq2_c$C.log 2>&1 || echo "sq2 pass failed c$C"
  python3 tools/sq_summary.py $O/pmc_sq1_c$C $O/pmc_sq2_c$C > $O/r04_c${C}_sq_counters.json
  cp $O/r04_c${C}_traffic.json $O/r04_c${C}_sq_counters.json profiles/
  rm -rf $O/pmc_fetch_c$C $O/pmc_write_c$C $O/pmc_sq1_c$C $O/pmc_sq2_c$C
done
for C in ${CONFIGS:-2 4 5} 10; do
  rocprofv3 --kernel-trace --stats -d $O/prof_bench_c$C -o t --output-format csv -- python3 bench.py --config $C --steps $([ $C = 2 ] && echo 3 || echo 1) --warmup 1 --no-cpu-baseline > $O/bench_prof_c$C.json 2> $O/bench_prof_c$C.err || echo "bench profile failed c$C"
  cp $(find $O/prof_bench_c$C -name "*kernel_stats.csv" | head -1) $O/r04_bench_c${C}_kernel_stats.csv
  [ $C = 2 ] && python3 tools/pix_launch_split.py $O/prof_bench_c$C > $O/r04_c2_pixrun_launches.txt
  rm -rf $O/prof_bench_c$C
done
for C in ${CONFIGS:-2 4 5}; do
  python3 bench.py --config $C --steps $([ $C = 2 ] && echo 20 || echo 3) --warmup 1 > $O/r04_bench_c$C.json 2> $O/r04_bench_c$C.err || echo "bench c$C failed"
  tail -2 $O/r04_bench_c$C.err
done
# what the recipes run (train_poly.sh:17 / train_cel.sh:8: --pc=1): the corrector on in the timed pass, and the `extra` blocks
python3 bench.py --config 2 --extra --steps 3 --warmup 1 --no-cpu-baseline > $O/r04_bench_c2_extra.json 2> $O/r04_bench_c2_extra.err
for C in 4 5; do python3 bench.py --config $C --extra --steps 1 --warmup 1 --no-cpu-baseline > $O/r04_bench_c${C}_extra.json 2> $O/r04_bench_c${C}_extra.err; done
python3 bench.py --config 10 --steps 3 --warmup 1 > $O/r04_bench_p10.json 2> $O/r04_bench_p10.err
python3 bench.py --config 10 --pc --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_bench_p10_pc.json 2> $O/r04_bench_p10_pc.err
python3 bench.py --config 2 --pc --steps 3 --warmup 1 --no-cpu-baseline > $O/r04_bench_c2_pc.json 2> $O/r04_bench_c2_pc.err
PIX_PHASES=1 python3 tools/pix_stamps.py > $O/r04_pix_stamps.txt 2>&1
ls -la $O
