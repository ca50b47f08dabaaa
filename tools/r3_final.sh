# Round-end run on one GPU box: the whole GPU suite, then the measurement pipeline (tools/r3_profiles.sh), the per-workgroup timeline of
# config 4's largest seg_gemm layers, the `extra` block of bench.py and the per-op stamps of the pixel kernel.
set -x
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/final_tests.log 2>&1
rc=$?; tail -4 gpurun_out/final_tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/r3_profiles.sh > gpurun_out/final_profiles.log 2>&1
tail -5 gpurun_out/final_profiles.log
python tools/gemm_timeline.py --config 4 --ops 8,11,21,109 > gpurun_out/r03/r03_seg_gemm_timeline_c4.txt 2>&1
python bench.py --extra --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/r03_bench_c2_extra.json 2> gpurun_out/r03/r03_bench_c2_extra.err
PIX_PHASES=1 python tools/pix_stamps.py > gpurun_out/r03/r03_pix_stamps.txt 2>&1
python - > gpurun_out/r03/r03_pix_notes.txt 2>&1 <<'PY'
import torch, bench
for cfg in (2, 4, 5):
    score, _ = bench.build_models(cfg, torch.device("cuda", 0)); W = bench.WORK[cfg]
    eng = score._engine_for(W["B"], W["hw"], W["hw"]) if cfg != 5 else score._engine_for(W["B"], W["hw"], W["hw"], True)
    print(f"config {cfg}: uses_pix {eng.info['uses_pix']}  fragment MB {eng.info['pix_fragment_mb']:.1f}  note: {eng.info['pix_note']!r}")
    score.release_engines()
PY
cat gpurun_out/r03/r03_pix_notes.txt
ls -la gpurun_out/r03
