#!/usr/bin/env python3
"""A/B of library builds on ONE box (MI355X boxes of the pool differ by a few percent, more than most kernel changes):
    python tools/ab_bench.py [--reps 2] [--config 2] NAME=path/to/lib.so ...      (NAME=default: the in-tree library)
Runs bench.py --steps 3 with each library in turn, `reps` rounds, and prints samples/s, the per-kernel ms of the forward and the loop's
ms per SDE step."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps, cfg = 2, "2"
while args and args[0].startswith("--"):
    if args[0] == "--reps": reps = int(args[1]); args = args[2:]
    elif args[0] == "--config": cfg = args[1]; args = args[2:]
    else: raise SystemExit("unknown flag " + args[0])
libs = [a.split("=", 1) for a in args]
for rep in range(reps):
    for name, path in libs:
        env = dict(os.environ)
        env.pop("SBM_LIB", None)
        extra = {}
        if path != "default":
            if path.startswith("env:"):                     # NAME=env:VAR=VALUE -> the in-tree library with an environment knob
                k, v = path[4:].split("=", 1); env[k] = v
            else:
                env["SBM_LIB"] = os.path.join(ROOT, path)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--config", cfg], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(name, "FAILED", r.stderr[-500:]); sys.exit(1)
        j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]); ro = j["roofline"]
        print(f"{name:10s} rep {rep}: {j['value']:7.2f} samples/s  loop {ro['loop_ms_per_sde_step'] * 1e3:7.1f} us/step  kernels {ro['per_kernel_ms']}", flush=True)
