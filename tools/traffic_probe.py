#!/usr/bin/env python3
"""Eager score-net forwards (no graph) of a BASELINE config for rocprofv3 --pmc passes, and the summariser of those passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o t --output-format csv -- python3 tools/traffic_probe.py --config 2
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o t --output-format csv -- python3 tools/traffic_probe.py --config 2
    python3 tools/traffic_probe.py --summarise gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r02_c2_traffic.json

The summary records the hash of the library the passes ran with: bench.py uses the numbers only for that very build.
"""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lib_hash():
    from score_based_multimodal_autoencoder_amd import _lib
    return _lib.source_hash()


if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    out = {}
    for d, key in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != key:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                a = acc.setdefault(name, [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
        for name, (tot, n) in acc.items():
            out.setdefault(name, {})[key] = {"launches": n, "avg_per_launch_raw": tot / n}
    print(json.dumps({"lib_hash": lib_hash(), "unit": "KB (FETCH_SIZE counts 64-byte requests in units that need x2 on gfx950: MI355X_MICROARCH.md)", "kernels": out}, indent=1))
    sys.exit(0)

import torch
cfg = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 2
import bench
score, _ = bench.build_models(cfg, torch.device("cuda", 0))
W = bench.WORK[cfg]
x = torch.randn(W["B"], W["n_mod"], W["hw"], W["hw"], device="cuda"); t = torch.rand(W["B"], device="cuda")
kw = {"z": torch.randn(W["B"], 512, device="cuda")} if cfg == 5 else {}
for _ in range(12 if cfg == 2 else 4):
    y = score(x, t, **kw)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
