#!/usr/bin/env python3
"""Eager config-2 score-net forwards (no graph) for a rocprofv3 --pmc pass:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o t --output-format csv -- python3 tools/traffic_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o t --output-format csv -- python3 tools/traffic_probe.py
then  python3 tools/traffic_probe.py --summarise gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_traffic.json
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

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
    print(json.dumps(out, indent=1))
    sys.exit(0)

import torch
from score_based_multimodal_autoencoder_amd import synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
x = torch.randn(256, 5, 8, 8, device="cuda"); t = torch.rand(256, device="cuda")
for _ in range(12):
    y = m(x, t)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
