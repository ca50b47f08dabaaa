#!/usr/bin/env python3
"""Per-kernel time of the launch mix of one sampler step (config-2 score net, B=256): HIP events on each kernel's own dispatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
eng = m._engine_for(B, 8, 8)
x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda"); o = torch.empty_like(x)
p = eng.profile(x, t, o, reps=30, sampler_mix=True)
tot = 0.0
for k, v in p.items():
    if v["launches"]:
        print(f"{k:16s} {v['launches']:3d} launches {1e3 * v['ms']:9.1f} us  ({1e3 * v['ms'] / v['launches']:7.1f} us each)")
        tot += v["ms"]
print(f"total {1e3 * tot:.1f} us per step; gemm flops (seg, chain, row) = {[round(f / 1e9, 2) for f in eng.gemm_flops(sampler_mix=True)]} GFLOP")
