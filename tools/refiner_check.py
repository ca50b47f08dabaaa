#!/usr/bin/env python3
"""The DiffuseVAE refiner's real network -- UNetModel(6, 128, 3, 2, (16,), 0.1, (1,2,2,3,4), num_heads=8) on 128x128 images
(eval_lat_celeba_hq_all.py:448) -- through the HIP path: error against the numpy oracle at B = 2, time per evaluation at --batch.

    python tools/refiner_check.py [--batch 64] [--reps 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import unet_openai as OU  # noqa: E402  (checker only)
from score_based_multimodal_autoencoder_amd import synth  # noqa: E402
from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
KW = dict(in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(16,), channel_mult=(1, 2, 2, 3, 4))
m = synth.fill_module_(UNetModel(6, 128, 3, 2, (16,), 0.1, (1, 2, 2, 3, 4), num_heads=8).eval()).cuda()
sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
x = synth.noise(3, 0, (2, 6, 128, 128))
t = np.array([999, 3], np.float32)
t0 = time.perf_counter()
want = OU.unet_openai_forward(sd, x, t, num_heads=8, **KW)
t_cpu = time.perf_counter() - t0
got = m(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
err = np.abs(got - want)
print(f"B=2 vs numpy oracle: max|err| = {err.max():.3e}, max|y| = {np.abs(want).max():.3e}, relative {err.max() / np.abs(want).max():.3e}; oracle {t_cpu:.1f} s on the host")
m.release_engines()
xb = torch.randn(a.batch, 6, 128, 128, device="cuda")
tb = torch.full((a.batch,), 500.0, device="cuda")
y = m(xb, tb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    y = m(xb, tb)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.reps * 1e3
eng = m._engine_for(a.batch, 128, 128, False)
fl = eng.info["flops_dense"]
print(f"B={a.batch}: {ms:.1f} ms per evaluation, {fl / 1e12:.2f} TFLOP dense -> {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3:.3f} of the fp32-MFMA peak); "
      f"workspace {eng.info['workspace_bytes'] / 2**30:.1f} GiB; finite: {bool(torch.isfinite(y).all())}")
