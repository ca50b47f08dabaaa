#!/usr/bin/env python3
"""Per-op timing table of one score-net forward (HIP events around every launch, eager).

    python tools/profile_ops.py [--batch 256] [--dim 64] [--channels 5] [--reps 20]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from score_based_multimodal_autoencoder_amd import _lib, synth  # noqa: E402
from score_based_multimodal_autoencoder_amd.unet_model import Unet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--channels", type=int, default=5)
ap.add_argument("--hw", type=int, default=8)
ap.add_argument("--mults", type=str, default="1,2,2,2")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--config", type=int, default=0, choices=(0, 2, 4, 5, 6), help="score net of a bench.py workload (overrides the shape flags); 6 = the DiffuseVAE refiner UNetModel(6->3) at 128x128 (eval_lat_celeba_hq_all.py:448), batch from --batch")
a = ap.parse_args()

if a.config == 4:
    a.batch, a.dim, a.channels, a.hw, a.mults = 64, 256, 3, 16, "1,2,2,2,2"
if a.config == 6:
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel  # noqa: E402
    a.channels, a.hw = 6, 128
    m = synth.fill_module_(UNetModel(6, 128, 3, 2, (16,), 0.1, (1, 2, 2, 3, 4), num_heads=8).eval()).cuda()
    eng = m._engine_for(a.batch, a.hw, a.hw, False)
elif a.config == 5:
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel  # noqa: E402
    a.batch, a.channels, a.hw = 64, 3, 16
    m = synth.fill_module_(UNetModel(3, 128, 3, 2, (), 0.1, (1, 2, 4, 8), num_heads=1, use_z=True, z_dim=512).eval()).cuda()
    eng = m._engine_for(a.batch, a.hw, a.hw, False)
else:
    m = synth.fill_module_(Unet(dim=a.dim, channels=a.channels, dim_mults=tuple(int(v) for v in a.mults.split(","))).eval()).cuda()
    eng = m._engine_for(a.batch, a.hw, a.hw)
x = torch.randn(a.batch, a.channels, a.hw, a.hw, device="cuda")
t = torch.rand(a.batch, device="cuda")
out = torch.empty((a.batch, 3 if a.config == 6 else a.channels, a.hw, a.hw), device="cuda")
n = eng.info["n_ops"]
ms = (ctypes.c_double * n)()
fl = (ctypes.c_double * n)()
L = _lib.lib()
_lib.check(L.sbm_debug_profile_ops(eng.h, x.data_ptr(), t.data_ptr(), out.data_ptr(), a.reps, ms, fl, n))
buf = ctypes.create_string_buffer(256)
tot = 0.0
print(f"{'#':>3} {'op':58s} {'us':>8} {'GFLOP':>8} {'TF/s':>7}")
for i in range(n):
    L.sbm_debug_op_name(eng.h, i, buf, 256)
    us = ms[i] * 1e3
    tot += us
    tf = fl[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 and fl[i] > 0 else 0
    print(f"{i:3d} {buf.value.decode()[:58]:58s} {us:8.1f} {fl[i] / 1e9:8.3f} {tf:7.1f}")
if os.environ.get("SBM_STAMPS"):
    print("in-kernel cycle stamps (middle block, wave 0): prologue / loop / epilogue | wait issue compute | chunks grid")
    st = (ctypes.c_uint64 * 16)()
    for i in range(n):
        L.sbm_debug_op_name(eng.h, i, buf, 256)
        if not buf.value.decode().startswith("0:"):
            continue
        _lib.check(L.sbm_debug_gemm_stamps(eng.h, x.data_ptr(), out.data_ptr(), i, st))
        v = list(st)
        print(f"{i:3d} {buf.value.decode()[:44]:44s} pro {v[0]:7d} loop {v[1]:8d} epi {v[2]:7d} | wait {v[3]:8d} clock {(v[0] + v[1] + v[2]) / max(v[4], 1) * 100:5.0f} MHz comp {v[5]:8d} | "
              f"chunks {v[6]:3d} grid {v[7]:5d}  cyc/chunk {v[5] / max(v[6], 1):6.0f} | prefill {v[8]:6d} gnload {v[9]:6d} gnred {v[10]:6d} setup {v[11]:5d}")
print(f"total {tot:.1f} us; dense FLOP {eng.info['flops_dense'] / 1e9:.2f} G, exec FLOP {eng.info['flops_exec'] / 1e9:.2f} G")
