#!/usr/bin/env python3
"""Timing of the FID stage at the evaluation's real size (pool3: 2048 features): the fp64-MFMA Gram accumulation, one
2048^3 fp64 GEMM, and the whole Frechet distance; against the fp64 matrix peak (78.6 TFLOP/s dense, v_mfma_f64_16x16x4_f64:
256 CUs x 4 SIMDs x 2048 flop / 64 cycles x 2.4 GHz) and, with --cpu, against the reference's numpy/scipy path on the host.

    python tools/fid_probe.py [--n 16384] [--dims 2048] [--cpu]
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from score_based_multimodal_autoencoder_amd import _lib, fid_score as F

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--dims", type=int, default=2048)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
PEAK = 78.6
D, n = a.dims, a.n
g = torch.Generator(device="cuda").manual_seed(1)
mix = torch.randn(D, D, device="cuda", generator=g) * (0.9 / D ** 0.5)
x = torch.relu(torch.randn(n, D, device="cuda", generator=g) @ mix + 0.20)
y = torch.relu(torch.randn(n, D, device="cuda", generator=g) @ mix + 0.25)
L = _lib.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

sx = F.ActivationStatistics(D)
ms = timed(lambda: sx.update(x))
fl = 2.0 * n * D * D
print(f"accumulate  n={n} dims={D}: {ms:8.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s fp64 = {fl / ms / 1e9 / PEAK:5.1%} of peak "
      f"(reads {n * D * 4 / 1e6:.0f} MB of features: {n * D * 4 / ms / 1e6:.0f} GB/s)")
A = torch.randn(D, D, dtype=torch.float64, device="cuda"); B = torch.randn(D, D, dtype=torch.float64, device="cuda"); C = torch.empty_like(A)
ms = timed(lambda: _lib.check(L.sbm_debug_dgemm_tn(A.data_ptr(), B.data_ptr(), C.data_ptr(), D, D, D, 1.0, 0.0, 0.0, st())), reps=10)
print(f"dgemm_tn    {D}^3           : {ms:8.3f} ms  {2.0 * D ** 3 / ms / 1e9:6.1f} TFLOP/s fp64 = {2.0 * D ** 3 / ms / 1e9 / PEAK:5.1%} of peak")
sx = F.ActivationStatistics(D).update(x); sy = F.ActivationStatistics(D).update(y)
(mx, cx), (my, cy) = sx.finalize(), sy.finalize()
torch.cuda.synchronize()
t0 = time.perf_counter(); fid, info = F.calculate_frechet_distance(mx, cx, my, cy, return_info=True); t1 = time.perf_counter()
gemms = 3 * sum(info["iters"]) + 2
print(f"frechet     dims={D}: {1e3 * (t1 - t0):8.1f} ms  fid {fid:.6f}  iterations {info['iters']}  ({gemms} GEMMs, "
      f"{gemms * 2.0 * D ** 3 / (t1 - t0) / 1e12:.1f} TFLOP/s fp64 overall)")
if a.cpu:
    from oracle import fid as OF
    xa, ya = x.cpu().numpy(), y.cpu().numpy()
    t0 = time.perf_counter(); m1, s1 = OF.activation_statistics(xa); m2, s2 = OF.activation_statistics(ya); t1 = time.perf_counter()
    want = OF.frechet_distance(m1, s1, m2, s2); t2 = time.perf_counter()
    print(f"cpu (numpy/scipy, {torch.get_num_threads()} threads): statistics {t1 - t0:.2f} s, frechet {t2 - t1:.2f} s, fid {want:.6f}  rel diff {abs(want - fid) / abs(want):.2e}")
