#!/usr/bin/env python3
"""Experiment: run the N-step sampler for B samples as G concurrent micro-batches (G engines, G streams, G host threads)."""
import argparse, ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--groups", type=str, default="1,2,4,8")
ap.add_argument("--n", type=int, default=200)
a = ap.parse_args()
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
sde = VPSDE(0.1, 5.0, 1000)
ts = torch.linspace(1.0, 1e-3, 1000, device="cuda")
L = _lib.lib()
for G in [int(g) for g in a.groups.split(",")]:
    b = a.batch // G
    engs = [_lib.Engine(_lib.SBM_UNET, _lib.unet_cfg_ints(64, 5, (1, 2, 2, 2), 8, 8), b, m.state_dict()) for _ in range(G)]
    xs = [torch.randn(b, 5, 8, 8, device="cuda") for _ in range(G)]
    xm = [torch.empty_like(x) for x in xs]
    cfg = sde._cfg()
    def run(i, n):
        scfg = _lib.SamplerCfg(0, 1, 0, 1, 7, i * b)
        _lib.check(L.sbm_sampler_run(engs[i].h, ctypes.byref(cfg), ctypes.byref(scfg), ts.data_ptr(), None, None, xs[i].data_ptr(), xm[i].data_ptr(), 0, n, None, None))
    def all_(n):
        th = [threading.Thread(target=run, args=(i, n)) for i in range(G)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    all_(20)
    t0 = time.perf_counter(); all_(a.n); dt = time.perf_counter() - t0
    print(f"G={G} (micro-batch {b}): {1e3 * dt / a.n:.3f} ms per SDE step for {a.batch} samples", flush=True)
    for e in engs: e.close()
