#!/usr/bin/env python3
"""Stress the chain kernel's synchronisation: many forwards with fresh inputs, each compared with the per-layer path."""
import os, sys
os.environ.setdefault("SBM_CHAIN_MIN_B", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
worst = 0.0
for B, dim in ((256, 64), (67, 8), (200, 32)):
    m = synth.fill_module_(Unet(dim=dim, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
    cfg = _lib.unet_cfg_ints(dim, 5, (1, 2, 2, 2), 8, 8)
    os.environ["SBM_CHAIN_PIX"] = "0"; ref = _lib.Engine(_lib.SBM_UNET, cfg, B, m.state_dict())
    os.environ["SBM_CHAIN_PIX"] = "16"; ch = _lib.Engine(_lib.SBM_UNET, cfg, B, m.state_dict())
    a = torch.empty(B, 5, 8, 8, device="cuda"); b = torch.empty_like(a)
    bad = 0
    for i in range(iters):
        x = torch.randn(B, 5, 8, 8, device="cuda") * (1 + i % 3); t = torch.rand(B, device="cuda")
        ref.forward(x, t, a); ch.forward(x, t, b)
        err = float((a - b).abs().max() / a.abs().max())
        worst = max(worst, err)
        if not err < 5e-5: bad += 1
    st = (ctypes.c_uint * 300)(); _lib.check(_lib.lib().sbm_debug_chain_sync(ch.h, st, 300))
    print(f"B={B} dim={dim}: {iters} forwards, {bad} mismatching, worst rel err {worst:.2e}, sync state clean: {not any(st)}", flush=True)
    ref.close(); ch.close()
