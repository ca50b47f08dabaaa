#!/usr/bin/env python3
"""Config 2 at full size (B = 256, 1000 reverse-SDE steps): how far do the samples move when (a) the batch is cut into two shards keyed by
their global sample offset, (b) other kernels carry the low-resolution levels (SBM_CHAIN_PIX = 0 / 16 / 64)?  Prints the differences
relative to max|x| -- the numbers behind the 1e-5 bounds of tests/test_gpu_parity.py::test_full_size_sampling_run_is_path_and_shard_independent."""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE, uncond_sampler
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
for b1 in (5.0, 20.0):
    sde = VPSDE(0.1, b1, 1000)
    x0 = torch.empty(256, 5, 8, 8, device="cuda")
    _lib.check(_lib.lib().sbm_fill_normal(x0.data_ptr(), 5 * 64, 256, 1234, 0, 0, 255, 1.0, None))
    outs = {}
    for pix in ("0", "16", "64"):
        os.environ["SBM_CHAIN_PIX"] = pix
        m.release_engines()
        outs[pix] = uncond_sampler((256, 5, 8, 8), m, "cuda", sde, x0=x0, seed=7, sample_offset=0).clone()
        halves = [uncond_sampler((128, 5, 8, 8), m, "cuda", sde, x0=x0[o:o + 128], seed=7, sample_offset=o) for o in (0, 128)]
        d = float((torch.cat(halves) - outs[pix]).abs().max()); sc = float(outs[pix].abs().max())
        print(f"beta1 {b1} chain_pix {pix}: max|x| {sc:.4g}; halves vs full: {d:.3g} = {d / sc:.3g} relative; bit-equal {torch.equal(torch.cat(halves), outs[pix])}", flush=True)
    for a, b in (("0", "16"), ("0", "64"), ("16", "64")):
        d = float((outs[a] - outs[b]).abs().max()); print(f"   paths {a} vs {b}: {d / float(outs[a].abs().max()):.3g} relative")
os.environ.pop("SBM_CHAIN_PIX", None)
