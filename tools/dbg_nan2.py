import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
B = 4
x0 = torch.from_numpy(synth.noise(9, 0, (B, 5, 8, 8))).cuda()
for tval in ("rand", 1.0, 0.5):
    t = torch.rand(B, device="cuda") if tval == "rand" else torch.full((B,), float(tval), device="cuda")
    outs = {}
    for env in ({}, {"SBM_PIX_VEC": "0"}):
        os.environ.update(env)
        try:
            m.release_engines()
            y1 = m(x0, t).clone(); y2 = m(x0, t).clone()
            outs[str(env)] = y1
            print(tval, env, "finite" if torch.isfinite(y1).all() else "NAN", "second call", "finite" if torch.isfinite(y2).all() else "NAN", float(y1.abs().max()) if torch.isfinite(y1).all() else "-", flush=True)
        finally:
            for k in env: os.environ.pop(k, None)
    a, b = list(outs.values())
    if torch.isfinite(a).all(): print("  max diff vec vs mfma", float((a - b).abs().max()))
