import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import synth, sde_helper2 as S
from score_based_multimodal_autoencoder_amd.unet_model import Unet
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
for B in (4, 256):
    x0 = torch.from_numpy(synth.noise(9, 0, (B, 5, 8, 8))).cuda()
    for N in (1, 2, 5):
        for kw in (dict(), dict(use_graph=False)):
            for env in ({}, {"SBM_PIX_VEC": "0"}, {"SBM_NO_FUSED_CVT": "1"}, {"SBM_NO_TIME_TABLE": "1"}):
                os.environ.update(env)
                try:
                    m.release_engines()
                    a = S.uncond_sampler((B, 5, 8, 8), m, "cuda", S.VPSDE(0.1, 5.0, N), x0=x0, seed=123, **kw)
                    print(B, N, kw, env, "nan" if not torch.isfinite(a).all() else "ok %.4f" % float(a.abs().max()), flush=True)
                finally:
                    for k in env: os.environ.pop(k, None)
