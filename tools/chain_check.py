#!/usr/bin/env python3
"""Compare the chain-kernel forward (SBM_CHAIN_PIX=k) with the per-layer forward (SBM_CHAIN_PIX=0), per sample."""
import os, sys
os.environ.setdefault("SBM_CHAIN_MIN_B", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 67
pix = sys.argv[2] if len(sys.argv) > 2 else "4"
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = synth.fill_module_(Unet(dim=dim, channels=2, dim_mults=(1, 2, 2, 2)).eval()).cuda()
x = torch.randn(B, 2, 8, 8, device="cuda"); t = torch.rand(B, device="cuda")
outs = {}
for k in ("0", pix):
    os.environ["SBM_CHAIN_PIX"] = k
    eng = _lib.Engine(_lib.SBM_UNET, _lib.unet_cfg_ints(dim, 2, (1, 2, 2, 2), 8, 8), B, m.state_dict())
    o = torch.empty_like(x)
    import ctypes
    for rep in range(3):
        eng.forward(x, t, o)
        st = (ctypes.c_uint * 300)()
        _lib.check(_lib.lib().sbm_debug_chain_sync(eng.h, st, 300))
        v = list(st)
        print('chain_pix', k, 'rep', rep, 'count', v[:8], 'arrived', v[8], 'done', v[9], 'abort', v[10], 'bar', v[16:144:16], 'left', v[144:272:16], flush=True)
    torch.cuda.synchronize()
    outs[k] = o.clone()
    eng.close()
err = (outs["0"] - outs[pix]).abs().amax(dim=(1, 2, 3))
print("max |y|", float(outs["0"].abs().max()), "per-sample max err:")
print(" ".join(f"{e:.1e}" for e in err.tolist()))
