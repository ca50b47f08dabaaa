"""Why does the per-kernel profile of config 2 read 3x slower right after a predictor-corrector pass?  (round 4, profiles/r04_bench_c2_pc.json)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from score_based_multimodal_autoencoder_amd import _lib
from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE

dev = torch.device("cuda", 0)
cfg = 2
W = bench.WORK[cfg]; B = W["B"]
score, decs = bench.build_models(cfg, dev)
sde = VPSDE(W["beta"][0], W["beta"][1], bench.N_SDE)
x0 = torch.empty((B, W["n_mod"], W["hw"], W["hw"]), device=dev)
_lib.check(_lib.lib().sbm_fill_normal(x0.data_ptr(), W["n_mod"] * 64, B, 1234, 0, 0, 255, 1.0, None))
eng = score._engine_for(B, 8, 8)
tt = torch.ones(B, device=dev) * 0.5
so = torch.empty_like(x0)

def prof(tag):
    p = eng.profile(x0.clone(), tt, so, reps=10, sampler_mix=True)
    print(tag, {k: round(v["ms"], 4) for k, v in p.items() if v["launches"]}, flush=True)

eng.forward(x0, tt, so)
prof("fresh          ")
z, outs, _ = bench.one_pass(cfg, score, decs, sde, x0, None, 7, 0, dev, pc=False)
print("pc=False: |z| max", float(z.abs().max()), "finite", bool(torch.isfinite(z).all()), flush=True)
prof("after pc=False ")
z, outs, _ = bench.one_pass(cfg, score, decs, sde, x0, None, 7, 0, dev, pc=True)
print("pc=True : |z| max", float(z.abs().max()), "finite", bool(torch.isfinite(z).all()), flush=True)
prof("after pc=True  ")
eng.forward(x0, tt, so)
print("plain forward: finite", bool(torch.isfinite(so).all()), flush=True)
prof("after a forward")
