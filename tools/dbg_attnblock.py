"""Phase stamps of attnblock_kernel (SBM_ATTNBLOCK_STAMPS=1): a few plain forwards of config 2's score net."""
import os, sys, torch
os.environ["SBM_ATTNBLOCK_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from score_based_multimodal_autoencoder_amd import _lib
dev = torch.device("cuda", 0)
W = bench.WORK[2]; B = W["B"]
score, decs = bench.build_models(2, dev)
x0 = torch.randn((B, 5, 8, 8), device=dev)
eng = score._engine_for(B, 8, 8)
tt = torch.ones(B, device=dev) * 0.5
so = torch.empty_like(x0)
for _ in range(6):
    eng.forward(x0, tt, so)
torch.cuda.synchronize()
