import ctypes, os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
B = 4
x = torch.from_numpy(synth.noise(9, 0, (B, 5, 8, 8))).cuda(); t = torch.full((B,), 0.5, device="cuda"); o = torch.empty_like(x)
eng = m._engine_for(B, 8, 8)
L = _lib.lib()
_lib.check(L.sbm_debug_forward_range(eng.h, x.data_ptr(), t.data_ptr(), o.data_ptr(), 0, 38, None))      # state in front of the 1x1 run
st = (ctypes.c_uint64 * 512)(); kinds = (ctypes.c_int * 512)(); nout = ctypes.c_int(0)
_lib.check(L.sbm_debug_pix_stamps(eng.h, 38, 3, st, kinds, 512, ctypes.byref(nout)))      # ops 38 .. 40 as one program
f = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
names = ["su", "sq", "mean", "rstd", "inv_r", "t(raw)", "qe0", "qe1"]
print("dumped program op", st[439] - 1, {n: f(st[440 + i]) for i, n in enumerate(names)}, "gn_stat", st[448], "ginv", f(st[449]), "qe2", f(st[450]))
