#!/usr/bin/env python3
"""Phases inside the ops of the one-position runs (build/stamp_vec.so, -DPIX_STAMP_VEC): start -> [GroupNorm moments] -> [slot loop] -> [reduce + epilogue + request]
-> [statistics + barrier] -> next op."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
B = 256
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
eng = m._engine_for(B, 8, 8)
x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda"); o = torch.empty_like(x)
for _ in range(3): eng.forward(x, t, o)
torch.cuda.synchronize()
L = _lib.lib(); n = eng.info["n_ops"]; buf = ctypes.create_string_buffer(256)
names = []
for i in range(n):
    L.sbm_debug_op_name(eng.h, i, buf, 256); names.append(buf.value.decode())
st = (ctypes.c_uint64 * 512)(); kinds = (ctypes.c_int * 512)(); nout = ctypes.c_int(0)
ok = [L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, 512, ctypes.byref(nout)) == 0 for i in range(n)]
first = max(i for i in range(n) if ok[i] and not ok[i - 1]); last = max(i for i in range(n) if ok[i])
_lib.check(L.sbm_debug_pix_stamps(eng.h, first, last - first + 1, st, kinds, 512, ctypes.byref(nout)))
full = list(st); k = list(kinds)[:nout.value]; CLK = 2100.0
j = first
for i in range(nout.value):
    kind = k[i] // 1000
    nm = "(stage)" if kind == 101 else names[j]
    if kind != 101: j += 1
    if kind in (102, 103, 104) and i < 96:
        ph = [full[128 + i * 4 + q] for q in range(4)]
        tot = (full[i + 1] - full[i]) / CLK
        if all(ph):
            print(f"{nm[:50]:50s} {kind} total {tot:5.2f} | head {(ph[0] - full[i]) / CLK:5.2f} moments {(ph[1] - ph[0]) / CLK:5.2f} slots {(ph[2] - ph[1]) / CLK:5.2f} epilogue+request {(ph[3] - ph[2]) / CLK:5.2f} stats+barrier {(full[i + 1] - ph[3]) / CLK:5.2f}")
        else:
            print(f"{nm[:50]:50s} {kind} total {tot:5.2f} | head {(ph[0] - full[i]) / CLK:5.2f} body {(ph[3] - ph[0]) / CLK:5.2f} stats+barrier {(full[i + 1] - ph[3]) / CLK:5.2f}")
