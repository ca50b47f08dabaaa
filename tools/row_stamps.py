#!/usr/bin/env python3
"""Per-op time inside the row kernel (stamps of workgroup 0), config-2 score net at B=256."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
eng = m._engine_for(B, 8, 8)
x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda"); o = torch.empty_like(x)
for _ in range(3): eng.forward(x, t, o)
torch.cuda.synchronize()
L = _lib.lib(); n = eng.info["n_ops"]; buf = ctypes.create_string_buffer(256)
names = []
for i in range(n):
    L.sbm_debug_op_name(eng.h, i, buf, 256); names.append(buf.value.decode())
# the row run = the ops between the two chain runs: find it by probing
st = (ctypes.c_uint64 * 256)(); kinds = (ctypes.c_int * 256)(); nout = ctypes.c_int(0)
first = None
for i in range(n):
    if L.sbm_debug_row_stamps(eng.h, i, 1, st, kinds, 256, ctypes.byref(nout)) == 0:
        if first is None: first = i
        last = i
print("row run: ops", first, "..", last)
_lib.check(L.sbm_debug_row_stamps(eng.h, first, last - first + 1, st, kinds, 256, ctypes.byref(nout)))
v = list(st)[:nout.value + 1]; k = list(kinds)[:nout.value]
j = first
for i in range(nout.value):
    kind, passes = k[i] // 1000, k[i] % 1000
    nm = "(stage)" if kind == 100 else names[j]
    if kind != 100: j += 1
    print(f"{nm[:50]:50s} {(v[i + 1] - v[i]) / 2400.0:8.2f} us" + (f"   {passes} pass(es)" if passes else ""))
print(f"total {(v[-1] - v[0]) / 2400.0:.1f} us for {nout.value} program ops")
