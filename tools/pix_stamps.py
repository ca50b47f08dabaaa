#!/usr/bin/env python3
"""Per-op time inside the pixel kernel (stamps of workgroup 0), config-2 score net at B=256: both runs (down / up)."""
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
st = (ctypes.c_uint64 * 256)(); kinds = (ctypes.c_int * 256)(); nout = ctypes.c_int(0)
ok = [L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, 256, ctypes.byref(nout)) == 0 for i in range(n)]
runs, i = [], 0
while i < n:
    if ok[i]:
        j = i
        while j + 1 < n and ok[j + 1]: j += 1
        runs.append((i, j)); i = j + 1
    else:
        i += 1
KN = {0: "gemm", 1: "dw", 2: "linattn", 4: "gn_res", 101: "(stage)"}
for first, last in runs:
    _lib.check(L.sbm_debug_pix_stamps(eng.h, first, last - first + 1, st, kinds, 256, ctypes.byref(nout)))
    full = list(st); v = full[:nout.value + 1]; k = list(kinds)[:nout.value]
    j = first
    print(f"run: ops {first}..{last}")
    for i in range(nout.value):
        kind, taps = k[i] // 1000, k[i] % 1000
        nm = "(stage)" if kind == 101 else names[j]
        if kind != 101: j += 1
        extra = ""
        if kind == 0 and i < 30 and os.environ.get("PIX_PHASES"):
            ph = [full[64 + i * 6 + j] for j in range(5)]
            if all(ph):
                extra = "   | taps+sync %.2f  fill %.2f  kloop %.2f  partials+sync %.2f  epilogue+sync %.2f (last pass)" % tuple((b_ - a_) / 2400.0 for a_, b_ in zip([v[i]] + ph[:4], ph))
        print(f"  {nm[:52]:52s} {KN.get(kind, kind):8} {(v[i + 1] - v[i]) / 2400.0:8.2f} us" + (f"   {taps} taps" if taps else "") + extra)
    print(f"  total {(v[-1] - v[0]) / 2400.0:.1f} us for {nout.value} program ops")
