#!/usr/bin/env python3
"""Per-op time inside the pixel kernel (stamps of workgroup 0), config-2 score net at B=256: every run of low-resolution ops.
With PIX_PHASES=1 every GEMM pass also shows its phases: K loop | partial tiles + barrier | epilogue + barrier."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
eng = m._engine_for(B, 8, 8)
print("uses_pix", eng.info["uses_pix"], "fragment MB %.1f" % eng.info["pix_fragment_mb"], "note:", eng.info["pix_note"])
if not eng.info["uses_pix"]:
    sys.exit(1)
x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda"); o = torch.empty_like(x)
for _ in range(3): eng.forward(x, t, o)
torch.cuda.synchronize()
L = _lib.lib(); n = eng.info["n_ops"]; buf = ctypes.create_string_buffer(256)
names = []
for i in range(n):
    L.sbm_debug_op_name(eng.h, i, buf, 256); names.append(buf.value.decode())
CAP = 512
st = (ctypes.c_uint64 * CAP)(); kinds = (ctypes.c_int * CAP)(); nout = ctypes.c_int(0)
ok = [L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, CAP, ctypes.byref(nout)) == 0 for i in range(n)]
runs, i = [], 0
while i < n:
    if ok[i]:
        j = i
        while j + 1 < n and ok[j + 1]: j += 1
        runs.append((i, j)); i = j + 1
    else:
        i += 1
KN = {0: "gemm", 1: "dw", 2: "linattn", 4: "gn_res", 101: "(stage)", 102: "vec", 15: "im2col"}
CLK = float(os.environ.get("PIX_CLOCK_MHZ", "2100"))          # s_memtime counts shader cycles; the clock under this load is ~2.1 GHz
for first, last in runs:
    _lib.check(L.sbm_debug_pix_stamps(eng.h, first, last - first + 1, st, kinds, CAP, ctypes.byref(nout)))
    full = list(st); v = full[:nout.value + 1]; k = list(kinds)[:nout.value]
    j = first; item = 0; tot = {}
    print(f"run: ops {first}..{last}")
    for i in range(nout.value):
        kind, passes = k[i] // 1000, k[i] % 1000
        nm = "(stage)" if kind == 101 else names[j]
        if kind != 101: j += 1
        extra = ""
        if kind == 0:
            if os.environ.get("PIX_PHASES"):
                for p in range(passes):
                    ph = [full[128 + (item + p) * 4 + q] for q in range(4)] if item + p < 96 else [0] * 4
                    if all(ph):
                        extra += "   | pre %.2f kloop %.2f  partials+sync %.2f  epilogue+sync %.2f" % ((ph[0] - v[i]) / CLK if p == 0 else 0.0, (ph[1] - ph[0]) / CLK, (ph[2] - ph[1]) / CLK, (ph[3] - ph[2]) / CLK)
            item += passes
        us = (v[i + 1] - v[i]) / CLK
        tot[KN.get(kind, kind)] = tot.get(KN.get(kind, kind), 0.0) + us
        print(f"  {nm[:52]:52s} {KN.get(kind, kind):8} {us:8.2f} us" + (f"   {passes} passes" if kind == 0 else "") + extra)
    print(f"  total {(v[-1] - v[0]) / CLK:.1f} us for {nout.value} program ops; by kind: " + ", ".join(f"{a} {b:.1f}" for a, b in tot.items()))
