#!/usr/bin/env python3
"""Per-op time inside the chain kernel (stamps of workgroup 0 of the first cluster), config-2 score net."""
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
# find the chain run: ask the library by trying ranges (ops marked chain are those the stamps call accepts)
st = (ctypes.c_uint64 * 1024)()
runs = []; cur = None
for i in range(n):
    if L.sbm_debug_chain_stamps(eng.h, i, 1, st) == 0:
        if cur is None: cur = [i, 0]
        cur[1] += 1
    elif cur is not None: runs.append(cur); cur = None
if cur: runs.append(cur)
print("chain runs:", runs)
first, count = max(runs, key=lambda r: r[1])
print("chain run: ops", first, "..", first + count - 1)
for rep in range(2):
    _lib.check(L.sbm_debug_chain_stamps(eng.h, first, count, st))
v = list(st)[:1 + 10 * count]
tot = (v[2 * count] - v[0]) / 2400.0
print(f"{'op':46s} {'work us':>8} {'wait us':>8}")
prev = v[0]
for k in range(count):
    w = (v[1 + 2 * k] - prev) / 2400.0; b = (v[2 + 2 * k] - v[1 + 2 * k]) / 2400.0
    tr = v[1 + 2 * count + 8 * k: 1 + 2 * count + 8 * k + 8]
    extra = ""
    if tr[0]:
        extra = "  | setup+GN %.2f  tables+epi-loads %.2f  K-loop %.2f  reduce+epilogue %.2f" % tuple((tr[i + 1] - tr[i]) / 2400.0 if tr[i + 1] and tr[i] else 0 for i in range(4))
    if tr[0] and tr[7] & 0xffff:
        nchunk = tr[7] & 0xffff
        extra += "  || wave0 last unit: %d chunks, per chunk: wait+stage %.0f cyc, mfma %.0f, refill-issue %.0f" % (nchunk, tr[5] / nchunk, tr[6] / nchunk, (tr[7] >> 16) / nchunk)
    print(f"{names[first + k][:46]:46s} {w:8.2f} {b:8.2f}{extra}")
    prev = v[2 + 2 * k]
print(f"total {tot:.1f} us for {count} ops")
