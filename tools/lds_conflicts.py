#!/usr/bin/env python3
"""Which ops of the pixel kernel make its LDS bank conflicts: every pixel-kernel op of the config-2 score net is run ALONE
(a one-op program of the sample-owned kernel, B = 256) three times in a fixed order, under

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d DIR -o t --output-format csv -- python3 tools/lds_conflicts.py
    python3 tools/lds_conflicts.py --summarise DIR

and the summary maps the dispatches back to the ops (dispatch order) and adds the counters up by op class.  A one-op program stages its
inputs from the workspace first (a few LDS stores), so the classes carry that small extra."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORDER = os.path.join(ROOT, "gpurun_out", "lds_conflicts_order.json")
REPS = 3


def klass(name, P):
    n = name.lower()
    if "ds_conv" in n: return f"dwconv7 {P}px"
    if ".core" in n: return f"linattn core {P}px"
    if "+residual" in n: return f"gn_res {P}px"
    return f"gemm {P}px" + (" (16-row MFMA)" if P >= 16 else (" (4-row MFMA)" if P > 1 else " (vector mat-vec)"))


if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    order = json.load(open(ORDER))
    rows = {}
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pixrun_kernel" not in r["Kernel_Name"]: continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    assert len(ids) >= len(order) * REPS, (len(ids), len(order))
    ids = ids[-len(order) * REPS:]                       # the engine's own warm-up forwards come first
    acc = {}
    for j, (name, P) in enumerate(order):
        for r in range(REPS):
            c = rows[ids[j * REPS + r]]
            a = acc.setdefault(klass(name, P), {"ops": set(), "conf": 0.0, "active": 0.0, "busy": 0.0})
            a["ops"].add(name); a["conf"] += c.get("SQ_LDS_BANK_CONFLICT", 0.0) / REPS; a["active"] += c.get("SQ_LDS_IDX_ACTIVE", 0.0) / REPS; a["busy"] += c.get("SQ_BUSY_CYCLES", 0.0) / REPS
    tc = sum(a["conf"] for a in acc.values()); ta = sum(a["active"] for a in acc.values())
    print(f"{'op class':34s} {'ops':>4s} {'conflict cycles':>16s} {'share of all':>13s} {'LDS active':>14s} {'conflict / active':>18s}")
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["conf"]):
        print(f"{k:34s} {len(a['ops']):4d} {a['conf']:16.0f} {a['conf'] / max(tc, 1):13.3f} {a['active']:14.0f} {a['conf'] / max(a['active'], 1):18.3f}")
    print(f"{'all':34s} {sum(len(a['ops']) for a in acc.values()):4d} {tc:16.0f} {1.0:13.3f} {ta:14.0f} {tc / max(ta, 1):18.3f}")
    sys.exit(0)

import ctypes
import torch
from score_based_multimodal_autoencoder_amd import _lib, synth
from score_based_multimodal_autoencoder_amd.unet_model import Unet
B = 256
m = synth.fill_module_(Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2)).eval()).cuda()
eng = m._engine_for(B, 8, 8)
x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda"); o = torch.empty_like(x)
eng.forward(x, t, o); torch.cuda.synchronize()
L = _lib.lib(); n = eng.info["n_ops"]; buf = ctypes.create_string_buffer(256)
st = (ctypes.c_uint64 * 512)(); kinds = (ctypes.c_int * 512)(); nout = ctypes.c_int(0)
info = (ctypes.c_double * 16)()
order = []
ops = []
for i in range(n):
    L.sbm_debug_op_name(eng.h, i, buf, 256); name = buf.value.decode()
    ops.append(name)
# pixel ops = those a one-op program can be built for; their pixel counts from the plan dump names are not exposed: take them from the level
# the op sits on (names carry the block: downs.0 / final 8x8, downs.1 / ups.2 4x4, downs.2 / ups.1 2x2, downs.3 / mid / ups.0 1x1; the
# strided convs belong to the level they WRITE)
def pixels(name):
    nm = name.split(":", 1)[-1]
    tbl = {"downs.0.3": 16, "downs.1.3": 4, "downs.2.3": 1, "ups.0.3": 4, "ups.1.3": 16, "ups.2.3": 64}
    for k, v in tbl.items():
        if nm.startswith(k): return v
    for k, v in (("downs.0", 64), ("final_conv", 64), ("downs.1", 16), ("ups.2", 16), ("downs.2", 4), ("ups.1", 4), ("downs.3", 1), ("mid", 1), ("ups.0", 1)):
        if nm.startswith(k): return v
    return 0
torch.cuda.synchronize()
for i in range(n):
    if L.sbm_debug_forward_range(eng.h, x.data_ptr(), t.data_ptr(), o.data_ptr(), i, i + 1, None) != 0: continue
torch.cuda.synchronize()
# now the measured dispatches: only ops the pixel kernel runs
probe = [i for i in range(n) if L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, 512, ctypes.byref(nout)) == 0]
torch.cuda.synchronize()
for i in probe:
    for r in range(REPS):
        _lib.check(L.sbm_debug_forward_range(eng.h, x.data_ptr(), t.data_ptr(), o.data_ptr(), i, i + 1, None))
    order.append((ops[i], pixels(ops[i])))
torch.cuda.synchronize()
os.makedirs(os.path.dirname(ORDER), exist_ok=True)
json.dump(order, open(ORDER, "w"))
print(f"{len(order)} pixel-kernel ops x {REPS} dispatches")
