#!/usr/bin/env python3
"""Op by op: the pixel-mode plan on the GPU (every marked op as a one-op program of the sample-owned kernel) against the scalar host
interpreter of the SAME plan (tests/interp.cpp with SBMREF_PIX=1).  Prints the error of every op; stops at the first that is wrong."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SBMREF_PIX"] = "1"
import numpy as np, torch
import planref
from score_based_multimodal_autoencoder_amd import _lib, arch, synth
name = sys.argv[1] if len(sys.argv) > 1 else "P2"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "unet_forward.npz")))
CASES = {"P2": dict(dim=64, channels=2, dim_mults=(1, 2, 2, 2)), "P5": dict(dim=64, channels=5, dim_mults=(1, 2, 2, 2))}
c = CASES[name]
x, t = g[f"{name}_x"], g[f"{name}_t"]
sd = synth.fill_state_dict(arch.unet_state_shapes(**c))
cfg = _lib.unet_cfg_ints(c["dim"], c["channels"], c["dim_mults"], x.shape[2], x.shape[3])
B = x.shape[0]
ref = planref.RefPlan(0, cfg, B, sd)
eng = _lib.Engine(0, cfg, B, sd)
print("uses_pix", eng.info["uses_pix"], eng.info["pix_note"])
L = _lib.lib()
n_ws = L.sbm_debug_workspace_floats(eng.h)
print("workspace floats", n_ws, ref.info["ws_floats"])
assert n_ws == ref.info["ws_floats"]
xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
out_d = torch.zeros(g[f"{name}_y"].shape, device="cuda"); out_h = np.zeros(g[f"{name}_y"].shape, np.float32)
ws_ref = ref.workspace(); host = np.empty(n_ws, np.float32)
bad = 0
for i in range(ref.info["n_ops"]):
    before = ws_ref.copy()
    ref.run_range(x, t, out_h, i, i + 1)
    _lib.check(L.sbm_debug_forward_range(eng.h, xd.data_ptr(), td.data_ptr(), out_d.data_ptr(), i, i + 1, None))
    _lib.check(L.sbm_debug_read_workspace(eng.h, 0, n_ws, host.ctypes.data))
    changed = np.isfinite(ws_ref) & ~((ws_ref == before) & np.isfinite(before))
    if not changed.any():
        print(f"op {i:3d} {ref.op_name(i)[:50]:50s} (no workspace change)"); continue
    want, got = ws_ref[changed], host[changed]
    scale = max(np.abs(want).max(), 1e-6); err = np.abs(got - want).max()
    flag = "" if err <= 3e-5 * scale else "   <-- WRONG"
    print(f"op {i:3d} {ref.op_name(i)[:50]:50s} changed {int(changed.sum()):8d}  err {err:.3e} scale {scale:.3e}{flag}")
    if flag:
        idx = np.flatnonzero(changed); w = np.abs(got - want); k = np.argsort(-w)[:8]
        print("     worst offsets (workspace floats):", [(int(idx[j]), float(got[j]), float(want[j])) for j in k])
        bad += 1
        if bad >= 3: break
        host_fix = host.copy(); host_fix[changed] = want          # continue from the reference state: one bug at a time
print("final", float(np.abs(out_d.cpu().numpy() - out_h).max()))

# ---- growing sub-ranges of every run of marked ops: ops [first, k) as ONE launch, the output of op k - 1 against the interpreter
print("\nruns (sub-ranges as single launches):")
n = ref.info["n_ops"]
st = (ctypes.c_uint64 * 512)(); kinds = (ctypes.c_int * 512)(); nout = ctypes.c_int(0)
marked = [L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, 512, ctypes.byref(nout)) == 0 for i in range(n)]
ref2 = planref.RefPlan(0, cfg, B, sd)
ws2 = ref2.workspace()
i = 0
while i < n:
    if not marked[i]:
        ref2.run_range(x, t, out_h, i, i + 1); i += 1; continue
    j = i
    while j + 1 < n and marked[j + 1]: j += 1
    base_state = ws2.copy()
    wrong = 0
    for k in range(i + 1, j + 2):
        before = ws2.copy()
        ref2.run_range(x, t, out_h, k - 1, k)
        changed = np.isfinite(ws2) & ~((ws2 == before) & np.isfinite(before))
        # GPU: restore the state in front of the run, execute [i, k) as one launch
        st0 = np.where(np.isfinite(base_state), base_state, 0.0).astype(np.float32)
        # (no write entry point for the workspace: replay the ops before the run on the GPU instead)
        _lib.check(L.sbm_debug_forward_range(eng.h, xd.data_ptr(), td.data_ptr(), out_d.data_ptr(), 0, i, None))
        _lib.check(L.sbm_debug_forward_range(eng.h, xd.data_ptr(), td.data_ptr(), out_d.data_ptr(), i, k, None))
        _lib.check(L.sbm_debug_read_workspace(eng.h, 0, n_ws, host.ctypes.data))
        nm = ref2.op_name(k - 1)
        if nm.startswith("1:"):                      # depth-wise: skip its statistics partials (layout differs by design), compare the tensor only
            pass
        want, got = ws2[changed], host[changed]
        if want.size == 0: continue
        # statistics regions are small; judge by the bulk: relative error of the 99.5th percentile
        scale = max(np.abs(want).max(), 1e-6); e = np.abs(got - want)
        frac_bad = float((e > 3e-5 * scale).mean())
        print(f"  [{i},{k}) last op {k - 1:3d} {nm[:46]:46s} changed {want.size:7d} max err {e.max():.3e} scale {scale:.3e} bad fraction {frac_bad:.4f}")
        if frac_bad > 0.02:
            idx = np.flatnonzero(changed); badi = idx[e > 3e-5 * scale] - idx.min()
            ldg = int(os.environ.get("OPCHECK_LD", "128"))
            pix = badi // (B * ldg); ch = badi % ldg; bb = (badi // ldg) % B
            lim = int(os.environ.get("OPCHECK_TENSOR_FLOATS", "6144"))
            order = np.argsort(-e); small = [j for j in order if (idx[j] - idx.min()) < lim and e[j] > 3e-5 * scale][:10]
            print("     tensor entries in error (pixel, sample, channel, got, want):", [(int((idx[j] - idx.min()) // (B * ldg)), int(((idx[j] - idx.min()) // ldg) % B), int((idx[j] - idx.min()) % ldg), float(got[j]), float(want[j])) for j in small])
            print("     wrong entries by pixel:", np.bincount(pix, minlength=64)[:70].tolist())
            print("     wrong entries by channel/8:", np.bincount(ch // 8, minlength=16).tolist(), " by sample:", np.bincount(bb).tolist())
            wrong += 1
            if wrong >= 2: break
    i = j + 1
