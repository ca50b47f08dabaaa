#!/usr/bin/env python3
"""Where and when the workgroups of one seg_gemm launch run (diagnostics): SBM_GEMM_TIMELINE makes sbm_debug_gemm_stamps dump every
workgroup's (start, end) on the 100 MHz clock and its HW_ID / XCC_ID.
    python tools/gemm_timeline.py --config 4 --ops 8,97"""
import argparse, collections, ctypes, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=4); ap.add_argument("--ops", type=str, default="8")
a = ap.parse_args()
path = tempfile.mktemp(suffix=".txt"); os.environ["SBM_GEMM_TIMELINE"] = path
import torch, bench
from score_based_multimodal_autoencoder_amd import _lib
score, _ = bench.build_models(a.config, torch.device("cuda", 0)); W = bench.WORK[a.config]
eng = score._engine_for(W["B"], W["hw"], W["hw"]) if a.config != 5 else score._engine_for(W["B"], W["hw"], W["hw"], True)
x = torch.randn(W["B"], W["n_mod"], W["hw"], W["hw"], device="cuda"); t = torch.rand(W["B"], device="cuda"); out = torch.empty_like(x)
if a.config == 5:
    z = torch.randn(W["B"], 512, device="cuda"); _lib.check(_lib.lib().sbm_model_set_aux(eng.h, z.data_ptr())); eng.forward_aux(x, t, z, out)
else:
    eng.forward(x, t, out)
torch.cuda.synchronize()
L = _lib.lib(); st = (ctypes.c_uint64 * 16)(); buf = ctypes.create_string_buffer(256)
for op in [int(v) for v in a.ops.split(",")]:
    open(path, "w").close()
    for _ in range(2):
        open(path, "w").close()
        _lib.check(L.sbm_debug_gemm_stamps(eng.h, x.data_ptr(), out.data_ptr(), op, st))
    L.sbm_debug_op_name(eng.h, op, buf, 256)
    rows = [ln.split() for ln in open(path).read().splitlines()[1:]]
    wg = [(int(r[0]), int(r[1]), int(r[2]), int(r[3], 16)) for r in rows]
    det = {int(r[0]): (int(r[4]), int(r[5]), int(r[6]), int(r[7])) for r in rows}      # wait cycles, compute cycles, chunks, loop cycles
    bych = collections.defaultdict(list)
    for b, s_, e, hw in wg:
        w, c, n, lp = det[b]
        tot = w >> 32; w &= 0xffffffff
        if n: bych[n].append(((e - s_) / 100, w / n, c / n, lp / n, tot / max(e - s_, 1) * 100, tot - lp))
    for n, v in sorted(bych.items()):
        med = lambda i: sorted(x[i] for x in v)[len(v) // 2]
        print(f"  {len(v):4d} workgroups with {n:4d} chunks: duration median {med(0):7.1f} us (max {max(x[0] for x in v):7.1f}); per chunk: wait {med(1):6.0f}  compute {med(2):6.0f}  loop {med(3):6.0f} cycles; in-kernel clock {med(4):5.0f} MHz; cycles outside the loop {med(5):7.0f}")
    if not wg:
        print(f"op {op}: no timeline (split-K or not a seg_gemm launch)"); continue
    t0 = min(w[1] for w in wg); t1 = max(w[2] for w in wg)
    cu = collections.defaultdict(list)
    for b, s, e, hw in wg:
        key = (hw >> 32 & 7, hw >> 13 & 7, hw >> 12 & 1, hw >> 8 & 15)         # XCC, SE, SH, CU
        cu[key].append((s - t0, e - t0, b))
    print(f"op {op} {buf.value.decode()[:50]}: {len(wg)} workgroups on {len(cu)} CUs, launch span {(t1 - t0) / 100:.1f} us")
    per = collections.Counter(len(v) for v in cu.values())
    print("  workgroups per CU:", dict(sorted(per.items())))
    starts = sorted(w[1] - t0 for w in wg); ends = sorted(w[2] - t0 for w in wg); dur = sorted(w[2] - w[1] for w in wg)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))] / 100
    print("  start us: p0 %.1f p50 %.1f p90 %.1f p100 %.1f | end us: p0 %.1f p50 %.1f p90 %.1f p100 %.1f | duration us: p0 %.1f p50 %.1f p90 %.1f p100 %.1f"
          % (q(starts, 0), q(starts, .5), q(starts, .9), q(starts, 1), q(ends, 0), q(ends, .5), q(ends, .9), q(ends, 1), q(dur, 0), q(dur, .5), q(dur, .9), q(dur, 1)))
    byx = collections.defaultdict(list)
    for b, s_, e, hw in wg: byx[hw >> 32 & 7].append((e - s_) / 100)
    print("  duration by XCD (min / median / max us):", {k: (round(min(v), 1), round(sorted(v)[len(v) // 2], 1), round(max(v), 1)) for k, v in sorted(byx.items())})
    conc = collections.Counter()
    for v in cu.values():
        ev = sorted([(s, 1) for s, e, b in v] + [(e, -1) for s, e, b in v]); c = 0; last = 0
        for tt, d in ev:
            conc[c] += tt - last; last = tt; c += d
        conc[0] += (t1 - t0) - last
    tot = sum(conc.values())
    print("  share of CU-time with k workgroups resident:", {k: round(v / tot, 3) for k, v in sorted(conc.items())})
    for key in sorted(cu)[:3]:
        print("  e.g. CU", key, [(round(s / 100, 1), round(e / 100, 1), b) for s, e, b in sorted(cu[key])])
os.unlink(path)
