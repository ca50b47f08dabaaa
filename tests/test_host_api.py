"""Host-side logic that needs no GPU: the drop-in modules' state_dict surface, the C-ABI library's
exported symbols, loud failure without a GPU, the SDE objects' scalar API, the CPU-baseline port, and
the world_size-2 sharding path over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from score_based_multimodal_autoencoder_amd import _lib, arch, synth
from score_based_multimodal_autoencoder_amd import sde_helper2 as S
from score_based_multimodal_autoencoder_amd.h_vae_model import CelebAAttrNewBN, CelebAAttrNewBNAE
from score_based_multimodal_autoencoder_amd.h_vae_model_copy import ResAE, ResAEN, ResVAE, ResVAEN
from score_based_multimodal_autoencoder_amd.unet_model import Unet


def _shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_unet_state_dict_surface():
    m = Unet(dim=64, channels=5, dim_mults=(1, 2, 2, 2), with_time_emb=True)
    want = {k: tuple(s) for k, s in arch.unet_entries(dim=64, channels=5, dim_mults=(1, 2, 2, 2))}
    assert _shapes(m) == want and list(m.state_dict()) == list(want)
    assert sum(p.numel() for p in m.parameters()) == 10_964_947          # P5 parameter count
    assert len(m.state_dict()) == 286
    # load_state_dict round trip with the reference's key names
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(want).items()}
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["downs.1.2.fn.fn.to_out.1.weight"], sd["downs.1.2.fn.fn.to_out.1.weight"])
    with pytest.raises(NotImplementedError):
        Unet(dim=8, use_convnext=False)


def test_decoder_state_dict_surface():
    for cls, v, enc, dec, ch, s, z in [(ResVAE, "ResVAE", arch.POLY_ENC, arch.POLY_DEC, 3, 32, 64),
                                       (ResAE, "ResAE", arch.POLY_ENC, arch.POLY_DEC, 3, 32, 64),
                                       (ResVAEN, "ResVAEN", arch.CEL_IMG_ENC, arch.CEL_IMG_DEC, 3, 128, 256),
                                       (ResAEN, "ResAEN", arch.CEL_MASK_ENC, arch.CEL_MASK_DEC, 1, 128, 256)]:
        m = cls(enc, dec, s, z, ch)
        assert _shapes(m) == {k: tuple(sh) for k, sh in arch.resvae_entries(v, enc, dec, s, z, ch)}
    assert _shapes(CelebAAttrNewBN(256)) == {k: tuple(s) for k, s in arch.attr_entries(256)}
    assert _shapes(CelebAAttrNewBNAE(256)) == {k: tuple(s) for k, s in arch.attr_entries(256, ae=True)}


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "sbmae.h")).read()
    declared = sorted(set(re.findall(r"\b(sbm_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sbmae.h but not exported"
    assert set(_lib.EXPORTS) <= set(declared)
    assert L.sbm_abi_version() == 3
    # the spec tables are host-only: usable without a GPU, and identical to the Python arch tables
    cfg = _lib.unet_cfg_ints(64, 5, (1, 2, 2, 2), 8, 8)
    assert _lib.specs(_lib.SBM_UNET, cfg) == [(k, int(np.prod(s))) for k, s in arch.unet_entries(dim=64, channels=5, dim_mults=(1, 2, 2, 2))]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    m = Unet(dim=8, channels=2, dim_mults=(1, 2))
    with pytest.raises(RuntimeError, match="no CPU path|GPU"):
        m(torch.zeros(1, 2, 4, 4), torch.zeros(1))
    with pytest.raises(RuntimeError):
        ResVAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3).eval().decoder(torch.zeros(2, 64))
    with pytest.raises(RuntimeError):
        S.uncond_sampler((2, 2, 8, 8), m, "cpu", S.VPSDE(0.1, 5, 10))
    with pytest.raises(RuntimeError):
        _lib.Engine(_lib.SBM_UNET, _lib.unet_cfg_ints(8, 2, (1, 2), 4, 4), 1,
                    synth.fill_state_dict(arch.unet_state_shapes(dim=8, channels=2, dim_mults=(1, 2))))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "score_based_multimodal_autoencoder_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
            assert "planref" not in src, f


@pytest.mark.parametrize("kind,b0,b1", [("vp", 0.1, 5.0), ("vp", 0.1, 20.0), ("subvp", 0.1, 20.0), ("ve", 0.01, 50.0)])
def test_sde_objects_scalar_api(golden, kind, b0, b1):
    g = golden("sde_steps.npz")
    N = 1000
    tag = f"{kind}_{b0}_{b1}_{N}"
    sde = {"vp": S.VPSDE, "subvp": S.subVPSDE, "ve": S.VESDE}[kind](b0, b1, N)
    ts = torch.from_numpy(g[tag + "_ts"])
    drift, diff = sde.sde(torch.ones(N, 1, 1, 1), ts)
    assert_close(drift.reshape(N).numpy(), g[tag + "_dcoef"], 1e-6, 0, "drift")
    assert_close(np.asarray(diff), g[tag + "_g"], 1e-6, 1e-9, "diffusion")
    mean, std = sde.marginal_prob(torch.ones(N, 1, 1, 1), ts)
    assert_close(mean.reshape(N).numpy(), g[tag + "_mpmean"], 1e-6, 0, "marginal mean")
    assert_close(np.asarray(std), g[tag + "_mpstd"], 1e-6, 1e-9, "marginal std")
    assert sde.T == 1 and sde.N == N
    if kind == "vp":
        np.testing.assert_array_equal(sde.alphas[torch.from_numpy(g[tag + "_aidx"])].numpy(), g[tag + "_alpha"])
    assert tuple(sde.prior_sampling((3, 2, 8, 8)).shape) == (3, 2, 8, 8)
    with pytest.raises(RuntimeError):                   # guidance is built (tests/test_gpu_guidance.py); CPU tensors still have no path
        S.em_predictor(torch.zeros(1), torch.zeros(1), None, sde, cl_g=object(), given="0", all_mods="01")


def test_cpu_baseline_port_matches_reference(golden):
    """oracle/torch_port.py (the cpu_baseline leg of bench.py) against the reference's outputs."""
    from oracle import torch_port as TP
    g = golden("unet_forward.npz")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(
        arch.unet_state_shapes(dim=64, channels=5, dim_mults=(1, 2, 2, 2))).items()}
    with torch.no_grad():
        y = TP.unet_forward(sd, torch.from_numpy(g["P5_x"]), torch.from_numpy(g["P5_t"]), 64, 4).numpy()
    assert_close(y, g["P5_y"], 1e-5, 1e-5 * np.abs(g["P5_y"]).max(), "torch port Unet")
    d = golden("decoders.npz")
    dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(
        arch.resvae_state_shapes("ResVAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3)).items()}
    with torch.no_grad():
        img = TP.poly_decoder(dsd, torch.from_numpy(d["ResVAE_z"]), arch.POLY_DEC).numpy()
    assert_close(img, d["ResVAE_y"], 1e-5, 1e-5 * np.abs(d["ResVAE_y"]).max(), "torch port decoder")
    s = golden("sde_steps.npz")
    ts = float(s["vp_0.1_5.0_100_ts"][50])
    xn, xm = TP.vp_em_step(torch.from_numpy(s["x"]), torch.ones(4) * ts, torch.from_numpy(s["score"]), torch.from_numpy(s["noise"]), 0.1, 5.0, 100)
    assert_close(xn.numpy(), s["vp_0.1_5.0_100_em50_x"], 1e-6, 1e-6, "torch port em step")


def test_sde_object_api_prior_logp_discretize_reverse(golden):
    """the parts of the SDE objects no script calls but the drop-in surface carries, against the real classes
    (sde_helper2.py:236-326, 364-381, 415-421, 457-473): prior_logp, discretize, reverse().sde / .discretize."""
    g = golden("sde_api.npz")
    x, sc, t = torch.from_numpy(g["x"]), torch.from_numpy(g["score"]), torch.from_numpy(g["t"])
    for tag, sde in (("vp", S.VPSDE(0.1, 20.0, 1000)), ("subvp", S.subVPSDE(0.1, 20.0, 1000)), ("ve", S.VESDE(0.01, 50.0, 1000))):
        assert_close(sde.prior_logp(x).numpy(), g[f"{tag}_logp"], 1e-6, 1e-4, tag + " prior_logp")
        f, G = sde.discretize(x, t)
        assert_close(f.numpy(), g[f"{tag}_disc_f"], 1e-6, 1e-7, tag + " discretize f")
        assert_close(G.numpy(), g[f"{tag}_disc_G"], 1e-6, 1e-7, tag + " discretize G")
        r = sde.reverse(lambda a, b: sc, False)
        d, gg = r.sde(x, t)
        assert_close(d.numpy(), g[f"{tag}_rev_drift"], 1e-6, 1e-5, tag + " reverse drift")
        assert_close(np.broadcast_to(gg.numpy(), g[f"{tag}_rev_g"].shape), g[f"{tag}_rev_g"], 1e-6, 1e-7, tag + " reverse diffusion")
        rf, rG = r.discretize(x, t)
        assert_close(rf.numpy(), g[f"{tag}_rev_disc_f"], 1e-6, 1e-6, tag + " reverse discretize f")
        assert_close(rG.numpy(), g[f"{tag}_rev_disc_G"], 1e-6, 1e-7, tag + " reverse discretize G")
        assert r.N == sde.N and r.T == sde.T


def test_default_sampler_key_is_fresh_per_call_and_follows_manual_seed():
    """with seed=None every sampler call draws a new Philox key from torch's default generator (independent noise per
    call, as the reference's per-call randn_like), and torch.manual_seed reproduces the sequence."""
    from score_based_multimodal_autoencoder_amd._module import fresh_seed
    torch.manual_seed(3)
    a = [fresh_seed(None) for _ in range(3)]
    torch.manual_seed(3)
    b = [fresh_seed(None) for _ in range(3)]
    assert a == b and len(set(a)) == 3
    assert fresh_seed(17) == 17 and fresh_seed(-1) == 0xFFFFFFFFFFFFFFFF


def test_polymnist_scalar_form_surface():
    """train_poly_unet_cont.py:174: uncond_sampler(sample_shape, model, device, T, eps, beta_0, beta_1, N, ...) exists with the
    reference's positional order and refuses the CPU (no fallback)."""
    import inspect
    from score_based_multimodal_autoencoder_amd import train_poly_unet_cont as P
    names = list(inspect.signature(P.uncond_sampler).parameters)
    assert names[:15] == ["sample_shape", "model", "device", "T", "eps", "beta_0", "beta_1", "N", "probability_flow", "pc", "n_steps",
                          "target_snr", "cl_g", "cl_s", "target"]
    assert list(inspect.signature(P.corrector).parameters)[:9] == ["x", "t", "score_fn", "beta_0", "beta_1", "N", "T", "n_steps", "target_snr"]
    with pytest.raises(RuntimeError):
        P.uncond_sampler((2, 2, 8, 8), lambda x, t: x, "cpu", N=5)
    with pytest.raises(NotImplementedError):
        P.uncond_sampler((2, 2, 8, 8), lambda x, t: x, "cuda", T=2, N=5)
    mean, std = P.marginal_prob(torch.ones(2, 1, 1, 1), torch.tensor([1.0, 0.5]), 0.1, 5.0)
    rm, rs = S.VPSDE(0.1, 5.0, 10).marginal_prob(torch.ones(2, 1, 1, 1), torch.tensor([1.0, 0.5]))
    assert torch.equal(mean, rm) and torch.equal(std, rs)


def test_guidance_module_state_dict_surface():
    from score_based_multimodal_autoencoder_amd.lat_sm2_model import ClwithTime2, ClwithTime3
    assert _shapes(ClwithTime2(n_mod=2, size_z=64, n_class=1)) == {k: tuple(s) for k, s in arch.clg_entries(2, 2, 64, 1)}
    assert _shapes(ClwithTime3(n_mod=2, size_z=256, n_class=1)) == {k: tuple(s) for k, s in arch.clg_entries(3, 2, 256, 1)}
    assert [n for n, _ in _lib.specs(_lib.SBM_CLG, [3, 2, 64, 1])] == [k for k, _ in arch.clg_entries(3, 2, 64, 1)]
    with pytest.raises(NotImplementedError):
        ClwithTime2(n_mod=3)


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from score_based_multimodal_autoencoder_amd import distributed as D
rank, world = D.init_from_env("gloo")
assert world == 2
G = 12
def sample_fn(start, count):           # a pure function of the GLOBAL row index, like the Philox-keyed sampler
    rows = torch.arange(start, start + count, dtype=torch.float32)
    return rows[:, None, None].expand(count, 2, 3) * 1.5 + 1.0, rows[:, None] * torch.ones(1, 5)
a, b = D.sample_sharded(sample_fn, G)
ra, rb = sample_fn(0, G)
assert torch.equal(a, ra) and torch.equal(b, rb), "gathered result differs from the unsharded one"
assert D.shard_range(G, rank, world) == (rank * 6, 6)
s = D.langevin_norm_allreduce(torch.tensor([1.0 + rank, 2.0]))
assert torch.equal(s, torch.tensor([3.0, 4.0]))
# one fresh key for all ranks, agreed explicitly (the sampling helpers themselves never issue a collective: fresh_seed is local)
from score_based_multimodal_autoencoder_amd._module import fresh_seed
torch.manual_seed(100 + rank)
local = fresh_seed(None)
torch.manual_seed(100 + rank)
key = D.shared_seed()
torch.manual_seed(100)
assert key == fresh_seed(None), "shared_seed is not rank 0's draw"
assert (key == local) == (rank == 0)
assert D.shared_seed(seed=(1 << 64) - 5 + rank) == (1 << 64) - 5
got = D.sample_sharded(lambda start, count, seed: torch.full((count, 1), float(seed % 1000)), G, share_seed=True)
assert got.shape == (G, 1) and bool((got == got[0]).all()), "ranks sampled under different keys"
if rank == 0:
    fresh_seed(None)          # a rank-conditional call must not hang: no collective inside
try:
    D.shard_range(13, rank, world); raise SystemExit("uneven split accepted")
except ValueError:
    pass
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_sharding_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_bench_launches_its_own_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself (fresh child processes, before
    any GPU call of the parent) and report the world size the process group really has.  --dry runs the launcher, the sharded path
    (distributed.sample_sharded), the single all-gather and the max-over-ranks timing on the CPU with gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["gathered_rows"] == 16 and j["rows_in_global_order"] and j["steps"] == 2
    # with a launcher's environment the script must NOT spawn again: one process, world size from the environment
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry"], env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_launcher_ends_the_siblings_of_a_rank_that_dies():
    """one rank dying at start-up must not leave the others in the rendezvous until the collective time-out: the launcher
    polls all children, terminates the rest, returns non-zero and shows the failed rank."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SBM_BENCH_DRY_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "rank 1 exited with 3" in r.stderr, (r.returncode, r.stderr[-1000:])
    assert time.monotonic() - t0 < 120


def test_pixel_kernel_register_audit(tmp_path):
    """csrc/pixrun.hip keeps its weight ring and epilogue operands in AGPRs with HARD-CODED indices (a40 .. a127), written and read by
    inline asm.  hipcc knows nothing about values that live there between two asm statements: once the kernel needs more than its 128
    VGPRs it spills into AGPRs -- lowest free index first -- and a spill into the ring is silent corruption (it happened during round 3:
    wrong first K steps behind every depth-wise op, no fault).  The kernel is therefore audited at build level: outside the asm blocks no
    instruction may name a40 or above, nothing may go to scratch, and the register total must fit two waves per SIMD."""
    import re
    src = os.path.join(ROOT, "score_based_multimodal_autoencoder_amd", "csrc", "pixrun.hip")
    out = tmp_path / "pixrun.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kern = text[text.index("pixrun_kernel"):]
    in_asm, bad = False, []
    for ln in kern.splitlines():
        if ";;#ASMSTART" in ln: in_asm = True; continue
        if ";;#ASMEND" in ln: in_asm = False; continue
        if in_asm or ln.lstrip().startswith((";", ".")): continue
        for m in re.finditer(r"\ba\[?(\d+)", ln):
            if int(m.group(1)) >= 40: bad.append(ln.strip())
    assert not bad, "compiler-generated instructions touch the reserved AGPRs a40..a127:\n" + "\n".join(bad[:10])
    meta = text[text.index("amdhsa.kernels:"):]            # one kernel in this translation unit
    agpr = int(re.search(r"\.agpr_count:\s+(\d+)", meta).group(1)); total = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
    assert agpr == 128 and total <= 256 and scratch == 0, (agpr, total, scratch)
    # the hand-written part is where it should be: 16 ring loads of four dwordx4 each + the epilogue operand reads
    assert len(re.findall(r"v_accvgpr_read_b32 v\d+, a1(0[4-9]|1\d|2[0-7])\b", kern)) == 24


def test_engine_fingerprint_sees_every_kind_of_weight_change_and_does_not_walk_the_tree_per_call():
    """The engine cache key: the tensor list is cached (the tree walk cost as much as a forward), and still every way a
    weight can change invalidates the engines: in-place update, `.data` swap, `load_state_dict`, a NEW nn.Parameter assigned to a
    sub-module, a buffer re-registered."""
    m = Unet(dim=8, channels=2, dim_mults=(1, 2)).eval()
    fp0 = m._fingerprint()
    lst = m.__dict__["_sbm_tensors"]
    assert m._fingerprint() == fp0 and m.__dict__["_sbm_tensors"] is lst            # second call: same list object, no walk
    p = next(m.parameters())
    with torch.no_grad():
        p.add_(1.0)
    fp1 = m._fingerprint()
    assert fp1 != fp0
    p.data = p.data.clone()
    fp2 = m._fingerprint()
    assert fp2 != fp1
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    fp3 = m._fingerprint()
    assert fp3 != fp2
    name = next(n for n, _ in m.named_parameters())
    owner = m
    for part in name.split(".")[:-1]:
        owner = getattr(owner, part)
    setattr(owner, name.split(".")[-1], torch.nn.Parameter(torch.zeros_like(p)))
    fp4 = m._fingerprint()
    assert fp4 != fp3 and m.__dict__["_sbm_tensors"] is not lst                       # the list was re-collected
    assert any(t is getattr(owner, name.split(".")[-1]) for t in m.__dict__["_sbm_tensors"])
    m.double()
    fp5 = m._fingerprint()
    assert fp5 != fp4
    # a sub-module replaced by an already-constructed one, and a write that bypasses registration (`_parameters[name] = t`)
    other = Unet(dim=8, channels=2, dim_mults=(1, 2)).eval().double()
    child = next(n for n, mm in m.named_children() if sum(1 for _ in mm.parameters()) > 0)
    setattr(m, child, getattr(other, child))
    fp6 = m._fingerprint()
    assert fp6 != fp5 and any(t is next(getattr(other, child).parameters()) for t in m.__dict__["_sbm_tensors"])
    name = next(n for n, _ in m.named_parameters())
    owner = m
    for part in name.split(".")[:-1]:
        owner = getattr(owner, part)
    new = torch.nn.Parameter(torch.ones_like(owner._parameters[name.split(".")[-1]]))
    owner._parameters[name.split(".")[-1]] = new                                       # no registration hook fires
    assert m._fingerprint() != fp6 and any(t is new for t in m.__dict__["_sbm_tensors"])


def test_release_engines_closes_all_and_empties_the_cache_even_when_a_close_raises():
    """Engine.close() may raise (tear-down error, unreported kernel failure); the cache must not keep dead engines."""
    m = Unet(dim=8, channels=2, dim_mults=(1, 2)).eval()
    closed = []

    class E:
        def __init__(self, bad): self.bad = bad
        def close(self):
            closed.append(self)
            if self.bad: raise RuntimeError("tear-down failed")
    m._sbm_engines = {1: E(True), 2: E(False), 3: E(True)}
    m._sbm_fingerprint = m._fingerprint()
    with pytest.raises(RuntimeError, match="tear-down failed"):
        m.release_engines()
    assert len(closed) == 3 and m._sbm_engines == {} and m._sbm_fingerprint is None
    m.release_engines()                                                                # nothing left: no error
