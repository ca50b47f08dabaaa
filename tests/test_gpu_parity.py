"""GPU parity: the HIP path (through the C ABI / the drop-in modules) against the golden fixtures
produced by the real reference, against the oracle on seeded inputs, and -- at the bench
configuration's full size -- through size-independent properties.

Stated tolerances: per-forward rtol=1e-5, atol=1e-5*max|y|; N-step trajectories with
injected noise rtol=1e-4, atol=1e-4*max|x|.
"""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

UNET_CASES = {
    "tiny": dict(dim=8, channels=2, dim_mults=(1, 2, 2, 2)),
    "tiny2lvl": dict(dim=16, channels=3, dim_mults=(1, 2)),
    "P2": dict(dim=64, channels=2, dim_mults=(1, 2, 2, 2)),
    "P5": dict(dim=64, channels=5, dim_mults=(1, 2, 2, 2)),
    "P10": dict(dim=128, channels=10, dim_mults=(1, 2, 2, 2)),
    "C3": dict(dim=256, channels=3, dim_mults=(1, 2, 2, 2, 2)),
}


@pytest.fixture(scope="module")
def pkg():
    import score_based_multimodal_autoencoder_amd as P
    from score_based_multimodal_autoencoder_amd import _lib, arch, h_vae_model, h_vae_model_copy, sde_helper2, synth, unet_model
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()                     # fails loudly if the HIP library is missing: there is no fallback
    return dict(lib=_lib, arch=arch, synth=synth, unet=unet_model, sde=sde_helper2, vae=h_vae_model_copy, attr=h_vae_model)


def _unet(pkg, name):
    c = UNET_CASES[name]
    m = pkg["unet"].Unet(dim=c["dim"], channels=c["channels"], dim_mults=c["dim_mults"]).eval()
    pkg["synth"].fill_module_(m)
    return m.cuda()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------
def test_no_cpu_path(pkg):
    m = pkg["unet"].Unet(dim=8, channels=2, dim_mults=(1, 2))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 4, 4), torch.zeros(1))


@pytest.mark.parametrize("name", ["tiny", "tiny2lvl"])
def test_unet_op_by_op_against_plan_interpreter(pkg, name, golden):
    """Every plan op's output on the GPU vs the scalar host interpreter of the same descriptors."""
    import planref
    g = golden("unet_forward.npz")
    c = UNET_CASES[name]
    x, t = g[f"{name}_x"], g[f"{name}_t"]
    sd = pkg["synth"].fill_state_dict(pkg["arch"].unet_state_shapes(**c))
    cfg = pkg["lib"].unet_cfg_ints(c["dim"], c["channels"], c["dim_mults"], x.shape[2], x.shape[3])
    B = x.shape[0]
    ref = planref.RefPlan(0, cfg, B, sd)
    eng = pkg["lib"].Engine(0, cfg, B, sd)
    L = pkg["lib"].lib()
    n_ws = L.sbm_debug_workspace_floats(eng.h)
    assert n_ws == ref.info["ws_floats"]
    xd, td = dev(x), dev(t)
    out_d = torch.zeros(g[f"{name}_y"].shape, device="cuda")
    out_h = np.zeros(g[f"{name}_y"].shape, np.float32)
    ws_ref = ref.workspace()
    host = np.empty(n_ws, np.float32)
    for i in range(ref.info["n_ops"]):
        before = ws_ref.copy()
        ref.run_range(x, t, out_h, i, i + 1)
        pkg["lib"].check(L.sbm_debug_forward_range(eng.h, xd.data_ptr(), td.data_ptr(), out_d.data_ptr(), i, i + 1, None))
        pkg["lib"].check(L.sbm_debug_read_workspace(eng.h, 0, n_ws, host.ctypes.data))
        changed = np.isfinite(ws_ref) & ~((ws_ref == before) & np.isfinite(before))
        if not changed.any():
            continue
        want, got = ws_ref[changed], host[changed]
        scale = max(np.abs(want).max(), 1e-6)
        err = np.abs(got - want).max()
        assert err <= 3e-5 * scale, f"op {i} ({ref.op_name(i)}): max err {err:.3e} vs scale {scale:.3e}"
    assert_close(out_d.cpu().numpy(), out_h, 1e-5, 3e-5 * np.abs(out_h).max(), "final output")


@pytest.mark.parametrize("name", ["tiny", "tiny2lvl", "P2", "P5", "P10", "C3"])
def test_unet_forward_matches_reference(pkg, golden, name):
    g = golden("unet_forward.npz")
    m = _unet(pkg, name)
    y = m(dev(g[f"{name}_x"]), dev(g[f"{name}_t"])).cpu().numpy()
    want = g[f"{name}_y"]
    assert_close(y, want, 1e-5, 1e-5 * np.abs(want).max(), name)


def test_unet_integer_sigma_index_time(pkg, golden):
    g = golden("unet_forward.npz")
    m = _unet(pkg, "tiny")
    y = m(dev(g["tiny_x"]), torch.from_numpy(g["tiny_ti"]).cuda()).cpu().numpy()
    assert_close(y, g["tiny_yi"], 1e-5, 1e-5 * np.abs(g["tiny_yi"]).max(), "sigma index time")


def test_unet_ragged_batch_and_padded_input(pkg):
    from oracle import unet as OU
    c = UNET_CASES["tiny"]
    sd = pkg["synth"].fill_state_dict(pkg["arch"].unet_state_shapes(**c))
    m = _unet(pkg, "tiny")
    for shape in [(5, 2, 6, 6), (67, 2, 8, 8), (1, 2, 8, 8)]:
        x = pkg["synth"].noise(3, shape[0], shape)
        t = np.linspace(1.0, 1e-3, shape[0]).astype(np.float32)
        want = OU.unet_forward(sd, x, t, **c)
        assert_close(m(dev(x), dev(t)).cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), str(shape))


def test_unet_full_batch_matches_oracle(pkg):
    """Config-2 shape: Unet(64, ch=5) at B=256 (128x128 / 128x64 tiles) vs the numpy oracle."""
    from oracle import unet as OU
    c = UNET_CASES["P5"]
    sd = pkg["synth"].fill_state_dict(pkg["arch"].unet_state_shapes(**c))
    m = _unet(pkg, "P5")
    x = pkg["synth"].noise(11, 0, (256, 5, 8, 8))
    t = np.linspace(1.0, 1e-3, 256).astype(np.float32)
    y = m(dev(x), dev(t)).cpu().numpy()
    want = OU.unet_forward(sd, x, t, **c)
    assert_close(y, want, 1e-5, 1e-5 * np.abs(want).max(), "B=256")
    # rows are independent: a 64-row sub-batch (different tile configuration) gives the same rows
    y64 = m(dev(x[:64]), dev(t[:64])).cpu().numpy()
    assert_close(y64, y[:64], 1e-6, 2e-6 * np.abs(want).max(), "sub-batch")
    # load_state_dict with new weights invalidates the packed engine
    sd2 = {k: torch.from_numpy(np.asarray(v)) * (0.5 if k.endswith("final_conv.1.weight") else 1.0) for k, v in sd.items()}
    m.load_state_dict(sd2)
    y2 = m(dev(x[:8]), dev(t[:8])).cpu().numpy()
    assert not np.allclose(y2, y[:8])


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ResVAE", "ResAE"])
def test_poly_decoders(pkg, golden, name):
    g = golden("decoders.npz")
    A = pkg["arch"]
    m = getattr(pkg["vae"], name)(A.POLY_ENC, A.POLY_DEC, 32, 64, 3).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    want = g[f"{name}_y"]
    assert_close(m.decoder(dev(g[f"{name}_z"])).cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), name)
    if name == "ResVAE":
        # the strided slice of the sampler state the reference decodes (train_poly_unet_cont.py:303)
        from oracle import decoders as OD
        sd = pkg["synth"].fill_state_dict(A.resvae_state_shapes(name, A.POLY_ENC, A.POLY_DEC, 32, 64, 3))
        state = pkg["synth"].noise(5, 0, (6, 5, 8, 8))
        zs = dev(state)[:, 3, :, :].view(6, 64)
        assert zs.stride(0) == 320
        want = OD.res_decoder(sd, state[:, 3].reshape(6, 64), A.POLY_ENC, A.POLY_DEC, 32, name)
        assert_close(m.decoder(zs).cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), "strided latents")
        with pytest.raises(RuntimeError):
            m.train().decoder(zs)


@pytest.mark.parametrize("name,cls,enc,dec,ch", [("ResVAEN_img", "ResVAEN", "CEL_IMG_ENC", "CEL_IMG_DEC", 3),
                                                 ("ResAEN_mask", "ResAEN", "CEL_MASK_ENC", "CEL_MASK_DEC", 1)])
def test_celeba_decoders(pkg, golden, name, cls, enc, dec, ch):
    g = golden("decoders.npz")
    A = pkg["arch"]
    m = getattr(pkg["vae"], cls)(getattr(A, enc), getattr(A, dec), 128, 256, ch).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    assert_close(m.decoder(dev(g[f"{name}_z"])).cpu().numpy(), g[f"{name}_y"], 1e-5, 1e-5, name)


@pytest.mark.parametrize("name", ["AttrBN", "AttrBNAE"])
def test_attr_decoders(pkg, golden, name):
    g = golden("decoders.npz")
    m = (pkg["attr"].CelebAAttrNewBN(256) if name == "AttrBN" else pkg["attr"].CelebAAttrNewBNAE(256)).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    want = g[f"{name}_y"]
    assert_close(m.decoder(dev(g[f"{name}_z"])).cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), name)


# ------------------------------------------------------------------------------------------------
SDE_TAGS = [("vp", 0.1, 5.0), ("vp", 1.0, 5.0), ("vp", 0.1, 20.0), ("subvp", 0.1, 20.0), ("ve", 0.01, 50.0)]


def _sde(pkg, kind, b0, b1, N):
    S = pkg["sde"]
    return {"vp": S.VPSDE, "subvp": S.subVPSDE, "ve": S.VESDE}[kind](b0, b1, N)


@pytest.mark.parametrize("kind,b0,b1", SDE_TAGS)
@pytest.mark.parametrize("N", [100, 1000])
def test_em_and_langevin_steps(pkg, golden, kind, b0, b1, N):
    g = golden("sde_steps.npz")
    S = pkg["sde"]
    tag = f"{kind}_{b0}_{b1}_{N}"
    sde = _sde(pkg, kind, b0, b1, N)
    x, score, noise = dev(g["x"]), dev(g["score"]), dev(g["noise"])
    for ti in (0, N // 2, N - 1):
        vt = torch.ones(4, device="cuda") * float(g[tag + "_ts"][ti])
        xn, xm = S.em_predictor(x, vt, lambda a, b: score, sde, noise=noise)
        assert_close(xm.cpu().numpy(), g[f"{tag}_em{ti}_mean"], 2e-6, 2e-6, "em mean")
        assert_close(xn.cpu().numpy(), g[f"{tag}_em{ti}_x"], 2e-6, 2e-6, "em x")
        if kind != "subvp":
            cn, cm = S.corrector(x, vt, lambda a, b: score, sde, 1, 0.16, noise=[noise])
            assert_close(cm.cpu().numpy(), g[f"{tag}_co{ti}_mean"], 3e-6, 3e-6, "corrector mean")
            assert_close(cn.cpu().numpy(), g[f"{tag}_co{ti}_x"], 3e-6, 3e-6, "corrector x")


def _traj_noise(pkg, n_draws, shape):
    return np.stack([pkg["synth"].noise(1234, k + 1, shape) for k in range(n_draws)])


@pytest.mark.parametrize("tag", ["vp_cp_pc0", "vp_cp_pc1", "vp_pc_pc0", "vp_pc_pc1", "ve_cp_pc1"])
@pytest.mark.parametrize("graph", [True, False])
def test_fused_sampler_trajectory(pkg, golden, tag, graph):
    """Config-1 shape (B=16, 2-mod, N=100): the whole loop in the HIP library with injected noise."""
    g = golden("trajectories.npz")
    S = pkg["sde"]
    kind, order, pcs = tag.split("_")
    pc = pcs == "pc1"
    m = _unet(pkg, "P2")
    sde = S.VPSDE(0.1, 5.0, 100) if kind == "vp" else S.VESDE(0.01, 5.0, 100)
    shape = (16, 2, 8, 8)
    x0 = pkg["synth"].noise(1234, 0, shape) * (np.float32(5.0) if kind == "ve" else np.float32(1.0))
    noise = dev(_traj_noise(pkg, 100 * (2 if pc else 1), shape))
    xm = S.uncond_sampler(shape, m, "cuda", sde, pc=pc, order=order, noise=noise, x0=torch.from_numpy(x0), use_graph=graph)
    want = g[f"{tag}_mean99"]
    assert_close(xm.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), tag)


def test_generic_score_fn_loop(pkg, golden):
    """uncond_sampler with an arbitrary score callable takes the Python loop over the HIP step kernels."""
    g = golden("trajectories.npz")
    S = pkg["sde"]
    m = _unet(pkg, "P2")
    shape = (16, 2, 8, 8)
    noise = dev(_traj_noise(pkg, 200, shape))
    xm = S.uncond_sampler(shape, lambda x, t: m(x, t), "cuda", S.VPSDE(0.1, 5.0, 100), pc=True, order="pc", noise=noise,
                          x0=torch.from_numpy(pkg["synth"].noise(1234, 0, shape)))
    want = g["vp_pc_pc1_mean99"]
    assert_close(xm.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), "python loop")


def test_reference_entry_point_replay(pkg, golden):
    """The unmodified reference uncond_sampler under torch.manual_seed(7): its prior and its 20 draws replayed."""
    g = golden("trajectories.npz")
    S = pkg["sde"]
    m = _unet(pkg, "P2")
    xm = S.uncond_sampler((4, 2, 8, 8), m, "cuda", S.VPSDE(0.1, 5.0, 20), noise=dev(g["ref_uncond_seed7_noise"]),
                          x0=torch.from_numpy(g["ref_uncond_seed7_prior"]))
    want = g["ref_uncond_seed7"]
    assert_close(xm.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), "uncond_sampler")
    # same host prior as the reference when nothing is injected
    torch.manual_seed(7)
    a = S.VPSDE(0.1, 5.0, 20).prior_sampling((4, 2, 8, 8)).numpy()
    np.testing.assert_array_equal(a, g["ref_uncond_seed7_prior"])


# ------------------------------------------------------------------------------------------------
def test_philox_normals(pkg):
    L = pkg["lib"]
    out = torch.empty(64, 4096, device="cuda")
    L.check(L.lib().sbm_fill_normal(out.data_ptr(), 4096, 64, 42, 0, 3, 0, 1.0, None))
    a = out.cpu().numpy().astype(np.float64)
    assert abs(a.mean()) < 0.01 and abs(a.var() - 1.0) < 0.02
    assert abs(np.mean(a ** 3)) < 0.05 and abs(np.mean(a ** 4) - 3.0) < 0.1
    # keyed by the GLOBAL sample index: rows [32, 64) equal a second call with sample_offset=32
    out2 = torch.empty(32, 4096, device="cuda")
    L.check(L.lib().sbm_fill_normal(out2.data_ptr(), 4096, 32, 42, 32, 3, 0, 1.0, None))
    assert torch.equal(out[32:], out2)
    # different step / draw / seed -> decorrelated
    for args in [(42, 0, 4, 0), (42, 0, 3, 1), (43, 0, 3, 0)]:
        o = torch.empty(64, 4096, device="cuda")
        L.check(L.lib().sbm_fill_normal(o.data_ptr(), 4096, 64, args[0], args[1], args[2], args[3], 1.0, None))
        assert abs(float((o * out).mean())) < 0.01


def test_sampler_determinism_and_sharding_invariance(pkg):
    """Bench-size properties (B=256, 5-mod): same seed -> same bits; splitting the batch over 'ranks'
    with sample_offset reproduces the unsharded rows exactly (EM-only has no cross-sample coupling)."""
    S = pkg["sde"]
    m = _unet(pkg, "P5")
    sde = S.VPSDE(0.1, 5.0, 1000)
    short = S.VPSDE(0.1, 5.0, 25)
    shape = (256, 5, 8, 8)
    x0 = torch.from_numpy(pkg["synth"].noise(9, 0, shape))
    a = S.uncond_sampler(shape, m, "cuda", short, x0=x0, seed=123)
    b = S.uncond_sampler(shape, m, "cuda", short, x0=x0, seed=123)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    c = S.uncond_sampler(shape, m, "cuda", short, x0=x0, seed=124)
    assert not torch.equal(a, c)
    parts = [S.uncond_sampler((128, 5, 8, 8), m, "cuda", short, x0=x0[r * 128:(r + 1) * 128], seed=123, sample_offset=r * 128)
             for r in range(2)]
    assert_close(torch.cat(parts).cpu().numpy(), a.cpu().numpy(), 1e-5, 1e-5 * float(a.abs().max()), "2-way shard")
    # the full 1000-step configuration runs and stays finite
    full = S.uncond_sampler(shape, m, "cuda", sde, x0=x0, seed=5)
    assert torch.isfinite(full).all()


def test_c_abi_error_paths(pkg):
    L = pkg["lib"]
    lib = L.lib()
    cfg = L.SdeCfg(0, 0.1, 5.0, 100, 0.16, 0)
    rc = lib.sbm_em_step(ctypes.byref(cfg), None, None, None, None, None, None, 4, 128, 0, 0, 0, 0, None)
    assert rc != 0 and b"bad arguments" in lib.sbm_last_error()
    with pytest.raises(RuntimeError):
        L.Engine(0, L.unet_cfg_ints(8, 2, (1, 2, 2, 2, 2, 2), 8, 8), 2,
                 pkg["synth"].fill_state_dict(pkg["arch"].unet_state_shapes(dim=8, channels=2, dim_mults=(1, 2, 2, 2, 2, 2))))


OA_CASES = {
    "oa_tiny": dict(in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(), channel_mult=(1, 2), z_dim=16, use_z=True),
    "oa_attn": dict(in_channels=2, model_channels=32, out_channels=2, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2), z_dim=16, use_z=True),
    "oa_C3c": dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(), channel_mult=(1, 2, 4, 8), z_dim=512, use_z=True),
}


@pytest.mark.parametrize("name", ["oa_tiny", "oa_attn", "oa_C3c"])
def test_unet_openai_matches_reference(pkg, golden, name):
    """unet_openai.UNetModel (config 5: conditional CelebA score net), with and without z_cond."""
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
    g = golden("unet_openai.npz")
    kw = OA_CASES[name]
    m = UNetModel(kw["in_channels"], kw["model_channels"], kw["out_channels"], kw["num_res_blocks"], kw["attention_resolutions"], 0.1,
                  kw["channel_mult"], num_heads=1, use_z=True, z_dim=kw["z_dim"]).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    y = m(dev(g[f"{name}_x"]), dev(g[f"{name}_t"]), z=dev(g[f"{name}_z"])).cpu().numpy()
    assert_close(y, g[f"{name}_y"], 1e-5, 1e-5 * np.abs(g[f"{name}_y"]).max(), name)
    y0 = m(dev(g[f"{name}_x"]), dev(g[f"{name}_t"])).cpu().numpy()
    assert_close(y0, g[f"{name}_y_noz"], 1e-5, 1e-5 * np.abs(g[f"{name}_y_noz"]).max(), name + " (z=None)")


# ------------------------------------------------------------------------------------------------
# chain kernel / time table: the fast paths against the plain per-layer path of the same library
def _sampler_run(pkg, m, x0, N, seed, **env):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        m.release_engines()
        sde = pkg["sde"].VPSDE(0.1, 5.0, N)
        return pkg["sde"].uncond_sampler(tuple(x0.shape), m, "cuda", sde, x0=x0, seed=seed)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        m.release_engines()


def test_time_table_is_bit_identical_to_per_step_time_path(pkg):
    """Tabulating the time MLP for all steps must not change a single bit of the trajectory."""
    m = _unet(pkg, "tiny")
    x0 = torch.randn(5, 2, 8, 8, device="cuda")
    a = _sampler_run(pkg, m, x0, 50, 3)
    b = _sampler_run(pkg, m, x0, 50, 3, SBM_NO_TIME_TABLE=1)
    assert torch.equal(a, b)


def test_update_kernels_that_advance_and_convert_leave_the_trajectory_bit_identical(pkg):
    """Round 4: the step's last update kernel advances the step counter (its last workgroup) and stores the new state in the score net's
    input layout as well, so a sampler step has no advance launch and no conversion launch.  Neither may change a bit: against the loop
    with the conversion op in every forward (SBM_NO_FUSED_CVT), predictor only and predictor-corrector in both orders, graph and eager."""
    import os
    m = _unet(pkg, "tiny")
    x0 = torch.randn(7, 2, 8, 8, device="cuda")
    sde = pkg["sde"].VPSDE(0.1, 5.0, 40)
    for kw in (dict(), dict(pc=True, order="cp"), dict(pc=True, order="pc"), dict(use_graph=False), dict(pc=True, order="pc", use_graph=False)):
        outs = []
        for env in ({}, {"SBM_NO_FUSED_CVT": "1"}):
            os.environ.update(env)
            try:
                m.release_engines()
                outs.append(pkg["sde"].uncond_sampler(tuple(x0.shape), m, "cuda", sde, x0=x0, seed=11, **kw).clone())
            finally:
                for k in env:
                    os.environ.pop(k, None)
        assert torch.equal(outs[0], outs[1]), kw
    m.release_engines()


def test_em_update_applied_by_the_score_nets_last_kernel_is_bit_identical(pkg):
    """Round 4: in a sampling loop the EM predictor is applied by the pixel-kernel run that ends the score net (the score never leaves LDS; same
    device functions as em_step_kernel, pixrun.hip kernel tail).  Against the loop with the update kernel of its own (SBM_NO_FUSED_EM): not a bit
    may differ -- predictor only, predictor-corrector in both orders, graph and eager, the conditional loop, x_mean included."""
    import os
    m = _unet(pkg, "P5")
    sde = pkg["sde"].VPSDE(0.1, 5.0, 40)
    for B in (7, 256):
        x0 = torch.randn(B, 5, 8, 8, device="cuda")
        for kw in (dict(), dict(pc=True, order="cp"), dict(pc=True, order="pc"), dict(use_graph=False), dict(probability_flow=True)):
            outs = []
            for env in ({}, {"SBM_NO_FUSED_EM": "1"}):
                os.environ.update(env)
                try:
                    m.release_engines()
                    assert m._engine_for(B, 8, 8).info["uses_pix"]
                    outs.append(pkg["sde"].uncond_sampler(tuple(x0.shape), m, "cuda", sde, x0=x0, seed=11, **kw).clone())
                finally:
                    for k in env:
                        os.environ.pop(k, None)
            assert torch.isfinite(outs[0]).all()
            assert torch.equal(outs[0], outs[1]), (B, kw)
    z_obs = torch.randn(7, 5, 64, device="cuda"); xc = torch.randn(7, 5, 8, 8, device="cuda")
    outs = []
    for env in ({}, {"SBM_NO_FUSED_EM": "1"}):
        os.environ.update(env)
        try:
            m.release_engines()
            outs.append(pkg["sde"].cond_sampler(z_obs.view(7, 5, 8, 8), [0, 2], m, sde, seed=3, noise_obs=True, x0=xc).clone())
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert torch.equal(outs[0], outs[1])
    m.release_engines()


def test_one_position_runs_match_the_mfma_path(pkg):
    """Round 4: the 1x1 level of the pixel kernel runs as vector-ALU mat-vecs streamed through the AGPR ring (PIX_VEC runs) instead of
    4-row MFMA passes (SBM_PIX_VEC=0): same results to float32 reassociation, forward and a 30-step sampler run, at B = 3 and at the bench's
    B = 256; and the op-level programs (every op alone) still run -- the stamps tool builds them."""
    import os
    m = _unet(pkg, "P5")
    sde = pkg["sde"].VPSDE(0.1, 5.0, 30)
    for B in (3, 256):
        x = torch.randn(B, 5, 8, 8, device="cuda"); t = torch.rand(B, device="cuda")
        res = {}
        for tag, env in (("vec", {}), ("mfma", {"SBM_PIX_VEC": "0"})):
            os.environ.update(env)
            try:
                m.release_engines()
                y = m(x, t).clone()
                assert m._engine_for(B, 8, 8).info["uses_pix"]
                res[tag] = (y, pkg["sde"].uncond_sampler((B, 5, 8, 8), m, "cuda", sde, x0=x, seed=5).clone())
            finally:
                for k in env:
                    os.environ.pop(k, None)
        for i in (0, 1):
            a, b = res["vec"][i], res["mfma"][i]
            assert float((a - b).abs().max()) <= (2e-5 if i == 0 else 1e-4) * float(b.abs().max()), (B, i)
    m.release_engines()


def test_first_last_conv_and_attention_block_inside_the_sample_owned_kernels(pkg):
    """Round 4: init_conv (im2col op + 1x1 conv) and the conv that writes the NCHW output run inside the pixel kernel's runs (SBM_PIX_IO=0: their
    own seg_gemm launches), the 8x8 attention block is ONE attnblock_kernel launch (SBM_ATTN_BLOCK=0: to_qkv + core as one kernel, to_out.0 and
    GroupNorm + residual in the pixel kernel).  Same results to float32 reassociation: forward and a 30-step sampler run, at B = 3 and B = 256,
    for the 5-modality net and for a 2-modality one (K = 98 of the im2col's 256 columns live)."""
    import os
    sde = pkg["sde"].VPSDE(0.1, 5.0, 30)
    for name, C in (("P5", 5), ("P2", 2)):
        m = _unet(pkg, name)
        for B in (3, 256):
            x = torch.randn(B, C, 8, 8, device="cuda"); t = torch.rand(B, device="cuda")
            res = {}
            for tag, env in (("new", {}), ("noio", {"SBM_PIX_IO": "0"}), ("noblock", {"SBM_ATTN_BLOCK": "0"}), ("neither", {"SBM_PIX_IO": "0", "SBM_ATTN_BLOCK": "0"})):
                os.environ.update(env)
                try:
                    m.release_engines()
                    y = m(x, t).clone()
                    info = m._engine_for(B, 8, 8).info
                    assert info["uses_pix"]
                    res[tag] = (y, pkg["sde"].uncond_sampler((B, C, 8, 8), m, "cuda", sde, x0=x, seed=5).clone())
                finally:
                    for k in env:
                        os.environ.pop(k, None)
            for tag in ("noio", "noblock", "neither"):
                for i in (0, 1):
                    a, b = res["new"][i], res[tag][i]
                    assert torch.isfinite(a).all()
                    assert float((a - b).abs().max()) <= (2e-5 if i == 0 else 1e-4) * float(b.abs().max()), (name, B, tag, i)
        m.release_engines()


def test_chain_kernel_matches_per_layer_kernels_at_bench_size(pkg):
    """Config-2 score net, B=256 (8 sample tiles = 8 XCD clusters): XCD-resident chain (the LEGACY path, SBM_PIX=0) vs one launch per
    layer."""
    import ctypes
    m = _unet(pkg, "P5")
    x = torch.randn(256, 5, 8, 8, device="cuda")
    t = torch.rand(256, device="cuda")
    import os
    outs = {}
    os.environ["SBM_PIX"] = "0"
    for pix in ("0", "16"):
        os.environ["SBM_CHAIN_PIX"] = pix
        try:
            m.release_engines()
            for _ in range(3):
                y = m(x, t)
            outs[pix] = y.clone()
            eng = m._engine_for(256, 8, 8)
            st = (ctypes.c_uint * 528)()                  # the whole ChainSync block: counters, arrival words, abort flag, barrier slots
            pkg["lib"].check(pkg["lib"].lib().sbm_debug_chain_sync(eng.h, st, 528))
            assert not any(st), "chain synchronisation state must be clean between launches (no abort, counters reset)"
        finally:
            os.environ.pop("SBM_CHAIN_PIX", None)
            m.release_engines()
    os.environ.pop("SBM_PIX", None)
    scale = float(outs["0"].abs().max())
    assert float((outs["0"] - outs["16"]).abs().max()) <= 2e-5 * scale
    assert torch.isfinite(outs["16"]).all()


def test_pixel_kernel_matches_per_layer_kernels_at_bench_size(pkg):
    """The DEFAULT path: the 4x4 / 2x2 / 1x1 levels as ONE launch of sample-owned workgroups (csrc/pixrun.hip: one sample per workgroup,
    activations in LDS, weights streamed in MFMA fragment order) against one launch per layer (SBM_CHAIN_PIX=0), config-2 score net at
    B = 256; a 20-step sampler run through both; and -- sample ownership -- a sample's low-resolution result does not depend on the batch
    it is in: the B = 64 engine reproduces the rows of the B = 256 one to the 8x8 levels' reassociation."""
    import ctypes
    import os
    from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE, uncond_sampler
    m = _unet(pkg, "P5")
    x = torch.randn(256, 5, 8, 8, device="cuda")
    t = torch.rand(256, device="cuda")
    sde = VPSDE(0.1, 5.0, 20)
    outs, runs = {}, {}
    for tag, env in (("layers", {"SBM_CHAIN_PIX": "0"}), ("pix", {})):
        os.environ.update(env)
        try:
            m.release_engines()
            for _ in range(2):
                y = m(x, t)
            outs[tag] = y.clone()
            eng = m._engine_for(256, 8, 8)
            if tag == "pix":
                assert eng.info["uses_pix"] and 5.0 < eng.info["pix_fragment_mb"] < 48.0, eng.info
                st = (ctypes.c_uint64 * 512)(); kinds = (ctypes.c_int * 512)(); nout = ctypes.c_int(0)
                L = pkg["lib"].lib()
                active = [i for i in range(eng.info["n_ops"]) if L.sbm_debug_pix_stamps(eng.h, i, 1, st, kinds, 512, ctypes.byref(nout)) == 0]
                assert len(active) >= 60, "the pixel kernel must be the one that ran the low-resolution ops (the one-position level included)"
                small = m(x[:64], t[:64]).clone()
                assert m._engine_for(64, 8, 8).info["uses_pix"]
                assert float((small - outs["pix"][:64]).abs().max()) <= 2e-5 * float(outs["pix"].abs().max())
            runs[tag] = uncond_sampler((256, 5, 8, 8), m, "cuda", sde, x0=x, seed=5).clone()
        finally:
            for k in env:
                os.environ.pop(k, None)
            m.release_engines()
    scale = float(outs["layers"].abs().max())
    assert float((outs["layers"] - outs["pix"]).abs().max()) <= 2e-5 * scale
    assert float((runs["layers"] - runs["pix"]).abs().max()) <= 1e-4 * float(runs["layers"].abs().max())


def test_full_size_sampling_run_is_path_and_shard_independent(pkg):
    """BASELINE config 2 at full size (B = 256, 1000 reverse-SDE steps, in-kernel Philox noise) at beta = (0.1, 20) -- the CelebA recipe's range
    (train_cel.sh:11), the stiffest the scripts use; PolyMNIST's own recipe is (1, 5) (train_poly.sh:17) and the bench runs SURVEY's (0.1, 5).  The
    samples do not depend on WHICH kernels carry the low-resolution levels (sample-owned pixel kernel down from 8x8 = the default, down
    from 4x4, or one launch per layer) beyond float32 reassociation -- measured 5e-7 of max|x| (tools/shard_check.py), bound 1e-5 -- nor
    on how the batch is split: on the default path every op from 8x8 down is sample-owned, so two shards of 128 keyed by their global
    sample offset reproduce the unsharded run BIT FOR BIT; with per-layer 8x8 kernels a 128-sample engine has another tile / split-K
    plan and agrees to reassociation (bound 1e-5)."""
    import os
    from score_based_multimodal_autoencoder_amd.sde_helper2 import VPSDE, uncond_sampler
    m = _unet(pkg, "P5")
    sde = VPSDE(0.1, 20.0, 1000)
    x0 = torch.empty(256, 5, 8, 8, device="cuda")
    pkg["lib"].check(pkg["lib"].lib().sbm_fill_normal(x0.data_ptr(), 5 * 64, 256, 1234, 0, 0, 255, 1.0, None))
    outs = {}
    for pix in ("0", "16", "64"):
        os.environ["SBM_CHAIN_PIX"] = pix
        try:
            m.release_engines()
            outs[pix] = uncond_sampler((256, 5, 8, 8), m, "cuda", sde, x0=x0, seed=7, sample_offset=0).clone()
            if pix != "0":
                again = uncond_sampler((256, 5, 8, 8), m, "cuda", sde, x0=x0, seed=7, sample_offset=0)
                assert torch.equal(again, outs[pix]), "the run must be deterministic"
                halves = torch.cat([uncond_sampler((128, 5, 8, 8), m, "cuda", sde, x0=x0[o:o + 128], seed=7, sample_offset=o) for o in (0, 128)])
                if pix == "64":
                    assert torch.equal(halves, outs[pix]), "default path: shards keyed by the global sample index == the unsharded run, bit for bit"
                else:
                    d = float((halves - outs[pix]).abs().max())
                    assert d <= 1e-5 * float(outs[pix].abs().max()), d
        finally:
            os.environ.pop("SBM_CHAIN_PIX", None)
            m.release_engines()
    assert all(torch.isfinite(v).all() for v in outs.values())
    scale = float(outs["0"].abs().max())
    for a, b in (("0", "16"), ("0", "64")):
        assert float((outs[a] - outs[b]).abs().max()) <= 1e-5 * scale, (a, b)


# ------------------------------------------------------------------------------------------------
# encoders (the conditional loop's source of observed latents)
@pytest.mark.parametrize("name", ["ResVAE", "ResAE"])
def test_poly_encoders(pkg, golden, name):
    g = golden("decoders.npz")
    arch = pkg["arch"]
    m = getattr(pkg["vae"], name)(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    e = m.encoder(dev(g[f"{name}_encx"]))
    mu = (e[0] if isinstance(e, tuple) else e).cpu().numpy()
    assert isinstance(e, tuple) == (name == "ResVAE")          # VAE: (mu, logvar); AE: mu  (h_vae_model_copy.py:114-116, 162-164)
    assert_close(mu, g[f"{name}_mu"], 1e-5, 1e-5 * np.abs(g[f"{name}_mu"]).max(), name + " mu")
    if name == "ResVAE":
        assert_close(e[1].cpu().numpy(), g[f"{name}_logvar"], 1e-5, 1e-5 * np.abs(g[f"{name}_logvar"]).max(), name + " logvar")
    with pytest.raises(ValueError):
        m.encoder(torch.zeros(2, 3, 16, 16, device="cuda"))


@pytest.mark.parametrize("name,cls,enc,dec,ch", [("ResVAEN_img", "ResVAEN", "CEL_IMG_ENC", "CEL_IMG_DEC", 3),
                                                 ("ResAEN_mask", "ResAEN", "CEL_MASK_ENC", "CEL_MASK_DEC", 1)])
def test_celeba_encoders(pkg, golden, name, cls, enc, dec, ch):
    g = golden("encoders.npz")
    arch = pkg["arch"]
    m = getattr(pkg["vae"], cls)(getattr(arch, enc), getattr(arch, dec), 128, 256, ch).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    e = m.encoder(dev(g[f"{name}_x"]))
    mu = (e[0] if isinstance(e, tuple) else e).cpu().numpy()
    assert_close(mu, g[f"{name}_mu"], 1e-5, 1e-5 * np.abs(g[f"{name}_mu"]).max(), name + " mu")
    if isinstance(e, tuple):
        assert_close(e[1].cpu().numpy(), g[f"{name}_logvar"], 1e-5, 1e-5 * np.abs(g[f"{name}_logvar"]).max(), name + " logvar")


@pytest.mark.parametrize("name", ["AttrBN", "AttrBNAE"])
def test_attr_encoders(pkg, golden, name):
    g = golden("encoders.npz")
    cls = pkg["attr"].CelebAAttrNewBN if name == "AttrBN" else pkg["attr"].CelebAAttrNewBNAE
    m = cls(256).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    e = m.encoder(dev(g[f"{name}_x"]))
    mu = (e[0] if isinstance(e, tuple) else e).cpu().numpy()
    assert isinstance(e, tuple) == (name == "AttrBN")
    assert_close(mu, g[f"{name}_mu"], 1e-5, 1e-5 * np.abs(g[f"{name}_mu"]).max(), name + " mu")
    if isinstance(e, tuple):
        assert_close(e[1].cpu().numpy(), g[f"{name}_logvar"], 1e-5, 1e-5 * np.abs(g[f"{name}_logvar"]).max(), name + " logvar")


@pytest.mark.parametrize("with_z,pc", [(True, False), (True, True), (False, False)])
def test_unet_openai_sampler_with_z_cond(pkg, with_z, pc):
    """Conditional CelebA path: z_cond reaches the score net as UNetModel(x, t, z=z_cond); the fused device
    loop and the step-by-step em_predictor/corrector calls give the same trajectory."""
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
    S = pkg["sde"]
    kw = OA_CASES["oa_tiny"]
    m = UNetModel(kw["in_channels"], kw["model_channels"], kw["out_channels"], kw["num_res_blocks"], kw["attention_resolutions"], 0.1,
                  kw["channel_mult"], num_heads=1, use_z=True, z_dim=kw["z_dim"]).eval()
    pkg["synth"].fill_module_(m)
    m = m.cuda()
    B, N = 4, 6
    sde = S.VPSDE(0.1, 5.0, N)
    x0 = torch.randn(B, 3, 8, 8, device="cuda")
    z = torch.randn(B, kw["z_dim"], device="cuda") if with_z else None
    dps = 2 if pc else 1
    noise = torch.randn(N * dps, B, 3, 8, 8, device="cuda")
    fused = S.uncond_sampler((B, 3, 8, 8), m, "cuda", sde, pc=pc, x0=x0, noise=noise, z_cond=z)
    # the same loop through the per-step API (order 'cp': corrector first, then predictor -- sde_helper2.py:115-128)
    x = x0.clone()
    ts = torch.linspace(sde.T, 1e-3, N, device="cuda")
    k = 0
    for i in range(N):
        vt = torch.ones(B, device="cuda") * ts[i]
        if pc:
            x, xm = S.corrector(x, vt, m, sde, 1, 0.16, z_cond=z, noise=[noise[k]]); k += 1
        x, xm = S.em_predictor(x, vt, m, sde, z_cond=z, noise=noise[k]); k += 1
    assert torch.isfinite(xm).all(), "the synthetic score net must keep this short chain finite"
    torch.testing.assert_close(fused, xm, rtol=1e-5, atol=1e-5)
    if with_z:
        other = S.uncond_sampler((B, 3, 8, 8), m, "cuda", sde, pc=pc, x0=x0, noise=noise, z_cond=torch.zeros_like(z))
        assert not torch.allclose(other, fused)


def test_cel_rep_conditioning_encoders(pkg, golden):
    """CelAttrRep / CelImgRep / CelMaskRep (encoder mu -> ProjectionHead) and get_conds."""
    g = golden("encoders.npz")
    H = pkg["attr"]
    models = {}
    for name, m in (("CelAttrRep", H.CelAttrRep(18, 512)), ("CelImgRep", H.CelImgRep(512)), ("CelMaskRep", H.CelMaskRep(512))):
        pkg["synth"].fill_module_(m.eval())
        m = m.cuda()
        models[name] = m
        y = m(dev(g[f"{name}_x"])).cpu().numpy()
        assert_close(y, g[f"{name}_y"], 1e-5, 1e-5 * np.abs(g[f"{name}_y"]).max(), name)
    conds = [models["CelImgRep"], models["CelMaskRep"], models["CelAttrRep"]]
    xs = [dev(g["CelImgRep_x"]), dev(g["CelMaskRep_x"]), dev(g["CelAttrRep_x"][:1])]
    z = H.get_conds(xs, conds, "02")
    want = (g["CelImgRep_y"] + g["CelAttrRep_y"][:1]) / 2
    assert_close(z.cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), "get_conds")
    assert H.get_conds(xs, conds, "") is None
