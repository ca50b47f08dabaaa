"""GPU parity of the alternative samplers and the conditional loop (f1) against goldens made by the
reference's own fid_upd10.gen_ae / calc_poly_uncond / train_poly_unet_cont.evaluate, through the C ABI."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TINY = dict(dim=8, dim_mults=(1, 2, 2, 2))


@pytest.fixture(scope="module")
def pkg():
    from score_based_multimodal_autoencoder_amd import _lib, arch, h_vae_model_copy, langevin, lat_sm2_model, sde_helper2, synth, unet_model
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return dict(lib=_lib, arch=arch, synth=synth, unet=unet_model, sde=sde_helper2, vae=h_vae_model_copy, lang=langevin, mlp=lat_sm2_model)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _filled(pkg, m):
    return pkg["synth"].fill_module_(m.eval()).cuda()


@pytest.mark.parametrize("tag,n_mod,size_z", [("mlp_small", 2, 16), ("mlp_poly5", 5, 64)])
def test_mlp_score_forward(pkg, golden, tag, n_mod, size_z):
    g = golden("alt_samplers.npz")
    m = _filled(pkg, pkg["mlp"].LSMPoly64_sm(n_mod=n_mod, size_z=size_z))
    y = m(dev(g[f"{tag}_x"])).cpu().numpy()
    assert_close(y, g[f"{tag}_y"], 1e-5, 1e-5 * np.abs(g[f"{tag}_y"]).max(), tag)


def test_mlp_score_state_dict_surface(pkg):
    m = pkg["mlp"].LSMPoly64_sm(n_mod=2, size_z=16)
    assert list(m.state_dict().keys()) == [k for k, _ in pkg["arch"].mlp_entries(2, 16)]


def test_fixed_step_kernel_masks_and_philox(pkg):
    """Observed modalities pass through bit-exactly; the in-kernel Philox draw equals sbm_fill_normal's for the same key."""
    L, lang = pkg["lib"], pkg["lang"]
    B, n_mod, per = 7, 5, 64
    z = torch.randn(B, n_mod, per, device="cuda")
    s = torch.randn(B, n_mod, per, device="cuda")
    out = lang.langevin_fixed_step(z, s, 0.25, 0.5, predicted=[1, 3], seed=99, sample_offset=11, step=4, draw=0, score_div=2.0)
    nz = torch.empty(B, n_mod * per, device="cuda")
    L.check(L.lib().sbm_fill_normal(nz.data_ptr(), n_mod * per, B, 99, 11, 4, 0, 1.0, L.stream_ptr()))
    nz = nz.view(B, n_mod, per)
    want = z.clone()
    for m in (1, 3):
        want[:, m] = (z[:, m] + 0.25 * (s[:, m] / 2.0)) + 0.5 * nz[:, m]
    torch.testing.assert_close(out, want, rtol=0, atol=1e-6)
    for m in (0, 2, 4):
        assert torch.equal(out[:, m], z[:, m])
    with pytest.raises(RuntimeError):
        lang.langevin_fixed_step(z.cpu(), s, 0.1, 0.1)
    with pytest.raises(ValueError):
        lang.langevin_fixed_step(z, s, 0.1, 0.1, predicted=[7])


@pytest.mark.parametrize("sched", [0, 1])
def test_mlp_langevin_matches_calc_poly_uncond(pkg, golden, sched):
    g = golden("alt_samplers.npz")
    arch = pkg["arch"]
    sm = _filled(pkg, pkg["mlp"].LSMPoly64_sm(n_mod=2, size_z=64))
    dec = _filled(pkg, pkg["vae"].ResVAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    draws = g[f"polyunc_s{sched}_noise"]
    z0 = np.stack([draws[0], draws[1]], axis=1)
    rest = draws[2:]
    if sched:   # last iteration draws (mod0 step, mod0 extra, mod1 step, mod1 extra): regroup into two joint draws
        body, tail = rest[:-4], rest[-4:]
        joint = [np.stack([body[2 * i], body[2 * i + 1]], axis=1) for i in range(5)]
        joint += [np.stack([tail[0], tail[2]], axis=1), np.stack([tail[1], tail[3]], axis=1)]
    else:
        joint = [np.stack([rest[2 * i], rest[2 * i + 1]], axis=1) for i in range(6)]
    z = pkg["lang"].mlp_langevin(sm, dev(z0), n_comp=6, lr1=0.03, lr2=0.001, schedule=bool(sched), noise=dev(np.stack(joint)))
    for m in range(2):
        y = dec.decoder(z[:, m].contiguous()).cpu().numpy()
        want = g[f"polyunc_s{sched}_out{m}"]
        assert_close(y, want, 1e-4, 1e-4 * np.abs(want).max(), f"calc_poly_uncond mod {m}")


@pytest.mark.parametrize("tag,n_mod,pred", [("genae_cond", 2, (1,)), ("genae_all2", 2, (0, 1)), ("genae_all10", 10, tuple(range(10)))])
def test_annealed_langevin_matches_gen_ae(pkg, golden, tag, n_mod, pred):
    """The full 200-level chain of fid_upd10.gen_ae (4000 / 400 score calls), intermediate states and decoded outputs."""
    g = golden("alt_samplers.npz")
    arch, synth = pkg["arch"], pkg["synth"]
    unet = _filled(pkg, pkg["unet"].Unet(channels=n_mod, **TINY))
    ae = _filled(pkg, pkg["vae"].ResAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    seed, B = int(g[f"{tag}_seed"][0]), 2
    k = 0
    z0 = np.zeros((B, n_mod, 64), np.float32)
    for m in range(n_mod):
        if m in pred:
            z0[:, m] = synth.noise(seed, k, (B, 64))
            k += 1
        else:
            z0[:, m] = ae.encoder(dev(g[f"{tag}_p"][m])).cpu().numpy()          # fid_upd10.py:83: observed latents from the encoder
    iters = 2 if len(pred) == 10 else 20
    steps = 200 * iters
    noise = np.zeros((steps, B, n_mod, 64), np.float32)
    for s in range(steps):
        for m in pred:
            noise[s, :, m] = synth.noise(seed, k, (B, 64))
            k += 1
    assert k == int(g[f"{tag}_seed"][1])
    seen = {}
    calls = [0]
    keep = {int(key[len(tag) + 4:]) for key in g if key.startswith(tag + "_zin")}

    def score(z, idx):
        if calls[0] in keep:
            seen[calls[0]] = z.cpu().numpy().copy()
        calls[0] += 1
        return 0.02 * unet(z, idx) - 0.2 * z               # the generator's tamed score (tests/golden/make_golden.py)

    z = pkg["lang"].annealed_langevin(score, dev(z0), predicted=list(pred), noise=dev(noise))
    assert calls[0] == steps
    for c, v in sorted(seen.items()):
        want = g[f"{tag}_zin{c}"]
        assert_close(v, want, 5e-4, 5e-4 * np.abs(want).max(), f"{tag} state at score call {c}")
    outs = [ae.decoder(z[:, m].contiguous()) for m in pred]
    y = (outs[0] if len(pred) == 1 else torch.cat(outs, dim=2)).cpu().numpy()
    assert_close(y, g[f"{tag}_y"], 1e-3, 1e-3 * np.abs(g[f"{tag}_y"]).max(), f"{tag} decoded")


def test_annealed_langevin_per_modality_coefficients_and_philox(pkg):
    """er/c per modality (the CelebA evaluator's er[mod], c[mod]) = separate masked updates; Philox run is reproducible."""
    lang = pkg["lang"]
    unet = _filled(pkg, pkg["unet"].Unet(channels=3, **TINY))
    score = lambda z, i: 0.02 * unet(z, i) - 0.2 * z
    z0 = torch.randn(4, 3, 64, device="cuda")
    sig = np.linspace(5, 0.1, 5)
    a = lang.annealed_langevin(score, z0, sigmas=sig, er=[2e-3, 2e-3, 1e-3], c=[0.5, 0.5, 0.2], iter_num=2, predicted=[1, 2], seed=5)
    b = lang.annealed_langevin(score, z0, sigmas=sig, er={1: 2e-3, 2: 1e-3}, c={1: 0.5, 2: 0.2}, iter_num=2, predicted=[1, 2], seed=5)
    assert torch.equal(a, b) and torch.equal(a[:, 0], z0[:, 0]) and not torch.equal(a[:, 1], z0[:, 1])
    c = lang.annealed_langevin(score, z0, sigmas=sig, er={1: 2e-3, 2: 1e-3}, c={1: 0.5, 2: 0.2}, iter_num=2, predicted=[1, 2], seed=6)
    assert not torch.equal(a, c)


def test_annealed_langevin_device_loop_equals_python_loop_bit_for_bit(pkg):
    """with this package's Unet as the score model the whole chain runs on the device (sbm_langevin_run: iteration counter in
    device memory, tabulated time path, one captured graph per iteration) -- the same kernels, coefficients and Philox stream as the
    Python loop, so the two agree bit for bit, with in-kernel and with injected noise; the model serves ordinary forwards afterwards."""
    lang = pkg["lang"]
    unet = _filled(pkg, pkg["unet"].Unet(channels=3, **TINY)).eval()
    z0 = torch.randn(4, 3, 64, device="cuda")
    sig = np.linspace(5, 0.1, 6)
    kw = dict(sigmas=sig, er={1: 2e-3, 2: 1e-3}, c={1: 0.5, 2: 0.2}, iter_num=3, predicted=[1, 2], seed=11)
    t = torch.full((4,), 2.0, device="cuda")
    before = unet(z0.view(4, 3, 8, 8), t).clone()
    a = lang.annealed_langevin(unet, z0, fused=True, **kw)
    b = lang.annealed_langevin(unet, z0, fused=False, **kw)
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a[:, 0], z0[:, 0]) and not torch.equal(a[:, 1], z0[:, 1])
    assert torch.equal(lang.annealed_langevin(unet, z0, **kw), a)                     # default: the device loop for a bare Unet
    nz = torch.randn(18, 4, 3, 64, device="cuda")
    kw.pop("seed")
    a2 = lang.annealed_langevin(unet, z0, noise=nz, fused=True, **kw)
    b2 = lang.annealed_langevin(unet, z0, noise=nz, fused=False, **kw)
    assert torch.equal(a2, b2) and not torch.equal(a2, a)
    assert torch.equal(unet(z0.view(4, 3, 8, 8), t), before)                          # table mode is off again
    z1, ms = lang.annealed_langevin(unet, z0, seed=11, return_timing=True, **kw)
    assert torch.equal(z1, a) and ms > 0


@pytest.mark.parametrize("schedule", [False, True])
def test_mlp_langevin_device_loop_equals_python_loop_bit_for_bit(pkg, schedule):
    """The MLP-score chain (fid_upd10.calc_poly_uncond :311-339) as one device-resident call (sbm_langevin_run on an SBM_MLP handle, with
    the schedule's closing step on the last score) against the Python loop over the same kernels: bit for bit, Philox and injected noise."""
    lang = pkg["lang"]
    from score_based_multimodal_autoencoder_amd.lat_sm2_model import LSMPoly64_sm
    mlp = pkg["synth"].fill_module_(LSMPoly64_sm(n_mod=3, size_z=64).eval()).cuda()
    z0 = torch.randn(6, 3, 64, device="cuda")
    kw = dict(n_comp=9, lr1=0.03, lr2=0.001, schedule=schedule, predicted=[0, 2])
    a = lang.mlp_langevin(mlp, z0, seed=21, fused=True, **kw)
    b = lang.mlp_langevin(mlp, z0, seed=21, fused=False, **kw)
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a[:, 1], z0[:, 1]) and not torch.equal(a[:, 0], z0[:, 0])
    assert torch.equal(lang.mlp_langevin(mlp, z0, seed=21, **kw), a)                  # default: the device loop for this package's MLP
    nz = torch.randn(10, 6, 3, 64, device="cuda")
    a2 = lang.mlp_langevin(mlp, z0, noise=nz, fused=True, **kw)
    b2 = lang.mlp_langevin(mlp, z0, noise=nz, fused=False, **kw)
    assert torch.equal(a2, b2) and not torch.equal(a2, a)
    y = mlp(z0.view(6, -1))
    assert torch.isfinite(y).all()                                                      # the handle serves ordinary forwards afterwards


@pytest.mark.parametrize("tag,noise_obs,pc", [("nobs1_pc1", True, True), ("nobs0_pc0", False, False)])
def test_conditional_loop_matches_evaluate(pkg, golden, tag, noise_obs, pc):
    g = golden("conditional.npz")
    arch, synth, S = pkg["arch"], pkg["synth"], pkg["sde"]
    unet = _filled(pkg, pkg["unet"].Unet(channels=2, **TINY))
    ae = _filled(pkg, pkg["vae"].ResAE(arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    sde = S.VPSDE(0.1, 5.0, 20)

    def image(state):
        tiles = [ae.decoder(state[:, m].reshape(1, 64).contiguous()).clamp(0.0, 1.0)[0].permute(1, 2, 0) for m in range(2)]
        return torch.cat(tiles, dim=1).cpu().numpy()

    xm = S.uncond_sampler((1, 2, 8, 8), unet, "cuda", sde, pc=pc, order="pc", x0=dev(g[f"{tag}_unc_prior"]), noise=dev(g[f"{tag}_unc_noise"]))
    assert_close(image(xm), g[f"{tag}_unc_img"], 5e-4, 5e-4, "unconditional image")
    z_obs = torch.zeros(1, 2, 8, 8, device="cuda")
    z_obs[0, 0] = ae.encoder(dev(g[f"{tag}_imgs"][0]))[0].reshape(8, 8)          # given = first modality, first sample
    x0 = np.zeros((1, 2, 8, 8), np.float32)
    x0[0, 1] = g[f"{tag}_cond_z0"].reshape(8, 8)
    out = S.cond_sampler(z_obs, [0], unet, sde, pc=pc, noise_obs=noise_obs, x0=dev(x0), noise=dev(g[f"{tag}_cond_noise"]))
    assert torch.equal(out[:, 0], z_obs[:, 0])
    assert_close(image(out), g[f"{tag}_cond_img"], 5e-4, 5e-4, "conditional image")


def test_renoise_observed_matches_marginal_prob(pkg):
    S = pkg["sde"]
    x = torch.randn(6, 3, 8, 8, device="cuda")
    zo = torch.randn(6, 3, 8, 8, device="cuda")
    t = torch.rand(6, device="cuda") * 0.9 + 0.05
    for sde in (S.VPSDE(0.1, 20.0, 1000), S.subVPSDE(0.1, 20.0, 1000), S.VESDE(0.01, 50.0, 1000)):
        out = S.renoise_observed(x, zo, t, sde, [0, 2])
        mean, std = sde.marginal_prob(zo, t)
        want = mean + std[:, None, None, None] * zo
        torch.testing.assert_close(out[:, 0], want[:, 0], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(out[:, 2], want[:, 2], rtol=2e-5, atol=2e-6)
        assert torch.equal(out[:, 1], x[:, 1])
