"""The CelebA conditional loops on the GPU against fixtures produced by the reference's own script functions, called
unmodified (tests/golden/make_golden.py celcond):

  * train_lat_celebhq_unet_cont2.calc_perf (:286-316): per-step re-noising of the given modalities, re-assembly, em_predictor
    [+ corrector], write-back of the predicted slices, final z_mean -> `sde_helper2.cond_sampler`, whose loop runs fused on the
    device (`sbm_cond_sampler_run`: one captured step replayed N times) for this package's score nets, and from Python over the
    same HIP step kernels for any other score callable;
  * eval_lat_celeba_hq_all.evaluate (:258-275): 500-level annealed Langevin with per-modality er[mod], c[mod]
    -> `langevin.annealed_langevin`.

Tolerances: N-step trajectories with injected noise rtol=1e-4, atol=1e-4*max|x|; the 1000-call annealed chain 5e-4.
"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

B, SZ, N = 3, 64, 20


@pytest.fixture(scope="module")
def pkg():
    from score_based_multimodal_autoencoder_amd import _lib, langevin, sde_helper2, synth, unet_model
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return dict(lib=_lib, lang=langevin, sde=sde_helper2, synth=synth, unet=unet_model)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _unet(pkg, dim):
    m = pkg["unet"].Unet(dim=dim, channels=3, dim_mults=(1, 2, 2, 2)).eval()
    pkg["synth"].fill_module_(m)
    return m.cuda()


CALC_PERF = [("g0_n1_pc1", "0", True, True), ("g12_n1_pc0", "12", True, False), ("g02_n0_pc1", "02", False, True), ("gnone_pc0", "", False, False)]


@pytest.mark.parametrize("tag,given,noise_obs,pc", CALC_PERF)
@pytest.mark.parametrize("route", ["fused", "python"])
def test_calc_perf_loop_matches_reference(pkg, golden, tag, given, noise_obs, pc, route):
    g = golden("celeba_cond.npz")
    S, synth = pkg["sde"], pkg["synth"]
    seed, count = (int(v) for v in g[f"cp_{tag}_seed"])
    unet = _unet(pkg, 16)
    k = 0
    z_obs = np.zeros((B, 3, 8, 8), np.float32)
    x0 = np.zeros((B, 3, 8, 8), np.float32)
    for m in range(3):
        if str(m) in given:
            z_obs[:, m] = g[f"mu{m}"].reshape(B, 8, 8)                        # encoder(...)[0] of the observed modality (:276-279)
        else:
            x0[:, m] = synth.noise(seed, k, (B, SZ)).reshape(B, 8, 8)         # sde.prior_sampling((B, size_z)) (:284)
            k += 1
    dps = 2 if pc else 1
    noise = np.stack([synth.noise(seed, k + j, (B, 3, 8, 8)) for j in range(N * dps)])
    assert k + N * dps == count
    model = unet if route == "fused" else (lambda x, t: unet(x, t))
    out = S.cond_sampler(dev(z_obs), [int(c) for c in given], model, S.VPSDE(0.1, 5.0, N), pc=pc, noise_obs=noise_obs, x0=dev(x0), noise=dev(noise))
    for m in range(3):
        want = g[f"cp_{tag}_z{m}"]
        assert_close(out[:, m].reshape(B, SZ).cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), f"{tag} modality {m} ({route})")


def test_fused_conditional_loop_graph_equals_eager_and_is_deterministic(pkg, golden):
    g = golden("celeba_cond.npz")
    S = pkg["sde"]
    unet = _unet(pkg, 16)
    z_obs = torch.zeros(B, 3, 8, 8, device="cuda")
    z_obs[:, 0] = dev(g["mu0"]).view(B, 8, 8)
    x0 = torch.randn(B, 3, 8, 8, device="cuda")
    kw = dict(pc=True, noise_obs=True, x0=x0, seed=77)
    a = S.cond_sampler(z_obs, [0], unet, S.VPSDE(0.1, 5.0, 50), **kw)
    b = S.cond_sampler(z_obs, [0], unet, S.VPSDE(0.1, 5.0, 50), use_graph=False, **kw)
    c = S.cond_sampler(z_obs, [0], unet, S.VPSDE(0.1, 5.0, 50), **kw)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(a[:, 0], z_obs[:, 0]), "observed modalities leave as their clean latents"
    d = S.cond_sampler(z_obs, [0], unet, S.VPSDE(0.1, 5.0, 50), pc=True, noise_obs=True, x0=x0)      # default seed: a fresh key per call
    assert not torch.equal(a, d)


@pytest.mark.parametrize("tag,given", [("g0", "0"), ("g12", "12"), ("gnone", "")])
def test_celeba_annealed_evaluator_matches_reference(pkg, golden, tag, given):
    """eval_lat_celeba_hq_all.evaluate: sigmas = linspace(5, 0.1, 500), n_comp = 2, er[mod] / c[mod] per modality."""
    g = golden("celeba_cond.npz")
    synth = pkg["synth"]
    seed, count = (int(v) for v in g[f"ev_{tag}_seed"])
    unet = _unet(pkg, 8)
    pred = [m for m in range(3) if str(m) not in given]
    k = 0
    z0 = np.zeros((B, 3, SZ), np.float32)
    for m in range(3):
        if m in pred:
            z0[:, m] = synth.noise(seed, k, (B, SZ))                          # torch.normal(0, 1, (B, size_z)) (:252)
            k += 1
        else:
            z0[:, m] = g[f"mu{m}"]
    steps = 500 * 2
    noise = np.zeros((steps, B, 3, SZ), np.float32)
    for s in range(steps):
        for m in pred:
            noise[s, :, m] = synth.noise(seed, k, (B, SZ))                    # randn_like(z[mod]) per predicted modality, in order (:272)
            k += 1
    assert k == count
    seen, calls = {}, [0]
    keep = {int(key[len(f"ev_{tag}_zin"):]) for key in g if key.startswith(f"ev_{tag}_zin")}

    def score(z, idx):
        if calls[0] in keep:
            seen[calls[0]] = z.cpu().numpy().copy()
        calls[0] += 1
        return 0.02 * unet(z, idx) - 0.2 * z               # the generator's tamed score (make_golden._Tamed)

    er = {m: float(g["ev_er"][m]) for m in range(3)}
    cc = {m: float(g["ev_c"][m]) for m in range(3)}
    z = pkg["lang"].annealed_langevin(score, dev(z0), sigmas=np.linspace(5, 0.1, 500), er=er, c=cc, iter_num=2, predicted=pred, noise=dev(noise))
    assert calls[0] == steps
    for c, v in sorted(seen.items()):
        want = g[f"ev_{tag}_zin{c}"]
        assert_close(v, want, 5e-4, 5e-4 * np.abs(want).max(), f"{tag} state at score call {c}")
    for m in range(3):
        want = g[f"ev_{tag}_z{m}"]
        assert_close(z[:, m].cpu().numpy(), want, 5e-4, 5e-4 * np.abs(want).max(), f"{tag} final latent of modality {m}")
