"""Classifier / energy guidance on the GPU against fixtures from the reference's own guidance code
(tests/golden/make_golden.py guid): RSDE.sde / corrector of sde_helper2.py (:283-312, :65-94) and the PolyMNIST script's random-pair
variant (train_poly_unet_cont.py:73-89, 129-146), differentiating the guidance MLP with torch.autograd; here the same network's
forward + input-VJP run in one HIP kernel (csrc/guidance.hip) and the score is updated on the device.

Tolerance: the MLP forward / gradient 1e-5 of max|.|; guided step outputs rtol=1e-4, atol=1e-4*max|x| (stated: north_star's fp32
tolerance for sampler steps).
"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def pkg():
    from score_based_multimodal_autoencoder_amd import _lib, lat_sm2_model, sde_helper2, synth, train_poly_unet_cont
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return dict(lib=_lib, lm=lat_sm2_model, sde=sde_helper2, synth=synth, poly=train_poly_unet_cont)


def _filled(pkg, cls, salt, **kw):
    m = cls(**kw).eval()
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(np.asarray(pkg["synth"].fill(f"{salt}.{k}", v.shape))).to(v.dtype).reshape(v.shape) for k, v in sd.items()})
    return m.cuda()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag,variant,sz,nc,ids", [("cl2_s64", 2, 64, 1, None), ("cl3_s64", 3, 64, 1, (0, 2)), ("cl2_s256c3", 2, 256, 3, None)])
def test_guidance_network_forward_and_input_gradient(pkg, golden, tag, variant, sz, nc, ids):
    g = golden("guidance.npz")
    cls = pkg["lm"].ClwithTime2 if variant == 2 else pkg["lm"].ClwithTime3
    m = _filled(pkg, cls, tag, n_mod=2, size_z=sz, n_class=nc)
    x, t = dev(g[f"{tag}_x"]), dev(g[f"{tag}_t"])
    y = m(x, t) if ids is None else m(x, t, *ids)
    assert_close(y.cpu().numpy(), g[f"{tag}_y"], 1e-5, 1e-5 * np.abs(g[f"{tag}_y"]).max(), "forward")
    # the gradient of mean(): through guide_ on a two-channel state holding the pair
    B = x.shape[0]
    state = x.view(B, 2, sz).contiguous().view(B, 2, 1, sz)
    score = torch.zeros_like(state)
    grad = m.guide_(state, t, score, 0, 1, 1.0, ids=ids or (0, 0), return_grad=True)
    want = g[f"{tag}_g"]
    assert_close(grad.cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), "d mean / d x")
    assert_close((-score).view(B, -1).cpu().numpy(), want, 1e-5, 1e-5 * np.abs(want).max(), "score -= cl_s * grad")


@pytest.mark.parametrize("given", ["0", "02", "12"])
def test_guided_em_predictor_and_corrector_match_reference(pkg, golden, given):
    """sde_helper2.em_predictor / corrector(..., cl_g={'01','02','12'}, cl_s, given, all_mods='012')."""
    g = golden("guidance.npz")
    S = pkg["sde"]
    cl = {k: _filled(pkg, pkg["lm"].ClwithTime2, "clg" + k, n_mod=2, size_z=64, n_class=1) for k in ("01", "02", "12")}
    x, score, noise, t = dev(g["s_x"]), dev(g["s_score"]), dev(g["s_noise"]), dev(g["s_t"])
    sde = S.VPSDE(0.1, 20.0, 1000)
    cl_s = float(g["s_cl_s"])
    xn, xm = S.em_predictor(x, t, lambda a, b: score.clone(), sde, cl_g=cl, cl_s=cl_s, given=given, all_mods="012", noise=noise)
    for got, key in ((xn, "em_x"), (xm, "em_mean")):
        want = g[f"s_{given}_{key}"]
        assert_close(got.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), key)
    assert np.abs(g[f"s_{given}_em_mean"] - g["s_none_em_mean"]).max() > 1e-2          # the fixture is sensitive to the guidance
    cn, cm = S.corrector(x, t, lambda a, b: score.clone(), sde, 1, 0.16, cl_g=cl, cl_s=cl_s, given=given, all_mods="012", noise=[noise])
    for got, key in ((cn, "co_x"), (cm, "co_mean")):
        want = g[f"s_{given}_{key}"]
        assert_close(got.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), key)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_polymnist_random_pair_guidance_matches_reference(pkg, golden, k):
    """train_poly_unet_cont.em_predictor / corrector: one random (given, predicted) pair per call (torch.randint on the default
    generator, so torch.manual_seed selects the same pair as in the reference run), ids passed, predicted score only."""
    g = golden("guidance.npz")
    P = pkg["poly"]
    cl3 = _filled(pkg, pkg["lm"].ClwithTime3, "clgpoly", n_mod=2, size_z=64, n_class=1)
    x, score, noise = dev(g["p_x"]), dev(g["p_score"]), dev(g["p_noise"])
    t = torch.ones(4, device="cuda") * 0.37
    cl_s = float(g["s_cl_s"])
    seed = int(g[f"p_{k}_seed"])
    torch.manual_seed(seed)
    xn, xm = P.em_predictor(x, t, lambda a, b: score.clone(), 0.1, 5.0, 100, False, cl3, cl_s, None, given="02", all_mods="0123", noise=noise)
    torch.manual_seed(seed)
    cn, cm = P.corrector(x, t, lambda a, b: score.clone(), 0.1, 5.0, 100, 1, 1, 0.16, cl3, cl_s, None, given="02", all_mods="0123", noise=[noise])
    for got, key in ((xn, "em_x"), (xm, "em_mean"), (cn, "co_x"), (cm, "co_mean")):
        want = g[f"p_{k}_{key}"]
        assert_close(got.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), key)


def test_any_torch_callable_is_accepted_like_the_reference(pkg, golden):
    """cl_g may be any differentiable callable, as in the reference (autograd on the GPU); this package's networks take the HIP VJP.
    Both routes must agree."""
    g = golden("guidance.npz")
    S = pkg["sde"]
    nn = torch.nn
    hip = _filled(pkg, pkg["lm"].ClwithTime2, "clg01", n_mod=2, size_z=64, n_class=1)

    class Plain(nn.Module):                       # the same function written with torch ops on the module's own parameters
        def __init__(self, src):
            super().__init__()
            self.sd = {k: v.detach().cuda() for k, v in src.state_dict().items()}

        def forward(self, x, t):
            sd = self.sd
            half = 32
            freq = torch.exp(torch.arange(half, device=x.device) * -(np.log(10000.0) / (half - 1)))
            e = t[:, None] * freq[None, :]
            temb = nn.functional.softplus(nn.functional.linear(torch.cat([e.sin(), e.cos()], dim=-1), sd["time_mlp.0.weight"], sd["time_mlp.0.bias"]))
            h = torch.cat([x, temb], dim=1)
            for i in (0, 2):
                h = nn.functional.softplus(nn.functional.linear(h, sd[f"layers.{i}.weight"], sd[f"layers.{i}.bias"]))
            return nn.functional.linear(h, sd["layers.4.weight"], sd["layers.4.bias"])

    x, score, t = dev(g["s_x"]), dev(g["s_score"]), dev(g["s_t"])
    a = S.apply_guidance_(x, t, score.clone(), {"01": hip}, 2000.0, "0", "01")
    b = S.apply_guidance_(x, t, score.clone(), {"01": Plain(hip)}, 2000.0, "0", "01")
    assert float((a - score).abs().max()) > 1e-3
    assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-4, 1e-5 * float(a.abs().max()), "HIP VJP vs torch.autograd")


@pytest.mark.parametrize("pc", [False, True])
@pytest.mark.parametrize("given", [[0], [0, 2]])
def test_guided_conditional_loop_on_the_device_equals_the_host_driven_loop(pkg, given, pc):
    """the guided conditional loop as ONE captured step replayed N times (sbm_guided_cond_sampler_run: re-noise, score net,
    guidance kernels, predictor [, score net, guidance, corrector]) against the same loop driven from Python, call by call, over the same
    kernels -- injected noise, so the two see the same draws.  The guidance must matter (the unguided run differs)."""
    from score_based_multimodal_autoencoder_amd.unet_model import Unet
    S = pkg["sde"]
    unet = pkg["synth"].fill_module_(Unet(dim=8, channels=3, dim_mults=(1, 2, 2, 2)).eval()).cuda()
    cl = {k: _filled(pkg, pkg["lm"].ClwithTime2, "clg" + k, n_mod=2, size_z=64, n_class=1) for k in ("01", "02", "12")}
    N, B = 12, 4
    sde = S.VPSDE(0.1, 5.0, N)
    z_obs = torch.randn(B, 3, 8, 8, device="cuda")
    x0 = torch.randn(B, 3, 8, 8, device="cuda")
    noise = torch.randn(N * (2 if pc else 1), B, 3, 8, 8, device="cuda")
    kw = dict(pc=pc, n_steps=1, noise_obs=True, x0=x0, noise=noise, cl_g=cl, cl_s=500.0, all_mods="012")
    a = S.cond_sampler(z_obs, given, unet, sde, **kw)
    b = S.cond_sampler(z_obs, given, unet, sde, fused_guidance=False, **kw)
    plain = S.cond_sampler(z_obs, given, unet, sde, pc=pc, n_steps=1, noise_obs=True, x0=x0, noise=noise)
    assert torch.isfinite(a).all()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-5 * scale, float((a - b).abs().max())
    assert float((a - plain).abs().max()) > 1e-3 * scale, "the guidance must change the trajectory"
    for m in given:
        assert torch.equal(a[:, m], z_obs[:, m])                                   # observed modalities leave as their clean latents
    # the graph and the eager form of the fused loop agree bit for bit
    c = S.cond_sampler(z_obs, given, unet, sde, use_graph=False, **kw)
    assert torch.equal(a, c)
