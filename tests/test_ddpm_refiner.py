"""DiffuseVAE refiner loop: oracle vs the reference's evaluate_diff_vae (CPU), the HIP step kernel and the drop-in
loop vs the oracle / the golden (GPU)."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import ddpm as OD
from score_based_multimodal_autoencoder_amd import synth

KW = dict(in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2))


def _draws(seed, shape):
    """The injected draws of the golden run, regenerated: draw k = synth.noise(seed, k, shape) (make_golden._Noise)."""
    k = [0]

    def draw():
        a = synth.noise(seed, k[0], shape)
        k[0] += 1
        return a
    return draw


def test_schedule_matches_closed_form():
    s = OD.schedule(1000)
    b = np.linspace(0.0001, 0.02, 1000)
    ab = np.cumprod(1 - b)
    assert_close(s["betas"], b, 1e-6, 0, "betas")
    assert_close(s["sqrt_one_minus_alphas_cumprod"], np.sqrt(1 - ab), 2e-3, 0, "sqrt(1 - alphabar)")      # float32: 1 - 0.9999 loses 3 digits at t = 0
    assert s["sqrt_posterior_variance"][0] == 0.0                 # no noise is added at t = 0


def test_oracle_loop_matches_reference_golden(golden):
    """The oracle loop around the oracle's own UNetModel restatement reproduces the reference function's output."""
    from oracle import unet_openai as OU
    from score_based_multimodal_autoencoder_amd import arch
    g = golden("ddpm_refiner.npz")
    sd = synth.fill_state_dict(dict(arch.unet_openai_entries(**KW)))
    seed, ndraw = int(g["seed"][0]), int(g["seed"][1])
    shape = g["vae_out"].shape
    seen = {}

    def eps_fn(x, t):
        k = 999 - t
        if f"xin{k}" in g:
            seen[k] = x
        return np.float32(0.1) * OU.unet_openai_forward(sd, x, np.full((shape[0],), t, np.float32), num_heads=2, **KW)

    y = OD.evaluate_diff_vae(g["vae_out"], eps_fn, _draws(seed, shape), 1000)
    assert ndraw == 1000
    for k, x in seen.items():                                     # states the model was called with, early / mid / last
        assert_close(x, g[f"xin{k}"], 2e-3, 2e-4, f"model input at call {k}")
    # 1000 steps amplify early differences by 1/sqrt(alphabar_T) = 157: float32 reassociation in the network shows at 1e-3
    assert_close(y, g["y"], 5e-3, 5e-3 * float(np.abs(g["y"]).max()), "refined sample")


@pytest.mark.gpu
def test_ddpm_step_kernel_matches_oracle_formula():
    import ctypes
    import torch
    from score_based_multimodal_autoencoder_amd import _lib
    s = OD.schedule(1000)
    B, chw = 3, 3 * 8 * 8
    x, eps, z = (synth.noise(5, i, (B, chw)) for i in range(3))
    xt, et, zt = (torch.from_numpy(a).cuda() for a in (x, eps, z))
    out = torch.empty_like(xt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for t in (999, 500, 1, 0):
        r, b, s1m, sg = (float(s[k][t]) for k in ("sqrt_recip_alphas", "betas", "sqrt_one_minus_alphas_cumprod", "sqrt_posterior_variance"))
        _lib.check(_lib.lib().sbm_ddpm_step(xt.data_ptr(), et.data_ptr(), zt.data_ptr(), out.data_ptr(), r, b, s1m, sg if t else 0.0, B, chw, 0, 0, t, st))
        f = np.float32
        want = f(r) * (x - f(b) * eps / f(s1m)) + (f(sg) * z if t else 0)
        assert_close(out.cpu().numpy(), want, 2e-6, 1e-6, f"step t={t}")
    # in-kernel Philox: deterministic, keyed by the global sample index (a shard reproduces its rows)
    a, b2 = torch.empty_like(xt), torch.empty_like(xt)
    _lib.check(_lib.lib().sbm_ddpm_step(xt.data_ptr(), et.data_ptr(), None, a.data_ptr(), 1.0, 0.0, 1.0, 1.0, B, chw, 9, 0, 7, st))
    _lib.check(_lib.lib().sbm_ddpm_step(xt[1:].data_ptr(), et[1:].data_ptr(), None, b2.data_ptr(), 1.0, 0.0, 1.0, 1.0, B - 1, chw, 9, 1, 7, st))
    torch.cuda.synchronize()
    assert torch.equal(a[1:], b2[:B - 1])
    zz = (a - xt).cpu().numpy()
    assert abs(zz.mean()) < 0.15 and abs(zz.std() - 1.0) < 0.15


@pytest.mark.gpu
def test_refiner_loop_matches_reference_golden(golden):
    """evaluate_diff_vae with this package's UNetModel (HIP kernels) as sm_model against the reference function's output."""
    import torch
    from score_based_multimodal_autoencoder_amd.diffuse_vae import evaluate_diff_vae
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
    g = golden("ddpm_refiner.npz")
    m = synth.fill_module_(UNetModel(KW["in_channels"], KW["model_channels"], KW["out_channels"], KW["num_res_blocks"], KW["attention_resolutions"], 0.1,
                                     KW["channel_mult"], num_heads=2).eval()).cuda()
    seed = int(g["seed"][0])
    shape = g["vae_out"].shape
    seen = {}
    calls = [0]

    def sm_model(x, t):
        if f"xin{calls[0]}" in g:
            seen[calls[0]] = x.cpu().numpy()
        calls[0] += 1
        return 0.1 * m(x, t)

    # draw 0 is the initial sample; the step at time t (t = 999 .. 1) uses draw 1000 - t
    noise = lambda t: torch.from_numpy(synth.noise(seed, 1000 - t, shape))
    y = evaluate_diff_vae(torch.from_numpy(g["vae_out"]).cuda(), sm_model, "cuda", noise=noise, sample0=torch.from_numpy(synth.noise(seed, 0, shape)))
    assert calls[0] == 1000
    for k, x in seen.items():
        assert_close(x, g[f"xin{k}"], 2e-3, 2e-4, f"model input at call {k}")
    assert_close(y.cpu().numpy(), g["y"], 5e-3, 5e-3 * float(np.abs(g["y"]).max()), "refined sample")
    # in-kernel noise: finite, deterministic
    y1 = evaluate_diff_vae(torch.from_numpy(g["vae_out"]).cuda(), sm_model, "cuda", timesteps=50, seed=3)
    y2 = evaluate_diff_vae(torch.from_numpy(g["vae_out"]).cuda(), sm_model, "cuda", timesteps=50, seed=3)
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)


@pytest.mark.gpu
def test_refiner_real_network_at_128x128_matches_oracle_and_loop_runs():
    """The refiner's REAL network (eval_lat_celeba_hq_all.py:448): UNetModel(img_ch*2 = 6, 128, 3, 2, attention_resolutions=(16,), 0.1,
    channel_mult=(1,2,2,3,4), num_heads=8) on 128x128 images -- 95 M parameters, 155 GFLOP per image and evaluation, attention with
    8 heads of 64 channels over the 8x8 level.  One evaluation of two images against the numpy oracle (the oracle takes ~5 s per
    image on the host), then a few steps of the DDPM loop around it: finite, deterministic, shards reproduce their rows."""
    import torch
    from oracle import unet_openai as OU
    from score_based_multimodal_autoencoder_amd.diffuse_vae import evaluate_diff_vae
    from score_based_multimodal_autoencoder_amd.unet_openai import UNetModel
    kw = dict(in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(16,), channel_mult=(1, 2, 2, 3, 4))
    m = synth.fill_module_(UNetModel(6, 128, 3, 2, (16,), 0.1, (1, 2, 2, 3, 4), num_heads=8).eval()).cuda()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    x = synth.noise(3, 0, (2, 6, 128, 128))
    t = np.array([999, 3], np.float32)
    want = OU.unet_openai_forward(sd, x, t, num_heads=8, **kw)
    got = m(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
    assert_close(got, want, 1e-4, REAL_NET_ATOL * float(np.abs(want).max()), "UNetModel(6->3) at 128x128")
    vae_out = torch.from_numpy(synth.noise(4, 0, (4, 3, 128, 128))).cuda()
    sm = lambda xx, tt: 0.1 * m(xx, tt)
    y = evaluate_diff_vae(vae_out, sm, "cuda", timesteps=3, seed=11)
    y2 = evaluate_diff_vae(vae_out, sm, "cuda", timesteps=3, seed=11)
    assert torch.isfinite(y).all() and torch.equal(y, y2)
    h = evaluate_diff_vae(vae_out[2:], sm, "cuda", timesteps=3, seed=11, sample_offset=2)
    assert torch.equal(h, y[2:]), "a shard keyed by its global sample offset reproduces its rows"
    m.release_engines()


REAL_NET_ATOL = 1e-5       # of max|y|, the per-forward tolerance of every other network (measured 3.6e-6: tools/refiner_check.py)
