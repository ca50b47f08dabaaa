"""Oracle restatement of the reference's alternative latent samplers and of the conditional SDE loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  fp32 arithmetic in the reference's operator order.
The state is the joint tensor (B, n_mod, size_z) with modalities in sorted-key order (the reference keeps a dict and
concatenates before every score call: fid_upd10.py:91, :321).

Parity pin: tests/golden/alt_samplers.npz and tests/golden/conditional.npz hold outputs of the reference's own functions
fid_upd10.gen_ae / fid_upd10.calc_poly_uncond / train_poly_unet_cont.evaluate run unmodified in the build container.
"""
import numpy as np

from . import sde as S

F32 = np.float32


def softplus(x):
    """nn.Softplus(beta=1, threshold=20)."""
    x = np.asarray(x, F32)
    return np.where(x > F32(20.0), x, np.log1p(np.exp(np.minimum(x, F32(20.0)).astype(np.float64)))).astype(F32)


def mlp_score(sd, x):
    """LSMPoly64_sm.forward lat_sm2_model.py:356-371: Linear at Sequential slots 0,2,..,12, Softplus between."""
    h = np.asarray(x, F32)
    for i in range(7):
        w, b = np.asarray(sd[f"layers.{2 * i}.weight"], F32), np.asarray(sd[f"layers.{2 * i}.bias"], F32)
        h = (h.astype(np.float64) @ w.T.astype(np.float64) + b).astype(F32)
        if i < 6:
            h = softplus(h)
    return h


def fixed_step(z, score, a, b, noise, predicted, score_div=1.0):
    """z[m] = z[m] + (a * (score[m] / score_div)) + b * noise[m] for m in predicted (fid_upd10.py:93-97, :335)."""
    z = np.asarray(z, F32)
    out = z.copy()
    for m in predicted:
        s = (np.asarray(score, F32).reshape(z.shape)[:, m] / F32(score_div)).astype(F32)
        out[:, m] = ((z[:, m] + (F32(a) * s).astype(F32)).astype(F32) + (F32(b) * np.asarray(noise, F32)[:, m]).astype(F32)).astype(F32)
    return out


def annealed_langevin(score_fn, z, predicted, noise_fn, sigmas=None, er=None, c=None, iter_num=None, hw=(8, 8), stop_after=None,
                      trace=None):
    """fid_upd10.gen_ae :56-99.  sigmas are float64 (np.linspace) in the reference; alpha = er*sigma_i^2/sigma_L^2 is a
    float64 0-dim tensor that enters the fp32 update as a scalar (rounded to fp32), cur_sigmas = sigmas.float().
    score_fn(z (B, n_mod, *hw), sigma_index (B,)) -> same shape.  noise_fn((B, size_z)) is called once per predicted
    modality per iteration, in sorted modality order (fid_upd10.py:95-97).  trace: dict filled with the state at the
    listed score-call numbers; stop_after: number of score calls to run."""
    z = np.asarray(z, F32).copy()
    B, n_mod, size_z = z.shape
    sig = np.linspace(5, 0.1, 200) if sigmas is None else np.asarray(sigmas, np.float64)
    dflt = (0.01, 0.7, 2) if len(predicted) == 10 else (2e-3, 0.5, 20)            # fid_upd10.py:64-71
    er = dflt[0] if er is None else er
    c = dflt[1] if c is None else c
    iter_num = dflt[2] if iter_num is None else iter_num
    calls = 0
    for i, s in enumerate(sig):
        idx = np.full((B,), i, dtype=F32)
        alpha = er * (s ** 2) / (sig[-1] ** 2)
        b = c * np.sqrt(2 * alpha)
        for _ in range(iter_num):
            if trace is not None and calls in trace:
                trace[calls] = z.reshape(B, n_mod, *hw).copy()
            if stop_after is not None and calls >= stop_after:
                return z
            score = np.asarray(score_fn(z.reshape(B, n_mod, *hw), idx), F32).reshape(B, n_mod, size_z)
            calls += 1
            noise = np.zeros_like(z)
            for m in predicted:
                noise[:, m] = noise_fn((B, size_z))
            z = fixed_step(z, score, F32(alpha), F32(b), noise, predicted, score_div=F32(s))
    return z


def mlp_langevin(score_fn, z, noise_fn, n_comp, lr1, lr2, schedule=False, predicted=None):
    """fid_upd10.calc_poly_uncond :319-339 (and the conditional twin :270-290 through `predicted`).  With `schedule` the
    last iteration applies, per modality, a second update of size 1*(i+1)/n_comp on the same score with a fresh draw."""
    z = np.asarray(z, F32).copy()
    B, n_mod, size_z = z.shape
    predicted = list(range(n_mod)) if predicted is None else list(predicted)
    for i in range(n_comp):
        score = np.asarray(score_fn(z.reshape(B, -1)), F32).reshape(B, n_mod, size_z)
        a = lr1 * ((i + 1) / n_comp) if schedule else lr1
        for m in predicted:
            nz = np.zeros_like(z)
            nz[:, m] = noise_fn((B, size_z))
            z = fixed_step(z, score, a, lr2, nz, [m])
            if schedule and i == n_comp - 1:
                nz[:, m] = noise_fn((B, size_z))
                z = fixed_step(z, score, 1 * ((i + 1) / n_comp), lr2, nz, [m])
    return z


def renoise_observed(cfg, x, z_obs, t, given):
    """train_poly_unet_cont.py:356-364: noised[mod] = mean + std * z[mod] with mean = exp(lmc) * z[mod] -- the reference
    scales the clean latent itself by std rather than drawing noise."""
    out = np.asarray(x, F32).copy()
    zo = np.asarray(z_obs, F32)
    mean, std = S.marginal_prob(cfg, zo, t)
    for m in given:
        out[:, m] = (mean[:, m] + (std[:, None, None] * zo[:, m]).astype(F32)).astype(F32)
    return out


def cond_sampler(cfg, score_fn, z_obs, given, x0, noise_fn, eps=1e-3, pc=False, n_steps=1, target_snr=0.16, noise_obs=False):
    """Conditional loop of train_poly_unet_cont.evaluate :340-381 ('pc' order: predictor then corrector).
    z_obs / x0: (B, n_mod, H, W); channels in `given` come from z_obs (re-noised each step when noise_obs), the others start
    at x0 and follow the reverse SDE.  Returns the final state: unobserved channels = last x_mean, observed = clean z_obs."""
    x = np.asarray(x0, F32).copy()
    zo = np.asarray(z_obs, F32)
    B = x.shape[0]
    for m in given:
        x[:, m] = zo[:, m]
    ts = S.linspace_f32(cfg.T, eps, cfg.N)
    x_mean = x
    pred = [m for m in range(x.shape[1]) if m not in given]
    for i in range(cfg.N):
        vec_t = (np.ones(B, dtype=F32) * ts[i]).astype(F32)
        if noise_obs:
            x_in = renoise_observed(cfg, x, zo, vec_t, given)
        else:
            x_in = x.copy()
            for m in given:
                x_in[:, m] = zo[:, m]
        nz = noise_fn(x_in.shape)
        x_upd, x_mean = S.em_step(cfg, x_in, vec_t, score_fn(x_in, vec_t), nz)
        if pc:
            for _ in range(n_steps):
                g = score_fn(x_upd, vec_t)
                x_upd, x_mean = S.langevin_step(cfg, x_upd, vec_t, g, noise_fn(x_upd.shape), target_snr)
        for m in pred:
            x[:, m] = x_upd[:, m]
    out = x.copy()
    for m in pred:
        out[:, m] = x_mean[:, m]
    for m in given:
        out[:, m] = zo[:, m]
    return out
