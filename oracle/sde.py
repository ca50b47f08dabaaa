"""Oracle restatement of the SDE sampler math (reference sde_helper2.py and the PolyMNIST
local copy train_poly_unet_cont.py:46-188).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  fp32 arithmetic in the reference's
operator order: Python-double scalars are rounded to fp32 before they meet an fp32 tensor,
exactly as PyTorch's wrapped-number promotion does.
"""
import math
import numpy as np

F32 = np.float32


def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in fp32 (ATen RangeFactories: symmetric halves)."""
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    lo = (start + step * i.astype(F32)).astype(F32)
    hi = (end - step * (steps - 1 - i).astype(F32)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


class SDECfg:
    """kind in {'vp','subvp','ve'}; (b0,b1) = (beta_min,beta_max) or (sigma_min,sigma_max)."""

    def __init__(self, kind="vp", b0=0.1, b1=20.0, N=1000, T=1.0):
        self.kind, self.b0, self.b1, self.N, self.T = kind, float(b0), float(b1), int(N), float(T)

    # sde_helper2.py:342-343
    def alphas(self):
        return (F32(1.0) - linspace_f32(self.b0 / self.N, self.b1 / self.N, self.N)).astype(F32)


def sde_coeffs(cfg, t):
    """Per-sample (drift_coef, diffusion): drift = drift_coef * x.

    VPSDE.sde sde_helper2.py:352-356; subVPSDE.sde :402-407; VESDE.sde :445-450."""
    t = np.asarray(t, dtype=F32)
    if cfg.kind in ("vp", "subvp"):
        beta_t = (F32(cfg.b0) + t * F32(cfg.b1 - cfg.b0)).astype(F32)
        dcoef = (F32(-0.5) * beta_t).astype(F32)
        if cfg.kind == "vp":
            g = np.sqrt(beta_t).astype(F32)
        else:
            e = np.exp((F32(-2 * cfg.b0) * t - F32(cfg.b1 - cfg.b0) * (t * t).astype(F32)).astype(F32)).astype(F32)
            g = np.sqrt((beta_t * (F32(1.0) - e)).astype(F32)).astype(F32)
        return dcoef, g
    # VE: sigma_min * (sigma_max/sigma_min)**t * sqrt(2 log(sigma_max/sigma_min))
    sigma = (F32(cfg.b0) * np.power(F32(cfg.b1 / cfg.b0), t).astype(F32)).astype(F32)
    g = (sigma * F32(math.sqrt(2 * (math.log(cfg.b1) - math.log(cfg.b0))))).astype(F32)
    return np.zeros_like(t), g


def marginal_prob(cfg, x, t):
    """sde_helper2.py:358-362 (VP), :409-413 (subVP: std = 1 - exp(2 lmc), no sqrt), :452-455 (VE)."""
    t = np.asarray(t, dtype=F32)
    x = np.asarray(x, dtype=F32)
    if cfg.kind == "ve":
        return x, (F32(cfg.b0) * np.power(F32(cfg.b1 / cfg.b0), t).astype(F32)).astype(F32)
    lmc = (F32(-0.25) * (t * t).astype(F32) * F32(cfg.b1 - cfg.b0) - F32(0.5) * t * F32(cfg.b0)).astype(F32)
    mean = (np.exp(lmc).astype(F32)[:, None, None, None] * x).astype(F32)
    e2 = np.exp((F32(2.0) * lmc).astype(F32)).astype(F32)
    std = np.sqrt(F32(1.0) - e2).astype(F32) if cfg.kind == "vp" else (F32(1.0) - e2).astype(F32)
    return mean, std


def em_step(cfg, x, t, score, z, probability_flow=False):
    """em_predictor + RSDE.sde (sde_helper2.py:45-52, 277-317; Poly :96-102, 52-94).

    Returns (x_new, x_mean).  `score` = score_fn(x, t), `z` = the randn_like draw."""
    x = np.asarray(x, F32); score = np.asarray(score, F32); z = np.asarray(z, F32)
    dcoef, g = sde_coeffs(cfg, t)
    b = lambda v: v[:, None, None, None]
    drift = (b(dcoef) * x).astype(F32) if cfg.kind != "ve" else np.zeros_like(x)
    g2s = ((b(g) ** 2).astype(F32) * score).astype(F32) * F32(0.5 if probability_flow else 1.0)
    drift = (drift - g2s).astype(F32)
    dt = -1.0 / cfg.N
    x_mean = (x + drift * F32(dt)).astype(F32)
    if probability_flow:
        x_new = (x_mean + (F32(0.0) * F32(math.sqrt(-dt))) * z).astype(F32)
    else:
        x_new = (x_mean + (b(g) * F32(math.sqrt(-dt))).astype(F32) * z).astype(F32)
    return x_new, x_mean


def corrector_alpha(cfg, t):
    """sde_helper2.py:56-60 / Poly :105-108."""
    t = np.asarray(t, F32)
    if cfg.kind == "ve":
        return np.ones_like(t)
    idx = ((t * F32(cfg.N - 1)).astype(F32) / F32(cfg.T)).astype(np.int64)
    return cfg.alphas()[idx]


def batch_mean_norm(v):
    """torch.norm(v.reshape(B,-1), dim=-1).mean()  (sde_helper2.py:97-98)."""
    v = np.asarray(v, F32)
    n = np.sqrt((v.reshape(v.shape[0], -1).astype(np.float64) ** 2).sum(axis=1)).astype(F32)
    return F32(n.astype(np.float64).mean())


def langevin_step(cfg, x, t, grad, noise, target_snr):
    """One inner iteration of `corrector` (sde_helper2.py:96-101)."""
    x = np.asarray(x, F32); grad = np.asarray(grad, F32); noise = np.asarray(noise, F32)
    alpha = corrector_alpha(cfg, t)
    gn, nn_ = batch_mean_norm(grad), batch_mean_norm(noise)
    r = F32(F32(target_snr) * nn_ / gn)
    step = ((r * r) * F32(2.0) * alpha).astype(F32)
    b = lambda v: v[:, None, None, None]
    x_mean = (x + b(step) * grad).astype(F32)
    x_new = (x_mean + b(np.sqrt((step * F32(2.0)).astype(F32)).astype(F32)) * noise).astype(F32)
    return x_new, x_mean


def sampler(cfg, score_fn, x0, noise_fn, eps=1e-3, probability_flow=False, pc=False, n_steps=1,
            target_snr=0.16, order="cp", checkpoints=None, stop_after=None):
    """uncond_sampler (sde_helper2.py:115-128, order 'cp' = corrector then predictor;
    train_poly_unet_cont.py:174-188, order 'pc').  Noise is injected: noise_fn(shape) is called in
    the reference's draw order (predictor draws BEFORE its score call, corrector AFTER).
    Returns the last executed sub-step's x_mean."""
    x = np.asarray(x0, F32)
    B = x.shape[0]
    ts = linspace_f32(cfg.T, eps, cfg.N)
    x_mean = x
    trace = {}

    def predictor(x):
        z = noise_fn(x.shape)
        s = score_fn(x, vec_t)
        return em_step(cfg, x, vec_t, s, z, probability_flow)

    def correct(x, x_mean):
        for _ in range(n_steps):
            g = score_fn(x, vec_t)
            n = noise_fn(x.shape)
            x, x_mean = langevin_step(cfg, x, vec_t, g, n, target_snr)
        return x, x_mean

    for i in range(cfg.N):
        vec_t = (np.ones(B, dtype=F32) * ts[i]).astype(F32)
        if order == "cp":
            if pc:
                x, x_mean = correct(x, x_mean)
            x, x_mean = predictor(x)
        else:
            x, x_mean = predictor(x)
            if pc:
                x, x_mean = correct(x, x_mean)
        if checkpoints is not None and i in checkpoints:
            trace[i] = (x.copy(), x_mean.copy())
        if stop_after is not None and i >= stop_after:
            break
    return (x_mean, trace) if checkpoints is not None else x_mean
