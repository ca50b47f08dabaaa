"""TEST INFRASTRUCTURE -- CPU restatement of the DiffuseVAE refiner's DDPM loop; never imported by the product.

Follows eval_lat_celeba_hq_all.py: linear_beta_schedule (:62-65), evaluate_diff_vae (:72-122).  The schedule tensors are built
with the same float32 torch expressions as the reference; the loop itself is numpy float32 in the reference's operation order.
Pinned by tests/golden/ddpm_refiner.npz (the reference function's own output).
"""
import numpy as np
import torch


def schedule(timesteps=1000):
    """-> dict of float32 arrays [timesteps]: betas, sqrt_recip_alphas, sqrt_one_minus_alphas_cumprod, sqrt_posterior_variance."""
    beta = torch.linspace(0.0001, 0.02, timesteps)                                        # :62-65
    abar = torch.cumprod(1. - beta, axis=0)                                               # alphas_cumprod (:83-84)
    abar_prev = torch.nn.functional.pad(abar[:-1], (1, 0), value=1.0)                     # :85
    pv = beta * (1. - abar_prev) / (1. - abar)                                            # posterior_variance (:93)
    return dict(betas=beta.numpy(), sqrt_recip_alphas=torch.sqrt(1.0 / (1. - beta)).numpy(),          # :86
                sqrt_one_minus_alphas_cumprod=torch.sqrt(1. - abar).numpy(),                           # :90
                sqrt_posterior_variance=torch.sqrt(pv).numpy())


def evaluate_diff_vae(vae_out, eps_fn, draw, timesteps=1000):
    """vae_out (B, C, H, W) in [0, 1]; eps_fn(x (B, 2C, H, W) float32, t int) -> (B, C, H, W); draw() -> next N(0,1) array of the
    sample's shape (first call: the initial sample, then one per step with t > 0).  Returns (sample + 1) / 2."""
    s = schedule(timesteps)
    f = np.float32
    sample = draw().astype(f)                                                             # :96
    v = (vae_out.astype(f) * f(2)) - f(1)                                                 # :97
    for t in range(timesteps - 1, -1, -1):
        eps = eps_fn(np.concatenate([sample, v], axis=1), t).astype(f)
        mean = s["sqrt_recip_alphas"][t] * (sample - s["betas"][t] * eps / s["sqrt_one_minus_alphas_cumprod"][t])   # :107-109
        sample = mean + s["sqrt_posterior_variance"][t] * draw().astype(f) if t > 0 else mean                    # :111-116
    return (sample + f(1)) / f(2)
