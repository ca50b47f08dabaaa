"""Drop-in for the DiffuseVAE refiner loop of the reference's evaluation script.

    evaluate_diff_vae(vae_out, sm_model, device)        eval_lat_celeba_hq_all.py:72-122
    linear_beta_schedule(timesteps)                      eval_lat_celeba_hq_all.py:62-65

1000 ancestral DDPM steps over the decoded image, each calling `sm_model(cat[sample, vae_out], t)` (in the reference a
`unet_openai.UNetModel(6 -> 3)`; here any callable on CUDA tensors, normally this package's `UNetModel`) and then ONE fused
update kernel (`sbm_ddpm_step`: mean, posterior noise from the in-kernel Philox stream keyed by the global sample index).
The schedule is built with the reference's own float32 expressions.  No CPU path.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._module import as_f32_cuda, fresh_seed


def linear_beta_schedule(timesteps):
    beta_start = 0.0001
    beta_end = 0.02
    return torch.linspace(beta_start, beta_end, timesteps)


@torch.no_grad()
def evaluate_diff_vae(vae_out, sm_model, device, timesteps=1000, *, noise=None, sample0=None, seed=None, sample_offset=0):
    """Keyword-only extras (defaults = reference behaviour with in-kernel noise): `noise(t)` -> injected N(0,1) tensor for step t
    (parity tests), `sample0` initial sample, `seed` / `sample_offset` Philox key (None: fresh per call from torch's default generator) and global index of row 0
    (sharded runs)."""
    seed = fresh_seed(seed)
    if hasattr(sm_model, "eval"):
        sm_model.eval()
    vae_out = as_f32_cuda(vae_out, "vae_out")
    B = vae_out.shape[0]
    chw = vae_out[0].numel()
    if chw % 4:
        raise ValueError("C*H*W must be a multiple of 4")
    # the reference's schedule (eval_lat_celeba_hq_all.py:80-93), same float32 torch expressions so the coefficients are bit-equal
    beta = linear_beta_schedule(timesteps)
    abar = torch.cumprod(1. - beta, axis=0)
    abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
    recip = torch.sqrt(1.0 / (1. - beta))                         # sqrt_recip_alphas
    s1m = torch.sqrt(1. - abar)                                   # sqrt_one_minus_alphas_cumprod
    sigma = torch.sqrt(beta * (1. - abar_prev) / (1. - abar))     # sqrt(posterior_variance)
    L = _lib.lib()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if sample0 is not None:
        sample = sample0.to(device=vae_out.device, dtype=torch.float32).contiguous().clone()
    else:
        sample = torch.empty_like(vae_out)                        # torch.normal(0, 1, vae_out.shape) (:96), from the Philox stream
        _lib.check(L.sbm_fill_normal(sample.data_ptr(), chw, B, seed, sample_offset, timesteps, 255, 1.0, st()))
    vae_out = (vae_out * 2) - 1
    nxt = torch.empty_like(sample)
    for t in range(timesteps - 1, -1, -1):
        t_batch = torch.full((B,), t, device=vae_out.device, dtype=torch.long)
        eps = sm_model(torch.cat([sample, vae_out], dim=1), t_batch)
        _lib._require_cuda(eps, "sm_model output")
        eps = eps.to(torch.float32).contiguous()
        z = None
        if t > 0 and noise is not None:
            z = noise(t).to(device=vae_out.device, dtype=torch.float32).contiguous()
        _lib.check(L.sbm_ddpm_step(sample.data_ptr(), eps.data_ptr(), z.data_ptr() if z is not None else None, nxt.data_ptr(),
                                   float(recip[t]), float(beta[t]), float(s1m[t]),
                                   float(sigma[t]) if t > 0 else 0.0, B, chw, seed, sample_offset, t, st()))
        sample, nxt = nxt, sample
    return (sample + 1) / 2
