"""Drop-in for the sampler section of the reference's PolyMNIST script (`train_poly_unet_cont.py:46-188`): the LOCAL copy of the
SDE sampler that takes beta_0 / beta_1 / N as scalars instead of an SDE object and runs predictor -> corrector.

    sde, marginal_prob                       train_poly_unet_cont.py:46-50, 167-172
    em_predictor, corrector                  train_poly_unet_cont.py:96-102, 104-165
    uncond_sampler                           train_poly_unet_cont.py:174-188

Same arithmetic as `sde_helper2` with `VPSDE(beta_0, beta_1, N)`; everything numeric runs in the HIP library (the whole loop
fused and graph-replayed when the score model is this package's `Unet`).  Only T = 1 is built (no caller passes another value).
"""
import torch

from . import sde_helper2 as _S


def _vp(beta_0, beta_1, N, T=1):
    if T != 1:
        raise NotImplementedError("T != 1: no caller of the reference uses it; the kernels assume T = 1")
    return _S.VPSDE(beta_0, beta_1, N)


def sde(x, t, beta_0, beta_1):
    """(drift, diffusion) of the forward VP SDE   [train_poly_unet_cont.py:46-50]"""
    return _vp(beta_0, beta_1, 1000).sde(x, t)


def marginal_prob(x, t, beta_0, beta_1):
    """(mean, std) of the perturbation kernel   [train_poly_unet_cont.py:167-172]"""
    return _vp(beta_0, beta_1, 1000).marginal_prob(x, t)


def em_predictor(x, t, score_fn, beta_0=0.1, beta_1=20, N=1000, probability_flow=False, cl_g=None, cl_s=None, target=None,
                 given=None, all_mods=None, **kw):
    """[train_poly_unet_cont.py:96-102]"""
    return _S.em_predictor(x, t, score_fn, _vp(beta_0, beta_1, N), probability_flow, cl_g, cl_s, target, given, all_mods,
                           poly_pairs=True, **kw)


def corrector(x, t, score_fn, beta_0, beta_1, N, T, n_steps, target_snr, cl_g=None, cl_s=None, target=None, given=None,
              all_mods=None, **kw):
    """[train_poly_unet_cont.py:104-165]"""
    return _S.corrector(x, t, score_fn, _vp(beta_0, beta_1, N, T), n_steps, target_snr, cl_g, cl_s, target, given, all_mods,
                        poly_pairs=True, **kw)


def uncond_sampler(sample_shape, model, device, T=1, eps=1e-3, beta_0=0.1, beta_1=20, N=1000, probability_flow=False, pc=False,
                   n_steps=1, target_snr=0.16, cl_g=None, cl_s=None, target=None, **kw):
    """[train_poly_unet_cont.py:174-188]: prior torch.randn(sample_shape), N x {predictor, [corrector]}, returns x_mean."""
    kw.setdefault("order", "pc")
    return _S.uncond_sampler(sample_shape, model, device, _vp(beta_0, beta_1, N, T), eps, probability_flow, pc, n_steps,
                             target_snr, cl_g, cl_s, target, **kw)
