"""The reference's two alternative latent samplers, on the joint state (B, n_mod, size_z).

* `annealed_langevin`  -- NCSN-style annealed Langevin dynamics over a sigma ladder with the Unet score taking the
  integer sigma index as its time input: fid_upd10.gen_ae :84-99 and eval_lat_celeba_hq_all.evaluate :258-275.
* `mlp_langevin`       -- fixed-step Langevin with the time-free MLP score: fid_upd10.calc_poly_uncond :311-339.

The reference keeps the state as a dict {modality: (B, size_z)} and concatenates it before every score call; here the
state is the concatenated tensor itself (modalities in sorted-key order) and `predicted` lists the modality indices
that move -- the others are the observed (conditioning) latents and pass through unchanged, as in the reference.
The update kernels are the HIP library's (`sbm_langevin_fixed_step`); there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._module import as_f32_cuda, fresh_seed


def _mask(n_mod, predicted):
    idx = range(n_mod) if predicted is None else predicted
    m = 0
    for i in idx:
        if not 0 <= int(i) < n_mod:
            raise ValueError(f"predicted modality {i} out of range")
        m |= 1 << int(i)
    return m


def langevin_fixed_step(z, score, a, b, predicted=None, noise=None, *, score_div=1.0, seed=0, sample_offset=0, step=0, draw=0, out=None):
    """z[:, m] += a*(score[:, m]/score_div) + b*N(0,1) for m in predicted (fid_upd10.py:97, :331)."""
    z = as_f32_cuda(z, "z")
    score = as_f32_cuda(score, "score")
    B, n_mod = z.shape[0], z.shape[1]
    per = z[0, 0].numel()
    if score.numel() != z.numel():
        raise ValueError("score and z differ in size")
    if noise is not None:
        noise = as_f32_cuda(noise, "noise")
        if noise.numel() != z.numel():
            raise ValueError("noise and z differ in size")
    out = torch.empty_like(z) if out is None else out
    _lib.check(_lib.lib().sbm_langevin_fixed_step(z.data_ptr(), score.data_ptr(), noise.data_ptr() if noise is not None else None,
                                                  out.data_ptr(), float(a), float(b), float(score_div), _mask(n_mod, predicted), B, n_mod, per,
                                                  int(seed), int(sample_offset), int(step), int(draw), _lib.stream_ptr()))
    return out


def default_sigmas():
    """fid_upd10.py:62: 200 noise levels from 5 down to 0.1 (float64, as np.linspace gives the reference)."""
    return np.linspace(5, 0.1, 200)


def _per_mod(v, n_mod, predicted):
    """scalar or per-modality sequence/dict -> list of (value, [modalities]) groups over the predicted modalities."""
    idx = list(range(n_mod)) if predicted is None else [int(i) for i in predicted]
    if isinstance(v, dict):
        vals = {i: float(v[i]) for i in idx}
    elif hasattr(v, "__len__"):
        vals = {i: float(v[i]) for i in idx}
    else:
        vals = {i: float(v) for i in idx}
    return vals


def _fused_score_net(score_model, hw):
    """True when `score_model` is this package's time-conditioned `Unet` called as score_model(z.view(B, n_mod, *hw), sigma_index) --
    the reference's own call (fid_upd10.py:91): the whole loop then runs on the device.  Any other callable takes the Python loop."""
    from .unet_model import Unet
    return isinstance(score_model, Unet) and score_model.with_time_emb and not score_model.training


def annealed_langevin(score_model, z, sigmas=None, er=None, c=None, iter_num=None, predicted=None, noise=None, *, seed=None,
                      sample_offset=0, hw=(8, 8), fused=None, return_timing=False):
    """Annealed Langevin over `sigmas` (fid_upd10.gen_ae :84-99; eval_lat_celeba_hq_all.evaluate :258-275).

    z (B, n_mod, size_z); score_model(z.view(B, n_mod, *hw), sigma_index) -> same shape.  Per level i and inner
    iteration: alpha = er*sigma_i^2/sigma_L^2 (float64, like the reference's np.linspace ladder); s = score/sigma_i;
    z[m] += alpha*s[m] + c*sqrt(2 alpha)*N for m in predicted.
    Defaults follow gen_ae: (er, c, iter_num) = (0.01, 0.7, 2) when 10 modalities are predicted, else (2e-3, 0.5, 20).
    er / c may be per-modality sequences or dicts (the CelebA evaluator's er[mod], c[mod]).
    noise: optional (len(sigmas)*iter_num, B, n_mod, size_z) injected normals (parity tests); default in-kernel Philox.
    fused: None (default) = the device-resident loop when `score_model` is this package's `Unet`, the Python loop over the same HIP
    kernels for any other callable; False forces the Python loop (the two agree bit for bit: tests/test_gpu_alt.py).
    """
    z = as_f32_cuda(z, "z").clone()
    B, n_mod = z.shape[0], z.shape[1]
    n_pred = n_mod if predicted is None else len(predicted)
    dflt = (0.01, 0.7, 2) if n_pred == 10 else (2e-3, 0.5, 20)
    ers = _per_mod(dflt[0] if er is None else er, n_mod, predicted)
    cs = _per_mod(dflt[1] if c is None else c, n_mod, predicted)
    iter_num = dflt[2] if iter_num is None else iter_num
    groups = {}
    for m in ers:
        groups.setdefault((ers[m], cs[m]), []).append(m)
    sig = default_sigmas() if sigmas is None else np.asarray(torch.as_tensor(sigmas).detach().cpu().numpy(), np.float64)
    seed = fresh_seed(seed)
    if fused is None:
        fused = _fused_score_net(score_model, hw)
    if fused:
        # device-resident loop (sbm_langevin_run): iteration counter on the device, time path tabulated once, one captured graph per
        # iteration replayed len(sigmas) * iter_num times -- the same kernels, coefficients and Philox stream as the loop below
        n_iter = sig.size * iter_num
        rows = np.zeros((n_iter, 2 * n_mod + 1), np.float32)
        times = np.repeat(np.arange(sig.size, dtype=np.float32), iter_num)
        mask = 0
        for (er_g, c_g), mods in groups.items():
            for i in range(sig.size):
                alpha = er_g * (sig[i] ** 2) / (sig[-1] ** 2)
                for m in mods:
                    rows[i * iter_num:(i + 1) * iter_num, m] = np.float32(alpha)
                    rows[i * iter_num:(i + 1) * iter_num, n_mod + m] = np.float32(c_g * np.sqrt(2 * alpha))
            for m in mods:
                mask |= 1 << m
        rows[:, 2 * n_mod] = np.repeat(sig.astype(np.float32), iter_num)
        eng = score_model._engine_for(B, hw[0], hw[1])
        coef = torch.from_numpy(rows).to(z.device); tdev = torch.from_numpy(times).to(z.device)
        if noise is not None:
            noise = as_f32_cuda(noise, "noise")
            if noise.shape[0] < n_iter:
                raise ValueError(f"noise needs {n_iter} draws")
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.lib().sbm_langevin_run(eng.h, tdev.data_ptr(), coef.data_ptr(), mask, n_mod, n_iter,
                                               noise.data_ptr() if noise is not None else None, z.data_ptr(), seed, int(sample_offset), 1, None,
                                               ctypes.byref(ms) if return_timing else None, _lib.stream_ptr()))
        return (z, ms.value) if return_timing else z
    single = len(groups) == 1
    buf = torch.empty_like(z) if single else None
    if return_timing:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    with torch.no_grad():
        k = 0
        for i in range(sig.size):
            sigma_index = torch.full((B,), i, device=z.device, dtype=torch.float32)       # the reference passes the integer index
            for _ in range(iter_num):
                score = score_model(z.view(B, n_mod, *hw), sigma_index)
                for (er_g, c_g), mods in groups.items():
                    alpha = er_g * (sig[i] ** 2) / (sig[-1] ** 2)
                    langevin_fixed_step(z, score, alpha, c_g * np.sqrt(2 * alpha), mods, None if noise is None else noise[k],
                                        score_div=np.float32(sig[i]), seed=seed, sample_offset=sample_offset, step=k, draw=0,
                                        out=buf if single else z)                        # element-wise: in place is safe
                if single:
                    z, buf = buf, z
                k += 1
    if return_timing:
        ev1.record(); ev1.synchronize()
        return z, ev0.elapsed_time(ev1)
    return z


def mlp_langevin(score_model, z, n_comp=50, lr1=0.03, lr2=0.001, schedule=False, predicted=None, noise=None, *, seed=None,
                 sample_offset=0, fused=None, return_timing=False):
    """Fixed-step Langevin with the MLP score (fid_upd10.calc_poly_uncond :311-339).

    z (B, n_mod, size_z); score_model(z.view(B, -1)) -> (B, n_mod*size_z).  With `schedule` the score step is
    lr1*(i+1)/n_comp and the last
    iteration applies a second update of size 1 with the same score (fid_upd10.py:336-339); noise then has n_comp+1 draws.
    fused: None (default) = the device-resident loop (sbm_langevin_run: iteration counter on the device, one captured graph per
    iteration) when `score_model` is this package's `LSMPoly64_sm`, the Python loop for any other callable; the two agree bit for bit."""
    from .lat_sm2_model import LSMPoly64_sm
    z = as_f32_cuda(z, "z").clone()
    B, n_mod = z.shape[0], z.shape[1]
    seed = fresh_seed(seed)
    if fused is None:
        fused = isinstance(score_model, LSMPoly64_sm) and not score_model.training
    if fused and n_comp > 0:
        rows = np.zeros((n_comp, 2 * n_mod + 1), np.float32)
        for i in range(n_comp):
            rows[i, :n_mod] = np.float32(lr1 * ((i + 1) / n_comp) if schedule else lr1)
        rows[:, n_mod:2 * n_mod] = np.float32(lr2)
        rows[:, 2 * n_mod] = 1.0
        tail = None
        if schedule:
            tl = np.zeros(2 * n_mod + 1, np.float32)
            tl[:n_mod] = np.float32(1 * (n_comp / n_comp)); tl[n_mod:2 * n_mod] = np.float32(lr2); tl[2 * n_mod] = 1.0
            tail = (ctypes.c_float * tl.size)(*tl.tolist())
        if noise is not None:
            noise = as_f32_cuda(noise, "noise")
            if noise.shape[0] < n_comp + (1 if schedule else 0):
                raise ValueError(f"noise needs {n_comp + (1 if schedule else 0)} draws")
        eng = score_model._engine_for(B)
        coef = torch.from_numpy(rows).to(z.device)
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.lib().sbm_langevin_run(eng.h, None, coef.data_ptr(), _mask(n_mod, predicted), n_mod, n_comp,
                                               noise.data_ptr() if noise is not None else None, z.data_ptr(), seed, int(sample_offset), 1, tail,
                                               ctypes.byref(ms) if return_timing else None, _lib.stream_ptr()))
        return (z, ms.value) if return_timing else z
    buf = torch.empty_like(z)
    if return_timing:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    with torch.no_grad():
        for i in range(n_comp):
            score = score_model(z.view(B, -1))
            a = lr1 * ((i + 1) / n_comp) if schedule else lr1
            langevin_fixed_step(z, score, a, lr2, predicted, None if noise is None else noise[i], seed=seed,
                                sample_offset=sample_offset, step=i, draw=0, out=buf)
            z, buf = buf, z
            if schedule and i == n_comp - 1:          # fid_upd10.py:338-339: one extra full-size step on the same (stale) score
                langevin_fixed_step(z, score, 1 * ((i + 1) / n_comp), lr2, predicted, None if noise is None else noise[n_comp],
                                    seed=seed, sample_offset=sample_offset, step=i, draw=1, out=buf)
                z, buf = buf, z
    if return_timing:
        ev1.record(); ev1.synchronize()
        return z, ev0.elapsed_time(ev1)
    return z
