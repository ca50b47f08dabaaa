"""Drop-in for the sampling half of the reference's `sde_helper2.py`:

    VPSDE / subVPSDE / VESDE          (sde_helper2.py:329-473)
    em_predictor, corrector           (sde_helper2.py:45-106)
    uncond_sampler                    (sde_helper2.py:115-128)

The z-updates (Euler-Maruyama predictor, Langevin corrector, their batch-mean norms) and -- when the
score model is this package's `Unet` -- the whole N-step loop run in the HIP library; the loop is
captured once into a hipGraph and replayed.  Classifier / energy guidance (`cl_g`, `cl_s`, `given`, `all_mods`;
SURVEY.md 8(f4)) is applied by `apply_guidance_` (HIP input-VJP kernel for this package's ClwithTime2/3).
Training-only helpers (loss_fn, JAX importance sampling) are out of scope.

Additions, all keyword-only and defaulted so reference call sites keep working:
    order  'cp' (corrector then predictor, sde_helper2.uncond_sampler) or
           'pc' (predictor then corrector, train_poly_unet_cont.uncond_sampler)
    noise  injected draws (parity tests): tensor [(N * draws_per_step), B, C, H, W] in execution order
    x0     initial state instead of sde.prior_sampling(shape)
    seed / sample_offset   Philox key (None: a fresh key per call from torch's default generator) and global index of
           row 0 (pc=False results do not depend on sharding; pc=True see `exact_sharded_norms`)
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._module import as_f32_cuda, fresh_seed


class SDE:
    """Common surface of the reference's SDE ABC (sde_helper2.py:191-326)."""
    kind = -1

    def __init__(self, N):
        self.N = N

    @property
    def T(self):
        return 1

    def _cfg(self, target_snr=0.16, probability_flow=False):
        return _lib.SdeCfg(self.kind, float(self.beta_0), float(self.beta_1), int(self.N), float(target_snr),
                           int(bool(probability_flow)))

    def prior_sampling(self, shape):
        return torch.randn(*shape)                       # host RNG, as the reference (sde_helper2.py:364)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.

    def discretize(self, x, t):
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * math.sqrt(dt)

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE/ODE object with the reference RSDE's `.sde` / `.discretize` (sde_helper2.py:255-326)."""
        return _Reverse(self, score_fn, probability_flow)

    # light-weight tensor helpers (API surface; not on the fused hot loop)
    def _beta_t(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)


class _Reverse:
    def __init__(self, fwd, score_fn, probability_flow):
        self.fwd, self.score_fn, self.probability_flow, self.N = fwd, score_fn, probability_flow, fwd.N

    @property
    def T(self):
        return self.fwd.T

    def sde(self, x, t, cl_g=None, cl_s=None, target=None, given=None, all_mods=None):
        drift, diffusion = self.fwd.sde(x, t)
        score = self.score_fn(x, t)
        if _guided(cl_g, given):
            score = apply_guidance_(as_f32_cuda(x, "x"), as_f32_cuda(t, "t"), as_f32_cuda(score, "score"), cl_g, cl_s, given, all_mods)
        drift = drift - diffusion[:, None, None, None] ** 2 * score * (0.5 if self.probability_flow else 1.)
        return drift, (torch.zeros_like(diffusion) if self.probability_flow else diffusion)

    def discretize(self, x, t):
        f, G = self.fwd.discretize(x, t)
        rev_f = f - G[:, None, None, None] ** 2 * self.score_fn(x, t) * (0.5 if self.probability_flow else 1.)
        return rev_f, (torch.zeros_like(G) if self.probability_flow else G)


class VPSDE(SDE):
    kind = 0

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

    def sde(self, x, t):
        beta_t = self._beta_t(t)
        return -0.5 * beta_t[:, None, None, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc[:, None, None, None]) * x, torch.sqrt(1. - torch.exp(2. * lmc))

    def discretize(self, x, t):
        """DDPM discretisation (sde_helper2.py:372-381)."""
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[timestep]
        alpha = self.alphas.to(x.device)[timestep]
        return torch.sqrt(alpha)[:, None, None, None] * x - x, torch.sqrt(beta)


class subVPSDE(SDE):
    kind = 1

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max
        # The reference's subVPSDE has no `.alphas`, so its corrector raises AttributeError; here the
        # corrector uses the same table as VPSDE (as in score_sde_pytorch, which the reference adapts).
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas

    def sde(self, x, t):
        beta_t = self._beta_t(t)
        discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return -0.5 * beta_t[:, None, None, None] * x, torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc)[:, None, None, None] * x, 1 - torch.exp(2. * lmc)


class VESDE(SDE):
    kind = 2

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.beta_0, self.beta_1 = sigma_min, sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    def sde(self, x, t):
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return torch.zeros_like(x), sigma * math.sqrt(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)))

    def marginal_prob(self, x, t):
        return x, self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2. * np.log(2 * np.pi * self.sigma_max ** 2) - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2)

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas.to(t.device)[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), self.discrete_sigmas[timestep - 1].to(t.device))
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)


# --------------------------------------------------------------------------------------------------
def apply_guidance_(x, t, score, cl_g, cl_s, given, all_mods, poly_pairs=False):
    """Classifier / energy guidance, in place on `score` (which the reference mutates too): sde_helper2.py:283-312 (RSDE.sde) and
    :65-94 (corrector) -- for every (given, predicted) pair among '01', '02', '12':
        new_x = cat[x[:, m1], x[:, m2]];  g = autograd.grad(cl_g[m1 m2](new_x.view(B, -1), t).mean(), new_x)
        score[:, m1] -= cl_s * g[:, 0];   score[:, m2] -= cl_s * g[:, 1]
    `poly_pairs`: the PolyMNIST script's variant (train_poly_unet_cont.py:73-89): ONE random (given, predicted) pair per call
    (two torch.randint draws), cl_g(new_x, t, id1, id2), and only the predicted modality's score is updated.
    Networks of this package (lat_sm2_model.ClwithTime2/3) take the HIP input-VJP kernel; any other callable goes through
    torch.autograd on the GPU exactly as in the reference."""
    if cl_g is None or given is None or not given:
        return score
    from .lat_sm2_model import _ClwithTime
    base = int(all_mods[0])
    predicted = "".join(m for m in all_mods if m not in given)
    jobs = []
    if poly_pairs:
        mod1 = given[torch.randint(len(given), (1,)).item()]
        mod2 = predicted[torch.randint(len(predicted), (1,)).item()]
        if cl_s is None:                                           # train_poly_unet_cont.py:86-88, 143-145: no scale, no update -- but the two
            return score                                           # randint draws above have been made, so the RNG stream matches
        jobs.append((cl_g, int(mod1) - base, int(mod2) - base, (False, True), True))
    else:
        for m1, m2 in (("0", "1"), ("0", "2"), ("1", "2")):
            if (m1 in given and m2 in predicted) or (m2 in given and m1 in predicted):
                jobs.append((cl_g[m1 + m2], int(m1) - base, int(m2) - base, (True, True), False))
    for net, c1, c2, upd, with_ids in jobs:
        if isinstance(net, _ClwithTime):
            net.guide_(x, t, score, c1, c2, cl_s, update=upd, ids=(c1, c2))
            continue
        with torch.enable_grad():                                   # a user's own torch module: the reference's autograd path
            new_x = torch.cat([x[:, c1].unsqueeze(1), x[:, c2].unsqueeze(1)], dim=1).detach().requires_grad_(True)
            flat = new_x.view(new_x.shape[0], -1)
            out = net(flat, t, c1, c2) if with_ids else net(flat, t)
            g = torch.autograd.grad(out.mean(), new_x)[0]
        if upd[0]:
            score[:, c1] -= cl_s * g[:, 0]
        if upd[1]:
            score[:, c2] -= cl_s * g[:, 1]
    return score


def _guide_jobs(cl_g, cl_s, given, all_mods, x):
    """The (network, channel 1, channel 2) list of apply_guidance_ for a fixed-pair run, or None when the run cannot be fused (a
    network that is not this package's HIP module, no scale, no modality names)."""
    from .lat_sm2_model import _ClwithTime
    if cl_s is None or all_mods is None or not isinstance(cl_g, dict):
        return None
    base = int(all_mods[0])
    gstr = "".join(all_mods[m] for m in given)
    predicted = "".join(m for m in all_mods if m not in gstr)
    jobs = []
    for m1, m2 in (("0", "1"), ("0", "2"), ("1", "2")):
        if (m1 in gstr and m2 in predicted) or (m2 in gstr and m1 in predicted):
            net = cl_g[m1 + m2]
            if not isinstance(net, _ClwithTime) or net.size_z != x[0, 0].numel():
                return None
            jobs.append((net, int(m1) - base, int(m2) - base))
    return jobs or None


def _guided(cl_g, given):
    return cl_g is not None and given is not None and bool(given)


def _alphas_dev(sde, device):
    if sde.kind == 2:
        return None
    cache = sde.__dict__.setdefault("_sbm_alphas", {})
    key = str(device)
    if key not in cache:
        cache[key] = sde.alphas.to(device=device, dtype=torch.float32).contiguous()
    return cache[key]


def _with_z(score_fn, z_cond):
    """The conditional CelebA script passes z_cond to em_predictor/corrector (train_lat_celebhq_unet_cont2_cond.py:225-226,
    307-309), which the reference's sde_helper2 does not accept (TypeError as shipped).  Intended meaning, defined
    here: score_fn(x, t) = UNetModel(x, t, z=z_cond)."""
    if z_cond is None:
        return score_fn
    return lambda x, t: score_fn(x, t, z=z_cond)


def em_predictor(x, t, score_fn, sde, probability_flow=False, cl_g=None, cl_s=None, target=None, given=None,
                 all_mods=None, z_cond=None, *, noise=None, score=None, philox=None, poly_pairs=False):
    """One reverse-SDE Euler-Maruyama step -> (x, x_mean)   [reference sde_helper2.py:45-52].

    `noise`: the randn_like draw (default: torch.randn_like(x), drawn BEFORE the score call like the reference).
    `score`: a precomputed score_fn(x, t) (skips the call).
    `philox`: (seed, sample_offset, step, draw) -- in-kernel counter-based noise keyed by the global sample index.
    `cl_g`, `cl_s`, `given`, `all_mods`: classifier guidance of the conditional callers (apply_guidance_)."""
    x = as_f32_cuda(x, "x")
    t = as_f32_cuda(t, "t")
    seed, off, step, draw = 0, 0, 0, 0
    if philox is not None:
        seed, off, step, draw = philox
        zp = None
    else:
        z = torch.randn_like(x) if noise is None else as_f32_cuda(noise, "noise")
        zp = z.data_ptr()
    s = as_f32_cuda(_with_z(score_fn, z_cond)(x, t) if score is None else score, "score")
    if _guided(cl_g, given):
        s = apply_guidance_(x, t, s if score is None else s.clone(), cl_g, cl_s, given, all_mods, poly_pairs)
    B = x.shape[0]
    x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
    cfg = sde._cfg(probability_flow=probability_flow)
    _lib.check(_lib.lib().sbm_em_step(ctypes.byref(cfg), x.data_ptr(), s.data_ptr(), zp, t.data_ptr(),
                                      x_new.data_ptr(), x_mean.data_ptr(), B, x[0].numel(), seed, off, step, draw,
                                      _lib.stream_ptr()))
    return x_new, x_mean


def corrector(x, t, score_fn, sde, n_steps, target_snr, cl_g=None, cl_s=None, target=None, given=None, all_mods=None,
              z_cond=None, *, noise=None, philox=None, exact_norms=False, norm_reduce=None, global_batch=None, poly_pairs=False):
    """n_steps Langevin corrector iterations -> (x, x_mean)   [reference sde_helper2.py:54-106].

    `noise`: optional list of n_steps injected draws (default torch.randn_like after each score call).
    `philox`: (seed, sample_offset, step, first_draw) -- draw the noise in-kernel from the counter-based stream keyed by the GLOBAL
      sample index instead (what the fused loop does; independent of how the batch is sharded).
    `exact_norms`: the step size uses the mean of the per-sample norms over the whole batch (sde_helper2.py:97-99); for a batch
      sharded over ranks this computes the shard's two sums in fp64, hands them to `norm_reduce(sums)` (default: all-reduce over
      the default process group; identity for one process) and divides by `global_batch` -- the single-device result, bit for bit."""
    x = as_f32_cuda(x, "x")
    t = as_f32_cuda(t, "t")
    B = x.shape[0]
    cfg = sde._cfg(target_snr=target_snr)
    alphas = _alphas_dev(sde, x.device)
    ap = alphas.data_ptr() if alphas is not None else None
    norms = torch.empty(2 * B, device=x.device, dtype=torch.float32)
    seed, off, step, d0 = philox if philox is not None else (0, 0, 0, 0)
    L = _lib.lib()
    x_mean = x
    for i in range(n_steps):
        g = as_f32_cuda(_with_z(score_fn, z_cond)(x, t), "score")
        if _guided(cl_g, given):
            g = apply_guidance_(x, t, g, cl_g, cl_s, given, all_mods, poly_pairs)
        if philox is not None:
            zp = None
        else:
            z = torch.randn_like(x) if noise is None else as_f32_cuda(noise[i], "noise")
            zp = z.data_ptr()
        x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
        if exact_norms:
            from . import distributed as D
            sums = torch.empty(2, device=x.device, dtype=torch.float64)
            _lib.check(L.sbm_langevin_norms(g.data_ptr(), zp, norms.data_ptr(), sums.data_ptr(), B, x[0].numel(), seed, off, step, d0 + i,
                                            _lib.stream_ptr()))
            sums = (norm_reduce or D.langevin_norm_allreduce)(sums)
            _lib.check(L.sbm_langevin_update(ctypes.byref(cfg), x.data_ptr(), g.data_ptr(), zp, t.data_ptr(), ap, x_new.data_ptr(),
                                             x_mean.data_ptr(), sums.data_ptr(), int(global_batch or B), B, x[0].numel(), seed, off, step,
                                             d0 + i, _lib.stream_ptr()))
        else:
            _lib.check(L.sbm_langevin_step(ctypes.byref(cfg), x.data_ptr(), g.data_ptr(), zp, t.data_ptr(), ap, x_new.data_ptr(),
                                           x_mean.data_ptr(), norms.data_ptr(), B, x[0].numel(), seed, off, step, d0 + i,
                                           _lib.stream_ptr()))
        x = x_new
    return x, x_mean


def _fused_ok(model, z_cond=None):
    from .unet_model import Unet
    from .unet_openai import UNetModel
    if isinstance(model, UNetModel):
        return not model.training or not model.dropout
    return isinstance(model, Unet) and model.with_time_emb and z_cond is None


def uncond_sampler(sample_shape, model, device, sde, eps=1e-3, probability_flow=False, pc=False, n_steps=1,
                   target_snr=0.16, cl_g=None, cl_s=None, target=None, *, order="cp", noise=None, x0=None, seed=None,
                   sample_offset=0, use_graph=True, return_timing=False, z_cond=None, exact_sharded_norms=False,
                   global_batch=None, norm_reduce=None):
    """N-step predictor(-corrector) sampling; returns the last sub-step's x_mean   [reference sde_helper2.py:115-128].

    `exact_sharded_norms` (with pc=True, batch sharded over ranks): the corrector's step size uses the mean norms of the GLOBAL
    batch (`global_batch` rows; two fp64 sums all-reduced per corrector iteration by `norm_reduce`, default the default process
    group) -- the single-device reference result whatever the world size.  Runs the loop from Python (a collective sits inside
    each step) with the same Philox stream as the fused loop.  Default (False): per-shard means, no in-loop collective.
    `cl_g` / `cl_s` / `target` are accepted and inert, as in the reference: its uncond_sampler never passes `given`, so the guidance
    branch (`cl_g is not None and given`) cannot fire (sde_helper2.py:124-126, 283)."""
    if order not in ("cp", "pc"):
        raise ValueError("order must be 'cp' (sde_helper2) or 'pc' (train_poly_unet_cont)")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("uncond_sampler runs on the GPU only (no CPU path)")
    with torch.no_grad():
        x = (sde.prior_sampling(sample_shape) if x0 is None else x0).to(device=device, dtype=torch.float32).contiguous()
        if x is x0:
            x = x.clone()
        timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
        B = x.shape[0]
        dps = 1 + (n_steps if pc else 0)
        if noise is not None:
            noise = as_f32_cuda(noise, "noise")
            if noise.shape[0] < sde.N * dps:
                raise ValueError(f"noise needs {sde.N * dps} draws")
        if exact_sharded_norms and pc:
            if noise is not None:
                raise ValueError("exact_sharded_norms draws its noise from the Philox stream (keyed by the global sample index)")
            seed = fresh_seed(seed)
            x_mean = x
            for i in range(sde.N):
                vec_t = torch.ones(B, device=device) * timesteps[i]
                kw = dict(z_cond=z_cond, exact_norms=True, norm_reduce=norm_reduce, global_batch=global_batch or B)
                if order == "cp":
                    x, x_mean = corrector(x, vec_t, model, sde, n_steps, target_snr, philox=(seed, sample_offset, i, 0), **kw)
                    x, x_mean = em_predictor(x, vec_t, model, sde, probability_flow, z_cond=z_cond, philox=(seed, sample_offset, i, n_steps))
                else:
                    x, x_mean = em_predictor(x, vec_t, model, sde, probability_flow, z_cond=z_cond, philox=(seed, sample_offset, i, 0))
                    x, x_mean = corrector(x, vec_t, model, sde, n_steps, target_snr, philox=(seed, sample_offset, i, 1), **kw)
            return x_mean
        if _fused_ok(model, z_cond):
            from .unet_openai import UNetModel
            if isinstance(model, UNetModel):
                eng = model._engine_for(B, x.shape[2], x.shape[3], z_cond is not None)
                zc = as_f32_cuda(z_cond, "z_cond") if z_cond is not None else None
                _lib.check(_lib.lib().sbm_model_set_aux(eng.h, zc.data_ptr() if zc is not None else None))
            else:
                eng = model._engine_for(B, x.shape[2], x.shape[3])
            cfg = sde._cfg(target_snr=target_snr, probability_flow=probability_flow)
            seed = fresh_seed(seed)
            scfg = _lib.SamplerCfg(int(bool(pc)), int(n_steps), 0 if order == "cp" else 1, int(bool(use_graph)),
                                   int(seed), int(sample_offset))
            alphas = _alphas_dev(sde, device) if pc else None
            x_mean = torch.empty_like(x)
            ms = ctypes.c_float(0.0)
            _lib.check(_lib.lib().sbm_sampler_run(eng.h, ctypes.byref(cfg), ctypes.byref(scfg), timesteps.data_ptr(),
                                                  alphas.data_ptr() if alphas is not None else None,
                                                  noise.data_ptr() if noise is not None else None, x.data_ptr(),
                                                  x_mean.data_ptr(), 0, sde.N, ctypes.byref(ms) if return_timing else None,
                                                  _lib.stream_ptr()))
            return (x_mean, ms.value) if return_timing else x_mean
        # generic score_fn: Python loop over the HIP step kernels
        k = 0
        x_mean = x
        for i in range(sde.N):
            vec_t = torch.ones(B, device=device) * timesteps[i]

            def corr(x):
                nonlocal k
                nz = None if noise is None else [noise[k + j] for j in range(n_steps)]
                k += n_steps if noise is not None else 0
                return corrector(x, vec_t, model, sde, n_steps, target_snr, z_cond=z_cond, noise=nz)

            def pred(x):
                nonlocal k
                nz = None if noise is None else noise[k]
                k += 1 if noise is not None else 0
                return em_predictor(x, vec_t, model, sde, probability_flow, z_cond=z_cond, noise=nz)

            if order == "cp":
                if pc:
                    x, x_mean = corr(x)
                x, x_mean = pred(x)
            else:
                x, x_mean = pred(x)
                if pc:
                    x, x_mean = corr(x)
        return x_mean


def renoise_observed(x, z_obs, t, sde, given, out=None):
    """x with the observed modality channels replaced by mean(t)*z_obs + std(t)*z_obs (train_poly_unet_cont.py:356-364:
    the reference scales the clean latent by std instead of drawing noise -- reproduced as is)."""
    x = as_f32_cuda(x, "x")
    z_obs = as_f32_cuda(z_obs, "z_obs")
    t = as_f32_cuda(t, "t")
    B, n_mod = x.shape[0], x.shape[1]
    mask = 0
    for m in given:
        mask |= 1 << int(m)
    out = torch.empty_like(x) if out is None else out
    cfg = sde._cfg()
    _lib.check(_lib.lib().sbm_renoise_observed(ctypes.byref(cfg), x.data_ptr(), z_obs.data_ptr(), t.data_ptr(), out.data_ptr(), mask, B, n_mod,
                                               x[0, 0].numel(), _lib.stream_ptr()))
    return out


def cond_sampler(z_obs, given, model, sde, eps=1e-3, pc=False, n_steps=1, target_snr=0.16, noise_obs=False, *, x0=None, noise=None,
                 z_cond=None, cl_g=None, cl_s=None, all_mods=None, poly_pairs=False, seed=None, sample_offset=0, use_graph=True,
                 return_timing=False, fused_guidance=True):
    """Conditional generation loop of the training scripts' evaluators (train_poly_unet_cont.evaluate :340-381,
    train_lat_celebhq_unet_cont2.calc_perf :277-318): channels in `given` hold the observed latents z_obs (re-noised to level t at
    every step when noise_obs), the others start from N(0,1) (or x0) and follow the reverse SDE, predictor then corrector.
    Returns the joint state (B, n_mod, H, W): unobserved channels = final x_mean, observed = clean z_obs.

    With this package's score nets the whole loop runs on the device as ONE captured step replayed N times (`sbm_cond_sampler_run`:
    re-noise + score + predictor [+ corrector] + write-back), also under `cl_g` guidance when every guidance network is this package's
    `ClwithTime2/3` and the pairs are fixed (`all_mods` naming the modalities, e.g. '012': `sbm_guided_cond_sampler_run`).  A generic
    score callable, a user's own torch guidance module, the PolyMNIST random-pair variant (`poly_pairs`) or `fused_guidance=False` drive
    the loop from Python over the same HIP step kernels.
    noise: injected draws [(N * (1 + pc * n_steps)), B, n_mod, H, W] in execution order (parity tests); default in-kernel Philox
    (fused) / torch.randn_like (generic)."""
    z_obs = as_f32_cuda(z_obs, "z_obs")
    device = z_obs.device
    B, n_mod = z_obs.shape[0], z_obs.shape[1]
    given = sorted(int(m) for m in given)
    keep = torch.zeros(n_mod, dtype=torch.bool, device=device)
    keep[given] = True
    keep = keep.view(1, n_mod, 1, 1)
    mask = 0
    for m in given:
        mask |= 1 << m
    with torch.no_grad():
        x = (torch.randn_like(z_obs) if x0 is None else as_f32_cuda(x0, "x0")).clone()
        x = torch.where(keep, z_obs, x)
        timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
        dps = 1 + (n_steps if pc else 0)
        if noise is not None:
            noise = as_f32_cuda(noise, "noise")
            if noise.shape[0] < sde.N * dps:
                raise ValueError(f"noise needs {sde.N * dps} draws")
        guided = cl_g is not None and len(given) > 0
        jobs = _guide_jobs(cl_g, cl_s, given, all_mods, x) if guided and not poly_pairs and fused_guidance else None
        if _fused_ok(model, z_cond) and (not guided or jobs is not None):
            from .unet_openai import UNetModel
            if isinstance(model, UNetModel):
                eng = model._engine_for(B, x.shape[2], x.shape[3], z_cond is not None)
                zc = as_f32_cuda(z_cond, "z_cond") if z_cond is not None else None
                _lib.check(_lib.lib().sbm_model_set_aux(eng.h, zc.data_ptr() if zc is not None else None))
            else:
                eng = model._engine_for(B, x.shape[2], x.shape[3])
            cfg = sde._cfg(target_snr=target_snr)
            scfg = _lib.SamplerCfg(int(bool(pc)), int(n_steps), 1, int(bool(use_graph)), fresh_seed(seed), int(sample_offset))
            ccfg = _lib.CondCfg(mask, int(bool(noise_obs)), n_mod)
            alphas = _alphas_dev(sde, device) if pc else None
            x_mean = torch.empty_like(x)
            ms = ctypes.c_float(0.0)
            if jobs:
                # guidance inside the captured step (sbm_guided_cond_sampler_run): the same clg kernel after every score evaluation
                arr = (_lib.GuideJob * len(jobs))(*[_lib.GuideJob(net._engine_for().h, c1, c2, c1, c2, 3) for net, c1, c2 in jobs])
                _lib.check(_lib.lib().sbm_guided_cond_sampler_run(eng.h, ctypes.byref(cfg), ctypes.byref(scfg), ctypes.byref(ccfg), z_obs.data_ptr(),
                                                                  arr, len(jobs), ctypes.c_float(float(cl_s)), timesteps.data_ptr(),
                                                                  alphas.data_ptr() if alphas is not None else None,
                                                                  noise.data_ptr() if noise is not None else None, x.data_ptr(), x_mean.data_ptr(), 0,
                                                                  sde.N, ctypes.byref(ms) if return_timing else None, _lib.stream_ptr()))
            else:
                _lib.check(_lib.lib().sbm_cond_sampler_run(eng.h, ctypes.byref(cfg), ctypes.byref(scfg), ctypes.byref(ccfg), z_obs.data_ptr(),
                                                           timesteps.data_ptr(), alphas.data_ptr() if alphas is not None else None,
                                                           noise.data_ptr() if noise is not None else None, x.data_ptr(), x_mean.data_ptr(), 0,
                                                           sde.N, ctypes.byref(ms) if return_timing else None, _lib.stream_ptr()))
            return (x_mean, ms.value) if return_timing else x_mean
        gstr = None
        if guided:
            if all_mods is None:
                raise ValueError("guidance needs all_mods (the modality characters of the joint state, e.g. '012')")
            gstr = "".join(all_mods[m] for m in given)
        x_mean = x
        k = 0
        for i in range(sde.N):
            vec_t = torch.ones(B, device=device) * timesteps[i]
            x_in = renoise_observed(x, z_obs, vec_t, sde, given) if noise_obs else torch.where(keep, z_obs, x)
            nz = None if noise is None else noise[k]
            k += 1
            x_upd, x_mean = em_predictor(x_in, vec_t, model, sde, cl_g=cl_g, cl_s=cl_s, given=gstr, all_mods=all_mods, z_cond=z_cond, noise=nz,
                                         poly_pairs=poly_pairs)
            if pc:
                nzs = None if noise is None else [noise[k + j] for j in range(n_steps)]
                k += n_steps
                x_upd, x_mean = corrector(x_upd, vec_t, model, sde, n_steps, target_snr, cl_g=cl_g, cl_s=cl_s, given=gstr, all_mods=all_mods,
                                          z_cond=z_cond, noise=nzs, poly_pairs=poly_pairs)
            x = x_upd
        return torch.where(keep, z_obs, x_mean)
