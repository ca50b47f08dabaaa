"""Drop-in for the reference's `lat_sm2_model.LSMPoly64_sm` (lat_sm2_model.py:348-371): the MLP score network of the
DSM/Langevin variant (train_lat_sm.py:180).  Seven Linear layers with Softplus between, on the concatenated latents
(B, n_mod*size_z); the forward runs as seven fused GEMM(+bias+softplus) launches of the HIP library.

Also `ClwithTime2` / `ClwithTime3`: the time-conditioned pairwise classifier / energy network the SDE scripts use for guidance
(`ClwithTime2(n_mod=2, size_z, n_class=1)`, train_lat_celebhq_unet_cont2.py:487; `ClwithTime3(...)`, train_poly_unet_cont.py:587).
The reference imports both names from this module but no longer defines them, so the architecture is this package's:
    temb = Softplus(Linear(64, 128)(SinusoidalPositionEmbeddings(64)(t)))          e1, e2 = mod_emb(id1), mod_emb(id2)   [ClwithTime3]
    out  = Linear(256, n_class)(Softplus(Linear(512, 256)(Softplus(Linear(D0, 512)(cat[x, temb, e1, e2])))))
Forward and the gradient of `out.mean()` with respect to x (what the guidance needs) run in one HIP kernel (csrc/guidance.hip).
"""
import ctypes

import torch

from . import _lib, arch
from ._module import ParamTreeModule, as_f32_cuda


class LSMPoly64_sm(ParamTreeModule):
    def __init__(self, n_mod=5, size_z=128):
        super().__init__()
        self.n_mod, self.size_z = n_mod, size_z
        if (n_mod * size_z) % 32:
            raise NotImplementedError("LSMPoly64_sm: n_mod*size_z must be a multiple of 32 on this backend")
        self._build_tree(arch.mlp_entries(n_mod, size_z))

    def _engine_for(self, B):
        return self._engine(("mlp", B), lambda sd: _lib.Engine(_lib.SBM_MLP, [self.n_mod, self.size_z], B, sd))

    def forward(self, x):
        """x (B, n_mod*size_z) -> score (B, n_mod*size_z)   [reference lat_sm2_model.py:370]"""
        x = as_f32_cuda(x, "LSMPoly64_sm input")
        if x.dim() != 2 or x.shape[1] != self.n_mod * self.size_z:
            raise ValueError(f"expected (B, {self.n_mod * self.size_z}), got {tuple(x.shape)}")
        out = torch.empty_like(x)
        return self._engine_for(x.shape[0]).forward(x, None, out)


class _ClwithTime(ParamTreeModule):
    variant = 2

    def __init__(self, n_mod=2, size_z=64, n_class=1):
        super().__init__()
        if n_mod != 2:
            raise NotImplementedError("the guidance network scores a PAIR of latents: n_mod must be 2 (every caller passes 2)")
        self.n_mod, self.size_z, self.n_class = n_mod, size_z, n_class
        self._build_tree(arch.clg_entries(self.variant, n_mod, size_z, n_class))

    def _engine_for(self):
        return self._engine(("clg",), lambda sd: _lib.Engine(_lib.SBM_CLG, [self.variant, self.n_mod, self.size_z, self.n_class], 1, sd))

    def _forward(self, x, t, id1, id2):
        x = as_f32_cuda(x, "guidance input")
        t = as_f32_cuda(t, "t").reshape(-1)
        B = x.shape[0]
        if x.dim() != 2 or x.shape[1] != self.n_mod * self.size_z or t.numel() != B:
            raise ValueError(f"expected x (B, {self.n_mod * self.size_z}) and t (B,)")
        out = torch.empty((B, self.n_class), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().sbm_clg_forward(self._engine_for().h, x.data_ptr(), t.data_ptr(), int(id1), int(id2), out.data_ptr(), B,
                                              _lib.stream_ptr()))
        return out

    def guide_(self, x, t, score, ch1, ch2, cl_s, update=(True, True), ids=(0, 0), return_grad=False):
        """score[:, ch1] -= cl_s * g[:, 0]; score[:, ch2] -= cl_s * g[:, 1] in place, g = d self(cat[x[:, ch1], x[:, ch2]], t).mean() / d pair
        (reference sde_helper2.py:286-294).  x, score: (B, C, H, W) with H*W == size_z."""
        x = as_f32_cuda(x, "x")
        t = as_f32_cuda(t, "t").reshape(-1)
        _lib._require_cuda(score, "score")
        if score.dtype != torch.float32 or not score.is_contiguous() or score.shape != x.shape:
            raise ValueError("score must be a contiguous fp32 tensor shaped like x (it is updated in place)")
        B, C = x.shape[0], x.shape[1]
        per = x[0, 0].numel()
        grad = torch.empty((B, 2 * per), device=x.device, dtype=torch.float32) if return_grad else None
        mask = (1 if update[0] else 0) | (2 if update[1] else 0)
        _lib.check(_lib.lib().sbm_clg_guide(self._engine_for().h, x.data_ptr(), t.data_ptr(), int(ch1), int(ch2), int(ids[0]), int(ids[1]),
                                            ctypes.c_float(float(cl_s)), mask, score.data_ptr(), grad.data_ptr() if grad is not None else None,
                                            B, C, per, _lib.stream_ptr()))
        return grad


class ClwithTime2(_ClwithTime):
    """cl_g['01'](new_x.view(B, -1), t) of the CelebA scripts (sde_helper2.py:290)."""
    variant = 2

    def forward(self, x, t):
        return self._forward(x, t, 0, 0)


class ClwithTime3(_ClwithTime):
    """cl_g(new_x.view(B, -1), t, id1, id2) of the PolyMNIST script (train_poly_unet_cont.py:82): also told WHICH two modalities."""
    variant = 3

    def forward(self, x, t, id1, id2):
        return self._forward(x, t, id1, id2)
