"""Drop-in for the reference's `lat_sm2_model.LSMPoly64_sm` (lat_sm2_model.py:348-371): the MLP score network of the
DSM/Langevin variant (train_lat_sm.py:180).  Seven Linear layers with Softplus between, on the concatenated latents
(B, n_mod*size_z); the forward runs as seven fused GEMM(+bias+softplus) launches of the HIP library.
"""
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
