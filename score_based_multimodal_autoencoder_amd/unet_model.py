"""Drop-in for the reference's `unet_model.Unet` (unet_model.py:189-323): same constructor, same
`state_dict` keys, same `forward(x, time)`; the forward runs as fused HIP kernels on gfx950.

Only the live configuration is provided: ConvNeXt blocks (`use_convnext=True`, the default no caller
overrides -- SURVEY.md D3).  `use_convnext=False` (dead code in the reference) raises.
"""
import torch

from . import _lib, arch
from ._module import ParamTreeModule, as_f32_cuda


class Unet(ParamTreeModule):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, with_time_emb=True,
                 resnet_block_groups=8, use_convnext=True, convnext_mult=2):
        super().__init__()
        if not use_convnext:
            raise NotImplementedError("Unet(use_convnext=False): the ResnetBlock variant is dead code in the reference "
                                      "(no caller selects it) and is not built")
        self.dim, self.channels, self.dim_mults = dim, channels, tuple(dim_mults)
        self.init_dim, self.out_dim = init_dim, out_dim
        self.with_time_emb, self.convnext_mult = with_time_emb, convnext_mult
        self._build_tree(arch.unet_entries(dim, channels, self.dim_mults, init_dim, out_dim, with_time_emb, convnext_mult))

    def _engine_for(self, B, H, W):
        cfg = _lib.unet_cfg_ints(self.dim, self.channels, self.dim_mults, H, W, self.init_dim, self.out_dim,
                                 self.with_time_emb, self.convnext_mult)
        return self._engine(("unet", B, H, W), lambda sd: _lib.Engine(_lib.SBM_UNET, cfg, B, sd))

    def forward(self, x, time=None):
        """x (B, channels, H, W), time (B,) -> (B, out_dim, H, W)   [reference unet_model.py:275]"""
        x = as_f32_cuda(x, "Unet input")
        B, C, H, W = x.shape
        if C != self.channels:
            raise ValueError(f"expected {self.channels} input channels, got {C}")
        t = None
        if self.with_time_emb:
            if time is None:
                raise ValueError("time embedding is enabled: `time` must be given")
            t = as_f32_cuda(time, "time").reshape(-1)       # integer sigma indices promote to fp32 as in the reference
            if t.numel() != B:
                raise ValueError("time must have one entry per sample")
        eng = self._engine_for(B, H, W)
        out_c = self.out_dim if self.out_dim is not None else self.channels
        out = torch.empty((B, out_c, H, W), device=x.device, dtype=torch.float32)
        return eng.forward(x, t, out)

    def flops_per_sample(self, H=8, W=8):
        """Dense-convention FLOP per sample per forward."""
        eng = self._engine_for(1, H, W)
        return eng.info["flops_dense"]
