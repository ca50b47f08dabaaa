"""Drop-in for the reference's `unet_openai.UNetModel` (unet_openai.py:361-575): the ADM-style score net of the
conditional CelebA script (`UNetModel(3, 128, 3, 2, (), 0.1, (1,2,4,8), num_heads=1, use_z=True, z_dim=512)`,
train_lat_celebhq_unet_cont2_cond.py:651-653).  Same constructor, same `state_dict` keys, same
`forward(x, timesteps, z=None, y=None)`; eval-mode semantics (Dropout is the identity); the forward runs as HIP
kernels.  Class-conditioning (`num_classes`), `use_scale_shift_norm` and 1-D/3-D variants have no live caller
and raise."""
import torch

from . import _lib, arch
from ._module import ParamTreeModule, as_f32_cuda


class UNetModel(ParamTreeModule):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, z_dim=None, num_classes=None, use_checkpoint=False,
                 num_heads=1, num_heads_upsample=-1, use_scale_shift_norm=False, use_z=False):
        super().__init__()
        if dims != 2 or not conv_resample or num_classes is not None or use_scale_shift_norm:
            raise NotImplementedError("only the live configuration is built: dims=2, conv_resample, no class conditioning, "
                                      "no scale-shift norm")
        if num_heads_upsample not in (-1, num_heads):
            raise NotImplementedError("num_heads_upsample != num_heads")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.channel_mult, self.num_heads, self.z_dim, self.use_z = tuple(channel_mult), num_heads, z_dim, bool(use_z)
        self.dropout = dropout
        self._build_tree(arch.unet_openai_entries(in_channels, model_channels, out_channels, num_res_blocks,
                                                  self.attention_resolutions, self.channel_mult, z_dim, self.use_z))

    def _engine_for(self, B, H, W, with_z):
        cfg = _lib.unet_oa_cfg_ints(self.in_channels, self.model_channels, self.out_channels, self.num_res_blocks,
                                    self.attention_resolutions, self.channel_mult, H, W, self.num_heads, self.z_dim, self.use_z)
        return self._engine(("oa", B, H, W, with_z), lambda sd: _lib.Engine(_lib.SBM_UNET_OA, cfg, B, sd, in_pitch=int(with_z)))

    def forward(self, x, timesteps, z=None, y=None, **kwargs):
        """x (B, in_channels, H, W), timesteps (B,), z (B, z_dim) or None -> (B, out_channels, H, W)  [unet_openai.py:538]"""
        if y is not None:
            raise NotImplementedError("class conditioning (y) has no live caller and is not built")
        if self.training and self.dropout:
            raise RuntimeError("the HIP path implements eval() mode (Dropout = identity); call .eval() as the reference's "
                               "calc_perf() does (train_lat_celebhq_unet_cont2_cond.py:241-244)")
        x = as_f32_cuda(x, "UNetModel input")
        B, C, H, W = x.shape
        t = as_f32_cuda(timesteps, "timesteps").reshape(-1)
        with_z = z is not None
        if with_z:
            if not self.use_z:
                raise AssertionError("z given but the model has no proj (use_z=False)")       # reference asserts too (:556)
            z = as_f32_cuda(z, "z")
            assert tuple(z.shape) == (B, self.z_dim)
        eng = self._engine_for(B, H, W, with_z)
        out = torch.empty((B, self.out_channels, H, W), device=x.device, dtype=torch.float32)
        return eng.forward_aux(x, t, z, out)
