"""Drop-ins for the per-modality autoencoders of the reference's `h_vae_model_copy.py`:
ResVAE / ResAE (PolyMNIST, :91-174) and ResVAEN / ResAEN (CelebA, :457-588).

Same constructors and `state_dict` keys.  `.decoder(z)` -- the only method on the sampling hot
path -- runs as fused HIP kernels (BatchNorm folded into the convolutions, nearest upsampling folded
into the consumer's reads).  `.encoder(x)` (the conditional loop's source of observed latents) runs the
same way.
"""
import torch

from . import _lib, arch
from ._module import ParamTreeModule, as_f32_cuda


class _ResBase(ParamTreeModule):
    _variant_name = "ResVAE"
    _variant_id = 0
    _returns_logvar = True

    def __init__(self, enc_channel_list, dec_channel_list, size_in=64, size_z=64, img_ch=3):
        super().__init__()
        self.enc_channel_list = [tuple(c) for c in enc_channel_list]
        self.dec_channel_list = [tuple(c) for c in dec_channel_list]
        self.size_in, self.size_z, self.img_ch = size_in, size_z, img_ch
        self._build_tree(arch.resvae_entries(self._variant_name, self.enc_channel_list, self.dec_channel_list,
                                             size_in, size_z, img_ch))

    def decoder(self, z):
        """z (B, size_z) -> (B, img_ch, size_in, size_in).  z may be the strided view
        z_out[:, ind].view(B, size_z) of the sampler state (train_poly_unet_cont.py:303)."""
        _lib._require_cuda(z, "decoder input")
        if self.training:
            raise RuntimeError("decoder runs with BatchNorm folded (eval mode); call .eval() as the reference's "
                               "evaluate() does (train_poly_unet_cont.py:294-295)")
        if z.dim() != 2 or z.shape[1] != self.size_z:
            raise ValueError(f"expected (B, {self.size_z}) latents")
        if z.dtype != torch.float32 or z.stride(1) != 1 or z.stride(0) % 4 or z.data_ptr() % 16:
            z = z.to(torch.float32).contiguous()
        B, pitch = z.shape[0], z.stride(0)
        cfg = _lib.resdec_cfg_ints(self._variant_id, self.enc_channel_list, self.dec_channel_list, self.size_in,
                                   self.size_z, self.img_ch)
        eng = self._engine(("dec", B, pitch), lambda sd: _lib.Engine(_lib.SBM_RESDEC, cfg, B, sd, in_pitch=pitch))
        out = torch.empty((B, self.img_ch, self.size_in, self.size_in), device=z.device, dtype=torch.float32)
        return eng.forward(z, None, out)

    def _encode(self, x):
        """image (B, img_ch, size_in, size_in) -> (mu, logvar)   [ResEncoder.forward h_vae_model_copy.py:63-71, ResEncoderN :401-409]"""
        x = as_f32_cuda(x, "encoder input")
        if self.training:
            raise RuntimeError("encoder runs with BatchNorm folded (eval mode); call .eval() first")
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.img_ch, self.size_in, self.size_in):
            raise ValueError(f"expected (B, {self.img_ch}, {self.size_in}, {self.size_in}) images")
        B = x.shape[0]
        cfg = _lib.resdec_cfg_ints(self._variant_id, self.enc_channel_list, self.dec_channel_list, self.size_in,
                                   self.size_z, self.img_ch)
        eng = self._engine(("enc", B), lambda sd: _lib.Engine(_lib.SBM_RESENC, cfg, B, sd))
        out = torch.empty((B, 2 * self.size_z), device=x.device, dtype=torch.float32)
        eng.forward(x, None, out)
        return out[:, :self.size_z], out[:, self.size_z:]

    def encoder(self, x):
        """VAE variants return (mu, logvar) (h_vae_model_copy.py:114-116, 480-482), AE variants mu (:162-164, 572-574)."""
        mu, logvar = self._encode(x)
        return (mu, logvar) if self._returns_logvar else mu.contiguous()

    def sample(self, amount, device):
        return self.decoder(torch.randn(amount, self.size_z).to(device))


class ResVAE(_ResBase):
    _variant_name, _variant_id = "ResVAE", 0


class ResAE(_ResBase):
    _variant_name, _variant_id = "ResAE", 0
    _returns_logvar = False


class ResVAEN(_ResBase):
    _variant_name, _variant_id = "ResVAEN", 1


class ResAEN(_ResBase):
    _variant_name, _variant_id = "ResAEN", 1
    _returns_logvar = False
