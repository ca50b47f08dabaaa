"""Drop-ins for the CelebA attribute autoencoders of the reference's `h_vae_model.py`:
CelebAAttrNewBN (:712-779) and CelebAAttrNewBNAE (:839-899).  `.decoder(z)` = 5 x (Linear,
BatchNorm1d, ReLU) + Linear -> 18 logits, as HIP GEMMs with the BatchNorm folded in."""
import torch

from . import _lib, arch
from ._module import ParamTreeModule, as_f32_cuda


class CelebAAttrNewBN(ParamTreeModule):
    _ae = False

    def __init__(self, size_z=64, att_size=18):
        super().__init__()
        self.size_z, self.att_size = size_z, att_size
        self._build_tree(arch.attr_entries(size_z, att_size, ae=self._ae))

    def decoder(self, z):
        _lib._require_cuda(z, "decoder input")
        if self.training:
            raise RuntimeError("decoder runs with BatchNorm folded (eval mode); call .eval() first")
        if z.dim() != 2 or z.shape[1] != self.size_z:
            raise ValueError(f"expected (B, {self.size_z}) latents")
        if z.dtype != torch.float32 or z.stride(1) != 1 or z.stride(0) % 4 or z.data_ptr() % 16:
            z = z.to(torch.float32).contiguous()
        B, pitch = z.shape[0], z.stride(0)
        eng = self._engine(("dec", B, pitch),
                           lambda sd: _lib.Engine(_lib.SBM_ATTRDEC, [self.size_z, self.att_size], B, sd, in_pitch=pitch))
        out = torch.empty((B, self.att_size), device=z.device, dtype=torch.float32)
        return eng.forward(z, None, out)

    def encoder(self, x):
        """attributes (B, att_size) -> (mu, logvar)  [h_vae_model.py:756-759]; the AE class returns mu  [:884-887]"""
        x = as_f32_cuda(x, "encoder input")
        if self.training:
            raise RuntimeError("encoder runs with BatchNorm folded (eval mode); call .eval() first")
        if x.dim() != 2 or x.shape[1] != self.att_size:
            raise ValueError(f"expected (B, {self.att_size}) attribute vectors")
        B = x.shape[0]
        eng = self._engine(("enc", B), lambda sd: _lib.Engine(_lib.SBM_ATTRENC, [self.size_z, self.att_size, int(self._ae)], B, sd))
        nout = self.size_z if self._ae else 2 * self.size_z
        out = torch.empty((B, nout), device=x.device, dtype=torch.float32)
        eng.forward(x, None, out)
        return out if self._ae else (out[:, :self.size_z], out[:, self.size_z:])


class CelebAAttrNewBNAE(CelebAAttrNewBN):
    _ae = True

    def __init__(self, size_z=64):
        super().__init__(size_z, 18)
