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


class _CelRepBase(ParamTreeModule):
    """Conditioning encoders of the conditional CelebA score model (train_lat_celebhq_unet_cont2_cond.get_conds :452-460):
    modality encoder (its mu output) followed by a ProjectionHead (Linear, GELU, Linear, +residual, LayerNorm)."""

    def _run(self, x, kind, cfg, shape_ok, what):
        x = as_f32_cuda(x, what)
        if self.training:
            raise RuntimeError("runs with BatchNorm folded and Dropout off (eval mode); call .eval() first")
        if not shape_ok(x):
            raise ValueError(f"unexpected {what} shape {tuple(x.shape)}")
        B = x.shape[0]
        eng = self._engine(("rep", B), lambda sd: _lib.Engine(kind, cfg, B, sd))
        out = torch.empty((B, self.size_z), device=x.device, dtype=torch.float32)
        return eng.forward(x, None, out)


class CelAttrRep(_CelRepBase):
    def __init__(self, att_size=18, size_z=512):
        super().__init__()
        self.att_size, self.size_z = att_size, size_z
        self._build_tree(arch.cel_attr_rep_entries(att_size, size_z))

    def forward(self, x):
        """attributes (B, att_size) -> (B, size_z)   [h_vae_model.py:998-1001]"""
        return self._run(x, _lib.SBM_ATTRENC, [self.size_z, self.att_size, 2], lambda t: t.dim() == 2 and t.shape[1] == self.att_size, "attributes")


class _CelImageRep(_CelRepBase):
    _enc, _ch = arch.CEL_IMG_REP_ENC, 3

    def __init__(self, size_z=512):
        super().__init__()
        self.size_z, self.size_in, self.img_ch = size_z, 128, self._ch
        self.enc_channel_list1 = [tuple(c) for c in self._enc]
        self._build_tree(arch.cel_rep_entries(self.enc_channel_list1, size_z, self._ch, 128))

    def forward(self, x):
        """image (B, img_ch, 128, 128) -> (B, size_z)   [h_vae_model.py:1052-1055, 1069-1072]"""
        cfg = _lib.resdec_cfg_ints(3, self.enc_channel_list1, [], 128, self.size_z, self._ch)      # variant 3 = N-variant encoder + projection head
        return self._run(x, _lib.SBM_RESENC, cfg, lambda t: t.dim() == 4 and tuple(t.shape[1:]) == (self._ch, 128, 128), "image")


class CelImgRep(_CelImageRep):
    _enc, _ch = arch.CEL_IMG_REP_ENC, 3


class CelMaskRep(_CelImageRep):
    _enc, _ch = arch.CEL_MASK_REP_ENC, 1


def get_conds(xs, cond_models, given):
    """Mean over the given modalities of their conditioning embeddings, or None (train_lat_celebhq_unet_cont2_cond.py:452-460)."""
    if len(given) == 0:
        return None
    with torch.no_grad():
        return torch.cat([cond_models[int(i)](xs[int(i)]).unsqueeze(0) for i in given]).mean(dim=0)
