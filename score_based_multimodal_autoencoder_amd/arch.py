"""Architecture tables: `state_dict` key names and shapes of the reference modules on the hot path.

Pure Python (no torch): used by the nn.Module shims to create parameters under the reference's key
names, by the tests to build closed-form weights, and mirrored by the C++ plan builder
(csrc/plan.cpp), which consumes tensors in exactly the order these generators yield them.
"""

POLY_ENC = [(64, 64, 64, 2), (64, 128, 128, 2), (128, 256, 256, 2)]
POLY_DEC = [(256, 128, 128, 2), (128, 128, 64, 2), (64, 64, 64, 2)]
CEL_IMG_ENC = [(64, 128, 128, 2), (128, 256, 256, 2), (256, 512, 512, 2)]
CEL_IMG_DEC = [(512, 512, 256, 2), (256, 256, 128, 2), (128, 128, 64, 2)]
CEL_MASK_ENC = [(64, 128, 128, 4), (128, 256, 256, 4)]
CEL_MASK_DEC = [(256, 256, 128, 4), (128, 128, 64, 4)]

ATTN_HEADS, ATTN_DIM_HEAD = 4, 32          # unet_model.py:127,152 defaults, never overridden


def unet_dims(dim, channels, dim_mults, init_dim=None, out_dim=None):
    """(init_dim, dims, in_out, out_dim) as unet_model.py:206-212,269 derive them."""
    init_dim = init_dim if init_dim is not None else dim // 3 * 2
    dims = [init_dim] + [dim * m for m in dim_mults]
    return init_dim, dims, list(zip(dims[:-1], dims[1:])), (out_dim if out_dim is not None else channels)


def _convnext(pre, cin, cout, time_dim, mult=2):
    """ConvNextBlock parameters (unet_model.py:95-113)."""
    if time_dim:
        yield pre + "mlp.1.weight", (cin, time_dim)
        yield pre + "mlp.1.bias", (cin,)
    yield pre + "ds_conv.weight", (cin, 1, 7, 7)
    yield pre + "ds_conv.bias", (cin,)
    yield pre + "net.0.weight", (cin,)
    yield pre + "net.0.bias", (cin,)
    yield pre + "net.1.weight", (cout * mult, cin, 3, 3)
    yield pre + "net.1.bias", (cout * mult,)
    yield pre + "net.3.weight", (cout * mult,)
    yield pre + "net.3.bias", (cout * mult,)
    yield pre + "net.4.weight", (cout, cout * mult, 3, 3)
    yield pre + "net.4.bias", (cout,)
    if cin != cout:
        yield pre + "res_conv.weight", (cout, cin, 1, 1)
        yield pre + "res_conv.bias", (cout,)


def _attn(pre, c, linear):
    """Residual(PreNorm(c, [Linear]Attention(c))) parameters (unet_model.py:126-187)."""
    hid = ATTN_HEADS * ATTN_DIM_HEAD
    yield pre + "fn.fn.to_qkv.weight", (hid * 3, c, 1, 1)
    if linear:
        yield pre + "fn.fn.to_out.0.weight", (c, hid, 1, 1)
        yield pre + "fn.fn.to_out.0.bias", (c,)
        yield pre + "fn.fn.to_out.1.weight", (c,)
        yield pre + "fn.fn.to_out.1.bias", (c,)
    else:
        yield pre + "fn.fn.to_out.weight", (c, hid, 1, 1)
        yield pre + "fn.fn.to_out.bias", (c,)
    yield pre + "fn.norm.weight", (c,)
    yield pre + "fn.norm.bias", (c,)


def unet_entries(dim, channels=3, dim_mults=(1, 2, 4, 8), init_dim=None, out_dim=None, with_time_emb=True,
                 convnext_mult=2):
    """(name, shape) in registration order of reference Unet.__init__ (unet_model.py:189-273)."""
    init_dim, dims, in_out, out_dim = unet_dims(dim, channels, dim_mults, init_dim, out_dim)
    td = dim * 4 if with_time_emb else 0
    yield "init_conv.weight", (init_dim, channels, 7, 7)
    yield "init_conv.bias", (init_dim,)
    if td:
        yield "time_mlp.1.weight", (td, dim)
        yield "time_mlp.1.bias", (td,)
        yield "time_mlp.3.weight", (td, td)
        yield "time_mlp.3.bias", (td,)
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        yield from _convnext(f"downs.{i}.0.", ci, co, td, convnext_mult)
        yield from _convnext(f"downs.{i}.1.", co, co, td, convnext_mult)
        yield from _attn(f"downs.{i}.2.", co, True)
        if i < n - 1:
            yield f"downs.{i}.3.weight", (co, co, 4, 4)
            yield f"downs.{i}.3.bias", (co,)
    mid = dims[-1]
    yield from _convnext("mid_block1.", mid, mid, td, convnext_mult)
    yield from _attn("mid_attn.", mid, False)
    yield from _convnext("mid_block2.", mid, mid, td, convnext_mult)
    for i, (ci, co) in enumerate(reversed(in_out[1:])):
        yield from _convnext(f"ups.{i}.0.", co * 2, ci, td, convnext_mult)
        yield from _convnext(f"ups.{i}.1.", ci, ci, td, convnext_mult)
        yield from _attn(f"ups.{i}.2.", ci, True)
        yield f"ups.{i}.3.weight", (ci, ci, 4, 4)          # ConvTranspose2d: (in, out, kh, kw)
        yield f"ups.{i}.3.bias", (ci,)
    yield from _convnext("final_conv.0.", dim, dim, 0, convnext_mult)
    yield "final_conv.1.weight", (out_dim, dim, 1, 1)
    yield "final_conv.1.bias", (out_dim,)


def unet_state_shapes(**kw):
    return dict(unet_entries(**kw))


def _bn(pre, c):
    yield pre + "weight", (c,)
    yield pre + "bias", (c,)
    yield pre + "running_mean", (c,)
    yield pre + "running_var", (c,)
    yield pre + "num_batches_tracked", ()


def _rblock(pre, cin, cmid, cout):
    """RBlock / RBlockN parameters (h_vae_model_copy.py:17-26, 355-364); size_conv always exists."""
    yield pre + "conv.0.weight", (cmid, cin, 3, 3)
    yield from _bn(pre + "conv.1.", cmid)
    yield pre + "conv.3.weight", (cout, cmid, 3, 3)
    yield from _bn(pre + "conv.4.", cout)
    yield pre + "size_conv.weight", (cout, cin, 1, 1)


def resvae_geometry(enc_channel_list, size_in):
    """(init_size, c_last, size_z_lin) as ResVAE.__init__ derives them (h_vae_model_copy.py:105-113)."""
    init_size = size_in
    for c in enc_channel_list:
        init_size //= c[3]
    c_last = enc_channel_list[-1][2]
    return init_size, c_last, init_size * init_size * c_last


def resvae_entries(variant, enc_channel_list, dec_channel_list, size_in=64, size_z=64, img_ch=3):
    """ResVAE / ResAE / ResVAEN / ResAEN state_dict (h_vae_model_copy.py:91-174, 457-588)."""
    enc_half = (size_in // 2)
    for c in enc_channel_list:
        enc_half //= c[3]
    enc_lin = enc_half * enc_half * (enc_channel_list[-1][2] // 2)
    c0 = enc_channel_list[0][0]
    yield "enc.ch_enc.0.weight", (c0, img_ch, 5, 5)
    yield "enc.ch_enc.0.bias", (c0,)
    yield from _bn("enc.ch_enc.1.", c0)
    for i, (ci, cm, co, _r) in enumerate(enc_channel_list):
        yield from _rblock(f"enc.r_blocks.{i}.", ci, cm, co)
    yield "enc.mu_lin.weight", (size_z, enc_lin)
    yield "enc.mu_lin.bias", (size_z,)
    yield "enc.logvar_lin.weight", (size_z, enc_lin)
    yield "enc.logvar_lin.bias", (size_z,)
    for i, (ci, cm, co, _r) in enumerate(dec_channel_list):
        yield from _rblock(f"dec.r_blocks.{i}.", ci, cm, co)
    c = dec_channel_list[-1][2]
    yield from _rblock("dec.ch_dec.0.", c, c, c)
    yield "dec.ch_dec.1.weight", (img_ch, c, 5, 5)
    yield "dec.ch_dec.1.bias", (img_ch,)
    _, _, zl = resvae_geometry(enc_channel_list, size_in)
    yield "z_lin.weight", (zl, size_z)
    yield "z_lin.bias", (zl,)


def resvae_state_shapes(variant, enc_channel_list, dec_channel_list, size_in=64, size_z=64, img_ch=3):
    return dict(resvae_entries(variant, enc_channel_list, dec_channel_list, size_in, size_z, img_ch))


ATTR_DEC_WIDTHS = (512, 512, 512, 256, 128)
ATTR_ENC_WIDTHS = (128, 256, 512, 512, 512)


def attr_entries(size_z=64, att_size=18, ae=False):
    """CelebAAttrNewBN / CelebAAttrNewBNAE state_dict (h_vae_model.py:712-757, 839-882)."""
    prev = att_size
    for i, w in enumerate(ATTR_ENC_WIDTHS):
        yield f"enc_net.{3 * i}.weight", (w, prev)
        yield f"enc_net.{3 * i}.bias", (w,)
        yield from _bn(f"enc_net.{3 * i + 1}.", w)
        prev = w
    yield "mu_lin.weight", (size_z, prev)
    yield "mu_lin.bias", (size_z,)
    if not ae:
        yield "logvar_lin.weight", (size_z, prev)
        yield "logvar_lin.bias", (size_z,)
    prev = size_z
    for i, w in enumerate(ATTR_DEC_WIDTHS):
        yield f"dec_net.{3 * i}.weight", (w, prev)
        yield f"dec_net.{3 * i}.bias", (w,)
        yield from _bn(f"dec_net.{3 * i + 1}.", w)
        prev = w
    yield "dec_net.15.weight", (att_size, prev)
    yield "dec_net.15.bias", (att_size,)


def attr_state_shapes(size_z=64, att_size=18, ae=False):
    return dict(attr_entries(size_z, att_size, ae))


# --------------------------------------------------------------------------------------------------
# unet_openai.UNetModel (reference unet_openai.py:361-575), the conditional CelebA score net (config 5)
# --------------------------------------------------------------------------------------------------
def unet_openai_structure(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions=(),
                          channel_mult=(1, 2, 4, 8)):
    """Block list mirroring UNetModel.__init__: returns (input_blocks, middle, output_blocks) where every block is
    a list of layers: ("conv", cin, cout) | ("res", cin, cout) | ("attn", ch) | ("down", ch) | ("up", ch)."""
    inputs = [[("conv", in_channels, model_channels)]]
    chans = [model_channels]
    ch, ds = model_channels, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [("res", ch, mult * model_channels)]
            ch = mult * model_channels
            if ds in attention_resolutions:
                layers.append(("attn", ch))
            inputs.append(layers)
            chans.append(ch)
        if level != len(channel_mult) - 1:
            inputs.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("attn", ch), ("res", ch, ch)]
    outputs = []
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            layers = [("res", ch + chans.pop(), model_channels * mult)]
            ch = model_channels * mult
            if ds in attention_resolutions:
                layers.append(("attn", ch))
            if level and i == num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def _oa_layer(pre, layer, emb):
    kind = layer[0]
    if kind == "conv":
        yield pre + "weight", (layer[2], layer[1], 3, 3)
        yield pre + "bias", (layer[2],)
    elif kind == "res":
        _, cin, cout = layer
        yield pre + "in_layers.0.weight", (cin,)
        yield pre + "in_layers.0.bias", (cin,)
        yield pre + "in_layers.2.weight", (cout, cin, 3, 3)
        yield pre + "in_layers.2.bias", (cout,)
        yield pre + "emb_layers.1.weight", (cout, emb)
        yield pre + "emb_layers.1.bias", (cout,)
        yield pre + "out_layers.0.weight", (cout,)
        yield pre + "out_layers.0.bias", (cout,)
        yield pre + "out_layers.3.weight", (cout, cout, 3, 3)
        yield pre + "out_layers.3.bias", (cout,)
        if cin != cout:
            yield pre + "skip_connection.weight", (cout, cin, 1, 1)
            yield pre + "skip_connection.bias", (cout,)
    elif kind == "attn":
        ch = layer[1]
        yield pre + "norm.weight", (ch,)
        yield pre + "norm.bias", (ch,)
        yield pre + "qkv.weight", (3 * ch, ch, 1)
        yield pre + "qkv.bias", (3 * ch,)
        yield pre + "proj_out.weight", (ch, ch, 1)
        yield pre + "proj_out.bias", (ch,)
    elif kind == "down":
        yield pre + "op.weight", (layer[1], layer[1], 3, 3)
        yield pre + "op.bias", (layer[1],)
    elif kind == "up":
        yield pre + "conv.weight", (layer[1], layer[1], 3, 3)
        yield pre + "conv.bias", (layer[1],)


def unet_openai_entries(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions=(),
                        channel_mult=(1, 2, 4, 8), z_dim=None, use_z=False):
    emb = model_channels * 4
    yield "time_embed.0.weight", (emb, model_channels)
    yield "time_embed.0.bias", (emb,)
    yield "time_embed.2.weight", (emb, emb)
    yield "time_embed.2.bias", (emb,)
    if use_z:
        yield "proj.0.weight", (emb, z_dim)
        yield "proj.0.bias", (emb,)
        yield "proj.2.weight", (emb, emb)
        yield "proj.2.bias", (emb,)
    inputs, middle, outputs = unet_openai_structure(in_channels, model_channels, out_channels, num_res_blocks,
                                                    attention_resolutions, channel_mult)
    for bi, layers in enumerate(inputs):
        for li, layer in enumerate(layers):
            yield from _oa_layer(f"input_blocks.{bi}.{li}.", layer, emb)
    for li, layer in enumerate(middle):
        yield from _oa_layer(f"middle_block.{li}.", layer, emb)
    for bi, layers in enumerate(outputs):
        for li, layer in enumerate(layers):
            yield from _oa_layer(f"output_blocks.{bi}.{li}.", layer, emb)
    yield "out.0.weight", (model_channels,)
    yield "out.0.bias", (model_channels,)
    yield "out.2.weight", (out_channels, model_channels, 3, 3)
    yield "out.2.bias", (out_channels,)


def unet_openai_state_shapes(**kw):
    return dict(unet_openai_entries(**kw))


_MLP_MUL = (1, 2, 6, 6, 4, 2, 1, 1)


def mlp_entries(n_mod=5, size_z=128):
    """state_dict of lat_sm2_model.LSMPoly64_sm (lat_sm2_model.py:348-371): nn.Sequential 'layers', Linear at even slots."""
    d = n_mod * size_z
    out = []
    for i in range(7):
        out.append((f"layers.{2 * i}.weight", (d * _MLP_MUL[i + 1], d * _MLP_MUL[i])))
        out.append((f"layers.{2 * i}.bias", (d * _MLP_MUL[i + 1],)))
    return out


def mlp_state_shapes(n_mod=5, size_z=128):
    return dict(mlp_entries(n_mod, size_z))


def clg_entries(variant, n_mod=2, size_z=64, n_class=1):
    """state_dict of lat_sm2_model.ClwithTime2 (variant 2) / ClwithTime3 (variant 3), this package's definition of the guidance
    MLP the reference imports but no longer ships; mirrors csrc/guidance.hip clg_tensor_specs."""
    d0 = n_mod * size_z + 128 + (64 if variant == 3 else 0)
    out = [("time_mlp.0.weight", (128, 64)), ("time_mlp.0.bias", (128,))]
    if variant == 3:
        out.append(("mod_emb.weight", (10, 32)))
    out += [("layers.0.weight", (512, d0)), ("layers.0.bias", (512,)), ("layers.2.weight", (256, 512)), ("layers.2.bias", (256,)),
            ("layers.4.weight", (n_class, 256)), ("layers.4.bias", (n_class,))]
    return out


def _proj_head(d):
    yield "proj.projection.weight", (d, d)
    yield "proj.projection.bias", (d,)
    yield "proj.fc.weight", (d, d)
    yield "proj.fc.bias", (d,)
    yield "proj.layer_norm.weight", (d,)
    yield "proj.layer_norm.bias", (d,)


CEL_IMG_REP_ENC = [(64, 128, 128, 2), (128, 256, 256, 2), (256, 512, 512, 2)]
CEL_MASK_REP_ENC = [(64, 128, 128, 4), (128, 256, 256, 4)]


def cel_rep_entries(enc_channel_list, size_z=512, img_ch=3, size_in=128):
    """CelImgRep / CelMaskRep state_dict (h_vae_model.py:1040-1072): ResEncoderN under `enc.` + ProjectionHead under `proj.`."""
    half = size_in // 2
    for c in enc_channel_list:
        half //= c[3]
    lin = half * half * (enc_channel_list[-1][2] // 2)
    c0 = enc_channel_list[0][0]
    yield "enc.ch_enc.0.weight", (c0, img_ch, 5, 5)
    yield "enc.ch_enc.0.bias", (c0,)
    yield from _bn("enc.ch_enc.1.", c0)
    for i, (ci, cm, co, _r) in enumerate(enc_channel_list):
        yield from _rblock(f"enc.r_blocks.{i}.", ci, cm, co)
    yield "enc.mu_lin.weight", (size_z, lin)
    yield "enc.mu_lin.bias", (size_z,)
    yield "enc.logvar_lin.weight", (size_z, lin)
    yield "enc.logvar_lin.bias", (size_z,)
    yield from _proj_head(size_z)


def cel_attr_rep_entries(att_size=18, size_z=512):
    """CelAttrRep state_dict (h_vae_model.py:987-1002): CelAttrEnc under `enc.` + ProjectionHead under `proj.`."""
    prev = att_size
    for i, w in enumerate(ATTR_ENC_WIDTHS):
        yield f"enc.enc_net.{3 * i}.weight", (w, prev)
        yield f"enc.enc_net.{3 * i}.bias", (w,)
        yield from _bn(f"enc.enc_net.{3 * i + 1}.", w)
        prev = w
    yield "enc.mu_lin.weight", (size_z, prev)
    yield "enc.mu_lin.bias", (size_z,)
    yield "enc.logvar_lin.weight", (size_z, prev)
    yield "enc.logvar_lin.bias", (size_z,)
    yield from _proj_head(size_z)
