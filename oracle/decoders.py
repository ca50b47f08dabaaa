"""Oracle restatement of the per-modality decoders (reference h_vae_model_copy.py, h_vae_model.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Modules are in eval() mode on the hot path
(train_poly_unet_cont.py:294-295), so BatchNorm uses its running statistics.
"""
import numpy as np
from . import np_ops as O

F32 = np.float32


def _bn(sd, pre, x):
    return O.batch_norm_eval(x, sd[pre + "weight"], sd[pre + "bias"], sd[pre + "running_mean"], sd[pre + "running_var"])


def rblock(sd, pre, x, in_w, out_w, up_rate=None, act="lrelu", up_mode="nearest"):
    """RBlock.forward h_vae_model_copy.py:30-39 (act='lrelu') / RBlockN.forward :368-377 (act='gelu')."""
    f = (lambda v: O.leaky_relu(v, 0.2)) if act == "lrelu" else O.gelu
    h = O.conv2d(x, sd[pre + "conv.0.weight"], None, pad=1)
    h = f(_bn(sd, pre + "conv.1.", h))
    h = O.conv2d(h, sd[pre + "conv.3.weight"], None, pad=1)
    h = _bn(sd, pre + "conv.4.", h)
    if in_w != out_w:
        x = O.conv2d(x, sd[pre + "size_conv.weight"], None)
    h = f((x + h).astype(F32))
    if up_rate is not None:
        h = O.upsample_nearest(h, up_rate) if up_mode == "nearest" else O.upsample_bilinear(h, up_rate)
    return h


def res_decoder(sd, z, enc_channel_list, dec_channel_list, size_in, variant="ResVAE"):
    """ResVAE/ResAE.decoder h_vae_model_copy.py:123-126,166-169 (+ResDecoder :73-89);
    ResVAEN/ResAEN.decoder :489-492,576-579 (+ResDecoderN :411-428: GELU, bilinear, Sigmoid)."""
    new = variant in ("ResVAEN", "ResAEN")
    init_size = size_in
    for c in enc_channel_list:
        init_size //= c[3]
    c_last = enc_channel_list[-1][2]
    h = O.relu(O.linear(z, sd["z_lin.weight"], sd["z_lin.bias"]))
    h = h.reshape(h.shape[0], c_last, init_size, init_size)
    for i, (cin, cmid, cout, up) in enumerate(dec_channel_list):
        h = rblock(sd, f"dec.r_blocks.{i}.", h, cin, cout, up,
                   act="gelu" if new else "lrelu", up_mode="bilinear" if new else "nearest")
    c = dec_channel_list[-1][2]
    h = rblock(sd, "dec.ch_dec.0.", h, c, c, None, act="lrelu")           # plain RBlock in both variants
    h = O.conv2d(h, sd["dec.ch_dec.1.weight"], sd["dec.ch_dec.1.bias"], pad=2)
    return O.sigmoid(h) if new else h


def attr_decoder(sd, z):
    """CelebAAttrNewBN[AE].decoder h_vae_model.py:739-756,768-769 / :865-890: 5x(Linear,BN1d,ReLU) + Linear."""
    h = np.asarray(z, F32)
    for i in range(5):
        h = O.linear(h, sd[f"dec_net.{3 * i}.weight"], sd[f"dec_net.{3 * i}.bias"])
        h = O.relu(_bn(sd, f"dec_net.{3 * i + 1}.", h))
    return O.linear(h, sd["dec_net.15.weight"], sd["dec_net.15.bias"])


def attr_encoder(sd, x):
    """CelebAAttrNewBN[AE].encoder h_vae_model.py:718-736,756-759 / :845-863,884-887: 5x(Linear,BN1d,ReLU) + mu_lin (+ logvar_lin)."""
    h = np.asarray(x, F32)
    for i in range(5):
        h = O.linear(h, sd[f"enc_net.{3 * i}.weight"], sd[f"enc_net.{3 * i}.bias"])
        h = O.relu(_bn(sd, f"enc_net.{3 * i + 1}.", h))
    mu = O.linear(h, sd["mu_lin.weight"], sd["mu_lin.bias"])
    lv = O.linear(h, sd["logvar_lin.weight"], sd["logvar_lin.bias"]) if "logvar_lin.weight" in sd else None
    return mu, lv


def res_encoder(sd, x, enc_channel_list, variant="ResVAE"):
    """ResEncoder.forward h_vae_model_copy.py:63-71 / ResEncoderN :401-409.  Returns (mu, logvar)."""
    new = variant in ("ResVAEN", "ResAEN")
    h = O.conv2d(x, sd["enc.ch_enc.0.weight"], sd["enc.ch_enc.0.bias"], pad=2)
    h = O.leaky_relu(_bn(sd, "enc.ch_enc.1.", h), 0.1 if new else 0.2)
    h = O.avg_pool2d(h, 2)
    f = O.gelu if new else (lambda v: O.leaky_relu(v, 0.2))
    for i, (cin, cmid, cout, down) in enumerate(enc_channel_list):
        pre = f"enc.r_blocks.{i}."
        r = O.conv2d(h, sd[pre + "conv.0.weight"], None, pad=1)
        r = f(_bn(sd, pre + "conv.1.", r))
        r = O.conv2d(r, sd[pre + "conv.3.weight"], None, pad=1)
        r = _bn(sd, pre + "conv.4.", r)
        s = O.conv2d(h, sd[pre + "size_conv.weight"], None) if cin != cout else h
        h = f((s + r).astype(F32))
        if down is not None:
            h = O.avg_pool2d(h, down)
    half = h.shape[1] // 2
    mu, lv = h[:, :half], h[:, half:]
    B = h.shape[0]
    mu = O.linear(mu.reshape(B, -1), sd["enc.mu_lin.weight"], sd["enc.mu_lin.bias"])
    lv = O.linear(lv.reshape(B, -1), sd["enc.logvar_lin.weight"], sd["enc.logvar_lin.bias"])
    return mu, lv
