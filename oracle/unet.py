"""Oracle restatement of the ConvNeXt score UNet (reference unet_model.py:189-323).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works on a ``state_dict`` of
numpy arrays with the reference's key names.
"""
import math
import numpy as np
from . import np_ops as O

F32 = np.float32


def unet_dims(dim, channels, dim_mults, init_dim=None, out_dim=None):
    """unet_model.py:206-212,269."""
    init_dim = init_dim if init_dim is not None else dim // 3 * 2
    dims = [init_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    return init_dim, dims, in_out, (out_dim if out_dim is not None else channels)


def sinusoidal_embedding(time, dim):
    """unet_model.py:40-47: [sin, cos], denominator half_dim-1, raw time (no x1000)."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = np.exp(np.arange(half, dtype=F32) * F32(-k)).astype(F32)
    arg = (np.asarray(time, dtype=F32)[:, None] * freqs[None, :]).astype(F32)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(F32)


def time_mlp(sd, time, dim):
    """unet_model.py:222-227: sinusoid -> Linear -> GELU -> Linear."""
    e = sinusoidal_embedding(time, dim)
    e = O.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = O.gelu(e)
    return O.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def convnext_block(sd, pre, x, t_emb):
    """unet_model.py:115-124."""
    C = x.shape[1]
    h = O.conv2d(x, sd[pre + "ds_conv.weight"], sd[pre + "ds_conv.bias"], pad=3, groups=C)
    if (pre + "mlp.1.weight") in sd and t_emb is not None:
        cond = O.linear(O.gelu(t_emb), sd[pre + "mlp.1.weight"], sd[pre + "mlp.1.bias"])
        h = h + cond[:, :, None, None]
    h = O.group_norm(h, 1, sd[pre + "net.0.weight"], sd[pre + "net.0.bias"])
    h = O.conv2d(h, sd[pre + "net.1.weight"], sd[pre + "net.1.bias"], pad=1)
    h = O.gelu(h)
    h = O.group_norm(h, 1, sd[pre + "net.3.weight"], sd[pre + "net.3.bias"])
    h = O.conv2d(h, sd[pre + "net.4.weight"], sd[pre + "net.4.bias"], pad=1)
    if (pre + "res_conv.weight") in sd:
        res = O.conv2d(x, sd[pre + "res_conv.weight"], sd[pre + "res_conv.bias"])
    else:
        res = x
    return (h + res).astype(F32)


def _split_heads(qkv, heads):
    B, C3, H, W = qkv.shape
    hid = C3 // 3
    q, k, v = qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:]
    rs = lambda t: t.reshape(B, heads, hid // heads, H * W)
    return rs(q), rs(k), rs(v)


def linear_attention(sd, pre, x, heads=4, dim_head=32):
    """unet_model.py:162-177.  q softmax over d (dim=-2), k softmax over n (dim=-1)."""
    B, C, H, W = x.shape
    qkv = O.conv2d(x, sd[pre + "to_qkv.weight"])
    q, k, v = _split_heads(qkv, heads)
    q = O.softmax(q, axis=-2)
    k = O.softmax(k, axis=-1)
    q = q * F32(dim_head ** -0.5)
    context = np.einsum("bhdn,bhen->bhde", k, v).astype(F32)
    out = np.einsum("bhde,bhdn->bhen", context, q).astype(F32)
    out = out.reshape(B, heads * dim_head, H, W)
    out = O.conv2d(out, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"])
    return O.group_norm(out, 1, sd[pre + "to_out.1.weight"], sd[pre + "to_out.1.bias"])


def attention(sd, pre, x, heads=4, dim_head=32):
    """unet_model.py:135-149."""
    B, C, H, W = x.shape
    qkv = O.conv2d(x, sd[pre + "to_qkv.weight"])
    q, k, v = _split_heads(qkv, heads)
    q = q * F32(dim_head ** -0.5)
    sim = np.einsum("bhdi,bhdj->bhij", q, k).astype(F32)
    attn = O.softmax(sim, axis=-1)                      # amax subtraction is inside softmax
    out = np.einsum("bhij,bhdj->bhid", attn, v).astype(F32)      # (B,h,n,d)
    out = out.transpose(0, 1, 3, 2).reshape(B, heads * dim_head, H, W)
    return O.conv2d(out, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def residual_prenorm(sd, pre, x, fn):
    """Residual(PreNorm(dim, fn)) -- unet_model.py:21-27,179-187."""
    xn = O.group_norm(x, 1, sd[pre + "fn.norm.weight"], sd[pre + "fn.norm.bias"])
    return (fn(sd, pre + "fn.fn.", xn) + x).astype(F32)


def _pad_pow2(n):
    return int(2 ** math.ceil(math.log2(n))) - n


def unet_forward(sd, x, time, dim, channels, dim_mults=(1, 2, 4, 8), init_dim=None, out_dim=None):
    """unet_model.py:275-323."""
    x = np.ascontiguousarray(x, dtype=F32)
    _, dims, in_out, _ = unet_dims(dim, channels, dim_mults, init_dim, out_dim)
    pw, ph = _pad_pow2(x.shape[-1]) // 2, _pad_pow2(x.shape[-2]) // 2      # :276-284
    if pw or ph:
        x = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    x = O.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], pad=3)
    t = time_mlp(sd, time, dim) if "time_mlp.1.weight" in sd else None
    hs = []
    n_res = len(in_out)
    for i in range(n_res):
        x = convnext_block(sd, f"downs.{i}.0.", x, t)
        x = convnext_block(sd, f"downs.{i}.1.", x, t)
        x = residual_prenorm(sd, f"downs.{i}.2.", x, linear_attention)
        hs.append(x)
        if i < n_res - 1:
            x = O.conv2d(x, sd[f"downs.{i}.3.weight"], sd[f"downs.{i}.3.bias"], stride=2, pad=1)
    x = convnext_block(sd, "mid_block1.", x, t)
    x = residual_prenorm(sd, "mid_attn.", x, attention)
    x = convnext_block(sd, "mid_block2.", x, t)
    for i in range(n_res - 1):
        x = np.concatenate([x, hs.pop()], axis=1)
        x = convnext_block(sd, f"ups.{i}.0.", x, t)
        x = convnext_block(sd, f"ups.{i}.1.", x, t)
        x = residual_prenorm(sd, f"ups.{i}.2.", x, linear_attention)
        # every ups level has an Upsample: `is_last` is never true there (unet_model.py:253-262)
        x = O.conv_transpose2d(x, sd[f"ups.{i}.3.weight"], sd[f"ups.{i}.3.bias"], stride=2, pad=1)
    x = convnext_block(sd, "final_conv.0.", x, None)
    x = O.conv2d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    if pw:
        x = x[..., pw:-pw]
    if ph:
        x = x[..., ph:-ph, :]
    return np.ascontiguousarray(x, dtype=F32)
