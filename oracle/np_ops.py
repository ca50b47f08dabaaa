"""Leaf ops of the oracle in plain numpy (fp32 data, NCHW).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each op restates the ATen
semantics the reference relies on; statistics are accumulated in float64 and
rounded once, which is at least as accurate as the reference's fp32 kernels.
"""
import numpy as np
from numpy.lib.stride_tricks import as_strided
from scipy.special import erf as _erf

F32 = np.float32


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


def _im2col(x, kh, kw, stride, pad):
    """x (B,C,H,W) -> cols (B, C*kh*kw, Ho*Wo) with zero padding."""
    B, C, H, W = x.shape
    if pad:
        xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), dtype=x.dtype)
        xp[:, :, pad:pad + H, pad:pad + W] = x
    else:
        xp = np.ascontiguousarray(x)
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    sb, sc, sh, sw = xp.strides
    v = as_strided(xp, shape=(B, C, kh, kw, Ho, Wo),
                   strides=(sb, sc, sh, sw, sh * stride, sw * stride), writeable=False)
    return v.reshape(B, C * kh * kw, Ho * Wo), Ho, Wo


def conv2d(x, w, b=None, stride=1, pad=0, groups=1):
    """torch.nn.functional.conv2d, zero padding.  w (Cout, Cin/groups, kh, kw)."""
    x = _f32(x)
    w = _f32(w)
    B, C, H, W = x.shape
    Cout, Cg, kh, kw = w.shape
    if groups == 1:
        cols, Ho, Wo = _im2col(x, kh, kw, stride, pad)
        out = np.matmul(w.reshape(Cout, -1)[None], cols)          # (B,Cout,HoWo)
        out = out.reshape(B, Cout, Ho, Wo)
    elif groups == C and Cg == 1 and Cout == C:
        # depthwise (unet_model.py:103 ds_conv)
        cols, Ho, Wo = _im2col(x, kh, kw, stride, pad)            # (B, C*kh*kw, HoWo)
        cols = cols.reshape(B, C, kh * kw, Ho * Wo)
        out = (cols * w.reshape(1, C, kh * kw, 1)).sum(axis=2, dtype=np.float64).reshape(B, C, Ho, Wo)
    else:
        raise NotImplementedError("groups")
    out = out.astype(F32, copy=False)
    if b is not None:
        out = out + _f32(b)[None, :, None, None]
    return out


def conv_transpose2d(x, w, b=None, stride=2, pad=1):
    """torch ConvTranspose2d (unet_model.py:30).  w (Cin, Cout, kh, kw).

    out[b,co,oy,ox] = sum_{ci,ky,kx} x[b,ci,iy,ix] * w[ci,co,ky,kx]  with oy = iy*stride - pad + ky.
    """
    x = _f32(x)
    w = _f32(w)
    B, Cin, H, W = x.shape
    _, Cout, kh, kw = w.shape
    Ho = (H - 1) * stride - 2 * pad + kh
    Wo = (W - 1) * stride - 2 * pad + kw
    full = np.zeros((B, Cout, (H - 1) * stride + kh, (W - 1) * stride + kw), dtype=np.float64)
    xr = x.transpose(0, 2, 3, 1).reshape(B * H * W, Cin)
    allk = np.matmul(xr, w.reshape(Cin, Cout * kh * kw)).reshape(B, H, W, Cout, kh, kw)   # one GEMM, all taps
    for ky in range(kh):
        for kx in range(kw):
            contrib = allk[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
            full[:, :, ky:ky + (H - 1) * stride + 1:stride, kx:kx + (W - 1) * stride + 1:stride] += contrib
    out = full[:, :, pad:pad + Ho, pad:pad + Wo].astype(F32)
    if b is not None:
        out = out + _f32(b)[None, :, None, None]
    return out


def linear(x, w, b=None):
    out = np.matmul(_f32(x), _f32(w).T)
    if b is not None:
        out = out + _f32(b)
    return out.astype(F32, copy=False)


def group_norm(x, groups, gamma, beta, eps=1e-5):
    """nn.GroupNorm: biased variance over (C/G, H, W) per sample and group."""
    x = _f32(x)
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1).astype(np.float64)
    mean = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)
    y = ((xg - mean) / np.sqrt(var + eps)).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    y = y * np.asarray(gamma, np.float64).reshape(shape) + np.asarray(beta, np.float64).reshape(shape)
    return y.astype(F32)


def batch_norm_eval(x, weight, bias, running_mean, running_var, eps=1e-5):
    """BatchNorm{1,2}d in eval mode (h_vae_model_copy.py:20,23; h_vae_model.py:741)."""
    x = _f32(x)
    shape = (1, -1) + (1,) * (x.ndim - 2)
    inv = 1.0 / np.sqrt(np.asarray(running_var, np.float64) + eps)
    y = (x.astype(np.float64) - np.asarray(running_mean, np.float64).reshape(shape)) * inv.reshape(shape)
    y = y * np.asarray(weight, np.float64).reshape(shape) + np.asarray(bias, np.float64).reshape(shape)
    return y.astype(F32)


def gelu(x):
    """nn.GELU() exact erf form (unet_model.py:98,108,225)."""
    x = _f32(x)
    return (0.5 * x * (1.0 + _erf(x / np.sqrt(2.0).astype(F32)))).astype(F32)


def silu(x):
    x = _f32(x)
    return (x / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def sigmoid(x):
    x = _f32(x)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def relu(x):
    return np.maximum(_f32(x), F32(0))


def leaky_relu(x, slope=0.2):
    x = _f32(x)
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def softmax(x, axis):
    x = _f32(x).astype(np.float64)
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def upsample_nearest(x, scale):
    """torch.nn.Upsample(scale_factor=s) default mode='nearest' (h_vae_model_copy.py:28)."""
    return np.repeat(np.repeat(_f32(x), scale, axis=2), scale, axis=3)


def upsample_bilinear(x, scale):
    """nn.Upsample(scale_factor=s, mode='bilinear'), align_corners=False (h_vae_model_copy.py:366).

    src = (dst + 0.5)/s - 0.5 clamped at 0; the upper neighbour is clamped to the last index.
    """
    x = _f32(x)
    B, C, H, W = x.shape

    def axis_weights(n_in):
        n_out = n_in * scale
        dst = np.arange(n_out, dtype=np.float64)
        src = np.maximum((dst + 0.5) / scale - 0.5, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0).astype(F32)
        l0 = (F32(1.0) - l1).astype(F32)
        return i0, i1, l0, l1

    y0, y1, wy0, wy1 = axis_weights(H)
    x0, x1, wx0, wx1 = axis_weights(W)
    rows = x[:, :, y0, :] * wy0[None, None, :, None] + x[:, :, y1, :] * wy1[None, None, :, None]
    out = rows[:, :, :, x0] * wx0[None, None, None, :] + rows[:, :, :, x1] * wx1[None, None, None, :]
    return out.astype(F32)


def avg_pool2d(x, k):
    x = _f32(x)
    B, C, H, W = x.shape
    return x.reshape(B, C, H // k, k, W // k, k).mean(axis=(3, 5)).astype(F32)
