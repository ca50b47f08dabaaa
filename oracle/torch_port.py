"""CPU port of the path on PyTorch's own CPU operators (ATen / oneDNN) -- the same operator sequence
the reference executes when it runs on a CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used (a) as a second, independent check of the
numpy oracle and (b) as the `cpu_baseline` leg of bench.py, because it is what the reference's CPU
path costs on the host cores (the numpy oracle would flatter the GPU).  Restated from the reference
modules cited in oracle/unet.py, oracle/sde.py and oracle/decoders.py; works on a state_dict of
torch tensors with the reference's key names.
"""
import math

import torch
import torch.nn.functional as F


def _convnext(sd, pre, x, t_emb):
    h = F.conv2d(x, sd[pre + "ds_conv.weight"], sd[pre + "ds_conv.bias"], padding=3, groups=x.shape[1])
    if (pre + "mlp.1.weight") in sd and t_emb is not None:
        h = h + F.linear(F.gelu(t_emb), sd[pre + "mlp.1.weight"], sd[pre + "mlp.1.bias"])[:, :, None, None]
    h = F.group_norm(h, 1, sd[pre + "net.0.weight"], sd[pre + "net.0.bias"])
    h = F.gelu(F.conv2d(h, sd[pre + "net.1.weight"], sd[pre + "net.1.bias"], padding=1))
    h = F.group_norm(h, 1, sd[pre + "net.3.weight"], sd[pre + "net.3.bias"])
    h = F.conv2d(h, sd[pre + "net.4.weight"], sd[pre + "net.4.bias"], padding=1)
    res = F.conv2d(x, sd[pre + "res_conv.weight"], sd[pre + "res_conv.bias"]) if (pre + "res_conv.weight") in sd else x
    return h + res


def _attn(sd, pre, x, linear):
    B, C, H, W = x.shape
    xn = F.group_norm(x, 1, sd[pre + "fn.norm.weight"], sd[pre + "fn.norm.bias"])
    q, k, v = F.conv2d(xn, sd[pre + "fn.fn.to_qkv.weight"]).reshape(B, 3, 4, 32, H * W).unbind(1)
    if linear:
        q = q.softmax(dim=-2) * (32 ** -0.5)
        k = k.softmax(dim=-1)
        ctx = torch.einsum("bhdn,bhen->bhde", k, v)
        out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, 128, H, W)
        out = F.conv2d(out, sd[pre + "fn.fn.to_out.0.weight"], sd[pre + "fn.fn.to_out.0.bias"])
        out = F.group_norm(out, 1, sd[pre + "fn.fn.to_out.1.weight"], sd[pre + "fn.fn.to_out.1.bias"])
    else:
        sim = torch.einsum("bhdi,bhdj->bhij", q * (32 ** -0.5), k)
        attn = (sim - sim.amax(dim=-1, keepdim=True)).softmax(dim=-1)
        out = torch.einsum("bhij,bhdj->bhid", attn, v).permute(0, 1, 3, 2).reshape(B, 128, H, W)
        out = F.conv2d(out, sd[pre + "fn.fn.to_out.weight"], sd[pre + "fn.fn.to_out.bias"])
    return out + x


def unet_forward(sd, x, time, dim, n_levels):
    """reference unet_model.py:275-323 (power-of-two inputs)."""
    half = dim // 2
    freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = time[:, None] * freqs[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    t = F.linear(F.gelu(F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    hs = []
    for i in range(n_levels):
        x = _convnext(sd, f"downs.{i}.0.", x, t)
        x = _convnext(sd, f"downs.{i}.1.", x, t)
        x = _attn(sd, f"downs.{i}.2.", x, True)
        hs.append(x)
        if i < n_levels - 1:
            x = F.conv2d(x, sd[f"downs.{i}.3.weight"], sd[f"downs.{i}.3.bias"], stride=2, padding=1)
    x = _convnext(sd, "mid_block1.", x, t)
    x = _attn(sd, "mid_attn.", x, False)
    x = _convnext(sd, "mid_block2.", x, t)
    for i in range(n_levels - 1):
        x = torch.cat((x, hs.pop()), dim=1)
        x = _convnext(sd, f"ups.{i}.0.", x, t)
        x = _convnext(sd, f"ups.{i}.1.", x, t)
        x = _attn(sd, f"ups.{i}.2.", x, True)
        x = F.conv_transpose2d(x, sd[f"ups.{i}.3.weight"], sd[f"ups.{i}.3.bias"], stride=2, padding=1)
    x = _convnext(sd, "final_conv.0.", x, None)
    return F.conv2d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])


def vp_em_step(x, t, score, z, b0, b1, N):
    """reference sde_helper2.py:45-52 + VPSDE.sde :352-356."""
    beta_t = b0 + t * (b1 - b0)
    drift = -0.5 * beta_t[:, None, None, None] * x
    g = torch.sqrt(beta_t)
    drift = drift - g[:, None, None, None] ** 2 * score
    x_mean = x + drift * (-1.0 / N)
    return x_mean + g[:, None, None, None] * math.sqrt(1.0 / N) * z, x_mean


def _rblock(sd, pre, x, cin, cout, up):
    bn = lambda p, v: F.batch_norm(v, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False)
    h = F.leaky_relu(bn(pre + "conv.1.", F.conv2d(x, sd[pre + "conv.0.weight"], padding=1)), 0.2)
    h = bn(pre + "conv.4.", F.conv2d(h, sd[pre + "conv.3.weight"], padding=1))
    if cin != cout:
        x = F.conv2d(x, sd[pre + "size_conv.weight"])
    h = F.leaky_relu(x + h, 0.2)
    return F.interpolate(h, scale_factor=up) if up else h


def poly_decoder(sd, z, dec_channel_list, c_last=256, init=4):
    """ResVAE.decoder, reference h_vae_model_copy.py:123-126."""
    h = F.relu(F.linear(z, sd["z_lin.weight"], sd["z_lin.bias"])).view(z.shape[0], c_last, init, init)
    for i, (ci, cm, co, up) in enumerate(dec_channel_list):
        h = _rblock(sd, f"dec.r_blocks.{i}.", h, ci, co, up)
    c = dec_channel_list[-1][2]
    h = _rblock(sd, "dec.ch_dec.0.", h, c, c, None)
    return F.conv2d(h, sd["dec.ch_dec.1.weight"], sd["dec.ch_dec.1.bias"], padding=2)
