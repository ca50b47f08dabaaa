#!/usr/bin/env python3
"""Generate the golden fixtures from the REAL reference modules (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference tree (/root/reference) is imported unmodified, with inert stub modules for the
`jax` / `torchvision` imports it never uses on the sampling path.  All
weights come from the closed-form filler (score_based_multimodal_autoencoder_amd/synth.py), so the
fixtures hold only inputs and expected outputs (data), never reference source.  The fixtures travel;
the reference does not.
"""
import os
import sys
import types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _n in ("jax", "jax.numpy", "torchvision", "torchvision.models"):
    sys.modules.setdefault(_n, types.ModuleType(_n))

import torch  # noqa: E402
import sde_helper2 as S  # noqa: E402  (reference)
from unet_model import Unet  # noqa: E402  (reference)
from h_vae_model_copy import ResVAE, ResAE, ResVAEN, ResAEN  # noqa: E402  (reference)
from h_vae_model import CelebAAttrNewBN, CelebAAttrNewBNAE  # noqa: E402  (reference)
from score_based_multimodal_autoencoder_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_grad_enabled(False)

POLY_ENC = [(64, 64, 64, 2), (64, 128, 128, 2), (128, 256, 256, 2)]
POLY_DEC = [(256, 128, 128, 2), (128, 128, 64, 2), (64, 64, 64, 2)]
CEL_IMG_ENC = [(64, 128, 128, 2), (128, 256, 256, 2), (256, 512, 512, 2)]
CEL_IMG_DEC = [(512, 512, 256, 2), (256, 256, 128, 2), (128, 128, 64, 2)]
CEL_MASK_ENC = [(64, 128, 128, 4), (128, 256, 256, 4)]
CEL_MASK_DEC = [(256, 256, 128, 4), (128, 128, 64, 4)]


def inp(name, shape, scale=1.0):
    return (synth.noise(0xC0FFEE, abs(hash_name(name)), shape) * scale).astype(np.float32)


def hash_name(name):
    import zlib
    return zlib.crc32(name.encode())


def save(fname, **arrays):
    path = os.path.join(OUT, fname)
    np.savez_compressed(path, **arrays)
    print(f"wrote {fname}: {len(arrays)} arrays [{os.path.getsize(path) / 1024:.1f} KiB]")


# ---------------------------------------------------------------------------------------------
def unet_cases():
    cases = {
        "tiny": dict(dim=8, channels=2, dim_mults=(1, 2, 2, 2), B=3, hw=(8, 8)),
        "tiny2lvl": dict(dim=16, channels=3, dim_mults=(1, 2), B=2, hw=(4, 4)),
        "P2": dict(dim=64, channels=2, dim_mults=(1, 2, 2, 2), B=2, hw=(8, 8)),
        "P5": dict(dim=64, channels=5, dim_mults=(1, 2, 2, 2), B=2, hw=(8, 8)),
        "P10": dict(dim=128, channels=10, dim_mults=(1, 2, 2, 2), B=1, hw=(8, 8)),
        "C3": dict(dim=256, channels=3, dim_mults=(1, 2, 2, 2, 2), B=1, hw=(16, 16)),
    }
    out = {}
    for name, c in cases.items():
        m = Unet(dim=c["dim"], channels=c["channels"], dim_mults=c["dim_mults"], with_time_emb=True).eval()
        synth.fill_module_(m)
        x = inp(f"unet.{name}.x", (c["B"], c["channels"]) + c["hw"])
        t = np.array([1.0, 0.5, 1e-3], dtype=np.float32)[: c["B"]]
        y = m(torch.from_numpy(x), torch.from_numpy(t)).numpy()
        out[f"{name}_x"], out[f"{name}_t"], out[f"{name}_y"] = x, t, y
        if name == "tiny":
            # integer "sigma index" time as the annealed-Langevin callers pass it (fid_upd10.py:86,96)
            ti = np.array([0, 7, 199], dtype=np.int64)
            out["tiny_ti"] = ti
            out["tiny_yi"] = m(torch.from_numpy(x), torch.from_numpy(ti)).numpy()
    save("unet_forward.npz", **out)


def sde_cases():
    out = {}
    x = inp("sde.x", (4, 2, 8, 8))
    score = inp("sde.score", (4, 2, 8, 8))
    noise = inp("sde.noise", (4, 2, 8, 8))
    for kind, cls, (b0, b1) in (("vp", S.VPSDE, (0.1, 5.0)), ("vp", S.VPSDE, (1.0, 5.0)), ("vp", S.VPSDE, (0.1, 20.0)),
                                ("subvp", S.subVPSDE, (0.1, 20.0)), ("ve", S.VESDE, (0.01, 50.0))):
        for N in (100, 1000):
            sde = cls(b0, b1, N)
            tag = f"{kind}_{b0}_{b1}_{N}"
            ts = torch.linspace(sde.T, 1e-3, N)
            out[tag + "_ts"] = ts.numpy()
            drift, diff = sde.sde(torch.ones(N, 1, 1, 1), ts)
            out[tag + "_dcoef"] = drift.reshape(N).numpy()
            out[tag + "_g"] = diff.numpy()
            mean, std = sde.marginal_prob(torch.ones(N, 1, 1, 1), ts)
            out[tag + "_mpmean"] = mean.reshape(N).numpy()
            out[tag + "_mpstd"] = std.numpy()
            if kind == "vp":      # subVPSDE has no .alphas in the reference: its corrector raises AttributeError
                idx = (ts * (N - 1) / sde.T).long()
                out[tag + "_alpha"] = sde.alphas[idx].numpy()
                out[tag + "_aidx"] = idx.numpy()
            # one predictor and one corrector step with fixed (x, score, noise)
            for ti in (0, N // 2, N - 1):
                vt = torch.ones(4) * ts[ti]
                orig = torch.randn_like
                torch.randn_like = lambda _x: torch.from_numpy(noise)
                try:
                    xn, xm = S.em_predictor(torch.from_numpy(x), vt, lambda a, b: torch.from_numpy(score), sde)
                    if kind != "subvp":
                        cn, cm = S.corrector(torch.from_numpy(x), vt, lambda a, b: torch.from_numpy(score).clone(), sde, 1, 0.16)
                        out[f"{tag}_co{ti}_x"], out[f"{tag}_co{ti}_mean"] = cn.numpy(), cm.numpy()
                finally:
                    torch.randn_like = orig
                out[f"{tag}_em{ti}_x"], out[f"{tag}_em{ti}_mean"] = xn.numpy(), xm.numpy()
    out["x"], out["score"], out["noise"] = x, score, noise
    save("sde_steps.npz", **out)


def trajectory_cases():
    """Config-1 shape: B=16, 2-mod, N=100, Unet(64, ch=2) with filler weights, injected noise."""
    m = Unet(dim=64, channels=2, dim_mults=(1, 2, 2, 2), with_time_emb=True).eval()
    synth.fill_module_(m)
    shape = (16, 2, 8, 8)
    out = {}
    for order in ("cp", "pc"):
        for pc in (False, True):
            for kind, mk in (("vp", lambda: S.VPSDE(0.1, 5.0, 100)), ("ve", lambda: S.VESDE(0.01, 5.0, 100))):
                if kind == "ve" and (order == "pc" or not pc):
                    continue
                sde = mk()
                seed = 1234
                counter = [0]

                def draw(_x=None):
                    counter[0] += 1
                    return torch.from_numpy(synth.noise(seed, counter[0], shape))

                x = torch.from_numpy(synth.noise(seed, 0, shape))
                if kind == "ve":
                    x = x * sde.sigma_max
                ts = torch.linspace(sde.T, 1e-3, sde.N)
                orig = torch.randn_like
                torch.randn_like = draw
                ck = {}
                try:
                    for i in range(sde.N):
                        vt = torch.ones(shape[0]) * ts[i]
                        if order == "cp":          # sde_helper2.uncond_sampler order
                            if pc:
                                x, xm = S.corrector(x, vt, m, sde, 1, 0.16)
                            x, xm = S.em_predictor(x, vt, m, sde)
                        else:                      # train_poly_unet_cont.uncond_sampler order
                            x, xm = S.em_predictor(x, vt, m, sde)
                            if pc:
                                x, xm = S.corrector(x, vt, m, sde, 1, 0.16)
                        if i in (0, 1, 10, 50, 99):
                            ck[i] = (x.numpy().copy(), xm.numpy().copy())
                finally:
                    torch.randn_like = orig
                tag = f"{kind}_{order}_pc{int(pc)}"
                for i, (a, b) in ck.items():
                    out[f"{tag}_x{i}"] = a
                    out[f"{tag}_mean{i}"] = b
                print(tag, "draws", counter[0], "final |x_mean| max", float(np.abs(ck[99][1]).max()))
    # the unmodified reference entry point with torch's own RNG stream, order 'cp', for the record
    torch.manual_seed(7)
    out["ref_uncond_seed7"] = S.uncond_sampler((4, 2, 8, 8), m, "cpu", S.VPSDE(0.1, 5.0, 20)).numpy()
    torch.manual_seed(7)
    out["ref_uncond_seed7_prior"] = torch.randn(4, 2, 8, 8).numpy()
    draws = []
    for _ in range(20):
        draws.append(torch.randn(4, 2, 8, 8).numpy())
    out["ref_uncond_seed7_noise"] = np.stack(draws)
    save("trajectories.npz", **out)


def decoder_cases():
    out = {}
    for name, cls in (("ResVAE", ResVAE), ("ResAE", ResAE)):
        m = cls(POLY_ENC, POLY_DEC, 32, 64, 3).eval()
        synth.fill_module_(m)
        z = inp(f"dec.{name}.z", (3, 64))
        out[f"{name}_z"], out[f"{name}_y"] = z, m.decoder(torch.from_numpy(z)).numpy()
        ximg = inp(f"enc.{name}.x", (2, 3, 32, 32), 0.5) + 0.5
        enc = m.encoder(torch.from_numpy(ximg))
        out[f"{name}_encx"] = ximg
        out[f"{name}_mu"] = (enc[0] if isinstance(enc, tuple) else enc).numpy()
        if isinstance(enc, tuple):
            out[f"{name}_logvar"] = enc[1].numpy()
    for name, cls, enc, dec, ch in (("ResVAEN_img", ResVAEN, CEL_IMG_ENC, CEL_IMG_DEC, 3),
                                    ("ResAEN_mask", ResAEN, CEL_MASK_ENC, CEL_MASK_DEC, 1)):
        m = cls(enc, dec, 128, 256, ch).eval()
        synth.fill_module_(m)
        z = inp(f"dec.{name}.z", (1, 256))
        y = m.decoder(torch.from_numpy(z)).numpy()
        out[f"{name}_z"] = z
        out[f"{name}_y"] = y
    for name, cls in (("AttrBN", CelebAAttrNewBN), ("AttrBNAE", CelebAAttrNewBNAE)):
        m = cls(256).eval()
        synth.fill_module_(m)
        z = inp(f"dec.{name}.z", (4, 256))
        out[f"{name}_z"], out[f"{name}_y"] = z, m.decoder(torch.from_numpy(z)).numpy()
    save("decoders.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["unet", "sde", "traj", "dec"]
    if "unet" in which:
        unet_cases()
    if "sde" in which:
        sde_cases()
    if "traj" in which:
        trajectory_cases()
    if "dec" in which:
        decoder_cases()


def unet_openai_cases():
    """unet_openai.UNetModel: two toy widths (one with attention inside the blocks) and the full config-5 net."""
    from unet_openai import UNetModel
    from score_based_multimodal_autoencoder_amd import arch
    out = {}
    for name, kw, B, hw in [
            ("oa_tiny", dict(in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(), channel_mult=(1, 2), z_dim=16, use_z=True), 2, 8),
            ("oa_attn", dict(in_channels=2, model_channels=32, out_channels=2, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2), z_dim=16, use_z=True), 2, 4),
            ("oa_C3c", dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(), channel_mult=(1, 2, 4, 8), z_dim=512, use_z=True), 1, 16)]:
        m = UNetModel(kw["in_channels"], kw["model_channels"], kw["out_channels"], kw["num_res_blocks"], kw["attention_resolutions"], 0.1,
                      kw["channel_mult"], num_heads=1, use_z=True, z_dim=kw["z_dim"]).eval()
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(s) for k, s in arch.unet_openai_entries(**kw)}
        synth.fill_module_(m)          # also re-randomises the zero_module convolutions
        k = len(out) // 5
        x = synth.noise(77, 5 * k, (B, kw["in_channels"], hw, hw)); t = np.array([0.7, 1e-3], np.float32)[:B]
        z = synth.noise(78, 5 * k, (B, kw["z_dim"]))
        out.update({name + "_x": x, name + "_t": t, name + "_z": z,
                    name + "_y": m(torch.from_numpy(x), torch.from_numpy(t), z=torch.from_numpy(z)).numpy(),
                    name + "_y_noz": m(torch.from_numpy(x), torch.from_numpy(t)).numpy()})
    save("unet_openai.npz", **out)


if __name__ == "__main__" and "openai" in sys.argv[1:]:
    unet_openai_cases()
