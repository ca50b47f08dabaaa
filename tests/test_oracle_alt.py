"""Oracle vs the reference's own script functions (alternative samplers a15/a16, conditional loop f1)."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import decoders as OD
from oracle import langevin as OL
from oracle import sde as OS
from oracle import unet as OU
from score_based_multimodal_autoencoder_amd import arch, synth

TINY = dict(dim=8, dim_mults=(1, 2, 2, 2))


class Draws:
    """The golden generator's noise: draw k of a run is synth.noise(seed, k, shape)."""

    def __init__(self, seed, start=0):
        self.seed, self.k = seed, start

    def __call__(self, shape):
        a = synth.noise(self.seed, self.k, tuple(shape))
        self.k += 1
        return a


def tamed_score(sd, n_mod):
    def f(z, idx):
        u = OU.unet_forward(sd, z, idx, channels=n_mod, **TINY)
        return (np.float32(0.02) * u - np.float32(0.2) * z).astype(np.float32)
    return f


@pytest.mark.parametrize("tag,n_mod,size_z", [("mlp_small", 2, 16), ("mlp_poly5", 5, 64)])
def test_mlp_score_forward(golden, tag, n_mod, size_z):
    g = golden("alt_samplers.npz")
    sd = synth.fill_state_dict(arch.mlp_state_shapes(n_mod, size_z))
    y = OL.mlp_score(sd, g[f"{tag}_x"])
    assert_close(y, g[f"{tag}_y"], 1e-5, 1e-5 * np.abs(g[f"{tag}_y"]).max(), tag)


@pytest.mark.parametrize("sched", [0, 1])
def test_mlp_langevin_matches_calc_poly_uncond(golden, sched):
    """fid_upd10.calc_poly_uncond run unmodified (2 modalities, n_comp 6), decoded through ResVAE."""
    g = golden("alt_samplers.npz")
    sd = synth.fill_state_dict(arch.mlp_state_shapes(2, 64))
    dsd = synth.fill_state_dict(arch.resvae_state_shapes("ResVAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    noise = g[f"polyunc_s{sched}_noise"]
    it = iter(noise[2:])
    z0 = np.stack([noise[0], noise[1]], axis=1)
    z = OL.mlp_langevin(lambda x: OL.mlp_score(sd, x), z0, lambda s: next(it), 6, 0.03, 0.001, schedule=bool(sched))
    assert next(it, None) is None
    for m in range(2):
        y = OD.res_decoder(dsd, z[:, m], arch.POLY_ENC, arch.POLY_DEC, 32, "ResVAE")
        want = g[f"polyunc_s{sched}_out{m}"]
        assert_close(y, want, 1e-4, 1e-4 * np.abs(want).max(), f"calc_poly_uncond mod {m}")


@pytest.mark.parametrize("tag,n_mod,pred,stop", [("genae_cond", 2, (1,), 99), ("genae_all2", 2, (0, 1), 99), ("genae_all10", 10, tuple(range(10)), 99)])
def test_annealed_langevin_matches_gen_ae(golden, tag, n_mod, pred, stop):
    """fid_upd10.gen_ae run unmodified; the states its score net was called with are the checkpoints.  The CPU suite follows
    the first 100 score calls (the GPU suite runs the chain to the end)."""
    g = golden("alt_samplers.npz")
    sd = synth.fill_state_dict(arch.unet_state_shapes(channels=n_mod, **TINY))
    esd = synth.fill_state_dict(arch.resvae_state_shapes("ResAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    seed = int(g[f"{tag}_seed"][0])
    draws = Draws(seed)
    B = 2
    z0 = np.zeros((B, n_mod, 64), np.float32)
    for m in range(n_mod):                                   # fid_upd10.py:77-83: N(0,1) for predicted, encoder otherwise
        if m in pred:
            z0[:, m] = draws((B, 64))
        else:
            z0[:, m] = OD.res_encoder(esd, g[f"{tag}_p"][m], arch.POLY_ENC, "ResAE")[0]
    cks = [int(k[len(tag) + 4:]) for k in g if k.startswith(tag + "_zin")]
    trace = {k: None for k in cks if k <= stop}
    OL.annealed_langevin(tamed_score(sd, n_mod), z0, list(pred), draws, stop_after=stop, trace=trace)
    for k, v in trace.items():
        want = g[f"{tag}_zin{k}"]
        assert_close(v, want, 2e-4, 2e-4 * np.abs(want).max(), f"{tag} state at score call {k}")


@pytest.mark.parametrize("tag,noise_obs,pc", [("nobs1_pc1", True, True), ("nobs0_pc0", False, False)])
def test_conditional_loop_matches_evaluate(golden, tag, noise_obs, pc):
    """train_poly_unet_cont.evaluate run unmodified (VP beta 0.1..5, N=20, 'pc' order): its unconditional and conditional images."""
    g = golden("conditional.npz")
    sd = synth.fill_state_dict(arch.unet_state_shapes(channels=2, **TINY))
    asd = synth.fill_state_dict(arch.resvae_state_shapes("ResAE", arch.POLY_ENC, arch.POLY_DEC, 32, 64, 3))
    cfg = OS.SDECfg("vp", 0.1, 5.0, 20)
    score = lambda x, t: OU.unet_forward(sd, x, t, channels=2, **TINY)
    dec = lambda z: OD.res_decoder(asd, z, arch.POLY_ENC, arch.POLY_DEC, 32, "ResAE")

    def image(state):                                        # train_poly_unet_cont.py:381-389: clip, HWC, modalities side by side
        tiles = [np.clip(dec(state[:, m].reshape(1, 64)), 0.0, 1.0)[0].transpose(1, 2, 0) for m in range(2)]
        return np.concatenate(tiles, axis=1)

    it = iter(g[f"{tag}_unc_noise"])
    xm = OS.sampler(cfg, score, g[f"{tag}_unc_prior"], lambda s: next(it), pc=pc, order="pc")
    assert_close(image(xm), g[f"{tag}_unc_img"], 2e-4, 2e-4, "unconditional image")

    z_obs = np.zeros((1, 2, 8, 8), np.float32)
    z_obs[0, 0] = OD.res_encoder(asd, g[f"{tag}_imgs"][0], arch.POLY_ENC, "ResAE")[0][0].reshape(8, 8)   # given = first modality, first sample
    x0 = np.zeros_like(z_obs)
    x0[0, 1] = g[f"{tag}_cond_z0"].reshape(8, 8)
    it = iter(g[f"{tag}_cond_noise"])
    out = OL.cond_sampler(cfg, score, z_obs, [0], x0, lambda s: next(it), pc=pc, noise_obs=noise_obs)
    assert next(it, None) is None
    assert_close(image(out), g[f"{tag}_cond_img"], 2e-4, 2e-4, "conditional image")
