"""BASELINE.json configs 4 and 5 as sampler WORKLOADS on the GPU: the N-step loops around the full-width
CelebA score nets -- `Unet(256, ch=3, (1,2,2,2,2))` on (B,3,16,16) latents (train_lat_celebhq_unet_cont2.py:472) and
`UNetModel(3,128,3,2,(),0.1,(1,2,4,8),use_z,z_dim=512)` + z_cond (train_lat_celebhq_unet_cont2_cond.py:651-653) -- against
fixtures produced by the REAL sde_helper2.uncond_sampler / em_predictor / corrector (tests/golden/make_golden.py cfg45), then
the CelebA decoders on the resulting latents, and -- at the configs' full size (B = 64, N = 1000) -- size-independent
properties: finite, deterministic, independent of how the batch is sharded.

Stated tolerance: N-step trajectories with injected noise rtol=1e-4, atol=1e-4*max|x|.  The fixture's score nets and noise are
TAMED (synth.tame_output_: last layer x 0.01, injected normals x 0.02), so the reference trajectory itself stays O(1) .. O(10) at the recipe's
beta = (0.1, 20) -- max|x_mean| 10 .. 20 -- and that tolerance checks elements of ordinary size.
"""
import contextlib
import os

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N_GOLD, SHAPE = 50, (2, 3, 16, 16)


@pytest.fixture(scope="module")
def pkg():
    from score_based_multimodal_autoencoder_amd import _lib, arch, h_vae_model, h_vae_model_copy, sde_helper2, synth, unet_model, unet_openai
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return dict(lib=_lib, arch=arch, synth=synth, unet=unet_model, oa=unet_openai, sde=sde_helper2, vae=h_vae_model_copy, attr=h_vae_model)


@pytest.fixture(scope="module")
def score4(pkg):
    m = pkg["unet"].Unet(dim=256, channels=3, dim_mults=(1, 2, 2, 2, 2)).eval()
    pkg["synth"].tame_output_(pkg["synth"].fill_module_(m))       # as the fixture's reference module (synth.tame_output_ says why)
    yield m.cuda()
    m.release_engines()


@pytest.fixture(scope="module")
def score5(pkg):
    m = pkg["oa"].UNetModel(3, 128, 3, 2, (), 0.1, (1, 2, 4, 8), num_heads=1, use_z=True, z_dim=512).eval()
    pkg["synth"].tame_output_(pkg["synth"].fill_module_(m))
    yield m.cuda()
    m.release_engines()


@contextlib.contextmanager
def production_chain_threshold():
    """conftest forces the XCD-resident chain kernel at every batch size; the workload properties are checked on the path a
    user gets (chain only for B >= 192)."""
    old = os.environ.get("SBM_CHAIN_MIN_B")
    os.environ["SBM_CHAIN_MIN_B"] = "192"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("SBM_CHAIN_MIN_B", None)
        else:
            os.environ["SBM_CHAIN_MIN_B"] = old


def _draws(pkg, seed, count, scale):
    """The injected normals of a fixture run: draw k = scale * synth.noise(seed, k, SHAPE); k = 0 is the prior.  `scale` is the fixture's
    NOISE_TAME (with the tamed last layer it keeps the reference trajectory O(1) .. O(10): synth.tame_output_)."""
    return [pkg["synth"].noise(int(seed), k, SHAPE) * np.float32(scale) for k in range(int(count))]


def _run_gold(pkg, g, tag, model, **kw):
    seed, count = g[tag + "_seed"]
    order, pc = tag.split("_")[1], tag.endswith("pc1")
    assert tuple(np.round(g["tame"], 6)) == (np.float32(pkg["synth"].OUTPUT_TAME).round(6), np.float32(pkg["synth"].NOISE_TAME).round(6)), "fixture and synth disagree on the taming scales"
    d = _draws(pkg, seed, count, g["tame"][1])
    sde = pkg["sde"].VPSDE(0.1, 20.0, N_GOLD)      # the CelebA recipe's beta range (train_cel.sh:11), N = 50: see make_golden.config45_cases
    noise = torch.from_numpy(np.stack(d[1:])).cuda()
    return pkg["sde"].uncond_sampler(SHAPE, model, "cuda", sde, pc=pc, n_steps=1, target_snr=0.16, order=order, noise=noise,
                                     x0=torch.from_numpy(d[0]), **kw)


@pytest.mark.parametrize("order", ["cp", "pc"])
@pytest.mark.parametrize("pc", [0, 1])
def test_config4_loop_matches_reference(pkg, golden, score4, order, pc):
    """sde_helper2.uncond_sampler (order cp) / calc_perf's predictor->corrector order around Unet(256, ch=3), N=50, beta = (0.1, 20)."""
    g = golden("config45.npz")
    tag = f"c4_{order}_pc{pc}"
    want = g[tag + "_xmean"]
    for use_graph in (True, False):
        got = _run_gold(pkg, g, tag, score4, use_graph=use_graph).cpu().numpy()
        assert_close(got, want, 1e-4, 1e-4 * np.abs(want).max(), f"{tag} graph={use_graph}")


@pytest.mark.parametrize("order", ["cp", "pc"])
@pytest.mark.parametrize("pc", [0, 1])
def test_config5_loop_matches_reference(pkg, golden, score5, order, pc):
    """The same loops around UNetModel(x, t, z=z_cond) (the semantics SURVEY D8 defines for the conditional script)."""
    g = golden("config45.npz")
    tag = f"c5_{order}_pc{pc}"
    want = g[tag + "_xmean"]
    zc = torch.from_numpy(g["c5_zcond"]).cuda()
    got = _run_gold(pkg, g, tag, score5, z_cond=zc).cpu().numpy()
    assert_close(got, want, 1e-4, 1e-4 * np.abs(want).max(), tag)


def test_config4_decoders_on_sampled_latents(pkg, golden, score4):
    """calc_perf's tail (train_lat_celebhq_unet_cont2.py:314-319): z[mod] = z_mean[:, ind].view(B, size_z) -> decoder, on the
    latents the GPU loop produced; compared with the reference decoders applied to the reference loop's latents."""
    g = golden("config45.npz")
    A = pkg["arch"]
    z = _run_gold(pkg, g, "c4_pc_pc1", score4)
    B = z.shape[0]
    img = pkg["synth"].fill_module_(pkg["vae"].ResVAEN(A.CEL_IMG_ENC, A.CEL_IMG_DEC, 128, 256, 3).eval()).cuda()
    msk = pkg["synth"].fill_module_(pkg["vae"].ResVAEN(A.CEL_MASK_ENC, A.CEL_MASK_DEC, 128, 256, 1).eval()).cuda()
    att = pkg["synth"].fill_module_(pkg["attr"].CelebAAttrNewBN(256).eval()).cuda()
    outs = {"img0": img.decoder(z[:, 0].reshape(B, 256))[:1], "mask": msk.decoder(z[:, 1].reshape(B, 256)), "attr": att.decoder(z[:, 2].reshape(B, 256))}
    for k, v in outs.items():
        want = g["c4_dec_" + k]
        assert_close(v.cpu().numpy(), want, 1e-3, 1e-3 * np.abs(want).max(), k)      # decoded images: 1e-3 relative
    for m in (img, msk, att):
        m.release_engines()


def _full_size_properties(pkg, model, **kw):
    """B = 64, N = 1000, in-kernel Philox noise: finite; the same key gives the same bits; two shards of 32 keyed by their global
    sample offset reproduce the unsharded rows bit for bit."""
    sde = pkg["sde"].VPSDE(0.1, 20.0, 1000)
    x0 = torch.empty(64, 3, 16, 16, device="cuda")
    pkg["lib"].check(pkg["lib"].lib().sbm_fill_normal(x0.data_ptr(), 3 * 256, 64, 99, 0, 0, 255, 1.0, None))
    zc = kw.pop("z_cond", None)

    def run(lo, n):
        extra = {} if zc is None else {"z_cond": zc[lo:lo + n].contiguous()}
        return pkg["sde"].uncond_sampler((n, 3, 16, 16), model, "cuda", sde, x0=x0[lo:lo + n].clone(), seed=4242, sample_offset=lo, **extra, **kw)

    a = run(0, 64)
    assert torch.isfinite(a).all()
    assert float(a.abs().max()) < 1e6      # (unit-variance Philox noise: |x| still grows over 1000 steps of this beta range; finite)
    b = run(0, 64)
    assert torch.equal(a, b), "same key, same bits"
    h0, h1 = run(0, 32), run(32, 32)
    assert torch.equal(torch.cat([h0, h1]), a), "two shards keyed by global sample index == the unsharded run"


def test_config4_full_size_properties(pkg, score4):
    with production_chain_threshold():
        score4.release_engines()
        _full_size_properties(pkg, score4)
        score4.release_engines()


def test_config5_full_size_properties(pkg, score5):
    zc = torch.empty(64, 512, device="cuda")
    pkg["lib"].check(pkg["lib"].lib().sbm_fill_normal(zc.data_ptr(), 512, 64, 98, 0, 0, 254, 1.0, None))
    with production_chain_threshold():
        _full_size_properties(pkg, score5, z_cond=zc)
