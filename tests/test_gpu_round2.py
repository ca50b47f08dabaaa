"""GPU checks of the host-API additions: per-call Philox keys, the exact sharded corrector, the PolyMNIST scalar-form sampler,
and the chain kernel's barrier time-out path (residency failure must surface as an error, never as silent garbage)."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def pkg():
    from score_based_multimodal_autoencoder_amd import _lib, sde_helper2, synth, train_poly_unet_cont, unet_model
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return dict(lib=_lib, sde=sde_helper2, synth=synth, unet=unet_model, poly=train_poly_unet_cont)


def _unet(pkg, dim=64, channels=2):
    m = pkg["unet"].Unet(dim=dim, channels=channels, dim_mults=(1, 2, 2, 2)).eval()
    pkg["synth"].fill_module_(m)
    return m.cuda()


def test_default_key_gives_fresh_noise_per_call_and_manual_seed_reproduces(pkg):
    """two default-seed calls must not replay the same in-loop noise; torch.manual_seed reproduces a run."""
    S = pkg["sde"]
    m = _unet(pkg, dim=8)
    x0 = torch.randn(4, 2, 8, 8)
    sde = S.VPSDE(0.1, 5.0, 20)
    a = S.uncond_sampler((4, 2, 8, 8), m, "cuda", sde, x0=x0)
    b = S.uncond_sampler((4, 2, 8, 8), m, "cuda", sde, x0=x0)
    assert not torch.equal(a, b), "successive default calls drew the same noise"
    torch.manual_seed(5)
    c = S.uncond_sampler((4, 2, 8, 8), m, "cuda", sde, x0=x0)
    torch.manual_seed(5)
    d = S.uncond_sampler((4, 2, 8, 8), m, "cuda", sde, x0=x0)
    assert torch.equal(c, d)


@pytest.mark.parametrize("order", ["cp", "pc"])
def test_exact_sharded_corrector_equals_unsharded_bit_for_bit(pkg, order):
    """sde_helper2.py:97-99: the corrector's step size uses the batch MEAN of the norms.  exact_sharded_norms all-reduces two fp64 sums
    per corrector iteration; here two half-batches run as two 'ranks' (threads, one model replica each) whose norm_reduce adds the
    partials: every row equals the unsharded run, bit for bit.  The default mode (per-shard means) must differ."""
    S = pkg["sde"]
    B, N = 8, 12
    sde = S.VPSDE(0.1, 5.0, N)
    x0 = torch.randn(B, 2, 8, 8)
    full = S.uncond_sampler((B, 2, 8, 8), _unet(pkg, dim=8), "cuda", sde, pc=True, order=order, x0=x0, seed=31, exact_sharded_norms=True)
    models = [_unet(pkg, dim=8), _unet(pkg, dim=8)]
    box, bar, outs = [None, None], threading.Barrier(2), [None, None]

    def worker(r):
        def reduce(sums):
            box[r] = sums.clone()
            torch.cuda.synchronize()
            bar.wait(timeout=60)
            total = box[0] + box[1]
            bar.wait(timeout=60)
            return total
        outs[r] = S.uncond_sampler((B // 2, 2, 8, 8), models[r], "cuda", sde, pc=True, order=order, x0=x0[r * 4:(r + 1) * 4], seed=31,
                                   sample_offset=r * 4, exact_sharded_norms=True, global_batch=B, norm_reduce=reduce)

    th = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert outs[0] is not None and outs[1] is not None
    assert torch.equal(torch.cat(outs), full), "sharded exact mode must reproduce the unsharded batch-mean corrector"
    # the fused loop with per-shard means is a different (documented) step size
    shard = S.uncond_sampler((B // 2, 2, 8, 8), models[0], "cuda", sde, pc=True, order=order, x0=x0[:4], seed=31, sample_offset=0)
    assert not torch.equal(shard, full[:4])
    # and on ONE device the exact mode agrees with the fused loop's own batch mean to rounding
    fused = S.uncond_sampler((B, 2, 8, 8), models[0], "cuda", sde, pc=True, order=order, x0=x0, seed=31)
    assert_close(full.cpu().numpy(), fused.cpu().numpy(), 1e-4, 1e-5 * float(fused.abs().max()), "exact mode vs fused loop, one device")


def test_polymnist_scalar_form_sampler_matches_reference_trajectory(pkg, golden):
    """train_poly_unet_cont.uncond_sampler(shape, model, device, T, eps, beta_0, beta_1, N, probability_flow, pc, ...): predictor then
    corrector, against the config-1 trajectory produced with the reference's functions in that order."""
    g = golden("trajectories.npz")
    P, synth = pkg["poly"], pkg["synth"]
    m = _unet(pkg)
    shape = (16, 2, 8, 8)
    x0 = synth.noise(1234, 0, shape)
    noise = torch.from_numpy(np.stack([synth.noise(1234, k, shape) for k in range(1, 201)])).cuda()
    xm = P.uncond_sampler(shape, m, "cuda", 1, 1e-3, 0.1, 5.0, 100, False, True, 1, 0.16, noise=noise, x0=torch.from_numpy(x0))
    want = g["vp_pc_pc1_mean99"]
    assert_close(xm.cpu().numpy(), want, 1e-4, 1e-4 * np.abs(want).max(), "scalar-form sampler")


def test_chain_barrier_timeout_is_reported_not_swallowed(pkg):
    """a chain-kernel barrier that cannot complete (not every workgroup resident) must surface.  The debug knob makes
    one workgroup skip its arrivals: every barrier times out (bounded spin), the grid drains, the kernel raises the host-mapped
    flag -- and the NEXT call on the handle fails loudly, resets the state, after which the model works again."""
    L = pkg["lib"].lib()
    m = _unet(pkg, dim=8)
    x = torch.randn(64, 2, 8, 8, device="cuda")
    t = torch.rand(64, device="cuda")
    good = m(x, t).clone()
    eng = m._engine_for(64, 8, 8)
    pkg["lib"].check(L.sbm_debug_chain_ctl(eng.h, 4096, 1))
    m(x, t)                                    # returns (asynchronous); its result is garbage
    torch.cuda.synchronize()
    pkg["lib"].check(L.sbm_debug_chain_ctl(eng.h, 0, 0))
    with pytest.raises(RuntimeError, match="timed out"):
        m(x, t)
    again = m(x, t)
    torch.cuda.synchronize()
    assert torch.equal(again, good), "after the reset the model must work as before"
    st = (ctypes.c_uint * 528)()
    pkg["lib"].check(L.sbm_debug_chain_sync(eng.h, st, 528))
    assert not any(st), "synchronisation state clean again"
