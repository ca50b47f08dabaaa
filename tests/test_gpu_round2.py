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
    """VERDICT r1 #4 / ADVICE r2: a chain-kernel barrier that cannot complete (not every workgroup resident) must surface.  The chain
    kernel is the LEGACY path now (SBM_PIX=0; the default sample-owned kernel has no inter-workgroup barrier at all).  The debug knob
    makes one workgroup skip its arrivals: every barrier times out (bounded spin), the grid drains, the kernel raises the host-mapped
    flag.  The call that enqueued the launch has long returned 0; what reports it is (a) Engine.check(sync=True) -- the way to vouch
    for an output no later call will -- or (b) the NEXT forward on the handle, or (c) close().  Each resets the state; the model then
    works again."""
    import os
    L = pkg["lib"].lib()
    os.environ["SBM_PIX"] = "0"
    try:
        m = _unet(pkg, dim=8)
        x = torch.randn(64, 2, 8, 8, device="cuda")
        t = torch.rand(64, device="cuda")
        good = m(x, t).clone()
        eng = m._engine_for(64, 8, 8)
        assert not eng.info["uses_pix"]
        eng.check(sync=True)                       # healthy: no error
        for how in ("check", "next call"):
            pkg["lib"].check(L.sbm_debug_chain_ctl(eng.h, 4096, 1))
            m(x, t)                                # returns (asynchronous); its result is garbage
            torch.cuda.synchronize()
            pkg["lib"].check(L.sbm_debug_chain_ctl(eng.h, 0, 0))
            with pytest.raises(RuntimeError, match="timed out"):
                if how == "check":
                    eng.check(sync=True)
                else:
                    m(x, t)
            again = m(x, t)
            torch.cuda.synchronize()
            assert torch.equal(again, good), "after the reset the model must work as before"
            st = (ctypes.c_uint * 528)()
            pkg["lib"].check(L.sbm_debug_chain_sync(eng.h, st, 528))
            assert not any(st), "synchronisation state clean again"
        # (c) an unreported time-out at destroy: close() raises (the handle is gone either way)
        pkg["lib"].check(L.sbm_debug_chain_ctl(eng.h, 4096, 1))
        m(x, t)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="timed out"):
            m.release_engines()
        m.release_engines()
    finally:
        os.environ.pop("SBM_PIX", None)


def test_handle_dropped_right_behind_asynchronous_work_is_released_safely(pkg):
    """VERDICT r2 item 2 (the abort recorded in gpurun_out/t7_r2.log): forwards and the single-kernel entry points are enqueued on the
    CALLER's stream and return at once, so a temporary model can be garbage-collected while its kernels still run.  sbm_model_destroy
    must wait for them (and report, not abort).  Enqueue a forward and eager Langevin work on a fresh handle and drop it immediately --
    both the default (pixel) and the legacy (chain) engines -- then keep using the device."""
    import gc
    import os
    from score_based_multimodal_autoencoder_amd import _lib
    L = pkg["lib"].lib()
    x = torch.randn(256, 2, 8, 8, device="cuda")
    t = torch.rand(256, device="cuda")
    for env in ({}, {"SBM_PIX": "0"}):
        os.environ.update(env)
        try:
            for rep in range(3):
                m = _unet(pkg, dim=8)
                y = m(x, t)                                            # asynchronous forward on the current stream
                noise = torch.randn_like(x)
                sums = torch.zeros(2, dtype=torch.float64, device="cuda")
                norms = torch.zeros(256, 2, device="cuda")
                pkg["lib"].check(L.sbm_langevin_norms(y.data_ptr(), noise.data_ptr(), norms.data_ptr(), sums.data_ptr(), 256, 2 * 64, 31, 0, 0, 1, _lib.stream_ptr()))
                for _ in range(4):
                    y = m(y, t)                                        # more work queued behind it
                eng = m._engine_for(256, 8, 8)
                m.release_engines()                                    # destroy NOW, kernels in flight; must neither abort nor raise
                assert eng.h is None
                del m, eng
                gc.collect()
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
        finally:
            for k in env:
                os.environ.pop(k, None)
