"""FID stage: the oracle against the reference's golden values (CPU), the fp64 device path against the oracle
(GPU), and the sharded accumulation over a world-size-2 gloo group (CPU)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from oracle import fid as OF

CASES = ("full", "singular", "wide")


# ---------------------------------------------------------------- CPU: oracle pinned by the reference's own function
@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_golden(golden, tag):
    g = golden("fid.npz")
    m1, s1 = OF.activation_statistics(g[f"{tag}.act1"])
    m2, s2 = OF.activation_statistics(g[f"{tag}.act2"])
    np.testing.assert_allclose(m1, g[f"{tag}.mu1"], rtol=0, atol=0)
    if f"{tag}.sigma1" in g:
        np.testing.assert_allclose(s1, g[f"{tag}.sigma1"], rtol=0, atol=0)
        np.testing.assert_allclose(s2, g[f"{tag}.sigma2"], rtol=0, atol=0)
    assert OF.frechet_distance(m1, s1, m2, s2) == pytest.approx(float(g[f"{tag}.fid"]), rel=1e-12)


def test_oracle_partial_sums_give_the_same_statistics(golden):
    g = golden("fid.npz")
    act = g["full.act1"]
    s_a, g_a, n_a = OF.partial_sums(act[:100])
    s_b, g_b, n_b = OF.partial_sums(act[100:])
    mu, sigma = OF.statistics_from_sums(s_a + s_b, g_a + g_b, n_a + n_b)
    want_mu, want_sigma = OF.activation_statistics(act)
    np.testing.assert_allclose(mu, want_mu, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(sigma, want_sigma, rtol=1e-11, atol=1e-14)


_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from oracle import fid as OF
    from score_based_multimodal_autoencoder_amd import distributed as D
    rank, world = D.init_from_env("gloo")
    g = dict(np.load(os.path.join({root!r}, "tests", "golden", "fid.npz")))
    act = g["full.act1"]
    lo, n = D.shard_range(act.shape[0], rank, world)
    s, gram, cnt = OF.partial_sums(act[lo:lo + n])
    flat = torch.from_numpy(np.concatenate([s, gram.ravel(), [float(cnt)]]))      # the layout ActivationStatistics.flat uses
    D.all_reduce_sum(flat)
    d = act.shape[1]
    f = flat.numpy()
    mu, sigma = OF.statistics_from_sums(f[:d], f[d:d + d * d].reshape(d, d), int(round(f[-1])))
    want_mu, want_sigma = OF.activation_statistics(act)
    assert int(round(f[-1])) == act.shape[0]
    np.testing.assert_allclose(mu, want_mu, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(sigma, want_sigma, rtol=1e-11, atol=1e-14)
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_sharded_statistics_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_c_abi_exports_fid_entry_points():
    import ctypes
    from score_based_multimodal_autoencoder_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    L = ctypes.CDLL(_lib.LIB_PATH)
    for sym in ("sbm_fid_accumulate", "sbm_fid_finalize_stats", "sbm_fid_work_doubles", "sbm_fid_frechet", "sbm_debug_dgemm_tn"):
        assert hasattr(L, sym), sym


# ---------------------------------------------------------------- GPU: the fp64 device path against the oracle
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 72, 333), (37, 160, 16), (256, 256, 2048)])
def test_dgemm_tn_matches_numpy(shape):
    import ctypes
    import torch
    from score_based_multimodal_autoencoder_amd import _lib
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N)
    at, b, c0 = rng.standard_normal((K, M)), rng.standard_normal((K, N)), rng.standard_normal((M, N))
    want = -0.5 * at.T @ b + 0.25 * c0 + 1.5 * np.eye(M, N)
    At, B, C = (torch.from_numpy(x).cuda() for x in (at, b, c0))
    _lib.check(_lib.lib().sbm_debug_dgemm_tn(At.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, -0.5, 0.25, 1.5,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    # fp64 accumulation in a different order from BLAS: a few ulps of the K-term sums
    np.testing.assert_allclose(C.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * np.sqrt(K))
    # the row-major-A form (the Newton-Schulz products)
    A2 = torch.from_numpy(np.ascontiguousarray(at.T)).cuda()
    C2 = torch.from_numpy(c0).cuda()
    _lib.check(_lib.lib().sbm_debug_dgemm_tn(A2.data_ptr(), B.data_ptr(), C2.data_ptr(), M, N, -K, -0.5, 0.25, 1.5,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(C2.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * np.sqrt(K))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_activation_statistics_match_oracle(golden, tag):
    import torch
    from score_based_multimodal_autoencoder_amd import fid_score as F
    g = golden("fid.npz")
    act = g[f"{tag}.act1"]
    st = F.ActivationStatistics(act.shape[1])
    for i in range(0, act.shape[0], 77):                       # ragged batches
        st.update(torch.from_numpy(act[i:i + 77]).cuda())
    assert st.n == act.shape[0]
    mu, sigma = st.finalize()
    want_mu, want_sigma = OF.activation_statistics(act)
    # fp64 throughout; the covariance is formed from raw sums (X^T X - n mu mu^T), np.cov from centred rows: ~n eps relative to mu^2
    np.testing.assert_allclose(mu.cpu().numpy(), want_mu, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(sigma.cpu().numpy(), want_sigma, rtol=1e-10, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_frechet_distance_matches_reference_golden(golden, tag):
    from score_based_multimodal_autoencoder_amd import fid_score as F
    g = golden("fid.npz")
    m1, s1 = OF.activation_statistics(g[f"{tag}.act1"])
    m2, s2 = OF.activation_statistics(g[f"{tag}.act2"])
    got, info = F.calculate_frechet_distance(m1, s1, m2, s2, return_info=True)
    # Newton-Schulz vs scipy's Schur sqrtm.  Full-rank products agree to ~1e-13; with fewer rows than dims the product is singular
    # and scipy's own result is only good to ~1e-7 relative (its sqrtm of a singular matrix), which bounds the comparison.
    tol = 1e-10 if tag == "full" else 2e-6
    assert got == pytest.approx(float(g[f"{tag}.fid"]), rel=tol), info
    assert not info["used_eps"] and max(info["iters"]) < 60


@pytest.mark.gpu
def test_fid_given_features_end_to_end_and_properties(golden):
    import torch
    from score_based_multimodal_autoencoder_amd import fid_score as F
    g = golden("fid.npz")
    a1, a2 = g["wide.act1"], g["wide.act2"]
    fid = F.calculate_fid_given_features(a1, a2, batch_size=64)
    assert fid == pytest.approx(float(g["wide.fid"]), rel=2e-6)
    # size-independent properties at the evaluation's real width (2048 features): d(X, X) = 0, symmetry
    rng = np.random.default_rng(5)
    mix = rng.standard_normal((2048, 2048)).astype(np.float32) * 0.02
    x = np.maximum(rng.standard_normal((4096, 2048)).astype(np.float32) @ mix + 0.2, 0).astype(np.float32)
    y = np.maximum(rng.standard_normal((4096, 2048)).astype(np.float32) @ mix + 0.25, 0).astype(np.float32)
    sx, sy = F.ActivationStatistics(2048).update(torch.from_numpy(x).cuda()), F.ActivationStatistics(2048).update(torch.from_numpy(y).cuda())
    (mx, cx), (my, cy) = sx.finalize(), sy.finalize()
    scale = float(torch.trace(cx) + torch.trace(cy))
    assert abs(F.calculate_frechet_distance(mx, cx, mx, cx)) < 1e-9 * scale
    dxy, dyx = F.calculate_frechet_distance(mx, cx, my, cy), F.calculate_frechet_distance(my, cy, mx, cx)
    assert dxy > 0 and dxy == pytest.approx(dyx, rel=1e-9)


@pytest.mark.gpu
def test_fid_rejects_cpu_tensors_and_bad_shapes():
    import torch
    from score_based_multimodal_autoencoder_amd import fid_score as F
    st = F.ActivationStatistics(16)
    with pytest.raises(RuntimeError):
        st.update(torch.zeros(4, 16))
    with pytest.raises(ValueError):
        st.update(torch.zeros(4, 8, device="cuda"))
    with pytest.raises(ValueError):
        st.finalize()
