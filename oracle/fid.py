"""TEST INFRASTRUCTURE -- CPU restatement of the reference's FID statistics; never imported by the product.

Follows pytorch_fid/fid_score.py: activations are held in a float64 array (:127, :147), mu = np.mean(act, axis=0) and
sigma = np.cov(act, rowvar=False) (:249-250, :273-279), calculate_frechet_distance (:187-240) with scipy's Schur-method
sqrtm.  Pinned by tests/golden/fid.npz, which holds the outputs of the reference's own calculate_frechet_distance.
"""
import numpy as np
from scipy import linalg


def activation_statistics(act):
    """fid_score.py:248-250 -- act is (n, dims); float32 features are widened first, as `pred_arr` (float64) does."""
    a = np.empty(act.shape, dtype=np.float64)
    a[:] = act
    return np.mean(a, axis=0), np.cov(a, rowvar=False)


def partial_sums(act):
    """(sum, Gram, n) of a shard in float64: the quantities the device path accumulates and all-reduces."""
    a = np.asarray(act, dtype=np.float64)
    return a.sum(axis=0), a.T @ a, a.shape[0]


def statistics_from_sums(s, g, n):
    mu = s / n
    return mu, (g - n * np.outer(mu, mu)) / (n - 1)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:187-240 (Sutherland's stable version)."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)          # :221
    if not np.isfinite(covmean).all():                                  # :222-227
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):                                        # :230-234
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))   # :236-240
