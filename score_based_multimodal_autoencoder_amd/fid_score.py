"""Drop-in for the statistics half of the reference's `pytorch_fid/fid_score.py`, on the GPU in fp64.

Same names and meaning as the reference where the reference has them:

    calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6)      fid_score.py:187-240
    calculate_activation_statistics(...)  -> ActivationStatistics        fid_score.py:243-251 (from features, see below)
    calculate_fid_given_features(act1, act2)                             fid_score.py:273-282 minus the Inception forward

The Inception network is not part of this package (no weights offline): everything starts from the (n, dims) feature rows the
reference collects in `pred_arr`.  Instead of growing that array on the host, `ActivationStatistics.update` accumulates
sum(x) and X^T X of each feature batch in fp64 on the device (one fp64-MFMA GEMM per batch); with several ranks each one
accumulates its shard and `all_reduce()` sums the buffers with ONE collective (dims + dims^2 + 1 doubles: 33.6 MB for pool3),
after which mean and covariance are exactly np.mean / np.cov(rowvar=False) of the concatenated features.

There is no CPU path: tensors must live on the GPU (numpy inputs are copied there) and a missing library raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f64(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    if not torch.is_tensor(x):
        x = torch.as_tensor(x, dtype=torch.float64)
    return x.to(device=device, dtype=torch.float64).contiguous()


def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this package has no CPU path")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


class ActivationStatistics:
    """Running fp64 sum / Gram matrix of feature rows; `finalize()` gives (mu, sigma) as the reference defines them."""

    def __init__(self, dims=2048, device=None):
        self.dims = int(dims)
        self.device = _device(device)
        # one flat buffer so that the cross-rank sum is a single collective: [sum (dims) | gram (dims^2) | n (1)]
        self.flat = torch.zeros(self.dims + self.dims * self.dims + 1, dtype=torch.float64, device=self.device)

    @property
    def n(self):
        return int(round(float(self.flat[-1].item())))

    def update(self, feats):
        """feats: (n, dims) float32 CUDA tensor (a batch of pool3 features).  fid_score.py:143-151 / :172-182."""
        _lib._require_cuda(feats, "features")
        if feats.dim() == 4:                                     # (n, dims, 1, 1) as the network returns them (:147 squeeze)
            feats = feats.reshape(feats.shape[0], -1)
        if feats.dim() != 2 or feats.shape[1] != self.dims:
            raise ValueError(f"features must be (n, {self.dims}), got {tuple(feats.shape)}")
        f = feats.to(torch.float32).contiguous()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            _lib.check(L.sbm_fid_accumulate(f.data_ptr(), f.shape[0], self.dims, f.stride(0), self.flat.data_ptr(),
                                            self.flat[self.dims:].data_ptr(), _stream()))
        self.flat[-1] += f.shape[0]
        return self

    def all_reduce(self, group=None):
        """Sum the accumulators over the ranks of `group` (RCCL over xGMI with backend "nccl"): the one exchange of the FID stage."""
        from . import distributed as D
        D.all_reduce_sum(self.flat, group)
        return self

    def finalize(self):
        """-> (mu, sigma): float64 CUDA tensors, np.mean(act, 0) and np.cov(act, rowvar=False) of everything accumulated."""
        n = self.n
        if n < 2:
            raise ValueError("need at least two feature rows for a covariance")
        mu = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        sigma = torch.empty(self.dims, self.dims, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sbm_fid_finalize_stats(self.flat.data_ptr(), self.flat[self.dims:].data_ptr(), n, self.dims,
                                                         mu.data_ptr(), sigma.data_ptr(), _stream()))
        return mu, sigma


def calculate_activation_statistics(act, batch_size=4096, device=None):
    """(mu, sigma) of a whole feature array, as numpy float64 (fid_score.py:243-251 with the activations already computed)."""
    device = _device(device)
    act_t = torch.as_tensor(act)
    st = ActivationStatistics(act_t.shape[1], device)
    for i in range(0, act_t.shape[0], batch_size):
        st.update(act_t[i:i + batch_size].to(device=device, dtype=torch.float32))
    mu, sigma = st.finalize()
    return mu.cpu().numpy(), sigma.cpu().numpy()


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6, max_iters=0, return_info=False, device=None):
    """d^2 = |mu1 - mu2|^2 + Tr(sigma1 + sigma2 - 2 sqrt(sigma1 sigma2))  (fid_score.py:187-240).

    Inputs: numpy arrays or tensors (moved to the GPU as float64).  The matrix square root is a Newton-Schulz iteration of fp64
    MFMA GEMMs (csrc/fid.hip); like the reference, a non-finite root is retried with `eps` on both diagonals."""
    device = _device(device)
    m1, m2 = _dev_f64(np.atleast_1d(mu1) if isinstance(mu1, np.ndarray) else mu1, device), _dev_f64(np.atleast_1d(mu2) if isinstance(mu2, np.ndarray) else mu2, device)
    s1, s2 = _dev_f64(np.atleast_2d(sigma1) if isinstance(sigma1, np.ndarray) else sigma1, device), _dev_f64(np.atleast_2d(sigma2) if isinstance(sigma2, np.ndarray) else sigma2, device)
    assert m1.shape == m2.shape, "Training and test mean vectors have different lengths"
    assert s1.shape == s2.shape, "Training and test covariances have different dimensions"
    dims = m1.shape[0]
    if s1.shape != (dims, dims):
        raise ValueError(f"covariances must be ({dims}, {dims})")
    L = _lib.lib()
    work = torch.empty(L.sbm_fid_work_doubles(dims), dtype=torch.float64, device=device)
    out = (ctypes.c_double * 5)()
    with torch.cuda.device(device):
        _lib.check(L.sbm_fid_frechet(m1.data_ptr(), s1.data_ptr(), m2.data_ptr(), s2.data_ptr(), dims, float(eps), int(max_iters),
                                     work.data_ptr(), out, _stream()))
    if out[4]:
        print("fid calculation produces singular product; adding %s to diagonal of cov estimates" % eps)      # fid_score.py:223-226
    if return_info:
        return float(out[0]), dict(tr_covmean=float(out[1]), iters=(int(out[2]), int(out[3])), used_eps=bool(out[4]))
    return float(out[0])


def calculate_fid_given_features(act1, act2, batch_size=4096, device=None):
    """calculate_fid_given_tensors (fid_score.py:264-282) with the activations already computed."""
    device = _device(device)
    stats = []
    for act in (act1, act2):
        act_t = torch.as_tensor(act)
        st = ActivationStatistics(act_t.shape[1], device)
        for i in range(0, act_t.shape[0], batch_size):
            st.update(act_t[i:i + batch_size].to(device=device, dtype=torch.float32))
        stats.append(st.finalize())
    (m1, s1), (m2, s2) = stats
    return calculate_frechet_distance(m1, s1, m2, s2, device=device)
