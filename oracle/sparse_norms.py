"""sparsemax and 1.5-entmax over an axis, numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference takes both from the third-party `entmax` package
(`from entmax import entmax15, sparsemax`, concept_attention/concept_attention_pipeline.py:5, used at :66-69), which
it neither pins in requirements.txt nor vendors, and which is not installable here; the reference holds no test,
fixture or output for these branches.  This file restates the PUBLISHED algorithms:

  * sparsemax -- Martins & Astudillo, "From Softmax to Sparsemax", ICML 2016, Algorithm 1: Euclidean projection
    onto the simplex: sort z descending, k = max{j : 1 + j z_(j) > sum_{i<=j} z_(i)},
    tau = (sum_{i<=k} z_(i) - 1) / k, p = max(z - tau, 0).
  * 1.5-entmax -- Peters, Niculae & Martins, "Sparse Sequence-to-Sequence Models", ACL 2019, Algorithm 2:
    p_i = max(z_i / 2 - tau, 0)^2 with tau such that sum p = 1: x = (z - max z)/2 sorted descending, for every
    prefix j: M_j = mean, ss_j = j (mean of squares - M_j^2), tau_j = M_j - sqrt(max((1 - ss_j)/j, 0)),
    k = #{j : tau_j <= x_(j)}, tau = tau_k.

Only tests/ may import this; it checks ca_heatmap_norm_accumulate and is pinned by the hand-derived known
answers and the defining properties in tests/test_sparse_norms_cpu.py.
"""
import numpy as np


def sparsemax(z, axis=-1):
    z = np.moveaxis(np.asarray(z, dtype=np.float64), axis, -1)
    z = z - z.max(-1, keepdims=True)
    srt = -np.sort(-z, axis=-1)
    cs = np.cumsum(srt, -1)
    rho = np.arange(1, z.shape[-1] + 1)
    support = 1 + rho * srt > cs
    k = support.sum(-1, keepdims=True)
    tau = (np.take_along_axis(cs, k - 1, -1) - 1) / k
    return np.moveaxis(np.maximum(z - tau, 0), -1, axis)


def entmax15(z, axis=-1):
    z = np.moveaxis(np.asarray(z, dtype=np.float64), axis, -1)
    x = (z - z.max(-1, keepdims=True)) / 2
    srt = -np.sort(-x, axis=-1)
    rho = np.arange(1, x.shape[-1] + 1)
    mean = np.cumsum(srt, -1) / rho
    mean_sq = np.cumsum(srt ** 2, -1) / rho
    ss = rho * (mean_sq - mean ** 2)
    tau = mean - np.sqrt(np.clip((1 - ss) / rho, 0, None))
    k = (tau <= srt).sum(-1, keepdims=True)
    tau_star = np.take_along_axis(tau, k - 1, -1)
    return np.moveaxis(np.maximum(x - tau_star, 0) ** 2, -1, axis)
