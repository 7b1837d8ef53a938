"""The oracle's sparsemax / 1.5-entmax (oracle/sparse_norms.py; parity with the `entmax` package UNPINNED) against
hand-derived known answers and the properties that define the two maps."""
import numpy as np
import pytest

from oracle.sparse_norms import entmax15, sparsemax


def test_sparsemax_known_answers():
    np.testing.assert_allclose(sparsemax([1.0, 0.5, -1.0]), [0.75, 0.25, 0.0], atol=1e-12)   # k=2, tau=.25
    np.testing.assert_allclose(sparsemax([3.0, 0.0, 0.0]), [1.0, 0.0, 0.0], atol=1e-12)      # one-hot
    np.testing.assert_allclose(sparsemax([0.0, 0.0, 0.0]), [1 / 3] * 3, atol=1e-12)           # uniform
    np.testing.assert_allclose(sparsemax([0.2, 0.1]), [0.55, 0.45], atol=1e-12)              # dense: z - tau


def test_entmax15_known_answers():
    np.testing.assert_allclose(entmax15([0.0, 0.0]), [0.5, 0.5], atol=1e-12)
    np.testing.assert_allclose(entmax15([4.0, 0.0]), [1.0, 0.0], atol=1e-12)
    # z = [1, 0]: x = [0, -.5], tau = -.25 - sqrt(.4375): p = (x - tau)^2
    t = -0.25 - np.sqrt(0.4375)
    np.testing.assert_allclose(entmax15([1.0, 0.0]), [t * t, (-0.5 - t) ** 2], atol=1e-12)


@pytest.mark.parametrize("fn", [sparsemax, entmax15])
def test_defining_properties(fn):
    rng = np.random.default_rng(0)
    for scale in (0.1, 1.0, 5.0):
        z = rng.normal(size=(200, 7)) * scale
        p = fn(z, axis=-1)
        assert (p >= 0).all()
        np.testing.assert_allclose(p.sum(-1), 1.0, atol=1e-10)
        np.testing.assert_allclose(fn(z + 3.7, axis=-1), p, atol=1e-10)      # shift invariance
        assert (np.argsort(-z, -1)[:, 0] == np.argmax(p, -1)).all()           # order preserving
        perm = rng.permutation(7)
        np.testing.assert_allclose(fn(z[:, perm], axis=-1), p[:, perm], atol=1e-12)
    # axis handling as the reference uses it (dim=-2 of [..., concepts, patches])
    z = rng.normal(size=(3, 5, 11))
    np.testing.assert_allclose(fn(z, axis=-2), np.swapaxes(fn(np.swapaxes(z, -1, -2), axis=-1), -1, -2))


def test_sparsemax_is_the_simplex_projection():
    """sparsemax(z) = argmin_{p in simplex} |p - z|^2: no random simplex point is closer."""
    rng = np.random.default_rng(1)
    z = rng.normal(size=(50, 5)) * 2
    p = sparsemax(z)
    d0 = ((p - z) ** 2).sum(-1)
    for _ in range(200):
        q = rng.dirichlet(np.ones(5), size=50)
        assert (((q - z) ** 2).sum(-1) >= d0 - 1e-12).all()


def test_entmax15_satisfies_its_fixed_point():
    """p_i = max(z_i/2 - tau, 0)^2 for one tau, sum p = 1 (the KKT form of Peters et al., eq. 9-10 at alpha = 1.5)."""
    rng = np.random.default_rng(2)
    z = rng.normal(size=(100, 6)) * 3
    p = entmax15(z)
    x = (z - z.max(-1, keepdims=True)) / 2
    sup = p > 0
    tau = np.where(sup, x - np.sqrt(p), np.nan)
    assert np.nanmax(np.nanmax(tau, -1) - np.nanmin(tau, -1)) < 1e-9           # one tau per row on the support
    t = np.nanmean(tau, -1, keepdims=True)
    assert (x[~sup] <= np.broadcast_to(t, x.shape)[~sup] + 1e-12).all()         # off-support logits are below it
