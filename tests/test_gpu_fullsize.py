"""BASELINE-sized checks (configs[1] shape: 1M x 128) through size-independent properties:
every tree is a permutation of the points, every cut satisfies max(left) <= thr == min(right)
with the margins of Internal.hs:496-501, the oracle's knn over the device-built forest returns
the same ids, and the MFMA / exact projection kernels agree on the leaf of (almost) every point."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rp():
    import rptree_amd
    return rptree_amd


def mixture(n, d, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    mu = (rng.random(n) < 0.5) * 2.0
    return (rng.standard_normal((n, d)) * 0.5 + mu[:, None]).astype(dtype)


def check_cuts(f, P, rng, n_nodes=24):
    topo = [r for r in f.topology() if not r[4]]
    for idx in rng.choice(len(topo), size=min(n_nodes, len(topo)), replace=False):
        level, heap, off, m, _ = (int(v) for v in topo[idx])
        for t in range(f.T):
            nh = m // 2
            p = P[t, level]
            left = p[f.perm[t, off:off + nh]]
            right = np.sort(p[f.perm[t, off + nh:off + m]])
            assert left.max() <= f.thr[t, heap] == right[0]
            assert f.mglo[t, heap] == left.max()
            assert f.mghi[t, heap] == right[1]


def test_c2_shape_exact_build_and_knn(rp, oracle):
    n, d, T, min_leaf, k = 1_000_000, 128, 4, 128, 10
    X = mixture(n, d, 1234)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    assert cfg.fpMaxTreeDepth == 13
    f = rp.forestBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, X)
    for t in range(T):
        assert np.array_equal(np.bincount(f.perm[t], minlength=n), np.ones(n, dtype=np.int64))
    P = f.proj()
    # exact-order projections: spot check against the reference order (Internal.hs:382)
    idx = np.nonzero(f.R[1, 3])[0]
    for i in (0, 12345, n - 1):
        assert P[1, 3, i] == oracle.inner_sd(idx, f.R[1, 3, idx], X[i])
    check_cuts(f, P, np.random.default_rng(0))
    # leaves: each bucket is ordered by the parent level's key (children inherit sorted order)
    leaves = [r for r in f.topology() if r[4]]
    for (level, heap, off, m, _) in leaves[:50] + leaves[-50:]:
        for t in range(T):
            kk = P[t, level - 1][f.perm[t, off:off + m]]
            assert (np.diff(kk) >= 0).all()
    # the oracle's knn over the device-built flat forest == device knn (ids), 64 queries
    Q = mixture(64, d, 4321)
    ids, dist, cnt = rp.knnBatch(k, f, Q)
    fo = oracle.Forest(n, d, f.R, f.L, min_leaf, f.perm, f.thr, f.mglo, f.mghi)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert np.array_equal(ids[i, :cnt[i]], wi)
        assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-12)
    # MFMA projections: same leaves for (almost) every point
    g = rp.forestBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, f.data,
                       mode=rp.RPT_PROJ_MFMA)
    leaf_off = np.array([o for (_, _, o, m, lf) in f.topology() if lf])
    flips = 0
    for t in range(T):
        ia, ib = np.empty(n, np.int64), np.empty(n, np.int64)
        ia[f.perm[t]] = np.arange(n)
        ib[g.perm[t]] = np.arange(n)
        flips += (np.searchsorted(leaf_off, ia, side="right") !=
                  np.searchsorted(leaf_off, ib, side="right")).sum()
    assert flips / (T * n) < 1e-3
    Pg = g.proj()
    scale = np.linalg.norm(X[:1000], axis=1)[None, None, :] * np.linalg.norm(f.R, axis=2)[:, :, None]
    assert (np.abs(Pg[:, :, :1000] - P[:, :, :1000]) <= 1e-5 * scale).all()


def test_f32_two_million_build_is_valid(rp):
    n, d, T, min_leaf = 2_000_000, 128, 2, 128
    X = mixture(n, d, 7, np.float32)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    f = rp.forestBatch(99, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, X)
    for t in range(T):
        assert np.array_equal(np.bincount(f.perm[t], minlength=n), np.ones(n, dtype=np.int64))
    P = f.proj()
    assert P.dtype == np.float32
    check_cuts(f, P, np.random.default_rng(1), n_nodes=16)
    ids, dist, cnt = rp.knnBatch(10, f, X[:32])
    assert (ids[:, 0] == np.arange(32)).all() and (dist[:, 0] == 0).all()   # self is found
