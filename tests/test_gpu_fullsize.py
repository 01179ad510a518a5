"""Large-shape checks through size-independent properties (the BASELINE configurations
themselves are in test_gpu_configs.py): every tree is a permutation of the points, every cut
satisfies max(left) <= thr == min(right) with the margins of Internal.hs:496-501."""
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
