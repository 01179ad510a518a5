"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer / index results must be IDENTICAL; exact-order projections must be
bit-identical; MFMA projections within 1e-5 * |x||r| (north_star tolerance, norm-relative
because a projection can cancel to ~0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rp():
    import rptree_amd
    return rptree_amd


@pytest.fixture(scope="module")
def ctx(rp):
    return rp.default_context()


@pytest.fixture
def option(ctx):
    """option(name, value): context manager switching one algorithm option of the context
    (rpt_ctx_set_option) for the enclosed calls — the fallback paths must stay exact too."""
    import contextlib

    @contextlib.contextmanager
    def switch(name, value):
        old = ctx.set_option(name, value)
        try:
            yield
        finally:
            ctx.set_option(name, old)
    return switch


def ref_inner_exact(X, r):
    """innerSD order (Internal.hs:382): acc = x_k*r_k + acc from the last index to the first,
    separate multiply and add (numpy element-wise ops are IEEE exact)."""
    acc = np.zeros(X.shape[0], dtype=X.dtype)
    for k in range(X.shape[1] - 1, -1, -1):
        if r[k] != 0:
            acc = X[:, k] * X.dtype.type(r[k]) + acc
    return acc


def sparse_R(rng, C, d, pnz):
    R = rng.standard_normal((C, d))
    R[rng.random((C, d)) >= pnz] = 0.0
    return R


# ------------------------------------------------------------------ projection batch
@pytest.mark.parametrize("n,d,C", [(1000, 16, 3), (5000, 128, 32), (777, 37, 33), (64, 5, 1),
                                   (1, 128, 40), (4099, 130, 13)])
def test_project_exact_f64_bit_identical(rp, ctx, n, d, C):
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d))
    R = sparse_R(rng, C, d, 0.47)
    P = rp.project(X, R, mode=rp.RPT_PROJ_EXACT, ctx=ctx)
    for c in range(C):
        assert np.array_equal(P[c], ref_inner_exact(X, R[c])), "column %d" % c


def test_project_exact_matches_oracle_inner_sd(rp, ctx, oracle):
    X = oracle.data_normal_dense2(1234, 300, 16)
    R, _ = oracle.forest_hyperplanes(7, 2, 4, 0.83, 16)
    P = rp.project(X, R.reshape(-1, 16), mode=rp.RPT_PROJ_EXACT, ctx=ctx)
    for c in range(8):
        idx = np.nonzero(R.reshape(-1, 16)[c])[0]
        for i in (0, 1, 150, 299):
            assert P[c, i] == oracle.inner_sd(idx, R.reshape(-1, 16)[c, idx], X[i])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 1e-5)])
# d = 128 with more than 32 hyperplanes: the wide (96 / 64 columns per pass) kernels and their tails
@pytest.mark.parametrize("n,d,C", [(1000, 16, 3), (5000, 128, 32), (777, 37, 33), (4099, 200, 17),
                                   (3001, 128, 33), (2100, 128, 64), (1999, 128, 100),
                                   (1500, 128, 230),
                                   # rows of k * 128 elements: one pass per 128-element K chunk
                                   (2000, 256, 100), (1500, 384, 40), (900, 512, 20),
                                   # 16-byte aligned rows of any length: short last chunk
                                   (3000, 64, 70), (1200, 784, 50), (2222, 100, 33)])
def test_project_mfma_within_tolerance(rp, ctx, dtype, tol, n, d, C):
    rng = np.random.default_rng(n * 3 + d)
    X = rng.standard_normal((n, d)).astype(dtype)
    R = sparse_R(rng, C, d, 0.47)
    P = rp.project(X, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    Rq = R.astype(dtype).astype(np.float64)          # the kernel rounds R to the compute type
    want = X.astype(np.float64) @ Rq.T
    scale = np.linalg.norm(X.astype(np.float64), axis=1)[:, None] * np.linalg.norm(Rq, axis=1)[None, :]
    err = np.abs(P.T.astype(np.float64) - want)
    assert (err <= tol * scale + 1e-300).all(), float((err / (scale + 1e-300)).max())


# d % 8 == 0: the bf16 matrix pipe with the hyperplanes split into TWO bf16 terms (|error| <= 2^-17 |x||r| by
# construction; option proj_bf16_terms = 3: three terms, f32-level agreement) — 128-column passes, 64-column
# tail, A resident in LDS up to d = 128 and streamed in chunks of four k-steps beyond, ragged last k-step,
# fewer points than one 256-point tile; otherwise the f32-MFMA kernels
@pytest.mark.parametrize("n,d,C", [(3000, 256, 32), (2500, 128, 100), (1111, 128, 52),
                                   (5000, 768, 150), (257, 72, 5), (100, 8, 1), (4097, 200, 129),
                                   (2048, 64, 300), (1000, 100, 40), (999, 36, 70)])
def test_project_mfma_bf16_input(rp, ctx, option, n, d, C):
    import torch
    rng = np.random.default_rng(5)
    Xf = rng.standard_normal((n, d)).astype(np.float32)
    xb = torch.from_numpy(Xf).to(torch.bfloat16)
    Xr = xb.to(torch.float32).numpy().astype(np.float64)     # exactly representable inputs
    xd = xb.cuda()
    torch.cuda.synchronize()
    ds = rp.Dataset.dense_device(ctx, xd.data_ptr(), n, d, rp.RPT_BF16, keep=xd)
    R = sparse_R(rng, C, d, 0.35)
    P = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    Rq = R.astype(np.float32).astype(np.float64)
    want = Xr @ Rq.T
    scale = np.linalg.norm(Xr, axis=1)[:, None] * np.linalg.norm(Rq, axis=1)[None, :]
    assert (np.abs(P.T - want) <= 1e-5 * scale).all()
    if d % 8 == 0:
        # two terms: every element of r within 2^-17 of itself (two roundings to 8 significant bits), so
        # 2^-17 |x||r| = 7.6e-6 by Cauchy-Schwarz, plus the f32 accumulation
        assert (np.abs(P.T - want) <= 8e-6 * scale).all()
        with option("proj_bf16_terms", 3):   # 24 bits of R: f32-level agreement
            P3 = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        assert (np.abs(P3.T - want) <= 2e-6 * scale).all()
        assert np.abs(P3.T - want).max() <= np.abs(P.T - want).max() * 1.5 + 1e-12
        with option("proj_bf16_terms", 8):   # the eight-wave workgroup shape: the same sums, bit for bit
            P8 = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
        assert np.array_equal(P8, P)


def test_project_bf16_both_kernels_agree(rp, ctx, option):
    """bf16 data: the bf16x3 matrix-pipe kernel and the f32-MFMA kernel on converted inputs
    compute the same contraction (both within 1e-5 |x||r|; here against each other), and a
    forest built on either has the same leaf assignment up to points within rounding of a median."""
    import torch
    rng = np.random.default_rng(17)
    n, d, C = 20000, 128, 96
    xb = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(torch.bfloat16).cuda()
    torch.cuda.synchronize()
    ds = rp.Dataset.dense_device(ctx, xb.data_ptr(), n, d, rp.RPT_BF16, keep=xb)
    R = sparse_R(rng, C, d, 0.4)
    Pa = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    with option("proj_bf16_f32", 1):
        Pb = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    Xr = xb.to(torch.float32).cpu().numpy().astype(np.float64)
    scale = np.linalg.norm(Xr, axis=1)[None, :] * np.linalg.norm(R, axis=1)[:, None]
    assert (np.abs(Pa.astype(np.float64) - Pb) <= 8e-6 * scale).all()      # (two hyperplane terms: 2^-17)
    with option("proj_bf16_terms", 3):
        Pc = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    assert (np.abs(Pc.astype(np.float64) - Pb) <= 2e-6 * scale).all()


def test_project_csr_exact(rp, ctx, oracle):
    rowptr, col, val = oracle.data_normal_sparse2(1234, 500, 40, 0.2)
    rng = np.random.default_rng(0)
    R = sparse_R(rng, 19, 40, 0.35)
    ds = rp.Dataset.csr(ctx, rowptr, col, val, 40)
    P = rp.project(ds, R, ctx=ctx)
    for c in (0, 7, 18):
        idx = np.nonzero(R[c])[0]
        for i in (0, 3, 250, 499):
            a, b = rowptr[i], rowptr[i + 1]
            assert P[c, i] == oracle.inner_ss(idx, R[c, idx], col[a:b], val[a:b])


@pytest.mark.parametrize("d,dtype", [(784, np.float64), (64, np.float64), (200, np.float32)])
def test_project_csr_dense_mfma(rp, ctx, oracle, option, d, dtype):
    """RPT_PROJ_MFMA on SVector rows (the tolerance mode): from 65 536 rows on, rows of d % 8 == 0
    elements are dense-ified as two bf16 terms and projected on the bf16 matrix pipe against the
    hyperplanes' three bf16 terms (launch_csr_dense_mfma).  Values within north_star's 1e-5 |x||r| of
    the f64 contraction — like the segmented kernel's one-FMA-per-term form (proj_csr_nodense), which
    the same call falls back to — and a forest built on them is a valid tree whose leaf assignment
    differs from the exact build's for < 1e-3 of the points."""
    n, C = 70000, 45
    rng = np.random.default_rng(7)
    nnz_row = max(3, d // 5)
    cols = np.sort(np.stack([rng.choice(d, nnz_row, replace=False) for _ in range(2000)]), axis=1)
    cols = cols[rng.integers(0, 2000, n)]
    rowptr = np.arange(n + 1, dtype=np.int64) * nnz_row
    col = cols.reshape(-1).astype(np.int32)
    val = (1.0 - rng.random(n * nnz_row)).astype(dtype)                 # U(0, 1] like C3
    R = sparse_R(rng, C, d, 0.35)
    ds = rp.Dataset.csr(ctx, rowptr, col, val, d)
    X = np.zeros((n, d))
    X[np.repeat(np.arange(n), nnz_row), col] = val.astype(np.float64)
    want = R @ X.T
    scale = np.linalg.norm(R, axis=1)[:, None] * np.linalg.norm(X, axis=1)[None, :]
    P = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    err = np.abs(P - want) / scale
    assert err.max() <= 1e-5, err.max()
    with option("proj_csr_nodense", 1):
        P2 = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    assert (np.abs(P2 - want) <= 1e-5 * scale).all()
    assert not np.array_equal(P, P2)                                    # two different kernels did run
    if d == 784:
        cfg = rp.rpTreeCfg(100, n, d)
        L, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
        Rf, _ = oracle.forest_hyperplanes(3, 2, L, pnz, d)
        fe = rp.forestBatch(0, L, 100, 2, pnz, d, ds, ctx=ctx, hyperplanes=Rf, mode=rp.RPT_PROJ_EXACT)
        fm = rp.forestBatch(0, L, 100, 2, pnz, d, ds, ctx=ctx, hyperplanes=Rf, mode=rp.RPT_PROJ_MFMA)
        leaf_off = np.array([o for (_, _, o, m, lf) in fe.topology() if lf])
        for t in range(2):
            assert np.array_equal(np.sort(fm.perm[t]), np.arange(n))
            inv_e = np.empty(n, np.int64); inv_e[fe.perm[t]] = np.arange(n)
            inv_m = np.empty(n, np.int64); inv_m[fm.perm[t]] = np.arange(n)
            flips = (np.searchsorted(leaf_off, inv_e, side="right") != np.searchsorted(leaf_off, inv_m, side="right")).mean()
            assert flips < 1e-3, flips


# ------------------------------------------------------------------ split on identical inputs
def test_split_segments_matches_partition_at_median(rp, ctx, oracle):
    rng = np.random.default_rng(11)
    n = 30000
    keys = np.round(rng.standard_normal(n), 1)       # heavy ties
    keys[rng.random(n) < 0.05] = 0.0
    keys[rng.random(n) < 0.02] = -0.0
    perm0 = rng.permutation(n).astype(np.int32)
    seg_len = [1, 2, 3, 4, 7, 8, 100, 4096, 4097, 9000, 12000]
    seg_off = np.concatenate([[0], np.cumsum(seg_len)[:-1]])
    perm, tm = rp.splitSegments(keys, perm0, seg_off, seg_len, ctx=ctx)
    for s, (o, l) in enumerate(zip(seg_off, seg_len)):
        ids = perm0[o:o + l]
        nh, order, thr, lo, hi = oracle.partition_at_median(keys[ids])
        assert np.array_equal(perm[o:o + l], ids[order]), "segment %d" % s
        assert (tm[s, 0], tm[s, 1], tm[s, 2]) == (thr, lo, hi)
    # untouched tail
    assert np.array_equal(perm[sum(seg_len):], perm0[sum(seg_len):])


# ------------------------------------------------------------------ forest build
def assert_forest_equal(f, fo):
    assert np.array_equal(f.perm, fo.perm)
    for a, b in ((f.thr, fo.thr), (f.mglo, fo.mglo), (f.mghi, fo.mghi)):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("n,d,T,min_leaf,L", [
    (1000, 16, 3, 20, None),       # SURVEY §8c golden (iii)
    (257, 5, 2, 128, 3),           # mixed leaf depths
    (20000, 32, 4, 50, None),      # big path (n > 4096) at the top levels
    (10000, 16, 2, 3000, 4),       # big nodes whose children are leaves
    (9000, 8, 2, 5000, 3),         # leaves larger than the LDS sort
    (3, 4, 2, 0, 4), (2, 4, 1, 0, 3), (1, 4, 1, 0, 2), (1, 4, 1, 5, 2), (50, 4, 2, 100, 5),
    (64, 3, 1, 1, 0),              # maxDepth 0: the root is a Tip
])
def test_forest_build_exact_identical(rp, ctx, oracle, n, d, T, min_leaf, L):
    X = oracle.data_normal_dense2(1234, n, d)
    if L is None:
        L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    else:
        pnz = 0.6
    R, _ = oracle.forest_hyperplanes(1235137, T, L, pnz, d)
    fo = oracle.forest_build_dense(X, R, min_leaf, want_proj=True)
    f = rp.forestBatch(1235137, L, min_leaf, T, pnz, d, X, ctx=ctx)
    assert np.array_equal(f.R, R)                     # host generator == oracle generator
    assert_forest_equal(f, fo)
    if L > 0 and n > min_leaf:
        P = f.proj()
        mask = ~np.isnan(fo.proj)
        assert np.array_equal(P[mask], fo.proj[mask])


def test_forest_build_sparse_with_ties_identical(rp, ctx, oracle):
    # SURVEY §8c golden (iv): sparse data x sparse hyperplanes -> many exact-zero projections
    rowptr, col, val = oracle.data_normal_sparse2(1234, 6000, 12, 0.25)
    R, _ = oracle.forest_hyperplanes(7, 3, 6, 0.3, 12)
    fo = oracle.forest_build_csr(rowptr, col, val, 12, R, 10, want_proj=True)
    assert (fo.proj[0, 0] == 0).sum() > 500
    f = rp.forestBatch(7, 6, 10, 3, 0.3, 12, (rowptr, col, val, 12), ctx=ctx)
    assert_forest_equal(f, fo)
    assert f.stats()["tie_nodes"] > 0


def test_forest_build_all_identical_points(rp, ctx, oracle):
    # every projection ties at every level: order falls back to the input order
    X = np.ones((5000, 6))
    R, _ = oracle.forest_hyperplanes(3, 2, 5, 0.9, 6)
    fo = oracle.forest_build_dense(X, R, 100)
    f = rp.forestBatch(3, 5, 100, 2, 0.9, 6, X, ctx=ctx)
    assert_forest_equal(f, fo)


def test_forest_build_mfma_mode_is_valid_tree(rp, ctx, oracle):
    """MFMA projections are not bit-identical, so compare structure: every point once per
    tree, thresholds consistent with the tree's own projections, leaf flips vs oracle rare."""
    n, d, T, min_leaf = 20000, 64, 4, 64
    X = oracle.data_normal_dense2(99, n, d)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(5, T, L, pnz, d)
    fo = oracle.forest_build_dense(X, R, min_leaf)
    f = rp.forestBatch(5, L, min_leaf, T, pnz, d, X, ctx=ctx, mode=rp.RPT_PROJ_MFMA)
    P = f.proj()
    topo = f.topology()
    for t in range(T):
        assert np.array_equal(np.sort(f.perm[t]), np.arange(n))
        for level, heap, off, m, leaf in topo:
            if leaf:
                continue
            nh = m // 2
            left = P[t, level][f.perm[t, off:off + nh]]
            right = P[t, level][f.perm[t, off + nh:off + m]]
            assert left.max() <= f.thr[t, heap] == right.min()
    # leaf assignment agreement with the exact-order oracle
    leaf_of = np.empty((T, n), dtype=np.int64)
    leaf_of_o = np.empty((T, n), dtype=np.int64)
    li = 0
    for level, heap, off, m, leaf in topo:
        if leaf:
            for t in range(T):
                leaf_of[t, f.perm[t, off:off + m]] = li
                leaf_of_o[t, fo.perm[t, off:off + m]] = li
            li += 1
    flips = (leaf_of != leaf_of_o).mean()
    assert flips < 1e-3, flips


def test_forest_build_f32(rp, ctx, oracle):
    n, d, T, min_leaf = 30000, 32, 3, 100
    X = oracle.data_normal_dense2(3, n, d).astype(np.float32)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    f = rp.forestBatch(11, L, min_leaf, T, pnz, d, X, ctx=ctx)
    P = f.proj()
    assert P.dtype == np.float32
    for t in range(T):
        assert np.array_equal(np.sort(f.perm[t]), np.arange(n))
    for level, heap, off, m, leaf in f.topology():
        if not leaf:
            nh = m // 2
            left = P[0, level][f.perm[0, off:off + nh]]
            right = P[0, level][f.perm[0, off + nh:off + m]]
            assert left.max() <= f.thr[0, heap] == right.min()


# ------------------------------------------------------------------ queries
@pytest.fixture(scope="module")
def small_forest(rp, ctx, oracle):
    n, d, T, min_leaf = 4000, 16, 5, 20
    X = oracle.data_normal_dense2(1234, n, d)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(1235137, T, L, pnz, d)
    fo = oracle.forest_build_dense(X, R, min_leaf)
    f = rp.forestBatch(1235137, L, min_leaf, T, pnz, d, X, ctx=ctx)
    Q = oracle.data_normal_dense2(4321, 32, d)
    return X, f, fo, Q


def test_candidates_identical(rp, small_forest, oracle):
    X, f, fo, Q = small_forest
    off, ids = rp.candidatesBatch(f, Q)
    for i in range(len(Q)):
        for t in range(f.T):
            want = oracle.candidates_dense(fo, Q[i], t)
            got = ids[off[i * f.T + t]:off[i * f.T + t + 1]]
            assert np.array_equal(got, want), (i, t)
    # data points as queries: proj == thr happens exactly -> the `otherwise` branch
    off, ids = rp.candidatesBatch(f, X[:16])
    for i in range(16):
        for t in range(f.T):
            want = oracle.candidates_dense(fo, X[i], t)
            assert np.array_equal(ids[off[i * f.T + t]:off[i * f.T + t + 1]], want)


@pytest.mark.parametrize("k", [1, 10, 50])
def test_knn_matches_oracle(rp, small_forest, oracle, k):
    X, f, fo, Q = small_forest
    ids, dist, cnt = rp.knnBatch(k, f, Q)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert cnt[i] == len(wi)
        assert np.array_equal(ids[i, :cnt[i]], wi), i       # duplicates kept, same order
        assert np.array_equal(dist[i, :cnt[i]], wd)   # the reference's bits
    # single-query reference-shaped call
    hits = rp.knn(rp.metricL2, k, f, rp.fromListDv(Q[0]))
    assert [h[1] for h in hits] == oracle.knn_dense(fo, X, Q[0], k)[0].tolist()


def test_knn_dedup(rp, small_forest, oracle):
    X, f, fo, Q = small_forest
    ids, dist, cnt = rp.knnBatch(10, f, Q, dedup=True)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(fo, X, Q[i], 10, dedup=True)
        assert np.array_equal(ids[i, :cnt[i]], wi)
        assert len(set(ids[i, :cnt[i]].tolist())) == cnt[i]


def test_knn_more_than_candidates(rp, ctx, oracle):
    X = oracle.data_normal_dense2(1, 40, 4)
    f = rp.forestBatch(2, 2, 5, 1, 1.0, 4, X, ctx=ctx)
    ids, dist, cnt = rp.knnBatch(30, f, X[:3])
    assert (cnt < 30).all() and (ids[np.arange(3), cnt] == -1).all()
    assert np.isinf(dist[0, cnt[0]])


def test_two_discs_reference_test(rp, ctx, oracle):
    # test/Data/RPTreeSpec.hs:50-85 on the device path
    n, T, min_leaf, k = 10000, 10, 20, 5
    X = oracle.data_circle2d2(42, n)
    cfg = rp.rpTreeCfg(min_leaf, n, 2)
    tts = rp.forestBatch(42, cfg.fpMaxTreeDepth, min_leaf, T, 1.0, 2, X, ctx=ctx)
    assert all(rp.treeSize(t) == n for t in tts)
    hits = rp.knn(rp.metricL2, k, tts, rp.fromListDv([0, 0]))
    assert max(h[0] for h in hits) < 1


def test_recall_with_matches_oracle(rp, small_forest, oracle):
    X, f, fo, Q = small_forest
    for i in range(4):
        assert rp.recallWith(rp.metricL2, f, 10, Q[i]) == pytest.approx(
            oracle.recall_with_dense(fo, X, Q[i], 10), abs=1e-12)


def test_brute_knn(rp, small_forest, oracle):
    X, f, fo, Q = small_forest
    ids, dist = rp.bruteKnn(f, Q[:8], 10)
    for i in range(8):
        wi, wd = oracle.brute_knn_dense(X, Q[i], 10)
        assert np.array_equal(ids[i], wi)
        assert np.array_equal(dist[i], wd)


def test_knn_csr(rp, ctx, oracle):
    n, d = 3000, 30
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.3)
    R, _ = oracle.forest_hyperplanes(9, 4, 6, 0.5, d)
    fo = oracle.forest_build_csr(rowptr, col, val, d, R, 25)
    f = rp.forestBatch(9, 6, 25, 4, 0.5, d, (rowptr, col, val, d), ctx=ctx)
    assert np.array_equal(f.perm, fo.perm)
    qr, qc, qv = oracle.data_normal_sparse2(6, 10, d, 0.3)
    ids, dist, cnt = rp.knnBatch(7, f, (qr, qc, qv, d))
    for i in range(10):
        a, b = qr[i], qr[i + 1]
        wi, wd = oracle.knn_csr(fo, rowptr, col, val, qc[a:b], qv[a:b], 7, true_l2=True)
        assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-9, atol=1e-12)
        # ids agree wherever distances are separated (sparse rows can tie exactly)
        if len(set(np.round(wd, 9))) == len(wd):
            assert np.array_equal(ids[i, :cnt[i]], wi)


def test_knn_merge_shards(rp, ctx, small_forest, oracle):
    """Multi-GPU merge on one device: two shard forests (trees 0-1 | 2-4) merged == full knn."""
    import ctypes as C
    import torch
    from rptree_amd import _lib
    X, f, fo, Q = small_forest
    k, nq = 10, len(Q)
    R = f.R
    fa = rp.forestBatch(0, f.L, f.min_leaf, 2, 0, f.d, X, ctx=ctx, hyperplanes=R[:2])
    fb = rp.forestBatch(0, f.L, f.min_leaf, 3, 0, f.d, X, ctx=ctx, hyperplanes=R[2:])
    parts = [rp.knnBatch(k, fa, Q), rp.knnBatch(k, fb, Q)]
    ids = torch.tensor(np.stack([p[0] for p in parts])).cuda()
    dist = torch.tensor(np.stack([p[1] for p in parts])).cuda()
    cnt = torch.tensor(np.stack([p[2] for p in parts])).cuda()
    oi = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rpt_knn_merge_dev(ctx._h, ids.data_ptr(), dist.data_ptr(),
                                            cnt.data_ptr(), 2, nq, k, 0, oi.data_ptr(),
                                            od.data_ptr(), oc.data_ptr()))
    ctx.sync()
    full = rp.knnBatch(k, f, Q)
    assert np.array_equal(oi.cpu().numpy(), full[0])
    assert np.array_equal(od.cpu().numpy(), full[1])


def test_knn_merge_records_equals_merge(rp, ctx, small_forest):
    """rpt_knn_merge_records_dev over packed exchange records (one all-gather) == rpt_knn_merge_dev
    over the three shard-major arrays; the record is filled directly by rpt_knn_dev."""
    import os
    import sys
    import torch
    from rptree_amd import _lib, sharded
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_rehearsal import ExchangeRecord
    X, f, fo, Q = small_forest
    k, nq = 10, len(Q)
    R = f.R
    shards = [rp.forestBatch(0, f.L, f.min_leaf, 2, 0, f.d, X, ctx=ctx, hyperplanes=R[:2]),
              rp.forestBatch(0, f.L, f.min_leaf, 3, 0, f.d, X, ctx=ctx, hyperplanes=R[2:])]
    ds, qs = rp.Dataset.of(ctx, X), rp.Dataset.of(ctx, Q)
    L_ = _lib.lib()
    recs = []
    for sh in shards:
        rec = ExchangeRecord(nq, k, torch.device("cuda", 0))
        _lib.check(L_.rpt_knn_dev(ctx._h, sh._h, ds._h, qs._h, k, 0, rec.ids.data_ptr(),
                                  rec.dist.data_ptr(), rec.count.data_ptr()))
        recs.append(rec)
    ctx.sync()
    assert recs[0].bytes % 16 == 0
    gathered = torch.stack([r.buf for r in recs]).contiguous()       # what the all-gather returns
    for g, sh in enumerate(shards):                                  # the views address shard g
        want = rp.knnBatch(k, sh, Q)
        vi, vd, vc = ExchangeRecord.views_of(gathered, g, nq, k)
        assert np.array_equal(vi.cpu().numpy(), want[0]) and np.array_equal(vd.cpu().numpy(), want[1])
        assert np.array_equal(vc.cpu().numpy(), want[2])
    outs = []
    for packed in (True, False):
        oi = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
        oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        if packed:
            _lib.check(L_.rpt_knn_merge_records_dev(ctx._h, gathered.data_ptr(), recs[0].bytes, 2,
                                                    nq, k, 0, oi.data_ptr(), od.data_ptr(),
                                                    oc.data_ptr()))
        else:
            ids = torch.stack([r.ids for r in recs]).contiguous()
            dd = torch.stack([r.dist for r in recs]).contiguous()
            cc = torch.stack([r.count for r in recs]).contiguous()
            torch.cuda.synchronize()
            _lib.check(L_.rpt_knn_merge_dev(ctx._h, ids.data_ptr(), dd.data_ptr(), cc.data_ptr(), 2,
                                            nq, k, 0, oi.data_ptr(), od.data_ptr(), oc.data_ptr()))
        ctx.sync()
        outs.append((oi.cpu().numpy(), od.cpu().numpy(), oc.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    full = rp.knnBatch(k, f, Q)
    assert np.array_equal(outs[0][0], full[0]) and np.array_equal(outs[0][1], full[1])
    with pytest.raises(_lib.RPTError):                               # record smaller than the layout
        _lib.check(L_.rpt_knn_merge_records_dev(ctx._h, gathered.data_ptr(), 8, 2, nq, k, 0,
                                                oi.data_ptr(), od.data_ptr(), oc.data_ptr()))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_knn_wave_and_workgroup_variants_agree_with_oracle(rp, ctx, oracle, option, dtype):
    """The fused query kernel has a one-wave-per-query variant (small shards) and a
    one-workgroup-per-query variant; both must give the oracle's ids/distances, including
    batches beyond the per-wave capacity (512 candidates), trees that outgrow their range
    slots (second traversal pass) and every duplicate rule."""
    n, d, T, ml, k = 6000, 24, 6, 60, 12
    X = oracle.data_normal_dense2(77, n, d)
    Q = oracle.data_normal_dense2(78, 40, d)
    if dtype == "f32":
        X, Q = X.astype(np.float32), Q.astype(np.float32)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(5, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    res = {}
    for force in ("1", "0"):
        with option("knn_wave", int(force)):
            res[force] = [rp.knnBatch(k, f, Q, dedup=m) for m in (0, 1, 2)]
    for m in range(3):
        for a, b in zip(res["1"][m], res["0"][m]):
            assert np.array_equal(a, b)
    if dtype == "f64":
        fo = oracle.forest_build_dense(X, R, ml)
        ids, dist, cnt = res["1"][0]
        for i in range(len(Q)):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-12, atol=0)
    # a wide-margin forest: many leaves per tree per query (slots overflow -> second pass)
    X2 = oracle.data_normal_dense2(79, 3000, 4)
    if dtype == "f32":
        X2 = X2.astype(np.float32)
    L2, _, pnz2 = oracle.tree_cfg(8, 3000, 4)
    R2, _ = oracle.forest_hyperplanes(6, 16, L2, pnz2, 4)
    f2 = rp.forestBatch(0, L2, 8, 16, pnz2, 4, X2, ctx=ctx, hyperplanes=R2, mode=rp.RPT_PROJ_EXACT)
    Q2 = X2[:64] * 1.0001
    out = {}
    for force in ("1", "0"):
        with option("knn_wave", int(force)):
            out[force] = rp.knnBatch(5, f2, Q2)
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("decimals", [None, 3, 1])
def test_large_pivot_bins_selection_path(rp, ctx, oracle, option, decimals):
    """Pivot bins above 128 points at the first levels (400 000 points: ~250 per bin) are split
    by SELECTION (sub-histogram of the bin, exact sort of the one sub-bin around the threshold);
    continuous keys, keys with moderate ties (3 decimals: ties inside the sorted sub-bin, decided
    by the earlier levels' keys) and heavy ties (1 decimal: the candidate set outgrows a wave and
    the full LDS sort takes over) must all give the oracle's forest, as must the build with the
    selection switched off."""
    n, d, T, ml = 400_000, 6, 2, 64
    X = oracle.data_normal_dense2(31, n, d)
    if decimals is not None:
        X = np.round(X, decimals)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    L = min(L, 9)
    R, _ = oracle.forest_hyperplanes(8, T, L, 1.0, d)
    if decimals is not None:
        R = np.round(R, 1)          # projections of rounded data on rounded hyperplanes tie often
    fo = oracle.forest_build_dense(X, R, ml)
    f = rp.forestBatch(0, L, ml, T, 1.0, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    assert_forest_equal(f, fo)
    with option("no_midselect", 1):
        g = rp.forestBatch(0, L, ml, T, 1.0, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    assert_forest_equal(g, fo)


@pytest.mark.parametrize("kind", ["cont", "ties", "dups"])
@pytest.mark.parametrize("k", [1, 10, 24])
def test_knn_f32_prefilter_is_exact(rp, ctx, oracle, option, kind, k):
    """f64 data, duplicates kept: the fused kernel ranks the candidates on an f32 shadow of X,
    computes exact distances for the best k + max(6, k/2) only and certifies the cut per query
    (falling back to the exact path when it cannot: heavy ties).  ids, distances and counts must
    equal the oracle's and the all-f64 kernel's, bit for bit — on continuous data, on rounded
    data (many exactly equal distances) and on data with repeated points."""
    n, d, T, ml = 20000, 16, 12, 100          # 12 x 100 candidates: the workgroup kernel
    if k == 10:
        T, ml = 6, 100                        # ... and 6 x 100: the one-wave-per-query kernel
    X = oracle.data_normal_dense2(55, n, d)
    if kind == "ties":
        X = np.round(X * 2) / 2
    elif kind == "dups":
        X[n // 2:] = X[:n - n // 2]             # every point twice
    rng = np.random.default_rng(3)
    Q = X[rng.integers(0, n, 48)] + (0.0 if kind != "cont" else 0.003)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(21, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    got = rp.knnBatch(k, f, Q)
    import ctypes as C
    from rptree_amd import _lib
    unc = C.c_int64(-1)
    _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    if kind == "cont" and k == 24:
        assert unc.value <= len(Q) // 8         # (the int8 tier: coarse for 16-element rows)
    # (k = 1: a query is a data point + 0.003, its source is found once per tree, and a cut inside
    # that group of equal distances cannot be certified: those queries are re-run in f64)
    with option("knn_no_pre32", 1):
        ref = rp.knnBatch(k, f, Q)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    with option("knn_no_pre16", 1):          # the f32 shadow alone
        g32 = rp.knnBatch(k, f, Q)
    for a, b in zip(g32, ref):
        assert np.array_equal(a, b)
    with option("knn_no_pre8", 1):           # the half shadow first (the default ranks on the int8 one)
        g16 = rp.knnBatch(k, f, Q)
        _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    if kind == "cont" and k == 24:
        assert unc.value == 0                   # every cut certified
    for a, b in zip(g16, ref):
        assert np.array_equal(a, b)
    fo = oracle.forest_build_dense(X, R, ml)
    ids, dist, cnt = got
    for i in range(0, len(Q), 4):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
        assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("path", ["default", "pre32", "no_pre32", "wave", "general"])
def test_knn_cut_between_two_candidates_an_ulp_apart(rp, ctx, oracle, option, path):
    """RPTree.hs:174 ranks ALL candidates on metricDDL2's left fold.  The batched distance passes
    reduce a row by a lane butterfly — the same value to an ulp, not the same bits — so the device
    keeps k + 8 candidates by that value, re-evaluates them as the fold and selects on THOSE bits.
    Here the cut falls, for every odd k, between two DIFFERENT rows whose squared terms are a
    permutation of each other (mathematically equal distances, sums that round differently): the
    member the reference keeps must be the one returned, on every f64 query path."""
    d, npairs, k_values = 128, 36, list(range(1, 64, 2))
    rng = np.random.default_rng(17)
    rows = []
    for j in range(npairs):
        x = rng.standard_normal(d)
        x *= (1.0 + 1e-3 * j) / np.linalg.norm(x)            # pair j clearly beyond pair j - 1
        rows += [x, x[rng.permutation(d)]]
    far = rng.standard_normal((300, d)) * 3.0
    X = np.ascontiguousarray(np.vstack([np.array(rows), far]))
    q = np.zeros((1, d))
    R = np.zeros((1, 0, d))                                  # maxDepth 0: ONE Tip, every point a candidate
    fo = oracle.forest_build_dense(X, R, 0)
    lf = np.array([oracle.metric_dd(x, q[0]) for x in X[:2 * npairs]])
    differ = int((lf[0::2] != lf[1::2]).sum())
    swapped = int((lf[1::2] < lf[0::2]).sum())
    assert differ >= 5 and swapped >= 2, (differ, swapped)   # the data really exercises the cut
    opts = {"default": {}, "pre32": {"knn_no_pre16": 1}, "no_pre32": {"knn_no_pre32": 1},
            "wave": {"knn_no_pre32": 1, "knn_wave": 1},
            "general": {"knn_general": 1}}[path]
    import contextlib
    with contextlib.ExitStack() as st:
        for name, v in opts.items():
            st.enter_context(option(name, v))
        f = rp.forestBatch(0, 0, 0, 1, 1.0, d, X, ctx=ctx, hyperplanes=R)
        for k in k_values:
            for dedup in (0, 1):
                ids, dist, cnt = rp.knnBatch(k, f, q, dedup=dedup)
                wi, wd = oracle.knn_dense(fo, X, q[0], k, dedup=dedup)
                assert cnt[0] == len(wi) and np.array_equal(ids[0, :cnt[0]], wi), (path, k, dedup)
                assert np.array_equal(dist[0, :cnt[0]], wd), (path, k, dedup)
            # knnPQ's nub (one entry per distance): two rows an ulp apart under one summation order
            # and equal under the other may or may not collapse (include/rptree_hip.h); what is
            # returned is strictly ascending and carries the fold's bits of the rows it names
            ids, dist, cnt = rp.knnBatch(k, f, q, dedup=rp.RPT_KNN_DEDUP_DISTANCE)
            m = int(cnt[0])
            assert m == k and (np.diff(dist[0, :m]) > 0).all()
            assert all(dist[0, i] == oracle.metric_dd(X[ids[0, i]], q[0]) for i in range(m))


def test_knn_f32_prefilter_uncertified_queries_rerun(rp, ctx, oracle):
    """Half of the points are one and the same point: for queries near it every distance at the
    prefilter's cut is equal, nothing can be certified, and exactly those queries are answered
    again by the all-f64 kernel — the others keep their prefiltered answers; all match the oracle."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml, k = 8000, 8, 12, 100, 10
    X = oracle.data_normal_dense2(91, n, d)
    X[n // 2:] = 1.0
    Q = np.concatenate([X[:20] + 0.001, np.ones((12, d))])
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(2, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    ids, dist, cnt = rp.knnBatch(k, f, Q)
    unc = C.c_int64(-1)
    _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    assert 12 <= unc.value < len(Q)
    fo = oracle.forest_build_dense(X, R, ml)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
        assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-12, atol=1e-15)


def test_knn_f32_prefilter_switches_itself_off_on_self_queries(rp, ctx, oracle):
    """Queries that ARE data points are found once per tree with equal distances; when more than a
    quarter of a batch cannot be certified the forest drops one ranking tier for the later batches
    (half shadow -> f32 shadow -> none: the third batch reports 0 uncertified because it never
    tries), and the answers stay the oracle's."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml, k = 8000, 8, 64, 100, 4      # 64 copies of the query's own point: more than either tier
                                              # keeps, also on its second, three times wider attempt
    X = oracle.data_normal_dense2(17, n, d)
    Q = X[:24].copy()
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(6, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X, R, ml)
    seen, tiers = [], []
    for rnd in range(3):
        ids, dist, cnt = rp.knnBatch(k, f, Q)
        unc, tier = C.c_int64(-1), C.c_int32(-1)
        _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        seen.append(unc.value)
        tiers.append(tier.value)
        for i in range(len(Q)):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.array_equal(dist[i, :cnt[i]], wd)
    assert tiers == [2, 1, 0], tiers
    assert seen[0] > len(Q) // 4 and seen[1] > len(Q) // 4 and seen[2] == 0


@pytest.mark.parametrize("shape", [(20000, 128, 12, 100), (20000, 128, 6, 100), (6000, 208, 12, 100),
                                   (3000, 1040, 6, 100)])
@pytest.mark.parametrize("kind", ["cont", "ties", "clip"])
def test_knn_int8_tier_is_exact(rp, ctx, oracle, option, shape, kind):
    """First ranking tier of dense f64 data with rows of 16 n elements: an int8 shadow under ONE
    scale, the query quantised alike, candidates ranked on the exact integer sum (qc - c)^2; the
    cut is certified through the triangle inequality (|q - s qc| + max row error) and uncertified
    queries are answered by the exact kernel.  Answers equal the all-f64 kernel's and the oracle's
    bit for bit: workgroup and wave kernel, rows of 8 / 13 / 65 sixteen-byte pieces, continuous and
    rounded data, and queries far outside the data's range (their elements clip at +-127)."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml = shape
    X = oracle.data_normal_dense2(31, n, d)
    if kind == "ties":
        X = np.round(X * 2) / 2
    rng = np.random.default_rng(11)
    Q = X[rng.integers(0, n, 40)] + 0.003
    if kind == "clip":
        Q[::3] *= 40.0                           # far beyond max |x|: clipped, eq is large, not certified
    cfg = rp.rpTreeCfg(ml, n, d)
    L, pnz = min(cfg.fpMaxTreeDepth, 8), cfg.fpProjNzDensity
    R, _ = oracle.forest_hyperplanes(5, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X, R, ml)
    for k in (1, 10, 24):
        got = rp.knnBatch(k, f, Q)
        tier, unc = C.c_int32(-1), C.c_int64(-1)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
        if tier.value != 3:                      # (a batch that failed too many cuts demotes the forest:
            assert kind == "clip" or d > 128, (tier.value, kind, k)   # clipped queries; long mixture rows)
        if kind == "cont" and d == 128:
            assert unc.value <= len(Q) // 4, unc.value
        with option("knn_no_pre32", 1):
            ref = rp.knnBatch(k, f, Q)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (k, kind)
        ids, dist, cnt = got
        for i in range(0, len(Q), 5):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.array_equal(dist[i, :cnt[i]], wd)


def test_knn_tiers_demote_one_at_a_time(rp, ctx, oracle, option):
    """Rows of 16 elements have all four tiers; self queries (64 copies of the nearest point across
    every cut; the int8 tier told to keep 12 instead of its 68) fail each in turn: int8 -> half ->
    f32 -> exact, the answers the oracle's throughout."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml, k = 8000, 16, 64, 100, 4     # (64 copies: beyond the second, three times wider attempt too)
    X = oracle.data_normal_dense2(17, n, d)
    Q = X[:24].copy()
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(6, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X, R, ml)
    tiers = []
    for rnd in range(4):
        with option("knn_kp8", 12):
            ids, dist, cnt = rp.knnBatch(k, f, Q)
        tier = C.c_int32(-1)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        tiers.append(tier.value)
        for i in range(len(Q)):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.array_equal(dist[i, :cnt[i]], wd)
    assert tiers == [3, 2, 1, 0], tiers


def test_knn_f32_prefilter_out_of_range_data(rp, ctx, oracle):
    """Values whose squares leave the f32 range (the shadow would hold inf): no prefilter, the
    all-f64 kernel answers; tiny values (f32 subnormals): certified or sent to the exact path —
    the oracle's result either way."""
    n, d, T, ml, k = 6000, 8, 12, 100, 5
    base = oracle.data_normal_dense2(9, n, d)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(4, T, L, pnz, d)
    for scale in (1e30, 1e17, 1e-19, 1e-21, 1e-42):
        X = base * scale
        Q = X[:16].copy()
        f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
        ids, dist, cnt = rp.knnBatch(k, f, Q)
        fo = oracle.forest_build_dense(X, R, ml)
        for i in range(len(Q)):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.array_equal(dist[i, :cnt[i]], wd) or np.allclose(dist[i, :cnt[i]], wd, rtol=1e-12, atol=0)


def test_forest_save_load_roundtrip(rp, ctx, small_forest, tmp_path):
    X, f, fo, Q = small_forest
    path = str(tmp_path / "forest.npz")
    rp.saveForest(path, f)
    g = rp.loadForest(path, X, ctx=ctx)
    assert np.array_equal(g.perm, f.perm) and np.array_equal(g.thr, f.thr, equal_nan=True)
    a, b = rp.knnBatch(10, f, Q), rp.knnBatch(10, g, Q)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    off_a, ids_a = rp.candidatesBatch(f, Q)
    off_b, ids_b = rp.candidatesBatch(g, Q)
    assert np.array_equal(off_a, off_b) and np.array_equal(ids_a, ids_b)


def assert_stream_equal(f, so, T):
    """device streamed forest == rpo_stream_forest_dense's heap arrays, bit for bit"""
    for t in range(T):
        assert np.array_equal(f.kind, so.kind[t])
        assert np.array_equal(f.leaf_off, so.leaf_off[t]) and np.array_equal(f.leaf_len, so.leaf_len[t])
        assert f.held == so.held[t]
        assert np.array_equal(f.perm[t, :f.held], so.leaf_ids[t, :f.held]), "tree %d" % t
        for name in ("thr", "mglo", "mghi"):
            assert np.array_equal(getattr(f, name)[t], getattr(so, name)[t], equal_nan=True), (name, t)


@pytest.mark.parametrize("n,d,T,ml,chunk", [
    (3000, 2, 4, 20, 30),        # the reference's own test shape: rpTreeCfg chunk = n / 100
    (4000, 12, 3, 20, 100),      # chunk divides n: nothing lost
    (4000, 12, 3, 20, 33),       # short last chunk: the data-loss branch (Internal.hs:277) fires
    (4000, 12, 2, 20, 3999),     # a last chunk of ONE point drops about half of the tree
    (5000, 8, 2, 0, 64),         # minLeaf 0: splits down to single points
    (20000, 16, 3, 50, 7000),    # chunk parts above the LDS sort's 4096 points (HBM merge path)
    (6000, 5, 2, 10, 6000),      # one chunk
])
def test_streaming_forest_is_the_reference_fold_over_chunks(rp, ctx, oracle, n, d, T, ml, chunk):
    """Conduit.hs:104-121 `forest` = chunkedAccum folding `insert` (Internal.hs:245-297): chunk-own
    medians averaged into the thresholds (:281), margins by (max, min) (:280), new points in front
    of a Tip's (:286), an empty chunk half dropping the subtree (:277) — device == oracle."""
    X = oracle.data_circle2d2(3, n) if d == 2 else oracle.data_normal_dense2(11, n, d)
    L, _, pnz = oracle.tree_cfg(max(ml, 1), n, d)
    R, _ = oracle.forest_hyperplanes(5, T, L, pnz, d)
    so = oracle.stream_forest_dense(X, R, ml, chunk)
    f = rp.forest(0, L, ml, T, chunk, pnz, d, X, ctx=ctx, hyperplanes=R)
    assert_stream_equal(f, so, T)
    assert f.dropped >= n - f.held
    if chunk >= n:                                   # one chunk = the batch build (:285-295)
        fb = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R)
        for t in range(T):
            got = np.concatenate(rp.leaves(f[t])) if n else np.zeros(0, np.int32)
            assert np.array_equal(got, fb.perm[t])
    elif n % chunk == 0:
        assert f.held == n and all(rp.treeSize(t) == n for t in f)   # RPTreeSpec.hs:93-94
        fo = oracle.forest_build_dense(X, R, ml)
        assert not np.array_equal(f.thr[0, 0], fo.thr[0, 0])        # NOT the batch thresholds
    elif chunk in (33, 3999):
        assert f.held < n                            # the reference's data-loss quirk, reproduced


def test_queries_on_a_streamed_forest(rp, ctx, oracle):
    """candidates / knn walk whatever tree they are given (RPTree.hs:168-176, 289-314): on the
    explicit topology of a streamed forest the device answers == the oracle's walk of the same
    heap arrays; a one-shot generator works as the source (Conduit)."""
    n, d, T, ml, chunk, k = 6000, 6, 4, 25, 60, 7
    X = oracle.data_normal_dense2(21, n, d)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    src = (rp.Embed(rp.fromListDv(x), ()) for x in X)
    f = rp.forest(0, L, ml, T, chunk, pnz, d, src, ctx=ctx, hyperplanes=R)
    so = oracle.stream_forest_dense(X, R, ml, chunk)
    assert_stream_equal(f, so, T)
    Q = oracle.data_normal_dense2(22, 40, d)
    off, cids = rp.candidatesBatch(f, Q)
    for dedup in (False, True):
        ids, dist, cnt = rp.knnBatch(k, f, Q, dedup=dedup)
        for i in range(len(Q)):
            for t in range(T):
                want = oracle.stream_candidates_dense(so, R, Q[i], t)
                assert np.array_equal(cids[off[i * T + t]:off[i * T + t + 1]], want)
            wi, wd = oracle.stream_knn_dense(so, R, X, Q[i], k, dedup=int(dedup))
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
            assert np.array_equal(dist[i, :cnt[i]], wd)
    # knnH walks any RPT too (RPTree.hs:199-217 over candidatesH :318-342): whole buckets in increasing
    # margin priority, the bucket taken last first, unsorted, not cut to k
    for kk in (1, k, 60, 400):
        hoff, hids, hdist = rp.knnHBatch(kk, f, Q[:16])
        for i in range(16):
            wi, wd = oracle.stream_knn_h_dense(so, R, X, Q[i], kk)
            assert np.array_equal(hids[hoff[i]:hoff[i + 1]], wi), (kk, i)
            assert np.array_equal(hdist[hoff[i]:hoff[i + 1]], wd)


@pytest.mark.parametrize("chunk", [100, 37, 3000])
def test_streaming_forest_of_svector_rows(rp, ctx, oracle, chunk):
    """`forest` is polymorphic in `Inner SVector v` (Conduit.hs:104-113): SVector rows go through the
    same fold of `insert` over chunks with innerSS as the inner product (Internal.hs:351-366).  Device
    == the oracle's fold bit for bit (kinds, offsets, Tip contents, thresholds, margins) when the chunk
    divides n, when it does not (data-loss branch) and for one chunk; candidates / kNN on the result
    == the oracle's walk wherever the distances separate."""
    n, d, T, ml, k = 3000, 24, 3, 20, 6
    rowptr, col, val = oracle.data_normal_sparse2(15, n, d, 0.3)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(5, T, L, pnz, d)
    so = oracle.stream_forest_csr(rowptr, col, val, d, R, ml, chunk)
    f = rp.forest(0, L, ml, T, chunk, pnz, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
    assert_stream_equal(f, so, T)
    if chunk >= n:
        fb = rp.forestBatch(0, L, ml, T, pnz, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
        for t in range(T):
            assert np.array_equal(np.concatenate(rp.leaves(f[t])), fb.perm[t])
    elif n % chunk == 0:
        assert f.held == n
    else:
        assert f.held < n                            # the reference's data-loss quirk on SVector rows too
    # queries: SVector queries against the oracle's walk with the query dense-ified (a zero of the query
    # contributes an exact zero to innerSD's sum: the same projections, the same decisions)
    qr, qc, qv = oracle.data_normal_sparse2(16, 8, d, 0.3)
    Qd = np.zeros((8, d))
    for i in range(8):
        Qd[i, qc[qr[i]:qr[i + 1]]] = qv[qr[i]:qr[i + 1]]
    off, cids = rp.candidatesBatch(f, (qr, qc, qv, d))
    ids, dist, cnt = rp.knnBatch(k, f, (qr, qc, qv, d))
    for i in range(8):
        for t in range(T):
            want = oracle.stream_candidates_dense(so, R, Qd[i], t)
            assert np.array_equal(cids[off[i * T + t]:off[i * T + t + 1]], want), (i, t)
        cand = cids[off[i * T]:off[(i + 1) * T]]
        assert cnt[i] == min(k, len(cand)) and set(ids[i, :cnt[i]].tolist()) <= set(cand.tolist())
        assert (np.diff(dist[i, :cnt[i]]) >= 0).all()


# ------------------------------------------------------------------ fallback paths
@pytest.mark.parametrize("n", [1000, 1800, 3000])
def test_heavy_ties_take_the_fallback_paths(rp, ctx, oracle, n):
    """Pivot bins larger than a wave slab (wsub -> subtree_kernel), than the block LDS
    (streaming -> general path) ... all must still reproduce the reference order."""
    rng = np.random.default_rng(n)
    X = np.round(rng.standard_normal((n, 3)), 0)           # ~7 distinct values per coordinate
    X[: n // 3] = 1.0                                      # one third identical points
    R, _ = oracle.forest_hyperplanes(11, 3, 7, 0.7, 3)
    fo = oracle.forest_build_dense(X, R, 8)
    f = rp.forestBatch(0, 7, 8, 3, 0, 3, X, ctx=ctx, hyperplanes=R)
    assert_forest_equal(f, fo)
    off, ids = rp.candidatesBatch(f, X[:5])
    for i in range(5):
        for t in range(3):
            assert np.array_equal(ids[off[i * 3 + t]:off[i * 3 + t + 1]],
                                  oracle.candidates_dense(fo, X[i], t))


def test_knn_general_path_large_k_and_many_ranges(rp, ctx, small_forest, oracle):
    X, f, fo, Q = small_forest
    ids, dist, cnt = rp.knnBatch(100, f, Q[:8])              # k > 64: bitonic-merge path
    for i in range(8):
        wi, wd = oracle.knn_dense(fo, X, Q[i], 100)
        assert np.array_equal(ids[i, :cnt[i]], wi)
    # more leaf ranges per query than the fused kernel's LDS slab (512): 600 one-level trees
    Xs = oracle.data_normal_dense2(5, 300, 4)
    R, _ = oracle.forest_hyperplanes(3, 600, 1, 1.0, 4)
    fs = rp.forestBatch(0, 1, 10, 600, 0, 4, Xs, ctx=ctx, hyperplanes=R)
    fos = oracle.forest_build_dense(Xs, R, 10)
    ids, dist, cnt = rp.knnBatch(5, fs, Xs[:4])
    for i in range(4):
        wi, wd = oracle.knn_dense(fos, Xs, Xs[i], 5)
        assert np.array_equal(ids[i, :cnt[i]], wi)


def test_knn_f32_data(rp, ctx, oracle):
    n, d = 5000, 24
    X = oracle.data_normal_dense2(8, n, d).astype(np.float32)
    cfg = rp.rpTreeCfg(30, n, d)
    f = rp.forestBatch(4, cfg.fpMaxTreeDepth, 30, 4, cfg.fpProjNzDensity, d, X, ctx=ctx)
    ids, dist, cnt = rp.knnBatch(5, f, X[:50])
    assert (ids[:, 0] == np.arange(50)).all() and (dist[:, 0] == 0).all()
    bi, bd = rp.bruteKnn(f, X[:50], 5)
    # every returned neighbour's distance is a true distance (f32 arithmetic)
    ref = np.sqrt(((X[ids[:, 1]].astype(np.float64) - X[:50].astype(np.float64)) ** 2).sum(1))
    assert np.allclose(dist[:, 1], ref, rtol=1e-5)
    assert (bi[:, 0] == np.arange(50)).all()


@pytest.mark.parametrize("shape", [(20000, 16, 12, 100), (20000, 16, 6, 100), (12000, 200, 8, 200),
                                   (6000, 208, 4, 200), (4000, 1040, 8, 200)])
@pytest.mark.parametrize("kind", ["cont", "ties", "self"])
def test_knn_f32_data_half_shadow_tier_changes_nothing(rp, ctx, option, oracle, shape, kind):
    """f32 data are ranked on an IEEE-half shadow first; the kept rows get the f32 distance the
    all-f32 kernel ranks every candidate on and the cut is certified per query (uncertified
    queries are answered again by the all-f32 kernel).  ids, distances and counts are the
    all-f32 kernel's bit for bit — workgroup kernel, wave kernel, long rows; continuous data,
    rounded data (many equal distances), queries that are data points."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml = shape
    X = oracle.data_normal_dense2(77, n, d).astype(np.float32)
    if kind == "ties":
        X = (np.round(X * 2) / 2).astype(np.float32)
    rng = np.random.default_rng(5)
    Q = X[rng.integers(0, n, 64)].copy()
    if kind != "self":
        Q = (Q + np.float32(0.003)).astype(np.float32)
    cfg = rp.rpTreeCfg(ml, n, d)
    f = rp.forestBatch(9, cfg.fpMaxTreeDepth, ml, T, cfg.fpProjNzDensity, d, X, ctx=ctx)
    for k in (1, 10, 24):
        got = rp.knnBatch(k, f, Q)
        tier, unc = C.c_int32(-1), C.c_int64(-1)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
        if k == 1:                                           # (later k: a demotion may have happened)
            assert tier.value == (3 if d % 16 == 0 else 2)   # the int8 (rows of 16 n bytes) / half tier ran
        if kind == "cont" and k == 24 and tier.value == 2:    # (the int8 tier's cut is wider: it may demote)
            assert unc.value <= len(Q) // 8
        with option("knn_no_pre16", 1):
            ref = rp.knnBatch(k, f, Q)
            _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
            assert tier.value == 0
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (k, kind)
        with option("knn_no_pre8", 1):
            g16 = rp.knnBatch(k, f, Q)
        for a, b in zip(g16, ref):
            assert np.array_equal(a, b), (k, kind)


# ------------------------------------------------------------------ knnH / knnPQ
def test_knnh_matches_oracle(rp, ctx, small_forest, oracle):
    """knnH (RPTree.hs:199-217): whole buckets of the lowest-margin-priority leaves, the bucket
    taken last first, unsorted and not cut to k."""
    X, f, fo, Q = small_forest
    for k in (1, 10, 60, 200):
        off, ids, dist = rp.knnHBatch(k, f, Q[:12])
        for i in range(12):
            wi, wd = oracle.knn_h_dense(fo, X, Q[i], k)
            got = ids[off[i]:off[i + 1]]
            assert np.array_equal(got, wi), (k, i)
            assert np.array_equal(dist[off[i]:off[i + 1]], wd)   # the reference's bits
    hits = rp.knnH(rp.metricL2, 10, f, Q[0])
    wi, wd = oracle.knn_h_dense(fo, X, Q[0], 10)
    assert [i for _, i in hits] == wi.tolist()


def test_knnh_sparse_matches_oracle(rp, ctx, oracle):
    n, d = 3000, 30
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.3)
    R, _ = oracle.forest_hyperplanes(9, 4, 6, 0.5, d)
    fo = oracle.forest_build_csr(rowptr, col, val, d, R, 25)
    f = rp.forestBatch(9, 6, 25, 4, 0.5, d, (rowptr, col, val, d), ctx=ctx)
    assert np.array_equal(f.perm, fo.perm)
    qr, qc, qv = oracle.data_normal_sparse2(6, 10, d, 0.3)
    off, ids, dist = rp.knnHBatch(30, f, (qr, qc, qv, d))
    for i in range(10):
        a, b = qr[i], qr[i + 1]
        wi, wd = oracle.knn_h_csr(fo, rowptr, col, val, qc[a:b], qv[a:b], 30, true_l2=True)
        assert np.array_equal(ids[off[i]:off[i + 1]], wi)
        assert np.allclose(dist[off[i]:off[i + 1]], wd, rtol=1e-9, atol=1e-12)


def test_knnpq_collapses_equal_distances(rp, ctx, small_forest, oracle):
    """knnPQ (RPTree.hs:181-194): one entry per distance value — the copies of a point that
    several trees return collapse, and so do distinct points at exactly the same distance."""
    X, f, fo, Q = small_forest
    ids, dist, cnt = rp.knnBatch(20, f, Q[:16], dedup=rp.RPT_KNN_DEDUP_DISTANCE)
    for i in range(16):
        wi, wd = oracle.knn_pq_dense(fo, X, Q[i], 20)
        assert np.array_equal(ids[i, :cnt[i]], wi)
        assert np.all(np.diff(dist[i, :cnt[i]]) > 0)
    hits = rp.knnPQ(rp.metricL2, 5, f, Q[0])
    assert [i for _, i in hits] == oracle.knn_pq_dense(fo, X, Q[0], 5)[0].tolist()


@pytest.mark.parametrize("n,min_leaf", [(60000, 1), (9000, 3), (2500, 1)])
def test_deep_trees_down_to_single_points(rp, ctx, oracle, n, min_leaf):
    """Trees split until the leaves hold one or two points: more levels below the streaming
    phase than one wave-kernel launch covers (its still-active nodes go through a second
    launch), leaves of every small size."""
    d, T = 6, 2
    X = oracle.data_normal_dense2(n, n, d)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(17, T, L, 1.0, d)
    fo = oracle.forest_build_dense(X, R, min_leaf)
    f = rp.forestBatch(0, L, min_leaf, T, 0, d, X, ctx=ctx, hyperplanes=R)
    assert_forest_equal(f, fo)
    ids, dist, cnt = rp.knnBatch(3, f, X[:20])
    for i in range(20):
        wi, wd = oracle.knn_dense(fo, X, X[i], 3)
        assert np.array_equal(ids[i, :cnt[i]], wi)


def test_random_shapes_short_sweep(rp, ctx):
    """15 seconds of tools/fuzz_parity.py: random (points, dimension, trees, minLeaf, depth, data
    kind) forests, device vs oracle, bit for bit."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                        "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(15.0, 20261003, ctx=ctx, verbose=False) > 20


def test_more_than_4096_bins_per_node(rp, ctx, oracle, option):
    """Very large nodes get up to 32768 value bins on the first streaming levels (two-stage
    pick); the option stream_big_node lowers the node size that triggers it."""
    n, d, T, min_leaf = 70000, 5, 3, 30
    X = oracle.data_normal_dense2(99, n, d)
    X[:5000] = np.round(X[:5000])                      # some ties as well
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(5, T, L, 1.0, d)
    fo = oracle.forest_build_dense(X, R, min_leaf)
    with option("stream_big_node", 1000):
        f = rp.forestBatch(0, L, min_leaf, T, 0, d, X, ctx=ctx, hyperplanes=R)
    assert_forest_equal(f, fo)


@pytest.mark.parametrize("dtype,mode", [(np.float64, "exact"), (np.float64, "mfma"), (np.float32, "auto")])
@pytest.mark.parametrize("n,maxnodes,kind", [
    (300000, 32, "cont"),     # 64 nodes of 4687 points after 6 streamed levels: three levels on codes
    (300000, 64, "cont"),     # 2343 points: two levels
    (300001, 128, "cont"),    # 1171 / 1172 points: one level, uneven sizes
    (262144, 16, "cont"),     # 8192 points: the kernel's capacity
    (300000, 32, "ties"),     # rounded keys: pivot codes shared by many points (pool, then the flagged redo)
    (300000, 32, "clump"),    # 3000 identical points: their node's pivot code overflows the pool (flagged redo)
])
def test_mid_size_nodes_on_packed_codes(rp, ctx, oracle, option, dtype, mode, n, maxnodes, kind):
    """Nodes between the wave kernel's 1024 points and 8192 (10 M-point shards after their
    streamed levels; here a smaller set with the streamed levels capped) select their medians on
    packed 16-bit codes (csub_kernel): perm, thresholds and margins identical to the oracle's and
    to the build with the kernel switched off, tie statistics included."""
    d, T, min_leaf = 8, 2, 40
    X = oracle.data_normal_dense2(4242, n, d)
    if kind == "ties":
        X = np.round(X * 8) / 8
    elif kind == "clump":
        X[:3000] = 0.5
    X = X.astype(dtype)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(77, T, L, 1.0 if kind != "cont" else pnz, d)
    pm = {"exact": rp.RPT_PROJ_EXACT, "mfma": rp.RPT_PROJ_MFMA, "auto": rp.RPT_PROJ_AUTO}[mode]
    import ctypes as C
    from rptree_amd import _lib
    with option("stream_maxnodes", maxnodes):
        f = rp.forestBatch(0, L, min_leaf, T, 0, d, X, ctx=ctx, hyperplanes=R, mode=pm)
        back, bad = C.c_int64(-1), C.c_int64(-1)
        _lib.check(_lib.lib().rpt_build_last_handed_back(ctx._h, C.byref(back), C.byref(bad)))
        assert bad.value == 0
        assert (back.value > 0) == (kind == "clump"), back.value
        with option("no_csub", 1):
            g = rp.forestBatch(0, L, min_leaf, T, 0, d, X, ctx=ctx, hyperplanes=R, mode=pm)
    assert_forest_equal(f, g)
    assert f.stats() == g.stats()
    if mode == "exact":
        fo = oracle.forest_build_dense(X.astype(np.float64), R, min_leaf)
        assert_forest_equal(f, fo)


@pytest.mark.parametrize("T", [200, 600, 1000])
def test_knn_many_trees_slots_and_second_traversal(rp, ctx, oracle, option, T):
    """The workgroup query kernel keeps kFR / T range slots per tree for its one-pass traversal
    (2 at T = 200, none above 512 trees: the emitting second traversal); thread = tree covers up to
    four trees per thread.  Answers equal the unfused general path's and the oracle's."""
    n, d, ml, k = 4000, 16, 40, 5
    X = oracle.data_normal_dense2(23, n, d)
    rng = np.random.default_rng(2)
    Q = X[rng.integers(0, n, 12)] + 0.01
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(3, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    got = rp.knnBatch(k, f, Q)
    with option("knn_general", 1):
        ref = rp.knnBatch(k, f, Q)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    fo = oracle.forest_build_dense(X, R, ml)
    for i in range(0, len(Q), 4):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert got[2][i] == len(wi) and np.array_equal(got[0][i, :got[2][i]], wi)
        assert np.array_equal(got[1][i, :got[2][i]], wd)


# ------------------------------------------------------------------ round 4: small-shard query kernels
def _tier_and_uncertified(ctx):
    import ctypes as C
    from rptree_amd import _lib
    tier, unc = C.c_int32(-1), C.c_int64(-1)
    _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
    _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    return tier.value, unc.value


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", [(20000, 128, 4, 100), (20000, 128, 8, 100), (12000, 64, 3, 60),
                                   (9000, 208, 5, 90)])
@pytest.mark.parametrize("kind", ["cont", "ties", "self"])
def test_knn_shard_kernels_are_exact(rp, ctx, oracle, option, dtype, shape, kind):
    """Small tree shards (what one of G GPUs holds) are answered by shard_ranges_kernel (traversal,
    lane = (query, tree)) + knn_shard_wave_kernel (one wave per query; exact distances only for the
    candidates the tier's bounds cannot exclude).  Same answers, bit for bit, as the round-3 one-wave
    kernel, the workgroup kernel, the all-exact kernel and (f64) the oracle: every ranking tier, one
    batch and several (8 x ~78 candidates > 512), three trees (63 lanes of the traversal), rows of
    4 / 8 / 13 sixteen-byte pieces, continuous / rounded data and queries that ARE data points (one
    copy per tree at distance 0)."""
    n, d, T, ml = shape
    X = oracle.data_normal_dense2(41, n, d)
    if kind == "ties":
        X = np.round(X * 2) / 2
    rng = np.random.default_rng(5)
    Q = X[rng.integers(0, n, 48)].copy()
    if kind != "self":
        Q += 0.004
    if dtype == "f32":
        X, Q = X.astype(np.float32), Q.astype(np.float32)
    cfg = rp.rpTreeCfg(ml, n, d)
    L, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X.astype(np.float64), R, ml) if dtype == "f64" else None
    tiers_seen = set()
    for k in (1, 10, 40):
        with option("knn_no_pre32", 1):
            ref = rp.knnBatch(k, f, Q)                       # all-exact distances
        variants = [("int8", {}), ("half", {"knn_no_pre8": 1})]
        if dtype == "f64":
            variants.append(("f32", {"knn_no_pre16": 1}))
        for name, opts in variants:
            import contextlib
            with contextlib.ExitStack() as st:
                st.enter_context(option("knn_wave", 1))
                for o, v in opts.items():
                    st.enter_context(option(o, v))
                got = rp.knnBatch(k, f, Q)
                tiers_seen.add(_tier_and_uncertified(ctx)[0])
                with option("knn_shard_old", 1):
                    old = rp.knnBatch(k, f, Q)
            for a, b, c in zip(got, ref, old):
                assert np.array_equal(a, b), (name, k, kind)
                assert np.array_equal(a, c), (name, k, kind)
        if fo is not None:
            ids, dist, cnt = ref
            for i in range(0, len(Q), 6):
                wi, wd = oracle.knn_dense(fo, X, Q[i], k)
                assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
                assert np.array_equal(dist[i, :cnt[i]], wd)
    assert tiers_seen & {2, 3}, tiers_seen


@pytest.mark.parametrize("T,nq", [(1, 5), (2, 1), (33, 64), (64, 37), (7, 130)])
def test_knn_shard_kernels_any_tree_count_and_batch_size(rp, ctx, oracle, option, T, nq):
    """shard_ranges_kernel packs 64 / T queries into a wave (lane = (query, tree)): one tree, trees that
    do not divide 64, 33 and 64 trees (one query per wave), query counts that leave the last wave and
    the last workgroup partly empty — the oracle's answers, every tier."""
    n, d, ml, k = 12000, 32, 25, 9
    X = oracle.data_normal_dense2(51, n, d)
    Q = oracle.data_normal_dense2(52, nq, d) * 0.9 + 0.1
    cfg = rp.rpTreeCfg(ml, n, d)
    L, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    R, _ = oracle.forest_hyperplanes(13, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X, R, ml)
    for opts in ({}, {"knn_no_pre8": 1}, {"knn_no_pre16": 1}):
        import contextlib
        with contextlib.ExitStack() as st:
            st.enter_context(option("knn_wave", 1))
            for o, v in opts.items():
                st.enter_context(option(o, v))
            ids, dist, cnt = rp.knnBatch(k, f, Q)
            tier, _ = _tier_and_uncertified(ctx)
        assert tier in (1, 2, 3)
        for i in range(nq):
            wi, wd = oracle.knn_dense(fo, X, Q[i], k)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi), (T, i, opts)
            assert np.array_equal(dist[i, :cnt[i]], wd)


def test_knn_shard_list_overflow_goes_to_the_exact_kernel(rp, ctx, oracle, option):
    """More candidates inside the band of the k-th estimate than the carried list holds (here: every
    point of a leaf is one of three values, hundreds of equal estimates) flag the query; the exact
    kernel answers it — the oracle's ids (ties in position order) and distances."""
    n, d, T, ml, k = 6000, 16, 4, 400, 10
    rng = np.random.default_rng(3)
    base = rng.standard_normal((3, d))
    X = base[rng.integers(0, 3, n)].copy()
    Q = base[[0, 1, 2, 0]] + 0.01
    cfg = rp.rpTreeCfg(ml, n, d)
    L, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    R, _ = oracle.forest_hyperplanes(4, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    fo = oracle.forest_build_dense(X, R, ml)
    with option("knn_wave", 1):
        ids, dist, cnt = rp.knnBatch(k, f, Q)
        tier, unc = _tier_and_uncertified(ctx)
    assert tier >= 1 and unc > 0, (tier, unc)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(fo, X, Q[i], k)
        assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi)
        assert np.array_equal(dist[i, :cnt[i]], wd)


@pytest.mark.parametrize("wave", [1, 0])
@pytest.mark.parametrize("pre", [1, 0])
def test_knn_nan_query_is_answered_and_hurts_nobody(rp, ctx, oracle, option, wave, pre):
    """A NaN in one query (the rank counting of the exact stage gave every NaN distance
    rank 0 and read uninitialised slots).  NaN distances now have a total order (behind every number,
    by position): the NaN query gets k in-range ids, the other queries of the batch their usual answers."""
    n, d, T, ml, k = 20000, 32, 4, 100, 10
    X = oracle.data_normal_dense2(8, n, d)
    Q = oracle.data_normal_dense2(9, 16, d)
    cfg = rp.rpTreeCfg(ml, n, d)
    L, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    R, _ = oracle.forest_hyperplanes(2, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    with option("knn_wave", wave), option("knn_no_pre32", 0 if pre else 1):
        clean = rp.knnBatch(k, f, Q)
        Qn = Q.copy()
        Qn[5, 7] = np.nan
        ids, dist, cnt = rp.knnBatch(k, f, Qn)
    keep = np.arange(len(Q)) != 5
    for a, b in zip((ids, dist, cnt), clean):
        assert np.array_equal(a[keep], b[keep])
    assert cnt[5] == k and np.all((ids[5] >= 0) & (ids[5] < n)) and len(set(ids[5].tolist())) >= 1
    assert np.all(np.isnan(dist[5]))
