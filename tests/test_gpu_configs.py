"""The five BASELINE.json configurations (C1..C5) through the C ABI on one MI355X,
each against the CPU oracle at the configuration's own shape.

  C1  10 000 x 16 f64, 1 tree, k = 10, 1 000 queries      everything bit-identical
  C2  1 M x 128 f64, 32 trees, k = 10                     oracle identity on two full trees,
                                                          permutation / cut properties on all 32,
                                                          kNN ids vs the oracle on the 32-tree forest
  C3  1 M x 784 CSR f64 (density 0.19), depth 13          exact build == oracle (innerSS order,
                                                          Internal.hs:351-366), candidates, kNN
  C4  10 M x 128 f32, the 8 trees of one GPU, depth 17    oracle on the exactly-upcast rows (2 trees):
                                                          leaf flips < 1e-3; cut properties (8 trees), kNN
  C5  10 M x 768 bf16, 4 of a GPU's 16 trees, depth 16,   same scheme (bf16 -> f64 is exact); the
      minLeaf 256, k = 50                                 oracle builds ONE tree of the 10 M upcast rows

C4 / C5 are multi-GPU configurations: a GPU holds 8 (16) of the trees and ALL points, so the
per-GPU work is what is tested here, at the configuration's own point count, depth and k.
Integer / index results must be identical where the arithmetic is the reference's (f64,
exact-order projections); f32 / bf16 data and the MFMA projections are build extensions checked
through the north-star tolerances (projection values within 1e-5 |x||r|, leaf assignment flips
< 1e-3).

What the kNN comparison of the reduced-precision configurations covers (knn_agreement): for
EVERY compared query the distances agree to rel_gap and the returned multiset is right wherever
the oracle's distances separate by more than rel_gap; position by position the id lists are
compared up to the first pair of DIFFERENT points whose oracle distances lie within rel_gap of
each other (the device ranks f32 distances there, the oracle f64 ones: either order is right).
The test prints how many positions that is and asserts it against the expectation stated at the
call site."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCPU = max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def rp():
    import rptree_amd
    return rptree_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def leaf_index(perm_row, leaf_off):
    """leaf number of every point from one tree's perm row"""
    n = perm_row.shape[0]
    pos = np.empty(n, dtype=np.int64)
    pos[perm_row] = np.arange(n)
    return np.searchsorted(leaf_off, pos, side="right")


def assert_permutation(perm_row):
    n = perm_row.shape[0]
    assert np.array_equal(np.bincount(perm_row, minlength=n), np.ones(n, dtype=np.int64))


def check_cuts(f, P, rng, n_nodes=24, trees=None):
    """Internal.hs:496-501 on the device's own projections: left half <= thr == min(right), the
    margins are the neighbours of the cut in sorted order."""
    topo = [r for r in f.topology() if not r[4]]
    for idx in rng.choice(len(topo), size=min(n_nodes, len(topo)), replace=False):
        level, heap, off, m, _ = (int(v) for v in topo[idx])
        for t in (range(f.T) if trees is None else trees):
            nh = m // 2
            p = P[t, level]
            left = p[f.perm[t, off:off + nh]]
            right = np.sort(p[f.perm[t, off + nh:off + m]])
            assert left.max() <= f.thr[t, heap] == right[0]
            assert f.mglo[t, heap] == left.max()
            assert f.mghi[t, heap] == right[1]


def flip_rate(perm_a, perm_b, leaf_off):
    return float((leaf_index(perm_a, leaf_off) != leaf_index(perm_b, leaf_off)).mean())


def knn_agreement(ids, dist, cnt, wi, wd, wc, k, rel_gap, extra=8):
    """device (ids, dist, cnt) vs oracle lists that hold k + extra entries (f64 arithmetic on the
    upcast rows; the device ranks f32 distances).  For EVERY query: the distances agree to
    rel_gap; every oracle entry clearly inside the cut (distance < d_k (1 - rel_gap)) is
    returned, and nothing is returned that is not within d_k (1 + rel_gap) — as multisets, the
    reference keeps duplicates (a point found in two trees appears twice).  Position by position
    the ids are compared up to the first near-tie between different points (see the module
    docstring).  Returns (positions compared one by one, positions in all, queries whose whole
    list was compared)."""
    from collections import Counter
    positions = total = whole = 0
    for i in range(ids.shape[0]):
        n_or = int(wc[i])
        m = min(n_or, k)
        assert cnt[i] == m
        assert np.allclose(dist[i, :m], wd[i, :m], rtol=rel_gap, atol=1e-30)
        total += m
        w, wid = wd[i, :n_or], wi[i, :n_or]
        got = Counter(ids[i, :m].tolist())
        # prefix of the oracle list free of near-ties between different points
        near = (np.diff(w) <= rel_gap * w[1:]) & (wid[1:] != wid[:-1])
        first = int(np.argmax(near)) if near.any() else len(w)   # entries [0, first) are separated
        p = min(first, m)
        assert np.array_equal(ids[i, :p], wi[i, :p]), "query %d" % i
        positions += p
        whole += p == m
        if m == k and n_or == k + extra and w[-1] <= w[k - 1] * (1 + rel_gap):
            continue                                  # the tie group at the cut is not closed
        dk = w[m - 1]
        must = Counter(wid[:m][w[:m] < dk * (1 - rel_gap)].tolist())
        allowed = Counter(wid[w <= dk * (1 + rel_gap)].tolist())
        assert not (must - got), "query %d: a clear neighbour is missing" % i
        assert not (got - allowed), "query %d: an entry beyond the cut was returned" % i
    return positions, total, whole


# ------------------------------------------------------------------------------------ C1
def test_c1_10k_x16_one_tree_everything_identical(rp, oracle):
    n, d, min_leaf, k, nq = 10_000, 16, 20, 10, 1_000
    X = oracle.data_normal_dense2(1234, n, d)
    Q = oracle.data_normal_dense2(4321, nq, d)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    assert cfg.fpMaxTreeDepth == 9 and abs(cfg.fpProjNzDensity - 0.8305) < 1e-4
    f = rp.treeBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, cfg.fpProjNzDensity, d, X)
    fo = oracle.forest_build_dense(X, f.R, min_leaf, want_proj=True)
    assert np.array_equal(f.perm, fo.perm)
    for name in ("thr", "mglo", "mghi"):
        assert np.array_equal(getattr(f, name), getattr(fo, name), equal_nan=True), name
    P = f.proj()
    assert np.array_equal(P[0, 0], fo.proj[0, 0])          # every point meets the root's vector
    ids, dist, cnt = rp.knnBatch(k, f, Q)
    wi, wd, wc = oracle.knn_dense_batch(fo, X, Q, k, threads=NCPU)
    assert np.array_equal(cnt, wc) and np.array_equal(ids, wi)
    assert np.array_equal(dist, wd)    # metricDDL2's own bits (left fold, Internal.hs:403-406)
    off, cids = rp.candidatesBatch(f, Q[:200])
    for i in range(200):
        assert np.array_equal(cids[off[i]:off[i + 1]], oracle.candidates_dense(fo, Q[i], 0))
    for i in range(5):
        assert rp.recallWith(rp.metricL2, f, k, Q[i]) == oracle.recall_with_dense(fo, X, Q[i], k)


# ------------------------------------------------------------------------------------ C2
def test_c2_1m_x128_32_trees(rp, oracle):
    n, d, T, min_leaf, k, nq = 1_000_000, 128, 32, 128, 10, 256
    X = oracle.data_normal_dense2(1234, n, d)
    Q = oracle.data_normal_dense2(4321, nq, d)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    assert cfg.fpMaxTreeDepth == 13 and abs(cfg.fpProjNzDensity - 0.4746) < 1e-4
    f = rp.forestBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, X)
    # (i) ALL 32 trees against the oracle, bit for bit (Internal.hs:484-505 at every node of the
    # forest; one oracle thread per tree: 6-7 s on the GPU box's host cores)
    fo = oracle.forest_build_dense(X, f.R, min_leaf, threads=min(T, NCPU))
    for t in range(T):
        assert np.array_equal(f.perm[t], fo.perm[t]), "tree %d" % t
        for name in ("thr", "mglo", "mghi"):
            assert np.array_equal(getattr(f, name)[t], getattr(fo, name)[t], equal_nan=True), (name, t)
    # (ii) every tree: a permutation, cuts consistent with the tree's own projections
    for t in range(T):
        assert_permutation(f.perm[t])
    P = f.proj()
    check_cuts(f, P, np.random.default_rng(0), n_nodes=16)
    # (iii) kNN over the whole 32-tree forest: the oracle walks the device-built flat forest
    ids, dist, cnt = rp.knnBatch(k, f, Q)
    ff = oracle.Forest(n, d, f.R, f.L, min_leaf, f.perm, f.thr, f.mglo, f.mghi)
    wi, wd, wc = oracle.knn_dense_batch(ff, X, Q, k, threads=NCPU)
    assert np.array_equal(cnt, wc) and np.array_equal(ids, wi)
    assert np.array_equal(dist, wd)    # metricDDL2's own bits (left fold, Internal.hs:403-406)
    # (iv) the timed mode (RPT_PROJ_MFMA): values within 1e-5 |x||r|, leaf flips < 1e-3 on all trees
    g = rp.forestBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d, f.data,
                       mode=rp.RPT_PROJ_MFMA)
    leaf_off = np.array([o for (_, _, o, m, lf) in f.topology() if lf])
    flips = np.mean([flip_rate(f.perm[t], g.perm[t], leaf_off) for t in range(T)])
    assert flips < 1e-3, flips
    for t in range(T):
        assert_permutation(g.perm[t])
    Pg = g.proj()
    scale = np.linalg.norm(X[:2000], axis=1)[None, None, :] * np.linalg.norm(f.R, axis=2)[:, :, None]
    assert (np.abs(Pg[:, :, :2000] - P[:, :, :2000]) <= 1e-5 * scale).all()
    check_cuts(g, Pg, np.random.default_rng(1), n_nodes=8)


# ------------------------------------------------------------------------------------ C3
def sparse_uniform_csr(torch, n, d, density, seed):
    """Bernoulli support + U(0,1] values, generated on the GPU for speed"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    cols, counts = [], []
    for r0 in range(0, n, 100_000):
        m = torch.rand((min(100_000, n - r0), d), device="cuda", generator=g) < density
        counts.append(m.sum(dim=1).cpu().numpy())
        cols.append(m.nonzero()[:, 1].to(torch.int32).cpu().numpy())
    col = np.concatenate(cols)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.concatenate(counts))
    val = 1.0 - np.random.default_rng(seed).random(int(rowptr[-1]))
    return rowptr, col, val


def test_c3_1m_x784_sparse(rp, oracle, torch):
    n, d, T, min_leaf, k, nq = 1_000_000, 784, 32, 128, 10, 32       # the whole forest of C3
    rowptr, col, val = sparse_uniform_csr(torch, n, d, 0.19, 1234)
    assert 0.185 < rowptr[-1] / (n * d) < 0.195
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    assert cfg.fpMaxTreeDepth == 13 and abs(cfg.fpProjNzDensity - 0.3455) < 1e-4
    f = rp.forestBatch(1235137, cfg.fpMaxTreeDepth, min_leaf, T, cfg.fpProjNzDensity, d,
                       (rowptr, col, val, d))
    fo = oracle.forest_build_csr(rowptr, col, val, d, f.R, min_leaf, threads=min(T, NCPU))
    assert np.array_equal(f.perm, fo.perm)
    for name in ("thr", "mglo", "mghi"):
        assert np.array_equal(getattr(f, name), getattr(fo, name), equal_nan=True), name
    # queries: fresh sparse rows of the same distribution
    qr, qc, qv = sparse_uniform_csr(torch, nq, d, 0.19, 4321)
    off, cids = rp.candidatesBatch(f, (qr, qc, qv, d))
    ids, dist, cnt = rp.knnBatch(k, f, (qr, qc, qv, d))
    compared = 0
    for i in range(nq):
        a, b = qr[i], qr[i + 1]
        for t in range(T):
            want = oracle.candidates_sparse(fo, qc[a:b], qv[a:b], t)
            assert np.array_equal(cids[off[i * T + t]:off[i * T + t + 1]], want)
        wi, wd = oracle.knn_csr(fo, rowptr, col, val, qc[a:b], qv[a:b], k + 1, true_l2=True)
        m = min(len(wi), k)
        assert cnt[i] == m
        assert np.allclose(dist[i, :m], wd[:m], rtol=1e-9, atol=1e-12)
        if (np.diff(wd) > 1e-9 * wd[1:]).all():      # ids wherever the distances separate
            assert np.array_equal(ids[i, :m], wi[:m])
            compared += 1
    assert compared >= nq // 2


# ------------------------------------------------------------------------------------ C4
def shard_check(rp, oracle, torch, Xd, Xh, T, min_leaf, k, nq, seed, rel_gap, min_same_cands,
                oracle_trees, min_positions, cut_nodes=12, min_close=0.97, tight=None):
    """Common part of C4 / C5: Xd = device tensor (f32 or bf16), Xh = the same rows on the host
    as float32 (exact).  Builds T trees in the default mode of the element type (MFMA) and checks
    all of them through their own projections and the trees `oracle_trees` against the oracle's
    build on the upcast rows (all host cores work inside those trees).  min_positions: the stated
    expectation for the share of kNN result positions compared one by one."""
    n, d = Xh.shape
    ctx = rp.default_context()
    ds = rp.Dataset.from_torch(ctx, Xd)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    _, R = rp.gen.forest_hyperplanes(seed, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)
    for t in range(T):
        assert_permutation(f.perm[t])
    P = f.proj()
    assert P.dtype == np.float32
    check_cuts(f, P, np.random.default_rng(seed), n_nodes=cut_nodes)
    # projection values against the f64 contraction of the upcast rows: 1e-5 |x||r| (north star)
    rows = np.random.default_rng(seed + 1).choice(n, size=2048, replace=False)
    Xs = Xh[rows].astype(np.float64)
    want = np.einsum("nd,tld->tln", Xs, R)
    scale = np.linalg.norm(Xs, axis=1)[None, None, :] * np.linalg.norm(R, axis=2)[:, :, None]
    assert (np.abs(P[:, :, rows] - want) <= 1e-5 * scale).all()
    del P
    # the reference's arithmetic on the exactly-upcast rows
    ot = list(oracle_trees)
    fo = oracle.forest_build_dense(Xh, R[ot], min_leaf, threads=max(NCPU, len(ot)))
    leaf_off = np.array([o for (_, _, o, m, lf) in f.topology() if lf])
    for j, t in enumerate(ot):
        fl = flip_rate(f.perm[t], fo.perm[j], leaf_off)
        print("tree %d: leaf flips vs the oracle %.2e" % (t, fl))
        assert fl < 1e-3, (t, fl)
    # thresholds: f32 projections of the same median point wherever no flipped point moved the
    # node's median rank (a moved rank shifts thr by one inter-point gap)
    tt = ~np.isnan(fo.thr)
    assert np.array_equal(tt, ~np.isnan(f.thr[ot]))
    rn = np.repeat(np.linalg.norm(R[ot], axis=2), [1 << l for l in range(f.L)], axis=1)
    xmax = float(np.linalg.norm(Xs, axis=1).max())
    close = np.abs(f.thr[ot] - fo.thr)[tt] <= 1e-5 * (xmax * rn)[tt]
    print("thresholds within 1e-5 |x||r| of the oracle's: %.4f of the nodes" % close.mean())
    assert close.mean() >= min_close, close.mean()
    if tight:
        # the same build under the tighter projection options (name -> (value, default)): its own bar
        for name, (val, _) in tight.items():
            ctx.set_option(name, val)
        try:
            f3 = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)
        finally:
            for name, (_, dflt) in tight.items():
                ctx.set_option(name, dflt)
        for j, t in enumerate(ot):
            fl3 = flip_rate(f3.perm[t], fo.perm[j], leaf_off)
            print("tree %d under %s: leaf flips vs the oracle %.2e" % (t, sorted(tight), fl3))
            assert fl3 < 1e-3, (t, fl3)
        close3 = np.abs(f3.thr[ot] - fo.thr)[tt] <= 1e-5 * (xmax * rn)[tt]
        print("thresholds under %s: %.4f of the nodes" % (sorted(tight), close3.mean()))
        assert close3.mean() >= 0.97, close3.mean()
        f3.close()
    del fo
    # queries: data points moved a little, in the data's element type
    qi = np.random.default_rng(seed + 2).choice(n, size=nq, replace=False)
    Qd = (Xd[torch.from_numpy(qi).cuda()].float() * 1.001 + 0.003).to(Xd.dtype).contiguous()
    Qh = Qd.float().cpu().numpy()
    qs = rp.Dataset.from_torch(ctx, Qd)
    ids, dist, cnt = rp.knnBatch(k, f, qs)
    # the oracle walks the DEVICE-built forest (f64 query projections, f64 distances)
    ff = oracle.Forest(n, d, R, f.L, min_leaf, f.perm, f.thr, f.mglo, f.mghi)
    off, cids = rp.candidatesBatch(f, qs)
    same = []
    for i in range(nq):
        want = np.concatenate([oracle.candidates_dense(ff, Qh[i].astype(np.float64), t)
                               for t in range(T)])
        same.append(np.array_equal(cids[off[i * T]:off[(i + 1) * T]], want))
    same = np.array(same)
    assert same.mean() >= min_same_cands, same.mean()
    wi, wd, wc = oracle.knn_dense_batch(ff, Xh, Qh.astype(np.float64), k + 8, threads=NCPU)
    sel = np.nonzero(same)[0]
    pos, tot, whole = knn_agreement(ids[sel], dist[sel], cnt[sel], wi[sel], wd[sel], wc[sel], k, rel_gap)
    print("kNN vs the oracle: %d queries with identical candidates of %d; %d of %d result positions "
          "compared one by one (%.0f %%), %d whole lists" % (len(sel), nq, pos, tot, 100.0 * pos / tot, whole))
    assert pos >= min_positions * tot, (pos, tot)
    return f


def test_c4_10m_x128_f32_shard(rp, oracle, torch):
    """BASELINE configs[3] on one of its 8 GPUs: all 10 M points, the GPU's 8 trees, depth 17."""
    n, d, T, min_leaf, k, nq = 10_000_000, 128, 8, 128, 10, 64
    g = torch.Generator(device="cuda").manual_seed(1234)
    coin = (torch.rand(n, 1, device="cuda", generator=g) < 0.5).float() * 2.0
    Xd = torch.randn(n, d, device="cuda", dtype=torch.float32, generator=g) * 0.5 + coin
    del coin
    Xh = Xd.cpu().numpy()
    assert rp.rpTreeCfg(min_leaf, n, d).fpMaxTreeDepth == 17
    # k = 10 of ~1000 candidates on continuous data: neighbouring distances within 1e-5 of each
    # other are rare, at least 80 % of the result positions are compared id by id
    shard_check(rp, oracle, torch, Xd, Xh, T, min_leaf, k, nq, 1235137, 1e-5, 0.9,
                oracle_trees=(0, T - 1), min_positions=0.8, cut_nodes=8)


# ------------------------------------------------------------------------------------ C5
def test_c5_10m_x768_bf16_depth16_k50(rp, oracle, torch):
    """BASELINE configs[4] at its real shape on one GPU: 10 M x 768 bf16 unit-norm rows (15.4 GB),
    depth 16, minLeaf 256, k = 50; 4 of the GPU's 16 trees (projection passes, the 32 768-bin first
    levels, the 11th streamed level and the subtree hand-over are per tree; the oracle builds one
    tree of the 10 M upcast rows with all host cores inside it)."""
    n, d, T, min_leaf, k, nq = 10_000_000, 768, 4, 256, 50, 64
    g = torch.Generator(device="cuda").manual_seed(99)
    Xd = torch.empty(n, d, device="cuda", dtype=torch.bfloat16)
    for r0 in range(0, n, 1_000_000):                    # 3 GB of f32 at a time
        x = torch.randn(1_000_000, d, device="cuda", dtype=torch.float32, generator=g)
        Xd[r0:r0 + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    del x
    Xh = np.empty((n, d), dtype=np.float32)
    for r0 in range(0, n, 1_000_000):
        Xh[r0:r0 + 1_000_000] = Xd[r0:r0 + 1_000_000].float().cpu().numpy()
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    assert cfg.fpMaxTreeDepth == 16 and abs(cfg.fpProjNzDensity - 0.3466) < 1e-4
    # k = 50 on unit-norm rows: the 58 oracle distances of a query lie within a few per cent of
    # each other, a near-tie (1e-5) somewhere in the list is likely; the prefix before it is
    # compared id by id: at least 50 % of all result positions (measured: 71 %)
    # bf16 rows meet the hyperplanes as TWO bf16 terms (|error| <= 2^-17 |x||r|, measured 4.6e-7): more points
    # within rounding of a median than under three terms (1.0e-7), so more nodes whose median rank moved —
    # both modes are checked against the same oracle tree, each against its own bar
    shard_check(rp, oracle, torch, Xd, Xh, T, min_leaf, k, nq, 1235137, 1e-5, 0.9,
                oracle_trees=(T - 1,), min_positions=0.5, cut_nodes=8, min_close=0.93,
                tight={"proj_bf16_terms": (3, 0)})


def test_bf16_forest_and_knn_small_all_paths(rp, oracle, torch):
    """bf16 data from HOST buffers (uint16 bit patterns through rpt_dataset_dense_host): build,
    candidates and kNN against the oracle on the upcast rows, small enough to enumerate."""
    n, d, T, min_leaf, k = 20_000, 64, 4, 40, 10
    rng = np.random.default_rng(3)
    Xb = rp.to_bf16(rng.standard_normal((n, d)).astype(np.float32))
    Xh = rp.from_bf16(Xb)
    ctx = rp.default_context()
    ds = rp.Dataset.dense(ctx, Xb, dtype=rp.RPT_BF16)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    _, R = rp.gen.forest_hyperplanes(5, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)
    fo = oracle.forest_build_dense(Xh, R, min_leaf)
    leaf_off = np.array([o for (_, _, o, m, lf) in f.topology() if lf])
    for t in range(T):
        assert_permutation(f.perm[t])
        assert flip_rate(f.perm[t], fo.perm[t], leaf_off) < 2e-3
    Q = Xh[:100] * 1.01
    ids, dist, cnt = rp.knnBatch(k, f, Q)                  # queries are rounded to bf16 on the way
    Qh = rp.from_bf16(rp.to_bf16(Q)).astype(np.float64)
    ff = oracle.Forest(n, d, R, f.L, min_leaf, f.perm, f.thr, f.mglo, f.mghi)
    wi, wd, wc = oracle.knn_dense_batch(ff, Xh, Qh, k + 8, threads=NCPU)
    off, cids = rp.candidatesBatch(f, Q)
    same = np.array([np.array_equal(
        cids[off[i * T]:off[(i + 1) * T]],
        np.concatenate([oracle.candidates_dense(ff, Qh[i], t) for t in range(T)]))
        for i in range(100)])
    assert same.mean() >= 0.9
    sel = np.nonzero(same)[0]
    pos, tot, whole = knn_agreement(ids[sel], dist[sel], cnt[sel], wi[sel], wd[sel], wc[sel], k, 1e-5)
    assert pos >= 0.8 * tot and whole >= 40, (pos, tot, whole)


@pytest.mark.parametrize("shape", [(20_000, 64, 12, 100, 10), (8_000, 768, 12, 100, 50)])
def test_bf16_int8_ranking_tier_changes_nothing(rp, torch, shape):
    """bf16 rows can be ranked on the int8 shadow first (opt-in: half of the bf16 bytes; the kept rows
    get the f32 distance the all-bf16 kernel ranks on, the cut certified per query): ids, distances
    and counts equal the all-bf16 kernel's bit for bit, rows of 4 and of 48 sixteen-byte pieces."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, min_leaf, k = shape
    rng = np.random.default_rng(13)
    Xb = rp.to_bf16((rng.standard_normal((n, d)) * 0.5 + rng.integers(0, 2, (n, 1)) * 2.0).astype(np.float32))
    Xh = rp.from_bf16(Xb)
    ctx = rp.default_context()
    ds = rp.Dataset.dense(ctx, Xb, dtype=rp.RPT_BF16)
    cfg = rp.rpTreeCfg(min_leaf, n, d)
    _, R = rp.gen.forest_hyperplanes(5, T, cfg.fpMaxTreeDepth, cfg.fpProjNzDensity, d)
    f = rp._build(ctx, ds, R, cfg.fpMaxTreeDepth, min_leaf, rp.RPT_PROJ_AUTO)
    Q = Xh[:64] * 1.01
    kp8 = ctx.set_option("knn_kp8", 200 if k == 50 else 58)   # (opt-in for bf16 rows)
    try:
        got = rp.knnBatch(k, f, Q)
    finally:
        ctx.set_option("knn_kp8", kp8)
    tier, unc = C.c_int32(-1), C.c_int64(-1)
    _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
    _lib.check(_lib.lib().rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
    assert tier.value == 3
    old = ctx.set_option("knn_no_pre8", 1)
    try:
        ref = rp.knnBatch(k, f, Q)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        assert tier.value == 0
    finally:
        ctx.set_option("knn_no_pre8", old)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    print("bf16 int8 tier: %d of %d queries uncertified" % (unc.value, len(Q)))
