"""CPU tests of the oracle's second-round restatements: the streaming `insert`
(Internal.hs:245-297 under Conduit.hs:147-176) with its data-loss branch, counts / keepCounts
(RPTree.hs:464-478), recallWith's Set-of-values semantics (RPTree.hs:276-282), and the typed /
threaded variants used by the full-size GPU tests and the bench's all-core baseline."""
import numpy as np


def small(oracle, n=4000, d=12, T=3, min_leaf=20, seed=7):
    X = oracle.data_normal_dense2(1234, n, d)
    L, _, pnz = oracle.tree_cfg(min_leaf, n, d)
    R, _ = oracle.forest_hyperplanes(seed, T, L, pnz, d)
    return X, R, L


# ---------------------------------------------------------------- streaming insert (8f-2)
def test_stream_one_chunk_is_the_batch_build(oracle):
    """A single chunk holding the whole source takes only the Tip branch (:285-295) = `create`."""
    X, R, L = small(oracle)
    n, ml = len(X), 20
    fb = oracle.forest_build_dense(X, R, ml)
    sf = oracle.stream_forest_dense(X, R, ml, chunk=n)
    topo = oracle.topology(n, L, ml)
    for t in range(R.shape[0]):
        assert sf.held[t] == n
        for level, heap, off, m, leaf in topo:
            if leaf:
                assert sf.kind[t, heap] == 2
                assert np.array_equal(sf.leaves(t)[heap], fb.perm[t, off:off + m])
            else:
                assert sf.kind[t, heap] == 1
                assert sf.thr[t, heap] == fb.thr[t, heap]
                assert (sf.mglo[t, heap], sf.mghi[t, heap]) == (fb.mglo[t, heap], fb.mghi[t, heap])


def test_stream_chunks_average_chunk_medians(oracle):
    """Two chunks: the root cut is the MEAN of the two chunk medians (:281 thr' = (thr0+thr)/2)
    and the margin the (max, min) of the chunk margins (:280) — not the batch median."""
    X, R, L = small(oracle, n=2000)
    ml = 20
    sf = oracle.stream_forest_dense(X, R, ml, chunk=1000)
    for t in range(R.shape[0]):
        idx = np.nonzero(R[t, 0])[0]
        p = np.array([oracle.inner_sd(idx, R[t, 0, idx], x) for x in X])
        a = oracle.partition_at_median(p[:1000])
        b = oracle.partition_at_median(p[1000:])
        assert sf.thr[t, 0] == (a[2] + b[2]) / 2
        assert sf.mglo[t, 0] == max(a[3], b[3]) and sf.mghi[t, 0] == min(a[4], b[4])
        batch = oracle.partition_at_median(p)
        assert sf.thr[t, 0] != batch[2]                      # the documented semantic difference
        assert sf.held[t] == 2000                            # large halves: nothing is lost
    # a Tip that keeps growing holds the LATER chunk first (:286 xs' = xs <> xs0, :288): depth
    # limit 0 makes the root such a Tip
    s0 = oracle.stream_forest_dense(X[:10], np.zeros((1, 0, X.shape[1])), 2, chunk=4)
    assert s0.leaves(0)[0].tolist() == [8, 9, 4, 5, 6, 7, 0, 1, 2, 3]


def test_stream_short_chunk_loses_data(oracle):
    """SURVEY 7.3-6, demonstrated.  When a chunk's half is EMPTY at a node that is already a Bin
    (n = 1 there: take 0 / drop 0), `partitionAtMedian` returns Nothing and `insert` answers
    `Tip () mempty` (:277): the whole subtree below, with every point stored in it, is replaced.
    Every full chunk sends the SAME number of points to a given node (sizes halve
    deterministically), so a node that always receives nothing never becomes a Bin and nothing is
    lost while all chunks have one size; the first chunk of a DIFFERENT size — C.chunksOf's short
    last chunk whenever chunk does not divide n — reaches Bins with empty halves."""
    X, R, L = small(oracle, n=4000)
    ml = 20
    for chunk in (1, 2, 5, 16, 100, 4000):                    # divisors of n: no short chunk
        assert (oracle.stream_forest_dense(X, R, ml, chunk).held == 4000).all(), chunk
    lost = {}
    for chunk in (3, 7, 33, 4000 - 1):                        # last chunks of 1, 3, 7, 1 points
        sf = oracle.stream_forest_dense(X, R, ml, chunk)
        assert (sf.held < 4000).all() and (sf.held > 0).all(), (chunk, sf.held)
        lost[chunk] = 4000 - int(sf.held[0])
        for t in range(R.shape[0]):                           # what is held is held once
            ids = np.concatenate(list(sf.leaves(t).values()))
            assert len(ids) == sf.held[t] == len(set(ids.tolist()))
    # a last chunk of ONE point empties the root's whole left subtree: about 2/3 (chunk 3: the
    # left children take 1 of 3) or 1/2 of everything inserted before it
    assert lost[3] > 2000 and lost[3999] > 1500


def test_stream_queries_on_one_chunk_equal_the_batch_queries(oracle):
    """candidates / knn over the heap arrays of a streamed forest == over the flat batch forest
    when the stream was one chunk (same tree, two layouts)."""
    X, R, L = small(oracle)
    ml, k = 20, 5
    fb = oracle.forest_build_dense(X, R, ml)
    sf = oracle.stream_forest_dense(X, R, ml, chunk=len(X))
    Q = oracle.data_normal_dense2(4321, 10, X.shape[1])
    for q in Q:
        for t in range(R.shape[0]):
            assert np.array_equal(oracle.stream_candidates_dense(sf, R, q, t),
                                  oracle.candidates_dense(fb, q, t))
        a, b = oracle.stream_knn_dense(sf, R, X, q, k), oracle.knn_dense(fb, X, q, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---------------------------------------------------------------- counts / keepCounts (8f-4)
def test_keep_counts(oracle):
    ids = [5, 3, 5, 1, 3, 5, 9]
    for thr, want in [(1, ([1, 3, 5, 9], [1, 2, 3, 1])), (2, ([3, 5], [2, 3])), (3, ([5], [3])),
                      (4, ([], []))]:
        gi, gc = oracle.keep_counts(ids, thr)
        assert gi.tolist() == want[0] and gc.tolist() == want[1]


def test_vote_knn_is_knn_over_the_kept_ids(oracle):
    X, R, L = small(oracle, T=8)
    ml, k = 20, 6
    f = oracle.forest_build_dense(X, R, ml)
    Q = oracle.data_normal_dense2(4321, 12, X.shape[1])
    for v in (1, 2, 4):
        ids, dist, cnt = oracle.knn_dense_batch(f, X, Q, k, vote_thr=v)
        for i in range(len(Q)):
            cand = np.concatenate([oracle.candidates_dense(f, Q[i], t) for t in range(f.T)])
            kept, _ = oracle.keep_counts(cand, v)
            dd = np.array([oracle.metric_dd(X[j], Q[i]) for j in kept])
            order = np.argsort(dd, kind="stable")[:k]
            assert cnt[i] == len(order)
            assert np.array_equal(ids[i, :cnt[i]], kept[order])
            assert np.array_equal(dist[i, :cnt[i]], dd[order])


# ---------------------------------------------------------------- recallWith on values
def test_recall_with_values_collapses_equal_points(oracle):
    X, R, L = small(oracle, n=2000)
    ml, k = 20, 5
    Xd = X.copy()
    Xd[1000:] = X[:1000]                              # every point twice (payload `()`)
    f = oracle.forest_build_dense(Xd, R, ml)
    q = Xd[17] * 1.0001
    by_id = oracle.recall_with_dense(f, Xd, q, k)
    by_val = oracle.recall_with_dense_values(f, Xd, q, k)
    # the k nearest IDS are k/2 distinct values (pairs): the value sets are smaller
    assert by_val <= by_id
    assert by_val <= (k + 1) // 2 / k + 1e-12
    # without duplicates both definitions agree
    g = oracle.forest_build_dense(X, R, ml)
    assert oracle.recall_with_dense(g, X, q, k) == oracle.recall_with_dense_values(g, X, q, k)


# ---------------------------------------------------------------- typed / threaded variants
def test_threads_and_float_rows_change_nothing(oracle):
    X, R, L = small(oracle, n=6000, T=5)
    ml = 20
    a = oracle.forest_build_dense(X, R, ml, want_proj=True)
    b = oracle.forest_build_dense(X, R, ml, want_proj=True, threads=4)
    for name in ("perm", "thr", "mglo", "mghi", "proj"):
        assert np.array_equal(getattr(a, name), getattr(b, name), equal_nan=True), name
    X32 = X.astype(np.float32)
    c = oracle.forest_build_dense(X32, R, ml, threads=2)                    # float rows, upcast on read
    d = oracle.forest_build_dense(X32.astype(np.float64), R, ml)            # the upcast copy
    for name in ("perm", "thr", "mglo", "mghi"):
        assert np.array_equal(getattr(c, name), getattr(d, name), equal_nan=True), name
    Q = oracle.data_normal_dense2(4321, 9, X.shape[1])
    ids, dist, cnt = oracle.knn_dense_batch(a, X, Q, 7, threads=3)
    for i in range(len(Q)):
        wi, wd = oracle.knn_dense(a, X, Q[i], 7)
        assert np.array_equal(ids[i, :cnt[i]], wi) and np.array_equal(dist[i, :cnt[i]], wd)
    i32, d32, c32 = oracle.knn_dense_batch(c, X32, Q, 7)
    i64, d64, c64 = oracle.knn_dense_batch(d, X32.astype(np.float64), Q, 7)
    assert np.array_equal(i32, i64) and np.array_equal(d32, d64)
    rowptr, col, val = oracle.data_normal_sparse2(5, 3000, 30, 0.3)
    Rs, _ = oracle.forest_hyperplanes(9, 4, 6, 0.5, 30)
    e = oracle.forest_build_csr(rowptr, col, val, 30, Rs, 25)
    g = oracle.forest_build_csr(rowptr, col, val, 30, Rs, 25, threads=4)
    assert np.array_equal(e.perm, g.perm) and np.array_equal(e.thr, g.thr, equal_nan=True)


def test_host_data_generator_word_stream(oracle):
    """rptree_amd.gen.normal_dense2 / normal_dense2_torch draw the oracle's SplitMix64 stream
    (numpy's and torch's log / cos may differ from libm's in the last bits)."""
    import torch
    from rptree_amd import gen
    want = oracle.data_normal_dense2(1234, 500, 16)
    a = gen.normal_dense2(1234, 500, 16, rows_per_chunk=128)
    b = gen.normal_dense2_torch(1234, 500, 16, torch.device("cpu"), rows_per_chunk=100).numpy()
    assert np.allclose(a, want, rtol=0, atol=1e-14) and np.allclose(b, want, rtol=0, atol=1e-14)
    # the mixture coin of every row is the stream's
    assert np.array_equal(np.round(a.mean(axis=1)), np.round(want.mean(axis=1)))
