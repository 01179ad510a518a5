"""GPU tests of the C-ABI additions of round 2: the multi-GPU entry points (rpt_comm_* /
rpt_forest_build_sharded / rpt_knn_sharded*, csrc/comm.hip on librccl) on the one GPU a test
box has, context options, shard merges beyond one launch's capacity, and the import of a forest
built ELSEWHERE (the oracle plays deserialiseRPForest's part, Internal.hs:185-196)."""
import ctypes as C

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


@pytest.fixture(scope="module")
def case(oracle):
    n, d, T, ml = 30000, 24, 6, 50
    X = oracle.data_normal_dense2(1234, n, d)
    Q = oracle.data_normal_dense2(4321, 300, d)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(1235137, T, L, pnz, d)
    return X, Q, R, L, ml


# --------------------------------------------------------------------------- multi-GPU entry points
def test_sharded_entry_points_one_gpu_equal_the_plain_ones(rp, ctx, case, oracle):
    """rpt_comm_init(1) + rpt_forest_build_sharded + rpt_knn_sharded(_dev) == rpt_forest_build +
    rpt_knn_* on the same device, bit for bit, and == the oracle (Internal.hs:234-240,
    RPTree.hs:174-176)."""
    import torch
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    k = 10
    plain = rp.forestBatch(0, L, ml, R.shape[0], 0, X.shape[1], X, ctx=ctx, hyperplanes=R)
    pi, pd, pc = rp.knnBatch(k, plain, Q)
    comm = sharded.Comm.local(1)
    assert (comm.nranks, comm.nlocal, comm.first_rank) == (1, 1, 0)
    c0 = comm.contexts[0]
    ds = rp.Dataset.dense(c0, X)
    qs = rp.Dataset.dense(c0, Q)
    sf = sharded.ShardedForest(comm, [ds], R, L, ml)
    loc, lo, nt = sf.local(0)
    assert (lo, nt) == (0, R.shape[0])
    assert np.array_equal(loc.perm, plain.perm)
    assert np.array_equal(loc.thr, plain.thr, equal_nan=True)
    si, sd, sc = sf.knn([qs], k)                                   # host variant
    assert np.array_equal(si, pi) and np.array_equal(sd, pd) and np.array_equal(sc, pc)
    oi = torch.empty((len(Q), k), dtype=torch.int32, device="cuda")
    od = torch.empty((len(Q), k), dtype=torch.float64, device="cuda")
    oc = torch.empty((len(Q),), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for flags in (rp.RPT_KNN_KEEP_DUPLICATES, rp.RPT_KNN_DEDUP):
        sf.knn_dev([qs], k, flags, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
        comm.sync()
        wi, wd, wc = rp.knnBatch(k, plain, Q, dedup=bool(flags))
        assert np.array_equal(oi.cpu().numpy(), wi) and np.array_equal(od.cpu().numpy(), wd)
        assert np.array_equal(oc.cpu().numpy(), wc)
    fo = oracle.forest_build_dense(X, R, ml)
    wi, wd, wc = oracle.knn_dense_batch(fo, X, Q, k, threads=4)
    assert np.array_equal(si, wi) and np.array_equal(sc, wc)
    sf.close()
    ds.close()
    qs.close()
    comm.close()


def test_comm_init_rank_with_a_unique_id(rp, ctx, case):
    """The one-process-per-GPU route: rpt_comm_unique_id + rpt_comm_init_rank (world of one)."""
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    uid = sharded.Comm.unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    comm = sharded.Comm.rank(ctx, 1, 0, uid)
    assert (comm.nranks, comm.nlocal, comm.first_rank) == (1, 1, 0)
    ds = rp.Dataset.dense(ctx, X)
    qs = rp.Dataset.dense(ctx, Q[:50])
    sf = sharded.ShardedForest(comm, [ds], R, L, ml)
    si, sd, sc = sf.knn([qs], 5)
    plain, _, _ = sf.local(0)
    pi, pd, pc = rp.knnBatch(5, plain, Q[:50])
    assert np.array_equal(si, pi) and np.array_equal(sd, pd) and np.array_equal(sc, pc)
    sf.close()
    comm.close()
    # the ctx was borrowed: still usable
    assert rp.project(X[:10], R[0, :1], ctx=ctx).shape == (1, 10)


def test_comm_from_a_torch_process_group(rp, ctx, case):
    """What bench.py does under torch.distributed.run: a gloo control plane, rank 0's RCCL id
    broadcast through it, rpt_comm_init_rank on every rank (here a world of one process)."""
    import os
    import torch.distributed as dist
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = sharded.Comm.from_process_group(ctx)
        assert (comm.nranks, comm.nlocal, comm.first_rank) == (1, 1, 0)
        ds = rp.Dataset.dense(ctx, X)
        qs = rp.Dataset.dense(ctx, Q[:20])
        sf = sharded.ShardedForest(comm, [ds], R, L, ml)
        si, sd, sc = sf.knn([qs], 5)
        plain, lo, nt = sf.local(0)
        assert (lo, nt) == (0, R.shape[0])
        pi, pd, pc = rp.knnBatch(5, plain, Q[:20])
        assert np.array_equal(si, pi) and np.array_equal(sd, pd) and np.array_equal(sc, pc)
        sf.close()
        comm.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_forced_exchange_runs_allgather_and_merge_on_one_rank(rp, ctx, case):
    """comm_force_exchange: a ONE-rank communicator takes the whole multi-GPU data path — record ->
    ncclAllGather on the ctx stream -> rpt_knn_merge_records_dev on the gathered buffer -> status
    scan — instead of returning the record.  Result == rpt_knn_dev, bit for bit, for duplicates
    kept / de-duplicated and nq in {0, 1, 10 000} (RPTree.hs:174-176)."""
    import torch
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    k = 10
    Qbig = np.concatenate([Q] * 34)[:10_000] * np.linspace(0.97, 1.03, 10_000)[:, None]
    for style in ("init", "init_rank"):
        if style == "init":
            comm = sharded.Comm.local(1)
            c0 = comm.contexts[0]
        else:
            comm = sharded.Comm.rank(ctx, 1, 0, sharded.Comm.unique_id())
            c0 = ctx
        old = c0.set_option("comm_force_exchange", 1)
        try:
            ds = rp.Dataset.dense(c0, X)
            sf = sharded.ShardedForest(comm, [ds], R, L, ml)
            plain, _, _ = sf.local(0)
            for nq in (0, 1, 10_000):
                qs = rp.Dataset.dense(c0, Qbig[:nq].reshape(nq, X.shape[1]))
                oi = torch.full((max(nq, 1), k), -7, dtype=torch.int32, device="cuda")
                od = torch.zeros((max(nq, 1), k), dtype=torch.float64, device="cuda")
                oc = torch.full((max(nq, 1),), -7, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                for flags in (rp.RPT_KNN_KEEP_DUPLICATES, rp.RPT_KNN_DEDUP):
                    sf.knn_dev([qs], k, flags, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
                    comm.sync()
                    if nq == 0:
                        continue
                    wi, wd, wc = rp.knnBatch(k, plain, qs, dedup=bool(flags))
                    assert np.array_equal(oi.cpu().numpy(), wi), (style, nq, flags)
                    assert np.array_equal(od.cpu().numpy(), wd)
                    assert np.array_equal(oc.cpu().numpy(), wc)
                si, sd, sc = sf.knn([qs], k)                       # host variant, same path
                if nq:
                    wi, wd, wc = rp.knnBatch(k, plain, qs)
                    assert np.array_equal(si, wi) and np.array_equal(sd, wd) and np.array_equal(sc, wc)
                qs.close()
            sf.close()
            ds.close()
        finally:
            c0.set_option("comm_force_exchange", old)
            comm.close()


def test_failed_rank_poisons_the_exchange_instead_of_hanging(rp, ctx, case):
    """Failure protocol of comm.hip: a rank whose query kernels fail still joins the all-gather with
    its record's status word set and returns its error; after the merge every count is -1 and
    rpt_comm_sync names the rank.  The next batch (no failure) is answered normally."""
    import torch
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    k, nq = 5, 64
    comm = sharded.Comm.rank(ctx, 1, 0, sharded.Comm.unique_id())
    old = ctx.set_option("comm_force_exchange", 1)
    try:
        ds = rp.Dataset.dense(ctx, X)
        qs = rp.Dataset.dense(ctx, Q[:nq])
        sf = sharded.ShardedForest(comm, [ds], R, L, ml)
        oi = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
        od = torch.zeros((nq, k), dtype=torch.float64, device="cuda")
        oc = torch.zeros((nq,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.set_option("comm_inject_failure", 1)
        with pytest.raises(rp.RPTError, match="comm_inject_failure"):
            sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
        with pytest.raises(rp.RPTError, match="rank 0 failed"):
            comm.sync()
        assert (oc.cpu().numpy() == -1).all()
        with pytest.raises(rp.RPTError, match="comm_inject_failure"):
            sf.knn([qs], k)                                        # host variant: the rank's own error
        ctx.set_option("comm_inject_failure", 0)
        si, sd, sc = sf.knn([qs], k)
        plain, _, _ = sf.local(0)
        wi, wd, wc = rp.knnBatch(k, plain, Q[:nq])
        assert np.array_equal(si, wi) and np.array_equal(sd, wd) and np.array_equal(sc, wc)
        sf.close()
        ds.close()
        qs.close()
    finally:
        ctx.set_option("comm_inject_failure", 0)
        ctx.set_option("comm_force_exchange", old)
        comm.close()


def test_a_failed_batch_is_reported_after_later_batches_succeeded(rp, ctx, case):
    """The failed word of an exchange accumulates until rpt_comm_sync reads it: batch 1 fails, batch 2
    succeeds, ONE sync after both still names the rank; the sync resets the word."""
    import torch
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    k, nq = 5, 64
    comm = sharded.Comm.rank(ctx, 1, 0, sharded.Comm.unique_id())
    old = ctx.set_option("comm_force_exchange", 1)
    try:
        ds = rp.Dataset.dense(ctx, X)
        qs = rp.Dataset.dense(ctx, Q[:nq])
        sf = sharded.ShardedForest(comm, [ds], R, L, ml)
        bufs = [(torch.zeros((nq, k), dtype=torch.int32, device="cuda"),
                 torch.zeros((nq, k), dtype=torch.float64, device="cuda"),
                 torch.zeros((nq,), dtype=torch.int32, device="cuda")) for _ in range(2)]
        torch.cuda.synchronize()
        ctx.set_option("comm_inject_failure", 1)
        with pytest.raises(rp.RPTError, match="comm_inject_failure"):
            sf.knn_dev([qs], k, 0, *[[b.data_ptr()] for b in bufs[0]])
        ctx.set_option("comm_inject_failure", 0)
        sf.knn_dev([qs], k, 0, *[[b.data_ptr()] for b in bufs[1]])       # a good batch, no sync in between
        with pytest.raises(rp.RPTError, match="rank 0 failed"):
            comm.sync()
        assert (bufs[0][2].cpu().numpy() == -1).all()
        plain, _, _ = sf.local(0)
        wi, wd, wc = rp.knnBatch(k, plain, Q[:nq])
        assert np.array_equal(bufs[1][0].cpu().numpy(), wi) and np.array_equal(bufs[1][2].cpu().numpy(), wc)
        sf.knn_dev([qs], k, 0, *[[b.data_ptr()] for b in bufs[1]])
        comm.sync()                                                       # the word was reset: no stale report
        sf.close()
        ds.close()
        qs.close()
    finally:
        ctx.set_option("comm_inject_failure", 0)
        ctx.set_option("comm_force_exchange", old)
        comm.close()


def test_a_stalled_exchange_aborts_the_communicator_instead_of_hanging(rp, ctx, case):
    """rpt_comm_sync polls hipStreamQuery + ncclCommGetAsyncError under a deadline instead of blocking in
    hipStreamSynchronize (a peer that never joins leaves an RCCL collective spinning).
    comm_stall_test makes a pending exchange count as past its deadline: the local communicator is
    aborted, the call returns RPT_E_INTERNAL, later sharded calls fail fast."""
    import torch
    from rptree_amd import sharded
    X, Q, R, L, ml = case
    k = 10
    Qbig = np.concatenate([Q] * 40)[:10_000]
    nq = len(Qbig)
    comm = sharded.Comm.rank(ctx, 1, 0, sharded.Comm.unique_id())
    old = ctx.set_option("comm_force_exchange", 1)
    try:
        ds = rp.Dataset.dense(ctx, X)
        qs = rp.Dataset.dense(ctx, Qbig)
        sf = sharded.ShardedForest(comm, [ds], R, L, ml)
        oi = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
        od = torch.zeros((nq, k), dtype=torch.float64, device="cuda")
        oc = torch.zeros((nq,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
        comm.sync()                                                       # the normal path: polls to completion
        ctx.set_option("comm_stall_test", 1)
        sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
        with pytest.raises(rp.RPTError, match="did not complete|aborted"):
            comm.sync()
        ctx.set_option("comm_stall_test", 0)
        with pytest.raises(rp.RPTError, match="aborted"):
            sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
        ctx.sync()
        sf.close()
        ds.close()
        qs.close()
    finally:
        ctx.set_option("comm_stall_test", 0)
        ctx.set_option("comm_force_exchange", old)
        comm.close()



def test_comm_argument_errors(rp, ctx, case):
    from rptree_amd import _lib, sharded
    L_ = _lib.lib()
    h = C.c_void_p()
    assert L_.rpt_comm_init(0, C.byref(h)) == -1
    have = C.c_int32()
    _lib.check(L_.rpt_device_count(C.byref(have)))
    assert L_.rpt_comm_init(have.value + 1, C.byref(h)) == -1
    assert b"visible" in L_.rpt_last_error()
    X, Q, R, L, ml = case
    comm = sharded.Comm.local(1)
    ds_other = rp.Dataset.dense(ctx, X[:1000])                     # lives on ANOTHER context
    with pytest.raises(rp.RPTError, match="communicator's"):
        sharded.ShardedForest(comm, [ds_other], R, L, ml)
    comm.close()


# --------------------------------------------------------------------------- merges
def merge_reference(gi, gd, gc, k, dedup):
    G, nq, _ = gi.shape
    oi = np.full((nq, k), -1, dtype=np.int32)
    od = np.full((nq, k), np.inf)
    oc = np.zeros(nq, dtype=np.int32)
    for q in range(nq):
        ent = [(gd[g, q, r], g, r, gi[g, q, r]) for g in range(G) for r in range(gc[g, q])]
        ent.sort(key=lambda e: (e[0], e[1], e[2]))
        seen, m = set(), 0
        for dd, _, _, i in ent:
            if m == k:
                break
            if dedup and i in seen:
                continue
            seen.add(i)
            oi[q, m], od[q, m] = i, dd
            m += 1
        oc[q] = m
    return oi, od, oc


@pytest.mark.parametrize("G,k", [(8, 1024), (3, 1024), (5, 700), (4, 1024), (8, 512)])
def test_merge_beyond_one_launch(rp, ctx, G, k):
    """G * k > 4096 entries per query: the shards are folded in one at a time; same order as the
    one-launch merge (distance, shard, rank), duplicates kept or dropped."""
    import torch
    from rptree_amd import _lib
    rng = np.random.default_rng(G * k)
    nq = 7
    # a point has ONE distance to a query (the de-duplicating merge relies on it): distances
    # are a function of (query, id), rounded so that many different points tie
    table = np.round(rng.random((nq, 3000)) * 50, 1)
    gi = rng.integers(0, 3000, size=(G, nq, k)).astype(np.int32)
    gd = np.take_along_axis(np.broadcast_to(table, (G, nq, 3000)), gi.astype(np.int64), axis=2)
    order = np.argsort(gd, axis=2, kind="stable")
    gi = np.take_along_axis(gi, order, axis=2)
    gd = np.take_along_axis(gd, order, axis=2)
    gc = rng.integers(k // 2, k + 1, size=(G, nq)).astype(np.int32)
    gc[0, 0] = 0
    ti, td, tc = (torch.from_numpy(a).cuda() for a in (gi, gd, gc))
    oi = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for flags in (0, 1):
        _lib.check(_lib.lib().rpt_knn_merge_dev(ctx._h, ti.data_ptr(), td.data_ptr(), tc.data_ptr(),
                                                G, nq, k, flags, oi.data_ptr(), od.data_ptr(),
                                                oc.data_ptr()))
        ctx.sync()
        wi, wd, wc = merge_reference(gi, gd, gc, k, bool(flags))
        assert np.array_equal(oc.cpu().numpy(), wc)
        for q in range(nq):
            m = wc[q]
            assert np.array_equal(oi.cpu().numpy()[q, :m], wi[q, :m])
            assert np.array_equal(od.cpu().numpy()[q, :m], wd[q, :m])


# --------------------------------------------------------------------------- options
def test_context_options(rp, ctx):
    assert ctx.get_option("knn_wave") == -1 and ctx.get_option("no_wsub") == 0
    old = ctx.set_option("no_wsub", 1)
    assert old == 0 and ctx.get_option("no_wsub") == 1
    ctx.set_option("no_wsub", 0)
    with pytest.raises(rp.RPTError, match="unknown option"):
        ctx.set_option("no_such_switch", 1)


def test_every_fallback_option_keeps_the_forest_identical(rp, ctx, oracle):
    n, d, T, ml = 50000, 10, 3, 30
    X = oracle.data_normal_dense2(5, n, d)
    X[:4000] = np.round(X[:4000], 1)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(8, T, L, pnz, d)
    fo = oracle.forest_build_dense(X, R, ml)
    for name in ("no_stream", "no_wsub", "no_wsort", "no_wpack", "no_wmid", "no_midselect", "proj_narrow"):
        old = ctx.set_option(name, 1)
        try:
            f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R)
        finally:
            ctx.set_option(name, old)
        assert np.array_equal(f.perm, fo.perm), name
        assert np.array_equal(f.thr, fo.thr, equal_nan=True), name


@pytest.mark.parametrize("kind", ["cont", "ties", "heavy", "constcol"])
@pytest.mark.parametrize("mode", ["exact", "mfma"])
def test_streaming_on_16bit_codes_is_exact(rp, ctx, oracle, kind, mode):
    """From 131 072 points on the streaming levels histogram 16-bit codes of the keys (csrc/
    codes.h) and read exact keys only for pivot bins and margins.  Exact mode: the forest must be
    the oracle's on continuous data, on data with ties, on heavy ties (a few distinct values: pivot
    bins outgrow everything, the general path takes over) and with a constant column; MFMA mode:
    identical to the build that streams on the keys themselves (option no_codes)."""
    n, d, T, ml = 200_000, 128, 3, 40
    X = oracle.data_normal_dense2(77, n, d)
    if kind == "ties":
        X = np.round(X, 1)
    elif kind == "heavy":
        X = np.round(X * 0.7)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(3, T, L, pnz, d)
    if kind == "constcol":
        R[1, 2] = 0.0                                    # every projection of that level is 0.0
        R[2, 0, :] = 0.0
        R[2, 0, 5] = 1e-300                              # denormal-range keys
    m = rp.RPT_PROJ_EXACT if mode == "exact" else rp.RPT_PROJ_MFMA
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=m)
    old = ctx.set_option("no_codes", 1)
    try:
        g = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=m)
    finally:
        ctx.set_option("no_codes", old)
    for name in ("perm", "thr", "mglo", "mghi"):
        assert np.array_equal(getattr(f, name), getattr(g, name), equal_nan=True), name
    if mode == "exact":
        fo = oracle.forest_build_dense(X, R, ml, threads=T)
        for name in ("perm", "thr", "mglo", "mghi"):
            assert np.array_equal(getattr(f, name), getattr(fo, name), equal_nan=True), name


def test_codes_after_the_projection_for_csr_and_bf16_rows(rp, ctx, oracle):
    """Projection kernels without a code epilogue (CSR rows) get their 16-bit codes from a pass over
    the stored keys (pcode_kernel): the streamed levels then run on codes as for dense rows.  CSR: the
    forest is the oracle's, with and without the pass; bf16 rows: identical forests with the codes of
    proj_bf16x3's epilogue, with the pass, and without codes (ties included: rounded values)."""
    n, d, T, ml = 160_000, 200, 2, 60
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.1)
    val = np.round(val, 1)                                   # many equal projections
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    fo = oracle.forest_build_csr(rowptr, col, val, d, R, ml, threads=8)
    for off in (0, 1):
        old = ctx.set_option("no_pcodes", off)
        try:
            f = rp.forestBatch(0, L, ml, T, pnz, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
        finally:
            ctx.set_option("no_pcodes", old)
        assert np.array_equal(f.perm, fo.perm), off
        for name in ("thr", "mglo", "mghi"):
            assert np.array_equal(getattr(f, name), getattr(fo, name), equal_nan=True), (name, off)
    # bf16 rows: codes from the stored keys (default), from proj_bf16x3's epilogue (proj_bf16_codes), or none —
    # short rows (hyperplanes resident in LDS, 64-column pass) and long ones (chunks of four k-steps, 128 columns)
    for (n, d, T, ml) in ((200_000, 64, 3, 50), (140_000, 264, 6, 40)):
        rng = np.random.default_rng(4)
        Xb = rp.to_bf16(np.round(rng.standard_normal((n, d)), 1).astype(np.float32))
        ds = rp.Dataset.dense(ctx, Xb, dtype=rp.RPT_BF16)
        L, _, pnz = oracle.tree_cfg(ml, n, d)
        R, _ = oracle.forest_hyperplanes(3, T, L, pnz, d)
        got = []
        for opts in ({}, {"proj_bf16_codes": 1}, {"no_pcodes": 1}):
            old = {k: ctx.set_option(k, v) for k, v in opts.items()}
            try:
                f = rp._build(ctx, ds, R, L, ml, rp.RPT_PROJ_AUTO)
            finally:
                for k, v in old.items():
                    ctx.set_option(k, v)
            got.append((f.perm.copy(), f.thr.copy(), f.mglo.copy(), f.mghi.copy()))
            f.close()
        for other in got[1:]:
            for a, b in zip(got[0], other):
                assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("d,dtype", [(600, np.float64), (784, np.float64), (300, np.float64),
                                     (784, np.float32), (1500, np.float32)])
def test_csr_projection_32_per_pass_is_bit_identical(rp, ctx, oracle, d, dtype):
    """CSR rows against 32 hyperplanes per pass: whole rows when the hyperplane tile fits LDS,
    otherwise two column halves (tail launch, then head launch continuing its sums) — the same
    right-nested sum as innerSS (Internal.hs:353-366) and as the 16-column kernel, bit for bit."""
    n, C = 5000, 72                      # 32 + 32 + 8 columns
    rng = np.random.default_rng(d)
    dens = 0.19
    m = rng.random((n, d)) < dens
    m[7] = False                         # an empty row
    m[8, :] = False
    m[8, d - 1] = True                   # a row living in the upper half only
    m[9, :] = False
    m[9, 0] = True                       # ... in the lower half only
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(m.sum(axis=1))
    col = np.nonzero(m)[1].astype(np.int32)
    val = (1.0 - rng.random(len(col))).astype(dtype)
    R = rng.standard_normal((C, d)) * (rng.random((C, d)) < 0.35)
    ds = rp.Dataset.csr(ctx, rowptr, col, val, d)
    P = rp.project(ds, R, mode=rp.RPT_PROJ_EXACT, ctx=ctx)
    old = ctx.set_option("proj_narrow", 1)
    try:
        Pn = rp.project(ds, R, mode=rp.RPT_PROJ_EXACT, ctx=ctx)
    finally:
        ctx.set_option("proj_narrow", old)
    assert np.array_equal(P, Pn)
    # RPT_PROJ_MFMA on CSR rows = the same kernel with one fused multiply-add per term: the
    # tolerance mode (1e-5 * |x| * |r|, include/rptree_hip.h), not bit-identical
    Pf = rp.project(ds, R, mode=rp.RPT_PROJ_MFMA, ctx=ctx)
    xn = np.sqrt(np.add.reduceat(np.square(val.astype(np.float64)), np.minimum(rowptr[:-1], len(val) - 1))
                 * (np.diff(rowptr) > 0))
    rn = np.linalg.norm(R, axis=1)
    assert np.all(np.abs(Pf.astype(np.float64) - P.astype(np.float64)) <= 1e-5 * rn[:, None] * xn[None, :] + 1e-300)
    assert not np.array_equal(Pf, P)
    if dtype == np.float64:
        for c in (0, 31, 32, 71):
            idx = np.nonzero(R[c])[0]
            for i in (0, 7, 8, 9, 1234, n - 1):
                a, b = rowptr[i], rowptr[i + 1]
                assert P[c, i] == oracle.inner_ss(idx, R[c, idx], col[a:b], val[a:b])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csr_knn_fused_equals_general_path_and_oracle(rp, ctx, oracle, dtype):
    """SVector data through the fused query kernel (traversal + CSR row distances + selection in
    one workgroup per query): the same ids, distances and counts as the unfused general path, for
    every duplicate rule; ids equal the oracle's (true L2) wherever its distances separate."""
    n, d, T, ml, k = 20000, 200, 8, 40, 10
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.2)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    f = rp.forestBatch(9, L, ml, T, pnz, d, (rowptr, col, val.astype(dtype), d), ctx=ctx, hyperplanes=R)
    qr, qc, qv = oracle.data_normal_sparse2(6, 64, d, 0.2)
    qs = (qr, qc, qv.astype(dtype), d)
    for dedup in (False, True, rp.RPT_KNN_DEDUP_DISTANCE):
        a = rp.knnBatch(k, f, qs, dedup=dedup)
        old = ctx.set_option("knn_general", 1)
        try:
            b = rp.knnBatch(k, f, qs, dedup=dedup)
        finally:
            ctx.set_option("knn_general", old)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), dedup
    if dtype == np.float64:
        fo = oracle.forest_build_csr(rowptr, col, val, d, R, ml)
        assert np.array_equal(f.perm, fo.perm)
        ids, dist, cnt = rp.knnBatch(k, f, qs)
        for i in range(64):
            a0, b0 = qr[i], qr[i + 1]
            wi, wd = oracle.knn_csr(fo, rowptr, col, val, qc[a0:b0], qv[a0:b0], k, true_l2=True)
            assert cnt[i] == len(wi)
            assert np.allclose(dist[i, :cnt[i]], wd, rtol=1e-9, atol=1e-12)
            if len(wd) > 1 and (np.diff(wd) > 1e-9 * wd[1:]).all():
                assert np.array_equal(ids[i, :cnt[i]], wi)


def test_csr_knn_with_the_reference_metric_is_bit_identical(rp, ctx, oracle):
    """RPT_KNN_METRIC_REFERENCE: SVector data ranked by the reference's own metricSSL2
    (Internal.hs:389-393 over binSS :435-450, which stops at the shorter vector's end) — ids AND
    distance bits equal `knn metricL2` of the oracle's faithful restatement (true_l2 = False), for
    every duplicate rule; and they differ from the true-L2 answer, as they should."""
    n, d, T, ml, k = 20000, 200, 8, 40, 10
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.2)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    f = rp.forestBatch(9, L, ml, T, pnz, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
    fo = oracle.forest_build_csr(rowptr, col, val, d, R, ml)
    assert np.array_equal(f.perm, fo.perm)
    qr, qc, qv = oracle.data_normal_sparse2(6, 48, d, 0.2)
    qs = (qr, qc, qv, d)
    differs = 0
    for dedup in (False, True, rp.RPT_KNN_DEDUP_DISTANCE):
        ids, dist, cnt = rp.knnBatch(k, f, qs, dedup=dedup, reference_metric=True)
        for i in range(48):
            a0, b0 = qr[i], qr[i + 1]
            wi, wd = oracle.knn_csr(fo, rowptr, col, val, qc[a0:b0], qv[a0:b0], k,
                                    dedup=(2 if dedup == rp.RPT_KNN_DEDUP_DISTANCE else int(dedup)),
                                    true_l2=False)
            assert cnt[i] == len(wi) and np.array_equal(ids[i, :cnt[i]], wi), (dedup, i)
            assert np.array_equal(dist[i, :cnt[i]], wd), (dedup, i)
        t_ids, t_dist, _ = rp.knnBatch(k, f, qs, dedup=dedup)
        differs += int((t_dist != dist).any())
    assert differs == 3                     # the truncated tails matter on this data


def test_csr_knn_f32_prefilter_is_exact(rp, ctx, oracle):
    """f64 SVector rows, duplicates kept: candidates are ranked on the (u16 column, f32 value)
    shadow of the rows, exact distances for the best k + 6, the cut certified per query on squared
    distances (uncertifiable queries re-run exactly).  Bit-identical answers to the all-exact
    kernel on continuous data, on values rounded to one decimal (equal distances everywhere) and
    on a data set with every row twice (ties exactly at the cut: the re-run path)."""
    n, d, T, ml, k = 30000, 300, 12, 40, 10
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.2)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    qr, qc, qv = oracle.data_normal_sparse2(6, 200, d, 0.2)
    # (b): rounded values; (c): rows 0..n/2 repeated as rows n/2..n
    half = n // 2
    h_end = rowptr[half]
    rp2 = np.concatenate([rowptr[:half + 1], rowptr[1:half + 1] + h_end])
    col2, val2 = np.concatenate([col[:h_end]] * 2), np.concatenate([val[:h_end]] * 2)
    # (d): forty copies of one row near the first query: equal distances across the cut
    a0, b0 = qr[0], qr[1]
    blk_c, blk_v = qc[a0:b0], qv[a0:b0] * 1.001
    rp3 = np.concatenate([rowptr, rowptr[-1] + (b0 - a0) * np.arange(1, 41)])
    col3, val3 = np.concatenate([col] + [blk_c] * 40), np.concatenate([val] + [blk_v] * 40)
    cases = [("continuous", rowptr, col, val, qv), ("rounded", rowptr, col, np.round(val, 1) + 0.05, np.round(qv, 1) + 0.05),
             ("every row twice", rp2, col2, val2, qv), ("forty copies", rp3, col3, val3, qv)]
    import ctypes as C
    from rptree_amd import _lib
    for name, rptr, cc, vv, qvv in cases:
        f = rp.forestBatch(9, L, ml, T, pnz, d, (rptr, cc, vv, d), ctx=ctx, hyperplanes=R)
        old = ctx.set_option("knn_no_pre32", 1)
        try:
            b = rp.knnBatch(k, f, (qr, qc, qvv, d))
        finally:
            ctx.set_option("knn_no_pre32", old)
        # the (u16, f32) shadow: opt-in, no faster than the exact kernel at C3
        on, no16 = ctx.set_option("knn_csr_pre32", 1), ctx.set_option("knn_no_pre16", 1)
        try:
            a = rp.knnBatch(k, f, (qr, qc, qvv, d))
        finally:
            ctx.set_option("knn_csr_pre32", on)
            ctx.set_option("knn_no_pre16", no16)
        unc = rp.knn_last_uncertified(ctx)
        tier = C.c_int32(-1)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        assert tier.value == 1
        for x, y in zip(a, b):
            assert np.array_equal(x, y), name
        if name == "continuous":
            assert unc == 0
        if name == "forty copies":
            assert unc > 0          # the copies tie across the prefilter's cut: the re-run path
        # the default: the fixed-width half table ranks first
        h = rp.knnBatch(k, f, (qr, qc, qvv, d))
        unc = rp.knn_last_uncertified(ctx)
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        assert tier.value == 2, name
        for x, y in zip(h, b):
            assert np.array_equal(x, y), name
        if name == "continuous":
            assert unc <= 2
        if name == "forty copies":
            assert unc > 0
        f.close()


def test_csr_half_table_is_skipped_when_it_cannot_hold_the_rows(rp, ctx, oracle):
    """Values beyond the half range, or one row so long that padding every row to its length would
    more than triple the nonzeros: no half table, the exact kernel answers (tier 0), same results
    as with the tier switched off."""
    import ctypes as C
    from rptree_amd import _lib
    n, d, T, ml, k = 6000, 300, 6, 40, 5
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.05)
    qr, qc, qv = oracle.data_normal_sparse2(6, 32, d, 0.05)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(9, T, L, pnz, d)
    # (a) a value of 1e5; (b) a full row in an otherwise 5 % dense set (n * 300 > 3 nnz + 2^20)
    big = val.copy()
    big[7] = 1e5
    rp2 = np.concatenate([rowptr, [rowptr[-1] + d]])
    col2 = np.concatenate([col, np.arange(d, dtype=col.dtype)])
    val2 = np.concatenate([val, np.full(d, 0.25)])
    tier = C.c_int32(-1)
    for name, rptr, cc, vv in (("huge value", rowptr, col, big), ("one full row", rp2, col2, val2)):
        f = rp.forestBatch(9, L, ml, T, pnz, d, (rptr, cc, vv, d), ctx=ctx, hyperplanes=R)
        a = rp.knnBatch(k, f, (qr, qc, qv, d))
        _lib.check(_lib.lib().rpt_knn_last_tier(ctx._h, C.byref(tier)))
        assert tier.value == 0, name
        old = ctx.set_option("knn_no_pre32", 1)
        try:
            b = rp.knnBatch(k, f, (qr, qc, qv, d))
        finally:
            ctx.set_option("knn_no_pre32", old)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), name
        f.close()


def test_csr_dataset_borrowed_from_hbm(rp, ctx, oracle):
    """rpt_dataset_csr_dev: CSR arrays that already live in HBM give the forest of the host-copied
    dataset (and of the oracle)."""
    import torch
    n, d = 4000, 30
    rowptr, col, val = oracle.data_normal_sparse2(5, n, d, 0.3)
    R, _ = oracle.forest_hyperplanes(9, 3, 6, 0.5, d)
    tr, tc, tv = (torch.from_numpy(a).cuda() for a in (rowptr, col, val))
    ds = rp.Dataset.csr_from_torch(ctx, tr, tc, tv, d)
    f = rp._build(ctx, ds, R, 6, 25, rp.RPT_PROJ_AUTO)
    fo = oracle.forest_build_csr(rowptr, col, val, d, R, 25)
    assert np.array_equal(f.perm, fo.perm) and np.array_equal(f.thr, fo.thr, equal_nan=True)
    qr, qc, qv = oracle.data_normal_sparse2(6, 8, d, 0.3)
    ids, dist, cnt = rp.knnBatch(5, f, (qr, qc, qv, d))
    g = rp.forestBatch(9, 6, 25, 3, 0.5, d, (rowptr, col, val, d), ctx=ctx, hyperplanes=R)
    ids2, dist2, cnt2 = rp.knnBatch(5, g, (qr, qc, qv, d))
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2) and np.array_equal(cnt, cnt2)


# --------------------------------------------------------------------------- voting (8f-4)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_knn_vote_matches_keep_counts(rp, ctx, oracle, dtype):
    """RPT_KNN_VOTE(v): distances only for the points found by at least v trees (counts /
    keepCounts, RPTree.hs:464-478, restated in the oracle), ascending id order, k best by
    (distance, id).  v = 1 is 'each candidate once'; a threshold no point reaches gives nothing."""
    n, d, T, ml, k = 40000, 20, 12, 60, 10
    X = oracle.data_normal_dense2(11, n, d)
    Q = X[:200] * 1.002 + 0.004
    if dtype == "f32":
        X, Q = X.astype(np.float32), Q.astype(np.float32)
    L, _, pnz = oracle.tree_cfg(ml, n, d)
    R, _ = oracle.forest_hyperplanes(13, T, L, pnz, d)
    f = rp.forestBatch(0, L, ml, T, pnz, d, X, ctx=ctx, hyperplanes=R, mode=rp.RPT_PROJ_EXACT)
    ff = oracle.Forest(n, d, R, L, ml, f.perm, f.thr, f.mglo, f.mghi)   # the device's own forest
    kept_any = 0
    for v in (1, 2, 3, 6, T, T + 1):
        ids, dist, cnt = rp.knnBatch(k, f, Q, vote=v)
        wi, wd, wc = oracle.knn_dense_batch(ff, X, Q.astype(np.float64), k, vote_thr=v, threads=4)
        if dtype == "f64":
            assert np.array_equal(cnt, wc) and np.array_equal(ids, wi), v
            assert np.allclose(dist, wd, rtol=1e-12)
        else:   # f32 distances: ids wherever the oracle's distances are separated
            assert np.array_equal(cnt, wc), v
            for i in range(len(Q)):
                m = wc[i]
                if m > 1 and (np.diff(wd[i, :m]) <= 1e-5 * wd[i, 1:m]).any():
                    continue
                assert np.array_equal(ids[i, :m], wi[i, :m]), (v, i)
        kept_any += int(wc.sum() > 0)
        if v == T + 1:
            assert (cnt == 0).all() and (ids == -1).all()
    assert kept_any >= 5
    # every id is returned once; fewer trees agree on far points: counts shrink with v
    c1 = rp.knnBatch(k, f, Q, vote=1)[2]
    c6 = rp.knnBatch(k, f, Q, vote=6)[2]
    assert (c6 <= c1).all()


# --------------------------------------------------------------------------- import (8f-1)
def test_import_of_an_oracle_built_forest(rp, ctx, case, oracle):
    """rpt_forest_import fed with arrays the DEVICE never produced (the oracle's = what a Haskell
    host would flatten out of deserialiseRPForest): candidates and kNN must be the oracle's."""
    X, Q, R, L, ml = case
    fo = oracle.forest_build_dense(X, R, ml)
    f = rp.importForest(ctx, X, R, ml, fo.perm, fo.thr, fo.mglo, fo.mghi, mode=rp.RPT_PROJ_EXACT)
    assert f.mode == rp.RPT_PROJ_EXACT
    T = R.shape[0]
    off, cids = rp.candidatesBatch(f, Q[:60])
    for i in range(60):
        for t in range(T):
            assert np.array_equal(cids[off[i * T + t]:off[i * T + t + 1]],
                                  oracle.candidates_dense(fo, Q[i], t))
    for k, dedup in ((10, False), (3, True)):
        ids, dist, cnt = rp.knnBatch(k, f, Q, dedup=dedup)
        wi, wd, wc = oracle.knn_dense_batch(fo, X, Q, k, dedup=int(dedup), threads=4)
        assert np.array_equal(ids, wi) and np.array_equal(cnt, wc)
        assert np.allclose(dist, wd, rtol=1e-12)
    off_h, ids_h, dist_h = rp.knnHBatch(5, f, Q[:20])
    for i in range(20):
        wi, wd = oracle.knn_h_dense(fo, X, Q[i], 5)
        assert np.array_equal(ids_h[off_h[i]:off_h[i + 1]], wi)
    with pytest.raises(rp.RPTError):
        f.proj()                                   # an imported forest holds no projections


def test_save_load_keeps_the_projection_mode(rp, ctx, case, tmp_path):
    X, Q, R, L, ml = case
    f = rp.forestBatch(0, L, ml, R.shape[0], 0, X.shape[1], X, ctx=ctx, hyperplanes=R,
                       mode=rp.RPT_PROJ_MFMA)
    assert f.mode == rp.RPT_PROJ_MFMA
    path = str(tmp_path / "f.npz")
    rp.saveForest(path, f)
    g = rp.loadForest(path, X, ctx=ctx)
    assert g.mode == rp.RPT_PROJ_MFMA
    a, b = rp.knnBatch(7, f, Q), rp.knnBatch(7, g, Q)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
