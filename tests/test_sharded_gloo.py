"""world_size-2 gloo test (CPU) of the multi-GPU protocol: tree -> shard mapping, the
all-gather layout [G][nq][k] and the stable merge order.  Each rank answers its shard with the
ORACLE (there is no GPU here); the exchange is rehearsed by tests/sharded_rehearsal.py (the
C ABI's record layout over gloo instead of RCCL); the merged result
must equal the oracle's knn over the full forest."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def merge_reference(gi, gd, gc, k, dedup):
    """numpy restatement of rpt_knn_merge_dev: order (dist, shard, rank-in-shard)."""
    G, nq, _ = gi.shape
    oi = np.full((nq, k), -1, dtype=np.int32)
    od = np.full((nq, k), np.inf)
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
    return oi, od


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "rp-tree_amd", "python"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from rptree_amd import sharded
    import sharded_rehearsal as rehearsal
    n, d, T, ml, k, nq = 3000, 12, 6, 25, 8, 20
    X = o.data_normal_dense2(1234, n, d)
    L, _, pnz = o.tree_cfg(ml, n, d)
    R, _ = o.forest_hyperplanes(99, T, L, pnz, d)
    Q = o.data_normal_dense2(4321, nq, d)
    lo, hi = sharded.tree_shard(T, world, rank)
    f_local = o.forest_build_dense(X, R[lo:hi], ml)
    ids = np.full((nq, k), -1, dtype=np.int32)
    dd = np.full((nq, k), np.inf)
    cnt = np.zeros(nq, dtype=np.int32)
    for i in range(nq):
        a, b = o.knn_dense(f_local, X, Q[i], k)
        ids[i, :len(a)], dd[i, :len(a)], cnt[i] = a, b, len(a)
    gi, gd, gc = rehearsal.gather_topk(torch.from_numpy(ids), torch.from_numpy(dd),
                                     torch.from_numpy(cnt))
    ok = True
    for dedup in (False, True):
        oi, od = merge_reference(gi.numpy(), gd.numpy(), gc.numpy(), k, dedup)
        f_full = o.forest_build_dense(X, R, ml)
        for i in range(nq):
            a, b = o.knn_dense(f_full, X, Q[i], k, dedup=dedup)
            ok = ok and np.array_equal(oi[i, :len(a)], a) and np.array_equal(od[i, :len(a)], b)
    # shard layout: slot g of the gathered tensor is rank g's list
    ok = ok and np.array_equal(gi[rank].numpy(), ids)
    # the packed exchange record (ONE all-gather): same content, shard-major, layout from the C ABI
    rec = rehearsal.ExchangeRecord(nq, k, torch.device("cpu"))
    rec.ids.copy_(torch.from_numpy(ids))
    rec.dist.copy_(torch.from_numpy(dd))
    rec.count.copy_(torch.from_numpy(cnt))
    gathered = rehearsal.gather_records(rec)
    ok = ok and tuple(gathered.shape) == (world, rec.bytes)
    for g in range(world):
        vi, vd, vc = rehearsal.ExchangeRecord.views_of(gathered, g, nq, k)
        ok = ok and torch.equal(vi, gi[g]) and torch.equal(vd, gd[g]) and torch.equal(vc, gc[g])
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_tree_shard():
    from rptree_amd import sharded
    assert [sharded.tree_shard(32, 8, r) for r in (0, 3, 7)] == [(0, 4), (12, 16), (28, 32)]
    # uneven forests: the split of rpt_forest_build_sharded (rank*T // world)
    assert [sharded.tree_shard(10, 4, r) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    with pytest.raises(ValueError):
        sharded.tree_shard(3, 4, 0)


def test_record_layout():
    from rptree_amd import sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_rehearsal as rehearsal
    nq, k = 1000, 10
    b, od, oi, oc = sharded.record_layout(nq, k)
    assert (od, oi, oc) == (0, nq * k * 8, nq * k * 12)
    # distances | ids | counts | one int32 status word, rounded up to 16 bytes
    assert b % 16 == 0 and nq * k * 12 + nq * 4 + 4 <= b < nq * k * 12 + nq * 4 + 4 + 16
    rec = rehearsal.ExchangeRecord(7, 3, torch.device("cpu"))
    rec.dist.fill_(1.5)
    rec.ids.fill_(-2)
    rec.count.fill_(3)
    raw = rec.buf.numpy()
    assert np.all(raw[:7 * 3 * 8].view(np.float64) == 1.5)
    assert np.all(raw[7 * 3 * 8:7 * 3 * 12].view(np.int32) == -2)
    assert np.all(raw[7 * 3 * 12:7 * 3 * 12 + 28].view(np.int32) == 3)


def test_sharded_knn_protocol_gloo(oracle):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(out) == {0: True, 1: True}
