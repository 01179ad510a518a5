"""CPU rehearsal of the exchange protocol of csrc/comm.hip (used by tests/test_sharded_gloo.py: world
size 2 over gloo, the shards answered by the oracle): the record layout, the shard-major gather and
the merge order are the C ABI's; only the transport differs (gloo instead of RCCL).  Test
infrastructure — the product's exchange is rpt_knn_sharded(_dev)."""
from rptree_amd.sharded import record_layout


def gather_topk(ids, dist_, cnt, group=None):
    """All-gather per-rank top-k lists -> shard-major tensors [G][nq][k], [G][nq][k], [G][nq]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    outs = []
    for x in (ids, dist_, cnt):
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)     # concatenation along dim 0
        outs.append(out.view((world,) + tuple(x.shape)))
    return tuple(outs)


class ExchangeRecord:
    """One shard's kNN result as a single byte buffer + typed views into it (dist [nq][k] f64,
    ids [nq][k] i32, count [nq] i32), in the layout of rpt_knn_record_layout."""

    def __init__(self, nq, k, device):
        import torch
        self.nq, self.k = nq, k
        self.bytes, od, oi, oc = record_layout(nq, k)
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=device)
        self.dist = self.buf[od:od + nq * k * 8].view(torch.float64).view(nq, k)
        self.ids = self.buf[oi:oi + nq * k * 4].view(torch.int32).view(nq, k)
        self.count = self.buf[oc:oc + nq * 4].view(torch.int32)

    @staticmethod
    def views_of(gathered, g, nq, k):
        """(ids, dist, count) views of shard g inside an all-gathered [G][bytes] tensor."""
        import torch
        _, od, oi, oc = record_layout(nq, k)
        row = gathered[g]
        return (row[oi:oi + nq * k * 4].view(torch.int32).view(nq, k),
                row[od:od + nq * k * 8].view(torch.float64).view(nq, k),
                row[oc:oc + nq * 4].view(torch.int32))


def gather_records(rec, group=None, out=None):
    """All-gather the ranks' exchange records -> uint8 tensor [G][bytes] (one collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, rec.bytes), dtype=torch.uint8, device=rec.buf.device)
    dist.all_gather_into_tensor(out.view(-1), rec.buf, group=group)
    return out
