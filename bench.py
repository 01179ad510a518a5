#!/usr/bin/env python3
"""bench.py — the hot path at BASELINE.json's configs[1] (C2): 1M x 128 dense f64 synthetic
mixture, 32-tree forest (minLeaf 128 -> depth 13, pnz 0.4746 by rpTreeCfg), k = 10, 10 000
queries, on N GPUs of one node.

A "step" = one forest build (projection batch + median splits of all levels) of the whole
32-tree forest; with N > 1 the trees are sharded in contiguous blocks (rank r builds trees
[r*T/N, (r+1)*T/N)), X is replicated — strong scaling, no collective in the build.
The kNN leg (same K steps, its own barrier-bracketed timed region) answers all queries on every
shard, all-gathers the per-shard top-k records over RCCL (one collective, ordered on the
device with the kernels around it) and merges them (rpt_knn_merge_records_dev).

Prints ONE JSON line on rank 0.  `value` = forest-build vectors/s with the data resident in
HBM; the kNN queries/s and recall@10 of the same run are in `knn` / `recall_at_10`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rp-tree_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# dense FP64 matrix peak: v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles per SIMD
# (32 flop/clk/SIMD, half the f32 16x16x4 rate of the guide's table) x 1024 SIMDs x 2.4 GHz
MFMA_F64_PEAK_TF = 78.6


def synth(n, d, seed, device):
    """Two-Gaussian mixture per vector, N(0,0.5) or N(2,0.5) (normalDense2, Gen.hs:132-137),
    drawn on the device so nothing crosses PCIe."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    mu = (torch.rand(n, generator=g, device=device) < 0.5).to(torch.float64) * 2.0
    x = torch.randn(n, d, generator=g, device=device, dtype=torch.float64) * 0.5
    x += mu[:, None]
    return x.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npoints", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", dest="d", type=int, default=128)
    ap.add_argument("--trees", type=int, default=32)
    ap.add_argument("--min-leaf", type=int, default=128)
    ap.add_argument("--knn-k", dest="k", type=int, default=10)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--mode", choices=["auto", "exact", "mfma"], default="mfma",
                    help="projection kernel of the timed build: mfma = the north-star MFMA tile "
                         "kernel (values within 1e-5*|x||r|), exact = reference summation order "
                         "(bit-identical leaf assignment); the other mode is timed as a side leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # rehearsal switch: several ranks on ONE GPU (RCCL refuses duplicate devices, so the
    # collective goes over gloo); never used by the driver
    one_gpu = os.environ.get("RPT_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import rptree_amd as rp
    from rptree_amd import _lib
    L_ = _lib.lib()
    ctx = rp.Context(local_rank)

    N, d, T, k, nq = args.n, args.d, args.trees, args.k, args.nq
    cfg = rp.rpTreeCfg(args.min_leaf, N, d)
    maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    mode = {"auto": rp.RPT_PROJ_AUTO, "exact": rp.RPT_PROJ_EXACT, "mfma": rp.RPT_PROJ_MFMA}[args.mode]
    if T % world != 0:
        raise SystemExit("trees must be divisible by the number of GPUs")
    Tl = T // world

    # ---- synthetic inputs, resident in HBM before any timed region ----
    X = synth(N, d, 1234, dev)
    Q = synth(nq, d, 4321, dev)
    torch.cuda.synchronize()
    ds = rp.Dataset.dense_device(ctx, X.data_ptr(), N, d, rp.RPT_F64, keep=X)
    qs = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq, d, rp.RPT_F64, keep=Q)
    _, R = rp.gen.forest_hyperplanes(1235137, T, maxd, pnz, d)       # host, Batch.hs:59-61
    Rl = np.ascontiguousarray(R[rank * Tl:(rank + 1) * Tl])          # this rank's trees

    def barrier():
        if world > 1:
            if one_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        ctx.sync()
        torch.cuda.synchronize()

    def build():
        return rp._build(ctx, ds, Rl, maxd, args.min_leaf, mode)

    # one exchange record per shard (distances | ids | counts back to back): ONE all-gather
    from rptree_amd import sharded
    rec = sharded.ExchangeRecord(nq, k, dev)
    ids_l, dist_l, cnt_l = rec.ids, rec.dist, rec.count
    if world > 1:
        rec_g = torch.empty((world, rec.bytes), dtype=torch.uint8, device=dev)
        ctx_stream = torch.cuda.ExternalStream(ctx.stream, device=dev)
    ids_o = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist_o = torch.empty((nq, k), dtype=torch.float64, device=dev)
    cnt_o = torch.empty((nq,), dtype=torch.int32, device=dev)

    exchange = {"mode": os.environ.get("RPT_BENCH_EXCHANGE", "stream-ordered")}

    def knn(forest, flags):
        _lib.check(L_.rpt_knn_dev(ctx._h, forest._h, ds._h, qs._h, k, flags, ids_l.data_ptr(),
                                  dist_l.data_ptr(), cnt_l.data_ptr()))
        if world == 1:
            return ids_l, dist_l, cnt_l
        if one_gpu:                                       # gloo rehearsal: stage through the host
            ctx.sync()
            sharded.gather_records(rec, out=rec_g, via_host=True)
            torch.cuda.synchronize()
        elif exchange["mode"] == "stream-ordered":
            # RCCL over xGMI, nq*k*12 + nq*4 B per rank; issued under the ctx stream, so the
            # collective waits for the shard's kernels and the merge waits for the collective
            # on the device — no host synchronisation in between
            with torch.cuda.stream(ctx_stream):
                sharded.gather_records(rec, out=rec_g)
        else:                                             # host-synced: the conservative order
            ctx.sync()
            sharded.gather_records(rec, out=rec_g)
            torch.cuda.synchronize()
        _lib.check(L_.rpt_knn_merge_records_dev(ctx._h, rec_g.data_ptr(), rec.bytes, world, nq, k,
                                                flags, ids_o.data_ptr(), dist_o.data_ptr(),
                                                cnt_o.data_ptr()))
        ctx.sync()
        return ids_o, dist_o, cnt_o

    def settle_exchange(forest):
        """The stream-ordered exchange must give the host-synced one's result on every rank;
        otherwise (or if it raises) the whole job uses the host-synced order."""
        if world == 1 or one_gpu or exchange["mode"] != "stream-ordered":
            return
        ok = 1
        try:
            exchange["mode"] = "host-synced"
            ref = [x.clone() for x in knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)]
            exchange["mode"] = "stream-ordered"
            for _ in range(3):
                got = knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
                ok &= int(all(torch.equal(a, b) for a, b in zip(ref, got)))
        except Exception as e:                            # noqa: BLE001
            sys.stderr.write("stream-ordered exchange failed (%s): host-synced\n" % e)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        exchange["mode"] = "stream-ordered" if int(flag.item()) == 1 else "host-synced"

    # ---- warmup ----
    forest = None
    for _ in range(max(args.warmup, 0)):
        if forest is not None:
            forest.close()
        forest = build()
        knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
    if forest is None:
        forest = build()
    settle_exchange(forest)

    # ---- timed region 1: K forest builds ----
    _lib.check(L_.rpt_prof_reset(ctx._h))
    _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if forest is not None:
            forest.close()
        forest = build()
    barrier()
    t_build = time.perf_counter() - t0

    import ctypes as C

    def read_prof():
        out = {}
        for name, which in (("project", 0), ("split", 1), ("knn_plan", 2), ("knn_topk", 3),
                            ("project_wide", 4)):
            ms, cnt = C.c_double(), C.c_int64()
            _lib.check(L_.rpt_prof_get(ctx._h, which, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    prof = read_prof()                       # projection / split spans of the build region only
    _lib.check(L_.rpt_prof_reset(ctx._h))

    # ---- timed region 2: K query batches ----
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
    barrier()
    t_knn = time.perf_counter() - t0
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))

    prof_q = read_prof()
    # ---- side leg: the same K builds with the OTHER projection kernel + leaf agreement ----
    alt_name = "exact" if args.mode == "mfma" else "mfma"
    alt_mode = rp.RPT_PROJ_EXACT if alt_name == "exact" else rp.RPT_PROJ_MFMA
    alt = rp._build(ctx, ds, Rl, maxd, args.min_leaf, alt_mode)       # warm
    alt.close()
    _lib.check(L_.rpt_prof_reset(ctx._h))
    _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        alt = rp._build(ctx, ds, Rl, maxd, args.min_leaf, alt_mode)
        if _ != args.steps - 1:
            alt.close()
    barrier()
    t_alt = time.perf_counter() - t0
    prof_alt = read_prof()
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))
    # leaf-assignment agreement between the two kernels (same leaf <=> same position range)
    topo = forest.topology()
    leaf_off = np.array([o for (_, _, o, n, lf) in topo if lf], dtype=np.int64)
    pa, pb = forest.perm, alt.perm

    def leaf_index(perm_row):
        inv = np.empty(N, dtype=np.int64)
        inv[perm_row] = np.arange(N)
        return np.searchsorted(leaf_off, inv, side="right")

    flips = 0
    nt_cmp = min(Tl, 4)
    for t in range(nt_cmp):
        flips += int((leaf_index(pa[t]) != leaf_index(pb[t])).sum())
    leaf_flip_rate = flips / float(nt_cmp * N)
    # exact-order build of the first trees, kept for the full-size comparison with the oracle
    ex_forest = alt if alt_name == "exact" else forest
    ex_perm = np.array(ex_forest.perm[:3])
    ex_thr = np.array(ex_forest.thr[:3])
    alt.close()

    tt = torch.tensor([t_build, t_knn], dtype=torch.float64, device="cpu" if one_gpu else dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_build, t_knn = float(tt[0]), float(tt[1])

    prof["knn_plan"], prof["knn_topk"] = prof_q["knn_plan"], prof_q["knn_topk"]
    cand_total = C.c_int64()
    _lib.check(L_.rpt_knn_last_candidates(ctx._h, C.byref(cand_total)))
    uncertified = C.c_int64()
    _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(uncertified)))

    # ---- recall (untimed) ----
    nq_eval = min(nq, 500)
    ids_d, _, cnt_d = knn(forest, rp.RPT_KNN_DEDUP)
    ids_d = ids_d[:nq_eval].cpu().numpy()
    recall_knn = recall_ref = None
    if rank == 0:
        qe = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq_eval, d, rp.RPT_F64, keep=Q)
        true_ids, _ = rp.bruteKnn(ds, qe, k, ctx=ctx)
        hit = sum(len(set(ids_d[i].tolist()) & set(true_ids[i].tolist())) for i in range(nq_eval))
        recall_knn = hit / (nq_eval * k)
        if world == 1:
            # the reference's recallWith (RPTree.hs:259-282): mean per-tree candidate recall
            ne = min(nq_eval, 100)
            qe2 = rp.Dataset.dense_device(ctx, Q.data_ptr(), ne, d, rp.RPT_F64, keep=Q)
            off, cids = rp.candidatesBatch(forest, qe2)
            acc = 0.0
            for i in range(ne):
                kk = set(true_ids[i].tolist())
                for t in range(Tl):
                    a, b = off[i * Tl + t], off[i * Tl + t + 1]
                    acc += len(kk & set(cids[a:b].tolist())) / k
            recall_ref = acc / (ne * Tl)

    # ---- CPU baseline: the oracle on a bounded sample, rank 0, N = 1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        Xh = X.cpu().numpy()
        nt = 3 if N >= 500_000 else min(T, 8)
        t0 = time.perf_counter()
        f_cpu = orc.forest_build_dense(Xh, R[:nt], args.min_leaf)
        t_cpu = time.perf_counter() - t0
        cpu_build = N / (t_cpu / nt * T)
        # full-size parity: the oracle's trees against the exact-order device build, bit for bit
        ncmp = min(nt, len(ex_perm))
        same_trees = sum(int(np.array_equal(f_cpu.perm[t], ex_perm[t]) and
                             np.array_equal(f_cpu.thr[t], ex_thr[t], equal_nan=True))
                         for t in range(ncmp))
        # queries: the oracle's knn over the FULL forest (the device-built flat arrays; they
        # are identical to the oracle's in exact mode) for a sample of queries
        fo = orc.Forest(N, d, R, maxd, args.min_leaf, forest.perm, forest.thr, forest.mglo,
                        forest.mghi)
        nqs = 100
        Qh = Q[:nqs].cpu().numpy()
        same = 0
        knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
        got = ids_l[:nqs].cpu().numpy()
        hit_ref = hit_gpu = 0
        t0 = time.perf_counter()
        for i in range(nqs):
            wi, _ = orc.knn_dense(fo, Xh, Qh[i], k)
            same += int(np.array_equal(wi, got[i, :len(wi)]))
            if i < nq_eval:   # recall@k of the reference's knn (duplicates kept) vs brute force
                kk = set(true_ids[i].tolist())
                hit_ref += len(kk & set(wi.tolist()))
                hit_gpu += len(kk & set(got[i].tolist()))
        t_cpuq = time.perf_counter() - t0
        cpu = {"value": cpu_build, "unit": "vectors/s", "cores": 1, "kind": "port",
               "sample": "oracle (C++ restatement, g++ -O2, 1 thread) building %d of the %d trees "
                         "on the same 1M x 128 data, scaled to %d trees; knn: %d queries over the "
                         "full forest" % (nt, T, T, nqs),
               "knn_queries_per_s": nqs / t_cpuq,
               "knn_ids_identical_to_gpu": "%d/%d" % (same, nqs),
               "trees_identical_to_gpu_exact_mode": "%d/%d" % (same_trees, ncmp),
               # recall@k of `knn` (duplicates kept) against brute force on the same queries:
               # the reference restatement vs the device
               "recall_at_k_reference_vs_gpu": [hit_ref / float(min(nqs, nq_eval) * k),
                                                hit_gpu / float(min(nqs, nq_eval) * k)]}

    if rank == 0:
        p_ms, p_n = prof["project"]
        w_ms, w_n = prof["project_wide"]
        cols_total = Tl * maxd                       # hyperplanes of one build on this rank
        if w_n > 0:
            # dominant kernel: proj_mfma_wide.  Column passes as planned by project.hip
            # (launch_mfma): 96 hyperplanes per read of X, the rest in the smallest shape that
            # holds it, a tail of <= 32 folded into the last full pass (128 columns)
            wide = []
            left = cols_total
            while left > 0:
                take = left if left <= 128 else 96
                if left <= 32:
                    take = 0                     # the 32-column kernel, not a wide launch
                    left = 0
                else:
                    wide.append(take)
                    left -= take
            cols = sum(wide) / len(wide)         # real hyperplanes per wide launch (average)
            kernel = "proj_mfma_wide (MFMA f64 16x16x4, passes of %s hyperplanes over X)" % (
                "+".join(str(c) for c in wide))
            avg_ms = w_ms / w_n
        else:
            cols = cols_total * args.steps / max(p_n, 1)
            kernel = "projection batch (%s), %.1f hyperplanes per launch" % (
                "proj_mfma_fast" if args.mode == "mfma" else "proj_exact_lds", cols)
            avg_ms = p_ms / max(p_n, 1)
        # bytes = N*d*s_x + d*T*s_x + N*T*s_p, flops = 2*N*d*T for a batch of T
        # hyperplanes -- evaluated at the hyperplanes ONE launch really covers (X is read once
        # per launch, not once per 32 hyperplanes)
        bytes_per_launch = N * d * 8 + d * cols * 8 + N * cols * 8
        flops_per_launch = 2.0 * N * d * cols
        hbm_achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        mfma_achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        hbm_frac = hbm_achieved / HBM_PEAK_GBS
        mfma_frac = mfma_achieved / MFMA_F64_PEAK_TF if args.mode == "mfma" else 0.0
        # HBM traffic of the same kernel from the PMC passes committed under profiles/
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per MI355X_MICROARCH.md); only
        # valid for the exact configuration it was measured on, otherwise null
        traffic = None
        mfma_busy = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            c = pj["config"]
            kname = ("proj_mfma_wide" if w_n > 0 else
                     "proj_mfma_fast" if args.mode == "mfma" else "proj_exact_lds")
            if (c["N"], c["d"]) == (N, d) and world == 1 and \
                    int(round(pj[kname]["cols"])) == int(round(cols)):
                traffic = pj[kname]["hbm_bytes_per_launch"]
                mfma_busy = pj[kname].get("mfma_util_pct")
        except Exception:
            traffic = None
        if mfma_frac > hbm_frac:
            roof = {"bound": "mfma", "achieved": mfma_achieved, "peak": MFMA_F64_PEAK_TF,
                    "unit": "TFLOP/s", "frac": mfma_frac}
        else:
            roof = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm_frac}
        roof.update({
            "kernel": kernel, "traffic": traffic, "avg_launch_ms": avg_ms,
            "launches": w_n if w_n > 0 else p_n,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_flops_per_launch": flops_per_launch,
            "hbm": {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_frac},
            "mfma_f64": {"achieved": mfma_achieved, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                         "frac": mfma_frac,
                         # rocprofv3 --pmc MfmaUtil of the same launches (profiles/), not live
                         "mfma_busy_pmc_pct": mfma_busy},
            # the reference formulation projects one tree level (32 hyperplanes) per read of X:
            # 8(d)'s 1.28 GB per level.  All projection launches of a build against that figure:
            "survey_8d_per_level": {
                "bytes_per_forest": maxd * (N * d * 8 + d * Tl * 8 + N * Tl * 8),
                "projection_ms_per_forest": p_ms / args.steps,
                "equivalent_GBps": maxd * (N * d * 8 + d * Tl * 8 + N * Tl * 8) /
                                   (p_ms / args.steps * 1e-3) / 1e9 if p_ms > 0 else 0.0},
        })
        out = {
            "metric": "forest-build vectors/s (1M x 128 dense, 32 trees); kNN queries/s and "
                      "recall@10 in `knn` / `recall_at_10`",
            "value": N * args.steps / t_build,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_build / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: %d x %d f64 two-Gaussian mixture, %d-tree forest, minLeaf %d, "
                                   "maxDepth %d, pnz %.4f, k=%d, %d queries; trees sharded %d/GPU, "
                                   "X replicated" % (N, d, T, args.min_leaf, maxd, pnz, k, nq, Tl),
                       "projection_mode": args.mode},
            "roofline": roof,
            "cpu_baseline": cpu,
            "knn": {"value": nq * args.steps / t_knn, "unit": "queries/s",
                    "ms_per_batch": t_knn / args.steps * 1e3, "semantics": "duplicates kept "
                    "(RPTree.hs:174-176)", "candidates_per_query": cand_total.value / nq,
                    "topk_kernel_ms": prof["knn_topk"][0] / max(prof["knn_topk"][1], 1),
                    "plan_ms": prof["knn_plan"][0] / max(prof["knn_plan"][1], 1),
                    "prefilter_uncertified_queries": uncertified.value,
                    "method": "all-f64 distances" if os.environ.get("RPT_KNN_NO_PRE32") else
                    "candidates ranked on an f32 shadow of X, exact f64 distances for the best "
                    "k+6, cut certified per query (exact fallback); results identical to the "
                    "all-f64 kernel",
                    "exchange": None if world == 1 else
                    ("gloo via host (rehearsal)" if one_gpu else
                     "one RCCL all-gather of %d B records, %s" % (rec.bytes, exchange["mode"]))},
            "recall_at_10": {"forest_knn_dedup_vs_brute_force": recall_knn,
                             "reference_recallWith_mean_per_tree": recall_ref,
                             "queries": nq_eval},
            "build_breakdown_ms": {"projection_total": p_ms / args.steps,
                                   "split_total": prof["split"][0] / args.steps},
            "other_projection_mode": {
                "mode": alt_name, "value": N * args.steps / t_alt, "unit": "vectors/s",
                "ms_per_step": t_alt / args.steps * 1e3,
                "projection_avg_launch_ms": prof_alt["project"][0] / max(prof_alt["project"][1], 1),
                "leaf_assignment_flip_rate_vs_timed_mode": leaf_flip_rate,
                "note": "exact = reference summation order, bit-identical to the oracle; "
                        "flips are points whose projection is within rounding of a median"},
            "forest_stats": forest.stats(),
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
