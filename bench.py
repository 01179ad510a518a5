#!/usr/bin/env python3
"""bench.py — the hot path at BASELINE.json's configs[1] (C2): 1M x 128 dense f64 synthetic
mixture, 32-tree forest (minLeaf 128 -> depth 13, pnz 0.4746 by rpTreeCfg), k = 10, 10 000
queries, on N GPUs of one node.

A "step" = one forest build (projection batch + median splits of all levels) of the whole
32-tree forest; with N > 1 the trees are sharded in contiguous blocks (rank r builds trees
[r*T/N, (r+1)*T/N)), X is replicated — strong scaling, no collective in the build.
The kNN leg (same K steps, its own barrier-bracketed timed region) answers all queries on every
shard, all-gathers the per-shard top-k records over RCCL (one ncclAllGather on the ctx streams,
ordered on the device with the kernels around it) and merges them — all inside the C ABI
(rpt_forest_build_sharded / rpt_knn_sharded_dev, csrc/comm.hip).

Two launch styles, the same data path:
  python bench.py --gpus N                       ONE process drives the N GPUs (rpt_comm_init)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
                                                 one process per GPU (rpt_comm_init_rank; the
                                                 RCCL id travels through the launcher's process
                                                 group, which is otherwise only used for barriers)

Prints ONE JSON line on rank 0.  `value` = forest-build vectors/s with the data resident in
HBM; the kNN queries/s and recall@10 of the same run are in `knn` / `recall_at_10`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rp-tree_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# Dense FP64 matrix peak.  The guide's table has no f64 row; AMD's MI355X data sheet gives
# 78.6 TFLOP/s "peak FP64 matrix", which is the instruction's own arithmetic:
# v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles per SIMD (32 flop/clk/SIMD, half the f32
# 16x16x4 rate of the guide's table) x 1024 SIMDs x 2.4 GHz = 78.6e12.
MFMA_F64_PEAK_TF = 78.6
# ... and what a chain-free stream of v_mfma_f64_16x16x4_f64 on every SIMD sustains on the chip
# (tools/mfma_f64_peak.hip, profiles/r03_mfma_f64_peak.txt: 44-47 TFLOP/s at 2 and 4 waves per
# SIMD, over 0.4 ms and over 72 ms).  Reported next to the data-sheet figure, never as `peak`.
MFMA_F64_STREAM_TF = 46.2
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix = FP32 vector rate
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (no sparsity)
PMC_FILES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json",
             "r01_pmc_traffic.json")   # newest first


# ---------------------------------------------------------------------------------------------
# other_configs: BASELINE configs[2..4] as ONE GPU sees them (C3 whole; the tree shard of one of
# the 8 GPUs for C4 / C5, at the configuration's own point count, depth and k), N = 1 only.
# Each leg is guarded: a failure is reported in its slot and cannot lose the C2 line.
# ---------------------------------------------------------------------------------------------
def _prof_read(L_, _lib, C, ctx):
    out = {}
    for name, which in (("project", 0), ("split", 1), ("knn_plan", 2), ("knn_topk", 3),
                        ("project_wide", 4)):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(L_.rpt_prof_get(ctx._h, which, C.byref(ms), C.byref(cnt)))
        out[name] = (ms.value, cnt.value)
    return out


def _timed_leg(rp, _lib, L_, C, torch, ctx, ds, qs, R, maxd, min_leaf, k, steps):
    """steps timed builds + steps timed query batches of one configuration on one device ->
    (forest, build ms, knn ms, kernel-class times of the builds, of the queries, candidates/query)"""
    nq = qs.n
    rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_AUTO).close()          # warm
    _lib.check(L_.rpt_prof_reset(ctx._h))
    _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    ctx.sync()
    t0 = time.perf_counter()
    f = None
    for _ in range(steps):
        if f is not None:
            f.close()
        f = rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_AUTO)
    ctx.sync()
    t_build = (time.perf_counter() - t0) / steps
    prof_b = _prof_read(L_, _lib, C, ctx)
    _lib.check(L_.rpt_prof_reset(ctx._h))
    dev = torch.device("cuda", ctx.device)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)

    def knn():
        _lib.check(L_.rpt_knn_dev(ctx._h, f._h, ds._h, qs._h, k, 0, C.c_void_p(ids.data_ptr()),
                                  C.c_void_p(dist.data_ptr()), C.c_void_p(cnt.data_ptr())))
        ctx.sync()

    knn()                                                                    # warm (+ shadows)
    _lib.check(L_.rpt_prof_reset(ctx._h))
    t0 = time.perf_counter()
    for _ in range(steps):
        knn()
    t_knn = (time.perf_counter() - t0) / steps
    prof_q = _prof_read(L_, _lib, C, ctx)
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))
    cand = C.c_int64()
    _lib.check(L_.rpt_knn_last_candidates(ctx._h, C.byref(cand)))
    return f, t_build * 1e3, t_knn * 1e3, prof_b, prof_q, cand.value / float(max(nq, 1)), (ids, dist, cnt)


def _ncores(T):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, n)


def _roof(kernel, avg_ms, launches, nbytes, flops, flop_peak_tf, note):
    hbm = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 and flops else 0.0
    hf, mf = hbm / HBM_PEAK_GBS, (tf / flop_peak_tf if flop_peak_tf else 0.0)
    r = ({"bound": "mfma", "achieved": tf, "peak": flop_peak_tf, "unit": "TFLOP/s", "frac": mf}
         if mf > hf else
         {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hf})
    r.update({"kernel": kernel, "avg_launch_ms": avg_ms, "launches": launches, "traffic": None,
              "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
              "hbm_frac": hf, "mfma_frac": mf, "note": note})
    return r


def _sparse_uniform_device(torch, dev, n, d, density, seed):
    """Bernoulli support + U(0,1] values, built on the device as CSR tensors"""
    g = torch.Generator(device=dev).manual_seed(seed)
    cols, counts = [], []
    for r0 in range(0, n, 100_000):
        m = torch.rand((min(100_000, n - r0), d), device=dev, generator=g) < density
        counts.append(m.sum(dim=1))
        cols.append(m.nonzero()[:, 1].to(torch.int32))
    col = torch.cat(cols)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.cat(counts), 0)
    val = 1.0 - torch.rand(col.numel(), dtype=torch.float64, device=dev, generator=g)
    return rowptr, col, val


def other_configs(which, steps, with_cpu):
    import ctypes as C
    import torch
    import rptree_amd as rp
    from rptree_amd import _lib, gen
    L_ = _lib.lib()
    ctx = rp.default_context()
    for kv in os.environ.get("RPT_BENCH_OPTIONS", "").split(","):   # A/B experiments: name=value,...
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    dev = torch.device("cuda", ctx.device)
    ncores = _ncores(0)
    out = {"note": "BASELINE configs[2..4] on ONE GPU: C3 whole, C4 / C5 as the tree shard of one of "
                   "their 8 GPUs (all points, T/8 trees); build in the API-default projection mode of "
                   "the element type; value = points / build time with the data resident in HBM",
           "steps": steps}

    def cpu_dense(orc, Xh, R, min_leaf, T, n_sample, fdev, Qh, k, L_full):
        """oracle baselines of a dense shard: 1 thread on the first n_sample rows scaled by
        point-levels, all cores INSIDE one full-size tree; knn over the device-built forest"""
        n, d = Xh.shape
        Ls = rp.rpTreeCfg(min_leaf, n_sample, d).fpMaxTreeDepth
        Rs = np.ascontiguousarray(R[:1, :Ls])
        t0 = time.perf_counter()
        orc.forest_build_dense(Xh[:n_sample], Rs, min_leaf, threads=1)
        t1 = time.perf_counter() - t0
        per_pl = t1 / (n_sample * Ls)                      # seconds per point-level, one tree
        v1 = n / (per_pl * n * L_full * T)
        t0 = time.perf_counter()
        fo = orc.forest_build_dense(Xh, R[:1], min_leaf, threads=ncores)
        tall = time.perf_counter() - t0
        ff = orc.Forest(n, d, R, L_full, min_leaf, fdev.perm, fdev.thr, fdev.mglo, fdev.mghi)
        nqs = len(Qh)
        t0 = time.perf_counter()
        orc.knn_dense_batch(ff, Xh, Qh, k, threads=1)
        tq = time.perf_counter() - t0
        return {"value": v1, "unit": "vectors/s", "cores": 1, "kind": "port",
                "sample": "oracle, 1 thread: ONE tree on the first %d rows (depth %d), %.1f s, "
                          "scaled by point-levels to %d rows x depth %d x %d trees; knn: %d queries "
                          "over the device-built forest" % (n_sample, Ls, t1, n, L_full, T, nqs),
                "knn_queries_per_s": nqs / tq,
                "cores_all": ncores, "value_all_cores": n / (tall * T),
                "sample_all_cores": "ONE full-size tree built with all %d threads inside it "
                                    "(%.1f s), scaled to %d trees" % (ncores, tall, T)}, fo

    # ------------------------------------------------------------------ C3
    if "c3" in which:
        try:
            n, d, T, min_leaf, k, nq = 1_000_000, 784, 32, 128, 10, 10_000
            cfg = rp.rpTreeCfg(min_leaf, n, d)
            maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
            rowptr, col, val = _sparse_uniform_device(torch, dev, n, d, 0.19, 1234)
            qr, qc, qv = _sparse_uniform_device(torch, dev, nq, d, 0.19, 4321)
            ds = rp.Dataset.csr_from_torch(ctx, rowptr, col, val, d)
            qs = rp.Dataset.csr_from_torch(ctx, qr, qc, qv, d)
            _, R = gen.forest_hyperplanes(1235137, T, maxd, pnz, d)
            f, b_ms, q_ms, pb, pq, cand, _ = _timed_leg(rp, _lib, L_, C, torch, ctx, ds, qs, R, maxd,
                                                        min_leaf, k, steps)
            nnz = int(val.numel())
            tier, unc = C.c_int32(), C.c_int64()
            _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
            _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
            kp = k + max(8, k // 2)
            ell_w = (int((rowptr[1:] - rowptr[:-1]).max().item()) + 3) & ~3     # slots per row of the half table
            row_b = (cand * ell_w * 4 + min(kp, cand) * (nnz / float(n)) * 12 if tier.value == 2
                     else cand * (nnz / float(n)) * 12)
            p_ms, p_n = pb["project"]
            per_build = p_n / steps
            passes = T * maxd / 32.0
            lpp = per_build / passes                     # launches per 32-hyperplane pass (2: halves)
            pass_bytes = nnz * 12 + (n + 1) * 8 + n * 32 * 8          # SURVEY 8d, one level of 32 trees
            res = {"workload": "C3: %d x %d CSR f64, density 0.19 (%d nonzeros), %d trees, minLeaf %d, "
                               "maxDepth %d, pnz %.4f, k=%d, %d queries" % (n, d, nnz, T, min_leaf, maxd, pnz, k, nq),
                   "dtype": "f64", "projection_mode": "exact (innerSS order, bit-identical)",
                   "build_ms": b_ms, "value": n / (b_ms * 1e-3), "unit": "vectors/s",
                   "knn_ms_per_batch": q_ms, "knn_queries_per_s": nq / (q_ms * 1e-3),
                   "candidates_per_query": cand,
                   "build_breakdown_ms": {"projection_total": p_ms / steps, "split_total": pb["split"][0] / steps},
                   "roofline": _roof("proj_csr_lds32<double> (32 hyperplanes per pass over the CSR arrays, a pass = "
                                     "%.0f launches over column halves)" % lpp,
                                     p_ms / max(p_n, 1) * lpp, int(per_build / lpp), pass_bytes,
                                     2.0 * nnz * 32, 0.0,
                                     "per PASS (SURVEY 8d: nnz*12 + (N+1)*8 + N*32*8 bytes, 2*nnz*32 flops); the "
                                     "kernel is instruction-issue bound, not HBM bound (DESIGN 4.1)"),
                   "knn_ranking_tier": tier.value, "knn_uncertified": unc.value,
                   "roofline_knn": _roof("knn_fused<CSR> (%s)" %
                                         ("candidates ranked on the fixed-width half table, exact f64 "
                                          "distances for the best %d" % kp if tier.value == 2
                                          else "exact f64 distances over SVector rows"),
                                         pq["knn_topk"][0] / steps, steps,   # (per batch: a re-run of uncertified queries is a 2nd launch)
                                         nq * row_b, 0.0, 0.0,
                                         "nq x (candidates x %d slots x 4 B + %d x mean row nonzeros x 12 B)"
                                         % (ell_w, kp) if tier.value == 2
                                         else "nq x candidates x mean row nonzeros x 12 B")}
            # side leg: the tolerance mode (RPT_PROJ_MFMA on SVector rows = dense-ified bf16 x 2 rows on
            # the matrix pipe, values within 1e-5 |x||r|; the timed C3 mode above stays the exact one)
            try:
                rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_MFMA).close()
                _lib.check(L_.rpt_prof_reset(ctx._h))
                _lib.check(L_.rpt_prof_enable(ctx._h, 1))
                ctx.sync()
                t0 = time.perf_counter()
                fm = None
                for _ in range(steps):
                    if fm is not None:
                        fm.close()
                    fm = rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_MFMA)
                ctx.sync()
                tm = (time.perf_counter() - t0) / steps * 1e3
                pm = _prof_read(L_, _lib, C, ctx)
                _lib.check(L_.rpt_prof_enable(ctx._h, 0))
                topo = f.topology()
                leaf_off = np.array([o for (_, _, o, nn, lf) in topo if lf], dtype=np.int64)
                pe, pmm = f.perm[0], fm.perm[0]
                ie = np.empty(n, dtype=np.int64); ie[pe] = np.arange(n)
                im = np.empty(n, dtype=np.int64); im[pmm] = np.arange(n)
                flips = float((np.searchsorted(leaf_off, ie, side="right") !=
                               np.searchsorted(leaf_off, im, side="right")).mean())
                fm.close()
                res["tolerance_mode"] = {
                    "mode": "RPT_PROJ_MFMA on CSR rows: rows dense-ified as two bf16 terms, hyperplanes as "
                            "three, v_mfma_f32_16x16x32_bf16, f32 accumulation (1e-5 |x||r|)",
                    "build_ms": tm, "projection_ms": pm["project"][0] / steps, "split_ms": pm["split"][0] / steps,
                    "projection_launches_per_build": pm["project"][1] / steps,
                    "leaf_flip_rate_vs_exact_tree0": flips}
            except Exception as e:      # noqa: BLE001
                res["tolerance_mode"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if with_cpu:
                from oracle import oracle as orc
                hr, hc, hv = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
                t0 = time.perf_counter()
                fo1 = orc.forest_build_csr(hr, hc, hv, d, R[:1], min_leaf, threads=1)
                t1 = time.perf_counter() - t0
                nt_all = min(T, ncores)
                t0 = time.perf_counter()
                fo_all = orc.forest_build_csr(hr, hc, hv, d, R[:nt_all], min_leaf, threads=ncores)
                tall = time.perf_counter() - t0
                same = bool(np.array_equal(fo1.perm[0], f.perm[0]) and
                            np.array_equal(fo1.thr[0], f.thr[0], equal_nan=True))
                fperm, fthr, fmglo, fmghi = f.perm, f.thr, f.mglo, f.mghi
                same_all = sum(int(np.array_equal(fo_all.perm[t], fperm[t]) and
                                   np.array_equal(fo_all.thr[t], fthr[t], equal_nan=True) and
                                   np.array_equal(fo_all.mglo[t], fmglo[t], equal_nan=True) and
                                   np.array_equal(fo_all.mghi[t], fmghi[t], equal_nan=True))
                               for t in range(nt_all))
                ff = orc.Forest(n, d, R, maxd, min_leaf, f.perm, f.thr, f.mglo, f.mghi)
                hqr, hqc, hqv = qr.cpu().numpy(), qc.cpu().numpy(), qv.cpu().numpy()
                nqs = 20
                t0 = time.perf_counter()
                for i in range(nqs):
                    orc.knn_csr(ff, hr, hc, hv, hqc[hqr[i]:hqr[i + 1]], hqv[hqr[i]:hqr[i + 1]], k, true_l2=True)
                tq = time.perf_counter() - t0
                res["cpu_baseline"] = {
                    "value": n / (t1 * T), "unit": "vectors/s", "cores": 1, "kind": "port",
                    "sample": "oracle, 1 thread: ONE full-size tree (%.1f s) scaled to %d trees; knn: %d "
                              "queries over the device-built forest" % (t1, T, nqs),
                    "knn_queries_per_s": nqs / tq, "cores_all": ncores,
                    "value_all_cores": n / (tall / nt_all * T),
                    "sample_all_cores": "%d trees on %d threads (%.1f s), scaled to %d" % (nt_all, ncores, tall, T),
                    "tree0_identical_to_gpu": same,
                    "trees_identical_to_gpu": "%d/%d" % (same_all, nt_all)}
            out["c3"] = res
            f.close()
            ds.close()
            qs.close()
            del rowptr, col, val, qr, qc, qv
        except Exception as e:          # noqa: BLE001 — a failed leg must not lose the C2 line
            out["c3"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ C4 shard
    if "c4" in which:
        try:
            n, d, Tall, G, min_leaf, k, nq = 10_000_000, 128, 64, 8, 128, 10, 100_000
            T = Tall // G
            cfg = rp.rpTreeCfg(min_leaf, n, d)
            maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
            g = torch.Generator(device=dev).manual_seed(1234)
            coin = (torch.rand(n, 1, device=dev, generator=g) < 0.5).float() * 2.0
            Xd = torch.randn(n, d, device=dev, dtype=torch.float32, generator=g) * 0.5 + coin
            del coin
            qi = torch.randint(0, n, (nq,), device=dev, generator=g)
            Qd = (Xd[qi] * 1.001 + 0.003).contiguous()
            torch.cuda.synchronize(dev)
            ds = rp.Dataset.from_torch(ctx, Xd)
            qs = rp.Dataset.from_torch(ctx, Qd)
            _, R = gen.forest_hyperplanes(1235137, T, maxd, pnz, d)
            f, b_ms, q_ms, pb, pq, cand, _ = _timed_leg(rp, _lib, L_, C, torch, ctx, ds, qs, R, maxd,
                                                        min_leaf, k, steps)
            w_ms, w_n = pb["project_wide"]
            cols = T * maxd / max(w_n / steps, 1)
            tier, unc = C.c_int32(), C.c_int64()
            _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
            _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
            kp = min(k + max(48, k), 47) if tier.value == 3 else k + max(8, k // 2)   # knn.hip: kp8 / kp16
            row_b = ((d * (1 if tier.value == 3 else 2) + min(1.0, kp / max(cand, 1.0)) * d * 4)
                     if tier.value >= 2 else d * 4)
            res = {"workload": "C4 shard: %d x %d f32 two-Gaussian mixture, %d of %d trees (one of %d GPUs), "
                               "minLeaf %d, maxDepth %d, pnz %.4f, k=%d, %d queries" %
                               (n, d, T, Tall, G, min_leaf, maxd, pnz, k, nq),
                   "dtype": "f32", "projection_mode": "mfma (f32 MFMA 16x16x4, 1e-5 |x||r|)",
                   "build_ms": b_ms, "value": n / (b_ms * 1e-3), "unit": "vectors/s",
                   "knn_ms_per_batch": q_ms, "knn_queries_per_s": nq / (q_ms * 1e-3),
                   "candidates_per_query": cand,
                   "build_breakdown_ms": {"projection_total": pb["project"][0] / steps,
                                          "split_total": pb["split"][0] / steps},
                   "roofline": _roof("proj_mfma_wide<float> (%.0f hyperplanes per launch on average)" % cols,
                                     w_ms / max(w_n, 1), int(w_n / steps),
                                     n * d * 4 + d * cols * 8 + n * cols * 4, 2.0 * n * d * cols,
                                     MFMA_F32_PEAK_TF, "SURVEY 8d formula at the hyperplanes one launch covers"),
                   "knn_ranking_tier": tier.value, "knn_uncertified": unc.value,
                   "roofline_knn": _roof("knn_fused_wave<float> (one wave per query%s)" %
                                         (", candidates ranked on the %s shadow, f32 distances for "
                                          "the best %d" % ("int8" if tier.value == 3 else "IEEE-half", kp)
                                          if tier.value >= 2 else ""),
                                         pq["knn_topk"][0] / steps, steps,   # (per batch: a re-run of uncertified queries is a 2nd launch)
                                         nq * cand * row_b, 0.0, 0.0,
                                         "nq x (candidates x d x 2 B + %d x d x 4 B)" % kp
                                         if tier.value == 2 else "nq x candidates x d x 4 B")}
            if with_cpu:
                from oracle import oracle as orc
                Xh = Xd.cpu().numpy()
                Qh = Qd[:50].cpu().numpy().astype(np.float64)
                res["cpu_baseline"], _ = cpu_dense(orc, Xh, R, min_leaf, T, 1_000_000, f, Qh, k, maxd)
                del Xh
            out["c4_shard"] = res
            f.close()
            ds.close()
            qs.close()
            del Xd, Qd
        except Exception as e:          # noqa: BLE001
            out["c4_shard"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ C5 shard
    if "c5" in which:
        try:
            n, d, Tall, G, min_leaf, k, nq = 10_000_000, 768, 128, 8, 256, 50, 100_000
            T = Tall // G
            cfg = rp.rpTreeCfg(min_leaf, n, d)
            maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
            g = torch.Generator(device=dev).manual_seed(99)
            Xd = torch.empty(n, d, device=dev, dtype=torch.bfloat16)
            for r0 in range(0, n, 1_000_000):
                x = torch.randn(1_000_000, d, device=dev, dtype=torch.float32, generator=g)
                Xd[r0:r0 + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
            del x
            qi = torch.randint(0, n, (nq,), device=dev, generator=g)
            Qd = (Xd[qi].float() * 1.001 + 0.003).to(torch.bfloat16).contiguous()
            torch.cuda.synchronize(dev)
            ds = rp.Dataset.from_torch(ctx, Xd)
            qs = rp.Dataset.from_torch(ctx, Qd)
            _, R = gen.forest_hyperplanes(1235137, T, maxd, pnz, d)
            f, b_ms, q_ms, pb, pq, cand, _ = _timed_leg(rp, _lib, L_, C, torch, ctx, ds, qs, R, maxd,
                                                        min_leaf, k, steps)
            w_ms, w_n = pb["project_wide"]
            cols = T * maxd / max(w_n / steps, 1)
            terms = 3 if ctx.get_option("proj_bf16_terms") == 3 else 2
            roof = _roof("proj_bf16x3 (bf16 rows x hyperplanes split into %d bf16 terms, %.0f hyperplanes "
                         "per launch)" % (terms, cols), w_ms / max(w_n, 1), int(w_n / steps),
                         n * d * 2 + d * cols * 8 + n * cols * 4, 2.0 * n * d * cols, MFMA_BF16_PEAK_TF,
                         "algorithmic flops 2*N*d*C against the dense bf16 MFMA peak; the kernel ISSUES %d "
                         "times that (r = r_hi + r_mid [+ r_lo]: |error| <= 2^-17 |x||r| with two terms, "
                         "inside the 1e-5 tolerance; option proj_bf16_terms = 3 keeps 24 bits)" % terms)
            roof["mfma_issued_frac"] = float(terms) * roof["mfma_frac"]
            tier, unc = C.c_int32(), C.c_int64()
            _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
            _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(unc)))
            kp = min(k + max(48, k), 223)                # entries the int8 tier keeps (knn.hip: kp8)
            row_b = d * 1 + min(1.0, kp / max(cand, 1.0)) * d * 2 if tier.value == 3 else d * 2
            res = {"workload": "C5 shard: %d x %d bf16 unit-norm rows, %d of %d trees (one of %d GPUs), minLeaf %d, "
                               "maxDepth %d, pnz %.4f, k=%d, %d queries" %
                               (n, d, T, Tall, G, min_leaf, maxd, pnz, k, nq),
                   "dtype": "bf16", "projection_mode": "mfma (bf16 MFMA 16x16x32, hyperplanes as %d bf16 terms, f32 accumulation, "
                                                       "1e-5 |x||r|)" % terms,
                   "build_ms": b_ms, "value": n / (b_ms * 1e-3), "unit": "vectors/s",
                   "knn_ms_per_batch": q_ms, "knn_queries_per_s": nq / (q_ms * 1e-3),
                   "candidates_per_query": cand,
                   "build_breakdown_ms": {"projection_total": pb["project"][0] / steps,
                                          "split_total": pb["split"][0] / steps},
                   "roofline": roof,
                   "knn_ranking_tier": tier.value, "knn_uncertified": unc.value,
                   "roofline_knn": _roof("knn_fused<bf16>%s" %
                                         (" (candidates ranked on the int8 shadow, f32 distances over the bf16 "
                                          "rows of the best %d)" % kp if tier.value == 3 else ""),
                                         pq["knn_topk"][0] / steps,
                                         steps, nq * cand * row_b, 0.0, 0.0,
                                         "nq x candidates x (d x 1 B int8 rows + k'/candidates x d x 2 B bf16 rows)"
                                         if tier.value == 3 else "nq x candidates x d x 2 B")}
            # side leg: the same build with THREE bf16 terms per hyperplane (the rounds 1-3 kernel's accuracy):
            # what the third term costs, and how many leaf assignments of tree 0 the two-term build moves
            if terms == 2:
                try:
                    old = ctx.set_option("proj_bf16_terms", 3)
                    try:
                        rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_AUTO).close()
                        _lib.check(L_.rpt_prof_reset(ctx._h))
                        _lib.check(L_.rpt_prof_enable(ctx._h, 1))
                        ctx.sync()
                        t0 = time.perf_counter()
                        f3 = None
                        for _ in range(steps):
                            if f3 is not None:
                                f3.close()
                            f3 = rp._build(ctx, ds, R, maxd, min_leaf, rp.RPT_PROJ_AUTO)
                        ctx.sync()
                        t3 = (time.perf_counter() - t0) / steps * 1e3
                        p3 = _prof_read(L_, _lib, C, ctx)
                        _lib.check(L_.rpt_prof_enable(ctx._h, 0))
                    finally:
                        ctx.set_option("proj_bf16_terms", old)
                    leaf_off = np.array([o for (_, _, o, nn, lf) in f.topology() if lf], dtype=np.int64)
                    i2 = np.empty(n, dtype=np.int64); i2[f.perm[0]] = np.arange(n)
                    i3 = np.empty(n, dtype=np.int64); i3[f3.perm[0]] = np.arange(n)
                    flips = float((np.searchsorted(leaf_off, i2, side="right") !=
                                   np.searchsorted(leaf_off, i3, side="right")).mean())
                    f3.close()
                    del i2, i3
                    res["three_term_mode"] = {
                        "mode": "option proj_bf16_terms = 3: hyperplanes as three bf16 terms (24 bits)",
                        "build_ms": t3, "projection_ms": p3["project"][0] / steps,
                        "split_ms": p3["split"][0] / steps,
                        "leaf_flip_rate_two_vs_three_terms_tree0": flips}
                except Exception as e:      # noqa: BLE001
                    res["three_term_mode"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if with_cpu:
                from oracle import oracle as orc
                Xh = np.empty((n, d), dtype=np.float32)
                for r0 in range(0, n, 1_000_000):
                    Xh[r0:r0 + 1_000_000] = Xd[r0:r0 + 1_000_000].float().cpu().numpy()
                Qh = Qd[:20].float().cpu().numpy().astype(np.float64)
                res["cpu_baseline"], _ = cpu_dense(orc, Xh, R, min_leaf, T, 300_000, f, Qh, k, maxd)
                del Xh
            out["c5_shard"] = res
            f.close()
            ds.close()
            qs.close()
            del Xd, Qd
        except Exception as e:          # noqa: BLE001
            out["c5_shard"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------
# shard_sweep: what ONE GPU of an N-GPU run does, measured on one GPU.  The trees of a forest are
# independent (Internal.hs:234-240) and knn concatenates per-tree candidates (RPTree.hs:174-176), so
# GPU r of G builds and queries trees [r*T/G, (r+1)*T/G) of the SAME point set: a run at G GPUs is as
# fast as its slowest shard plus the exchange.  For C2 (T = 32) and C4 (T = 64) this leg times the
# shard sizes T/8, T/4, T/2 and T: build ms, kNN ms per batch of 10 000 and of 100 000 queries, both
# through rpt_knn_sharded_dev on a one-rank communicator WITH the exchange forced (record ->
# ncclAllGather -> merge -> status scan: the code path of every rank; what one GPU cannot show is
# the transfer over xGMI and G records from G devices).  `projected_speedup_8gpu` = time of the
# whole forest on one GPU / time of the T/8 shard incl. the forced exchange — a PROJECTION from
# one-GPU measurements, never a measured scaling number (the driver's SCALE run is that).
# ---------------------------------------------------------------------------------------------
def shard_sweep(which, steps):
    import ctypes as C
    import torch
    import rptree_amd as rp
    from rptree_amd import _lib, gen, sharded
    L_ = _lib.lib()
    comm = sharded.Comm.local(1)
    ctx = comm.contexts[0]
    dev = torch.device("cuda", ctx.device)
    out = {"note": "one GPU, tree shards of the whole point set (what 1 of G GPUs holds); kNN through "
                   "rpt_knn_sharded_dev with comm_force_exchange = 1 (one-rank ncclAllGather + merge); every "
                   "figure is the best of the timed calls (the mean is next to it: *_mean); "
                   "projected_speedup_8gpu = whole forest on one GPU / the T/8 shard, a projection",
           "steps": steps}

    def leg(name, Xd, Qs, T_all, min_leaf, k, mode, dt):
        n, d = Xd.shape
        cfg = rp.rpTreeCfg(min_leaf, n, d)
        maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
        ds = rp.Dataset.from_torch(ctx, Xd)
        _, R = gen.forest_hyperplanes(1235137, T_all, maxd, pnz, d)
        rows = []
        for T in (T_all // 8, T_all // 4, T_all // 2, T_all):
            row = {"trees": T}
            try:
                sharded.ShardedForest(comm, [ds], R[:T], maxd, min_leaf, mode).close()      # warm
                comm.sync()
                tb = []
                for _ in range(steps):                 # timed one by one: a 40 GB forest's first
                    t0 = time.perf_counter()           # allocations can stall a build by 100 ms
                    sf = sharded.ShardedForest(comm, [ds], R[:T], maxd, min_leaf, mode)
                    comm.sync()
                    tb.append((time.perf_counter() - t0) * 1e3)
                    if _ < steps - 1:
                        sf.close()
                row["build_ms"] = min(tb)
                row["build_ms_mean"] = sum(tb) / len(tb)
                for nq, Qd in Qs:
                    qs = rp.Dataset.from_torch(ctx, Qd)
                    oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
                    od = torch.empty((nq, k), dtype=torch.float64, device=dev)
                    oc = torch.empty((nq,), dtype=torch.int32, device=dev)
                    torch.cuda.synchronize(dev)
                    for force in (0, 1):
                        ctx.set_option("comm_force_exchange", force)
                        for _ in range(2):
                            sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
                            comm.sync()
                        tq = []
                        for _ in range(max(steps, 5)):
                            t0 = time.perf_counter()
                            sf.knn_dev([qs], k, 0, [oi.data_ptr()], [od.data_ptr()], [oc.data_ptr()])
                            comm.sync()
                            tq.append((time.perf_counter() - t0) * 1e3)
                        key = "knn_ms_nq%d%s" % (nq, "_forced_exchange" if force else "")
                        row[key] = min(tq)
                        row[key + "_mean"] = sum(tq) / len(tq)
                    ctx.set_option("comm_force_exchange", 0)
                    tier = C.c_int32()
                    _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(tier)))
                    row["knn_ranking_tier"] = tier.value
                    qs.close()
                    del oi, od, oc
                sf.close()
            except Exception as e:          # noqa: BLE001
                row["error"] = "%s: %s" % (type(e).__name__, e)
                ctx.set_option("comm_force_exchange", 0)
            rows.append(row)
        res = {"workload": "%s: %d x %d %s, %d trees in all, minLeaf %d, maxDepth %d, k=%d" %
                           (name, n, d, dt, T_all, min_leaf, maxd, k), "shards": rows}
        try:
            whole, eighth = rows[-1], rows[0]
            proj = {"build": whole["build_ms"] / eighth["build_ms"]}
            for nq, _ in Qs:
                proj["knn_nq%d" % nq] = whole["knn_ms_nq%d" % nq] / eighth["knn_ms_nq%d_forced_exchange" % nq]
            res["projected_speedup_8gpu"] = proj
        except Exception:                   # noqa: BLE001
            res["projected_speedup_8gpu"] = None
        ds.close()
        return res

    if "c2" in which:
        try:
            X = gen.normal_dense2_torch(1234, 1_000_000, 128, dev)
            Qb = gen.normal_dense2_torch(4321, 100_000, 128, dev)
            torch.cuda.synchronize(dev)
            out["c2"] = leg("C2", X, [(10_000, Qb[:10_000].contiguous()), (100_000, Qb)], 32, 128, 10,
                            rp.RPT_PROJ_MFMA, "f64")
            del X, Qb
        except Exception as e:              # noqa: BLE001
            out["c2"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    if "c4" in which:
        try:
            n, d = 10_000_000, 128
            g = torch.Generator(device=dev).manual_seed(1234)
            coin = (torch.rand(n, 1, device=dev, generator=g) < 0.5).float() * 2.0
            Xd = torch.randn(n, d, device=dev, dtype=torch.float32, generator=g) * 0.5 + coin
            del coin
            qi = torch.randint(0, n, (100_000,), device=dev, generator=g)
            Qd = (Xd[qi] * 1.001 + 0.003).contiguous()
            torch.cuda.synchronize(dev)
            out["c4"] = leg("C4", Xd, [(10_000, Qd[:10_000].contiguous()), (100_000, Qd)], 64, 128, 10,
                            rp.RPT_PROJ_AUTO, "f32")
            del Xd, Qd
        except Exception as e:              # noqa: BLE001
            out["c4"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    comm.close()
    return out


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
    ap.add_argument("--other-configs", default="c3,c4,c5",
                    help="comma list of the other BASELINE configurations to time after C2 at N = 1 "
                         "(c3, c4, c5; 'none' to skip): build / kNN / roofline / cpu_baseline each")
    ap.add_argument("--shard-sweep", default="c2,c4",
                    help="comma list of the configurations whose tree shards (T/8 .. T trees on one GPU) are "
                         "timed after C2 at N = 1 (c2, c4; 'none' to skip): build / kNN at 10 000 and 100 000 "
                         "queries incl. the forced one-rank exchange, projected_speedup_8gpu")
    ap.add_argument("--_other-child", dest="other_child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--_sweep-child", dest="sweep_child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sweep_child:       # child process of the shard_sweep leg: ONE JSON line on stdout
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        res = shard_sweep([w for w in args.shard_sweep.split(",") if w], max(1, min(args.steps, 3)))
        os.write(json_fd, (json.dumps(res) + "\n").encode())
        return
    if args.other_child:       # child process of the other_configs leg: ONE JSON line on stdout
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        res = other_configs([w for w in args.other_configs.split(",") if w], max(1, min(args.steps, 5)),
                            not args.no_cpu_baseline)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
        return

    rank = int(os.environ.get("RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launcher = env_world > 1
    if launcher and env_world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, env_world))
    world = args.gpus
    if world < 1:
        raise SystemExit("--gpus must be >= 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries exactly ONE line, the JSON result: native libraries (RCCL prints a version
    # banner on stdout when its first communicator is made) and stray prints go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import rptree_amd as rp
    from rptree_amd import _lib, gen, sharded
    L_ = _lib.lib()

    # ---- devices and the communicator (C ABI, librccl) ----
    if launcher:
        # control plane only (rendezvous, barriers, max over ranks): gloo.  The data path's
        # collective is the library's own RCCL communicator.
        dist.init_process_group("gloo")
        torch.cuda.set_device(local_rank)
        ctx0 = rp.Context(local_rank)
        comm = sharded.Comm.from_process_group(ctx0)
        launch = "one process per GPU (torch.distributed.run), rpt_comm_init_rank"
    else:
        have = C.c_int32()
        if L_.rpt_device_count(C.byref(have)) != 0:
            raise SystemExit("--gpus %d: no HIP device visible to this process (%s)"
                             % (world, L_.rpt_last_error().decode()))
        if have.value < world:
            raise SystemExit("--gpus %d: only %d HIP device(s) visible to this process"
                             % (world, have.value))
        comm = sharded.Comm.local(world)
        launch = "one process, rpt_comm_init(%d)" % world
    ctxs = comm.contexts
    nloc = comm.nlocal
    devs = [torch.device("cuda", c.device) for c in ctxs]
    ctx = ctxs[0]                       # kernel timing and the untimed evaluation legs: device 0

    N, d, T, k, nq = args.n, args.d, args.trees, args.k, args.nq
    cfg = rp.rpTreeCfg(args.min_leaf, N, d)
    maxd, pnz = cfg.fpMaxTreeDepth, cfg.fpProjNzDensity
    mode = {"auto": rp.RPT_PROJ_AUTO, "exact": rp.RPT_PROJ_EXACT, "mfma": rp.RPT_PROJ_MFMA}[args.mode]
    if T < world:
        raise SystemExit("need at least one tree per GPU")
    lo0, hi0 = sharded.tree_shard(T, world, comm.first_rank)
    Tl = hi0 - lo0                      # trees on this process's first device

    # ---- synthetic inputs (SplitMix64 streams, seeds 1234 / 4321 / 1235137), drawn on
    # the device and resident in HBM before any timed region ----
    X0 = gen.normal_dense2_torch(1234, N, d, devs[0])
    Q0 = gen.normal_dense2_torch(4321, nq, d, devs[0])
    Xs = [X0] + [X0.to(dv) for dv in devs[1:]]           # replicas over xGMI
    Qs = [Q0] + [Q0.to(dv) for dv in devs[1:]]
    for dv in devs:
        torch.cuda.synchronize(dv)
    dss = [rp.Dataset.from_torch(c, x) for c, x in zip(ctxs, Xs)]
    qss = [rp.Dataset.from_torch(c, q) for c, q in zip(ctxs, Qs)]
    ds, qs, X, Q = dss[0], qss[0], X0, Q0
    _, R = gen.forest_hyperplanes(1235137, T, maxd, pnz, d)          # host, Batch.hs:59-61

    def barrier():
        comm.sync()
        for dv in devs:
            torch.cuda.synchronize(dv)
        if launcher:
            dist.barrier()

    def build(m=mode):
        return sharded.ShardedForest(comm, dss, R, maxd, args.min_leaf, m)

    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dv),
             torch.empty((nq, k), dtype=torch.float64, device=dv),
             torch.empty((nq,), dtype=torch.int32, device=dv)) for dv in devs]
    for dv in devs:
        torch.cuda.synchronize(dv)
    o_ids = [o[0].data_ptr() for o in outs]
    o_dist = [o[1].data_ptr() for o in outs]
    o_cnt = [o[2].data_ptr() for o in outs]

    def knn(forest, flags):
        """kernels -> ONE ncclAllGather -> merge, all enqueued on the ctx streams; one host
        synchronisation at the end.  Returns device 0's (ids, dist, count)."""
        forest.knn_dev(qss, k, flags, o_ids, o_dist, o_cnt)
        comm.sync()
        return outs[0]

    # ---- warmup ----
    forest = None
    for _ in range(max(args.warmup, 0)):
        if forest is not None:
            forest.close()
        forest = build()
        knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
    if forest is None:
        forest = build()

    def read_prof():
        out = {}
        for name, which in (("project", 0), ("split", 1), ("knn_plan", 2), ("knn_topk", 3),
                            ("project_wide", 4)):
            ms, cnt = C.c_double(), C.c_int64()
            _lib.check(L_.rpt_prof_get(ctx._h, which, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

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
    build(alt_mode).close()                                          # warm
    _lib.check(L_.rpt_prof_reset(ctx._h))
    _lib.check(L_.rpt_prof_enable(ctx._h, 1))
    barrier()
    t0 = time.perf_counter()
    alt = None
    for _ in range(args.steps):
        if alt is not None:
            alt.close()
        alt = build(alt_mode)
    barrier()
    t_alt = time.perf_counter() - t0
    prof_alt = read_prof()
    _lib.check(L_.rpt_prof_enable(ctx._h, 0))
    # leaf-assignment agreement between the two kernels (same leaf <=> same position range)
    f_loc, _, _ = forest.local(0)
    a_loc, _, _ = alt.local(0)
    topo = f_loc.topology()
    leaf_off = np.array([o for (_, _, o, n, lf) in topo if lf], dtype=np.int64)
    pa, pb = f_loc.perm, a_loc.perm

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
    ex_loc = a_loc if alt_name == "exact" else f_loc
    ex_perm = np.array(ex_loc.perm)            # every tree of this rank (C2: 32 x 4 MB)
    ex_thr, ex_mglo, ex_mghi = np.array(ex_loc.thr), np.array(ex_loc.mglo), np.array(ex_loc.mghi)
    alt.close()

    # ---- the survey's own projection batch (ONE level x 32 trees = 32 hyperplanes per
    # read of X, 1.280 GB at C2): the same K builds with 32 hyperplanes per pass (proj_narrow:
    # 13 x proj_mfma_fast at C2).  The timed headline reads X once per 96-128 hyperplanes, which is
    # faster overall; this leg is the unit north_star quotes its ">= 60 % of HBM" on ----
    per_level = None
    if args.mode == "mfma" and world == 1:
        per_level = {}
        for tag, opts in (("with_codes", {"proj_narrow": 1}), ("no_codes", {"proj_narrow": 1, "no_codes": 1})):
            olds = {o: ctx.set_option(o, v) for o, v in opts.items()}
            try:
                build(rp.RPT_PROJ_MFMA).close()
                _lib.check(L_.rpt_prof_reset(ctx._h))
                _lib.check(L_.rpt_prof_enable(ctx._h, 1))
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    build(rp.RPT_PROJ_MFMA).close()
                barrier()
                t_nar = (time.perf_counter() - t0) / args.steps
                pn = read_prof()
                _lib.check(L_.rpt_prof_enable(ctx._h, 0))
                n_ms, n_n = pn["project"]
                lb = N * d * 8 + d * 32 * 8 + N * 32 * 8
                avg = n_ms / max(n_n, 1)
                per_level[tag] = {
                    "kernel": "proj_mfma_fast (MFMA f64 16x16x4, 32 hyperplanes per pass over X%s)" %
                              ("; the pass also writes the split's 16-bit codes" if tag == "with_codes" else ""),
                    "launches_per_build": n_n / args.steps, "avg_launch_ms": avg,
                    "algorithmic_bytes_per_launch": lb,
                    "achieved": lb / (avg * 1e-3) / 1e9 if avg > 0 else 0.0, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": lb / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0,
                    "bound": "hbm", "build_ms": t_nar * 1e3,
                    "projection_ms_per_build": n_ms / args.steps, "split_ms_per_build": pn["split"][0] / args.steps}
            except Exception as e:      # noqa: BLE001
                per_level[tag] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                for o, v in olds.items():
                    ctx.set_option(o, v)
        per_level["note"] = ("SURVEY 8d's per-level projection batch (N*d*8 + d*32*8 + N*32*8 = 1.280 GB, "
                             "8.19 GFLOP); HIP events around every launch; profiles/r04_narrow_* hold the "
                             "rocprofv3 kernel trace and FETCH/WRITE of the same launches")

    if launcher:
        tt = torch.tensor([t_build, t_knn], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_build, t_knn = float(tt[0]), float(tt[1])

    prof["knn_plan"], prof["knn_topk"] = prof_q["knn_plan"], prof_q["knn_topk"]
    cand_total = C.c_int64()
    _lib.check(L_.rpt_knn_last_candidates(ctx._h, C.byref(cand_total)))
    uncertified = C.c_int64()
    _lib.check(L_.rpt_knn_last_uncertified(ctx._h, C.byref(uncertified)))
    knn_tier = C.c_int32()         # 0 all-f64, 1 f32 shadow, 2 IEEE-half shadow, 3 int8 shadow (timed batches)
    _lib.check(L_.rpt_knn_last_tier(ctx._h, C.byref(knn_tier)))

    # ---- recall (untimed) ----
    nq_eval = min(nq, 500)
    ids_d, _, cnt_d = knn(forest, rp.RPT_KNN_DEDUP)
    ids_d = ids_d[:nq_eval].cpu().numpy()
    recall_knn = recall_ref = None
    true_ids = None
    if rank == 0:
        qe = rp.Dataset.dense_device(ctx, Q.data_ptr(), nq_eval, d, rp.RPT_F64, keep=Q)
        true_ids, _ = rp.bruteKnn(ds, qe, k, ctx=ctx)
        hit = sum(len(set(ids_d[i].tolist()) & set(true_ids[i].tolist())) for i in range(nq_eval))
        recall_knn = hit / (nq_eval * k)
        if world == 1:
            # the reference's recallWith (RPTree.hs:259-282): mean per-tree candidate recall
            ne = min(nq_eval, 100)
            qe2 = rp.Dataset.dense_device(ctx, Q.data_ptr(), ne, d, rp.RPT_F64, keep=Q)
            off, cids = rp.candidatesBatch(f_loc, qe2)
            acc = 0.0
            for i in range(ne):
                kk = set(true_ids[i].tolist())
                for t in range(Tl):
                    a, b = off[i * Tl + t], off[i * Tl + t + 1]
                    acc += len(kk & set(cids[a:b].tolist())) / k
            recall_ref = acc / (ne * Tl)

    # ---- PCIe-inclusive rates (SURVEY 8d "also including H2D"), rank 0, N = 1: host buffers in,
    # host results out, through the *_host entry points; never part of `value` ----
    h2d = None
    if rank == 0 and world == 1:
        Xh = X.cpu().numpy()
        Qh = Q.cpu().numpy()
        best_b = best_q = 1e30
        parts = None
        for _ in range(2):
            t0 = time.perf_counter()
            dsh = rp.Dataset.dense(ctx, Xh)
            t1 = time.perf_counter()
            fh = rp._build(ctx, dsh, R, maxd, args.min_leaf, mode)
            ctx.sync()
            t2 = time.perf_counter()
            fh.perm, fh.thr                                   # copy-out accessors (perm + node arrays)
            t3 = time.perf_counter()
            if t3 - t0 < best_b:
                best_b, parts = t3 - t0, (t1 - t0, t2 - t1, t3 - t2)
            t0 = time.perf_counter()
            rp.knnBatch(k, fh, Qh)                            # host queries in, host results out
            best_q = min(best_q, time.perf_counter() - t0)
            fh.close()
            dsh.close()
        h2d = {"build_vectors_per_s": N / best_b, "knn_queries_per_s": nq / best_q,
               "upload_ms": parts[0] * 1e3, "build_ms": parts[1] * 1e3,
               "download_perm_nodes_ms": parts[2] * 1e3, "knn_ms_per_batch": best_q * 1e3,
               "note": "pageable host buffers: upload of X + build + download of perm and node "
                       "arrays; knn = rpt_knn_host incl. query upload and result download (the "
                       "first of the two passes also builds the f32 shadow)"}

    # ---- CPU baseline: the oracle on a bounded sample, rank 0, N = 1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        Xh = X.cpu().numpy()
        # threads actually used by the all-core leg: the CPUs this process may run on, at most
        # one per tree (trees are the unit of parallelism)
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        ncores = max(1, min(ncores, T))
        # (i) the reference is single-threaded: 1 thread, a few trees, scaled to the forest
        nt = 3 if N >= 500_000 else min(T, 8)
        t0 = time.perf_counter()
        f_cpu = orc.forest_build_dense(Xh, R[:nt], args.min_leaf, threads=1)
        t_cpu = time.perf_counter() - t0
        cpu_build = N / (t_cpu / nt * T)
        # (ii) courtesy upper baseline: trees in parallel over all host cores
        nt_all = T                       # the whole forest on all cores
        t0 = time.perf_counter()
        f_all = orc.forest_build_dense(Xh, R[:nt_all], args.min_leaf, threads=ncores)
        t_all = time.perf_counter() - t0
        cpu_build_all = N / (t_all / nt_all * T)
        assert np.array_equal(f_all.perm[:nt], f_cpu.perm)
        # full-size parity: EVERY tree the all-core leg built against the exact-order device build,
        # bit for bit (leaf assignment, thresholds, margins: Internal.hs:484-505 at every node)
        ncmp = min(nt_all, len(ex_perm))
        same_trees = sum(int(np.array_equal(f_all.perm[t], ex_perm[t]) and
                             np.array_equal(f_all.thr[t], ex_thr[t], equal_nan=True) and
                             np.array_equal(f_all.mglo[t], ex_mglo[t], equal_nan=True) and
                             np.array_equal(f_all.mghi[t], ex_mghi[t], equal_nan=True))
                         for t in range(ncmp))
        # queries: the oracle's knn over the FULL forest (the device-built flat arrays; they
        # are identical to the oracle's in exact mode) for a sample of queries
        fo = orc.Forest(N, d, R, maxd, args.min_leaf, f_loc.perm, f_loc.thr, f_loc.mglo, f_loc.mghi)
        nqs = 100
        Qh = Q[:max(nqs, 2000)].cpu().numpy()
        knn(forest, rp.RPT_KNN_KEEP_DUPLICATES)
        got = outs[0][0][:nqs].cpu().numpy()
        t0 = time.perf_counter()
        wi, _, wc = orc.knn_dense_batch(fo, Xh, Qh[:nqs], k, threads=1)
        t_cpuq = time.perf_counter() - t0
        same = sum(int(np.array_equal(wi[i, :wc[i]], got[i, :wc[i]])) for i in range(nqs))
        hit_ref = sum(len(set(true_ids[i].tolist()) & set(wi[i].tolist())) for i in range(min(nqs, nq_eval)))
        hit_gpu = sum(len(set(true_ids[i].tolist()) & set(got[i].tolist())) for i in range(min(nqs, nq_eval)))
        nq_all = len(Qh)
        t0 = time.perf_counter()
        orc.knn_dense_batch(fo, Xh, Qh, k, threads=ncores)
        t_cpuq_all = time.perf_counter() - t0
        cpu = {"value": cpu_build, "unit": "vectors/s", "cores": 1, "kind": "port",
               "sample": "oracle (C++ restatement of the reference, g++ -O2, 1 thread — the "
                         "reference is single-threaded) building %d of the %d trees on the same "
                         "%d x %d data, scaled to %d trees; knn: %d queries over the full forest"
                         % (nt, T, N, d, T, nqs),
               "knn_queries_per_s": nqs / t_cpuq,
               "cores_all": ncores,
               "value_all_cores": cpu_build_all,
               "knn_queries_per_s_all_cores": nq_all / t_cpuq_all,
               "sample_all_cores": "courtesy upper baseline (SURVEY 8d-ii): all %d trees built "
                                   "concurrently on %d threads (%.1f s, nothing scaled); %d queries "
                                   "answered concurrently" % (nt_all, ncores, t_all, nq_all),
               "knn_ids_identical_to_gpu": "%d/%d" % (same, nqs),
               "trees_identical_to_gpu_exact_mode": "%d/%d" % (same_trees, ncmp),
               # recall@k of `knn` (duplicates kept) against brute force on the same queries:
               # the reference restatement vs the device
               "recall_at_k_reference_vs_gpu": [hit_ref / float(min(nqs, nq_eval) * k),
                                                hit_gpu / float(min(nqs, nq_eval) * k)]}

    # ---- the exchange on the hardware this run has: a 4-tree shard (what one of 8 GPUs holds at
    # C2) answered through record -> ncclAllGather -> merge on the ONE-rank communicator
    # (comm_force_exchange), against the same shard without the exchange ----
    exch = None
    if rank == 0 and world == 1 and T >= 8:
        try:
            Ts = T // 8
            sf4 = sharded.ShardedForest(comm, dss, R[:Ts], maxd, args.min_leaf, mode)
            tms = {}
            for force in (0, 1):
                ctx.set_option("comm_force_exchange", force)
                for _ in range(3):
                    knn(sf4, rp.RPT_KNN_KEEP_DUPLICATES)
                barrier()
                t0 = time.perf_counter()
                for _ in range(max(args.steps, 5)):
                    knn(sf4, rp.RPT_KNN_KEEP_DUPLICATES)
                barrier()
                tms[force] = (time.perf_counter() - t0) / max(args.steps, 5) * 1e3
            ctx.set_option("comm_force_exchange", 0)
            t0 = time.perf_counter()
            for _ in range(max(args.steps, 5)):
                sharded.ShardedForest(comm, dss, R[:Ts], maxd, args.min_leaf, mode).close()
            barrier()
            b4 = (time.perf_counter() - t0) / max(args.steps, 5) * 1e3
            exch = {"trees": Ts, "queries": nq, "k": k, "record_bytes": sharded.record_layout(nq, k)[0],
                    "knn_ms_without_exchange": tms[0], "knn_ms_with_forced_exchange": tms[1],
                    "exchange_one_rank_ms": tms[1] - tms[0], "build_ms": b4,
                    "note": "one rank: the all-gather moves the record inside the device; what is "
                            "measured is the enqueue + merge + status-scan cost of the path every rank "
                            "of an N-GPU run takes, not xGMI transfer time"}
            sf4.close()
        except Exception as e:      # noqa: BLE001
            exch = {"error": "%s: %s" % (type(e).__name__, e)}
            ctx.set_option("comm_force_exchange", 0)

    # ---- the other BASELINE configurations, in a child process (a crash or hang there cannot
    # lose this line) ----
    other = None
    if rank == 0 and world == 1 and args.other_configs not in ("", "none"):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--_other-child", "--other-configs",
               args.other_configs, "--steps", str(args.steps)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=420)
            line = pr.stdout.decode().strip().splitlines()[-1] if pr.stdout.strip() else ""
            other = json.loads(line) if line else {"error": "child exited with %d, no output" % pr.returncode}
        except subprocess.TimeoutExpired:
            other = {"error": "other_configs child exceeded 420 s"}
        except Exception as e:      # noqa: BLE001
            other = {"error": "%s: %s" % (type(e).__name__, e)}

    sweep = None
    if rank == 0 and world == 1 and args.shard_sweep not in ("", "none"):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--_sweep-child", "--shard-sweep", args.shard_sweep,
               "--steps", str(args.steps)]
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            line = pr.stdout.decode().strip().splitlines()[-1] if pr.stdout.strip() else ""
            sweep = json.loads(line) if line else {"error": "child exited with %d, no output" % pr.returncode}
        except subprocess.TimeoutExpired:
            sweep = {"error": "shard_sweep child exceeded 300 s"}
        except Exception as e:      # noqa: BLE001
            sweep = {"error": "%s: %s" % (type(e).__name__, e)}

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
        # HBM traffic of the same kernels from the PMC passes committed under profiles/
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per MI355X_MICROARCH.md).  NOT
        # measured by this run: `traffic_source` names the file; only used for the exact
        # configuration it was measured on, otherwise null
        traffic = mfma_busy = knn_traffic = traffic_source = None
        for fn in PMC_FILES:
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                c = pj["config"]
                kname = ("proj_mfma_wide" if w_n > 0 else
                         "proj_mfma_fast" if args.mode == "mfma" else "proj_exact_lds")
                if (c["N"], c["d"]) == (N, d) and world == 1 and \
                        int(round(pj[kname]["cols"])) == int(round(cols)):
                    traffic = pj[kname]["hbm_bytes_per_launch"]
                    mfma_busy = pj[kname].get("mfma_util_pct")
                    if (c.get("T"), c.get("nq"), c.get("k")) == (T, nq, k):
                        knn_traffic = pj.get("knn_fused", {}).get("hbm_bytes_per_launch")
                    traffic_source = "profiles/%s (%s)" % (fn, pj.get("build", "committed PMC passes"))
                    break
            except Exception:
                continue
        if mfma_frac > hbm_frac:
            roof = {"bound": "mfma", "achieved": mfma_achieved, "peak": MFMA_F64_PEAK_TF,
                    "unit": "TFLOP/s", "frac": mfma_frac}
        else:
            roof = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm_frac}
        roof.update({
            "kernel": kernel, "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": avg_ms,
            "launches": w_n if w_n > 0 else p_n,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_flops_per_launch": flops_per_launch,
            "hbm": {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_frac},
            "mfma_f64": {"achieved": mfma_achieved, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                         "frac": mfma_frac,
                         "peak_source": "AMD MI355X data sheet, FP64 matrix; = 2048 flop / 64 clk "
                                        "/ SIMD x 1024 SIMDs x 2.4 GHz",
                         "measured_stream_peak": MFMA_F64_STREAM_TF,
                         "frac_of_measured_stream": mfma_achieved / MFMA_F64_STREAM_TF
                         if args.mode == "mfma" else 0.0,
                         "measured_stream_source": "profiles/r03_mfma_f64_peak.txt: a naive loop of "
                                                   "independent v_mfma_f64_16x16x4_f64 on every SIMD "
                                                   "sustains 44-47 TFLOP/s at a measured 2.29 GHz (104 "
                                                   "elapsed cycles per MFMA; this kernel: 88 at the "
                                                   "2.11 GHz it runs at) - reported, NOT a roof",
                         "shader_clock_ghz_under_this_kernel": 2.11,
                         "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs / duration, profiles/r03_pmc.csv",
                         # rocprofv3 --pmc MfmaUtil of the same launches (profiles/), not live
                         "mfma_busy_pmc_pct": mfma_busy},
            # the reference formulation projects one tree level (32 hyperplanes) per read of X:
            # 8(d)'s 1.28 GB per level.  All projection launches of a build against that figure
            # are a SPEED-UP over the per-level formulation, not a roofline fraction:
            "survey_8d_per_level": {
                "bytes_per_forest": maxd * (N * d * 8 + d * Tl * 8 + N * Tl * 8),
                "projection_ms_per_forest": p_ms / args.steps,
                "speedup_vs_per_level_at_hbm_peak":
                    (maxd * (N * d * 8 + d * Tl * 8 + N * Tl * 8) / (HBM_PEAK_GBS * 1e9)) /
                    (p_ms / args.steps * 1e-3) if p_ms > 0 else 0.0},
        })
        # second roofline object: the query kernel against the bytes it really gathers
        cand_q = cand_total.value / float(max(nq, 1))
        tier = knn_tier.value
        pre32 = tier > 0
        sb = {3: 1, 2: 2}.get(tier, 4)                  # bytes per element of the ranking shadow
        kp = (min(k + max(48, k), 223) if tier == 3 else       # knn.hip: kp8 / kp16 / prefilter_keep
              k + max(8, k // 2) if tier == 2 else k + max(6, k // 2))
        # per BATCH (the re-run of the few uncertified queries is a second launch inside a batch)
        topk_ms = prof["knn_topk"][0] / max(args.steps, 1)
        knn_bytes = nq * (cand_q * d * sb + kp * d * 8) if pre32 else nq * cand_q * d * 8
        if tier != 3:
            knn_traffic = None                          # the committed PMC passes (r03, r04) are the int8 tier's
        knn_ach = knn_bytes / (topk_ms * 1e-3) / 1e9 if topk_ms > 0 else 0.0
        roof_knn = {"bound": "hbm", "achieved": knn_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": knn_ach / HBM_PEAK_GBS, "traffic": knn_traffic,
                    "traffic_source": traffic_source if knn_traffic else None,
                    "kernel": "knn_fused (%s-shadow ranking + exact f64 distances of the best k')"
                    % {3: "int8", 2: "half"}.get(tier, "f32") if pre32 else "knn_fused (all-f64 distances)",
                    "avg_launch_ms": topk_ms,
                    "algorithmic_bytes_per_launch": knn_bytes,
                    "bytes_formula": "nq x (candidates x d x %d B shadow rows + k' x d x 8 B exact "
                                     "rows), k' = %d" % (sb, kp) if pre32 else
                                     "nq x candidates x d x 8 B (SURVEY 8d)",
                    "candidates_per_query": cand_q}
        out = {
            "metric": "forest-build vectors/s (%d x %d dense f64, %d trees; projection mode %s = "
                      "RPT_PROJ_%s, the API default RPT_PROJ_AUTO on f64 data is the exact-order "
                      "kernel timed in `other_projection_mode`); kNN queries/s and recall@%d in "
                      "`knn` / `recall_at_10`" % (N, d, T, args.mode, args.mode.upper(), k),
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
                       "projection_mode": args.mode,
                       "generator": "SplitMix64 streams of SURVEY 8(d) (data 1234, queries 4321, "
                                    "forest 1235137), drawn on the device"},
            "launch": launch,
            "rccl_ranks": comm.nranks,
            "roofline": roof,
            "roofline_knn": roof_knn,
            "cpu_baseline": cpu,
            "h2d_inclusive": h2d,
            "knn": {"value": nq * args.steps / t_knn, "unit": "queries/s",
                    "ms_per_batch": t_knn / args.steps * 1e3, "semantics": "duplicates kept "
                    "(RPTree.hs:174-176)", "candidates_per_query": cand_q,
                    "topk_kernel_ms": topk_ms,
                    "plan_ms": prof["knn_plan"][0] / max(prof["knn_plan"][1], 1),
                    "prefilter_uncertified_queries": uncertified.value,
                    "ranking_tier": tier,
                    "method": "all-f64 distances" if not pre32 else
                    "candidates ranked on %s of X, exact left-fold f64 distances for the best %d, "
                    "cut certified per query (exact fallback); results identical to the all-f64 "
                    "kernel" % ({3: "an int8 shadow (one scale, integer ranking values)",
                                 2: "an IEEE-half shadow"}.get(tier, "an f32 shadow"), kp),
                    "exchange": None if world == 1 else
                    "one ncclAllGather of %d B records per rank on the ctx streams "
                    "(rpt_knn_sharded_dev), merge on every device" % sharded.record_layout(nq, k)[0],
                    "exchange_one_rank": exch},
            "recall_at_10": {"forest_knn_dedup_vs_brute_force": recall_knn,
                             "reference_recallWith_mean_per_tree": recall_ref,
                             "queries": nq_eval,
                             "note": "C2 as the survey fixed it (32 trees, one leaf of ~122 points "
                                     "per tree, d = 128 mixture) has a low absolute recall; the "
                                     "+-1 % target is met by identity of the returned ids with "
                                     "the reference restatement (cpu_baseline)"},
            "build_breakdown_ms": {"projection_total": p_ms / args.steps,
                                   "split_total": prof["split"][0] / args.steps},
            "other_projection_mode": {
                "mode": alt_name, "value": N * args.steps / t_alt, "unit": "vectors/s",
                "ms_per_step": t_alt / args.steps * 1e3,
                "projection_avg_launch_ms": prof_alt["project"][0] / max(prof_alt["project"][1], 1),
                "leaf_assignment_flip_rate_vs_timed_mode": leaf_flip_rate,
                "note": "exact = reference summation order, bit-identical to the oracle; "
                        "flips are points whose projection is within rounding of a median"},
            "forest_stats": f_loc.stats(),
            "other_configs": other,
            "shard_sweep": sweep,
            "roofline_per_level": per_level,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    forest.close()
    if launcher:
        dist.barrier()
    comm.close()
    if launcher:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
