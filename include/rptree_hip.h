/*
 * rptree_hip.h — C ABI of the MI355X (gfx950) implementation of rp-tree's random-projection
 * hot path.  This is the drop-in boundary: the entry points are what a Haskell
 * `foreign import ccall` layer under the unchanged Data.RPTree API binds.
 *
 * The reference (ocramz/rp-tree v0.7.1) is closed pure Haskell with no FFI or plugin
 * interface; each entry point therefore names the reference FUNCTION it
 * replaces (file:line relative to the reference checkout).
 *
 * Conventions
 *   - Plain C: opaque handles, raw pointers, sizes.  No torch / C++ types.
 *   - Every function returns an int32 status: 0 = ok, negative = error (RPT_E_*).
 *     rpt_last_error() returns a message for the calling thread.  Nothing throws or aborts
 *     across the ABI: every entry point catches C++ exceptions of its host-side planners
 *     (std::bad_alloc -> RPT_E_NOMEM, anything else -> RPT_E_INTERNAL).  There is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with RPT_E_HIP.
 *   - Ownership: the caller owns every host buffer it passes or receives.  The library owns
 *     device memory behind the opaque handles; release it with the matching *_free.
 *   - Pointers named *_host are host memory, *_dev are device (HBM) addresses of the ctx's
 *     device, e.g. torch tensor .data_ptr() values.
 *   - Threading: re-entrant per rpt_ctx (one ctx = one device + one stream); a ctx is not
 *     thread-safe.  All work of a ctx is enqueued on its stream; entry points that return
 *     host data synchronise that stream, *_dev entry points do not (call rpt_ctx_sync).
 *
 * Flat forest layout (identical to oracle/rptree_oracle.h):
 *   perm[T][N]  int32   point ids; per tree the concatenation of the leaf buckets in
 *                       left-to-right order, each bucket in the reference's order
 *                       (Internal.hs:495,504-505: children inherit the stably sorted order).
 *   thr, mglo, mghi  double [T][2^L-1], heap order (root 0, children 2h+1, 2h+2); NaN where
 *                       the slot is not a Bin.  = _rpThreshold / _rpMargin of `RPT`
 *                       (Internal.hs:139-145), values per Internal.hs:496-501.
 *   The topology (Bin vs Tip, every node's offset and size) is a pure function of
 *   (N, minLeaf, maxDepth): Internal.hs:289 and :495,503; rpt_topology() enumerates it.
 */
#ifndef RPTREE_HIP_H
#define RPTREE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPT_ABI_VERSION 1

/* status codes */
#define RPT_OK 0
#define RPT_E_ARG (-1)      /* invalid argument */
#define RPT_E_HIP (-2)      /* HIP runtime / device error (incl. no device) */
#define RPT_E_NOMEM (-3)    /* host or device allocation failed */
#define RPT_E_UNSUPPORTED (-4)
#define RPT_E_INTERNAL (-5)

/* element types of datasets / queries */
#define RPT_F64 0           /* the reference's only type (Double) */
#define RPT_F32 1           /* build extension */
#define RPT_BF16 2          /* build extension */

/* projection modes (rpt_project, rpt_forest_build flags) */
#define RPT_PROJ_AUTO 0     /* f64 data: EXACT; f32/bf16 data: MFMA */
#define RPT_PROJ_EXACT 1    /* f64 VALU, the reference's summation order and no FMA
                               (Internal.hs:382): bit-identical to innerSD/innerSS */
#define RPT_PROJ_MFMA 2     /* MFMA tiles (f64/f32/bf16 inputs), k-ordered fma chain:
                               within 1e-5*|x||r| of the reference value.  CSR rows: dense-ified
                               as two bf16 terms on the bf16 matrix pipe (from 65 536 rows on,
                               d % 8 == 0), else the exact-order kernel with ONE fused
                               multiply-add per term instead of the reference's two roundings
                               (same tolerance) */

/* knn flags */
#define RPT_KNN_KEEP_DUPLICATES 0 /* the reference: RPTree.hs:174-176 never de-duplicates */
#define RPT_KNN_DEDUP 1           /* extension: each point id at most once */
#define RPT_KNN_DEDUP_DISTANCE 2  /* knnPQ (RPTree.hs:181-194): `nub` keeps one entry per DISTANCE */
/* Voting (the MRPT vote threshold; the reference carries it as the commented-out `counts` /
 * `keepCounts` sketch, RPTree.hs:464-478): only the points found in at least v of the trees'
 * candidate lists get a distance; they are taken in ascending id order (the order of
 * M.foldrWithKey in keepCounts), each once, and the k best by (distance, id) are returned.
 * Dense data, k <= 64; or-ed into the knn flags: RPT_KNN_VOTE(v), v in [1, 65535]. */
#define RPT_KNN_VOTE(v) ((int32_t)(v) << 8)
/* SVector (CSR) data: distances by the reference's own metricSSL2 (Internal.hs:389-393 over diffSS /
 * binSS :435-450) — the merge of the two index lists STOPS when either vector is exhausted, so the
 * tail of the longer one is silently dropped — instead of the true Euclidean
 * distance, for results identical to `knn metricL2` of the reference on SVector data.  Bit-exact
 * (left fold in merge order); one thread per candidate, general query path.  Ignored for dense
 * data (metricDDL2 is what they get anyway). */
#define RPT_KNN_METRIC_REFERENCE (1 << 24)

typedef struct rpt_ctx rpt_ctx;
typedef struct rpt_dataset rpt_dataset;
typedef struct rpt_forest rpt_forest;

/* ---- library / context ---- */
int32_t rpt_abi_version(void);
const char* rpt_last_error(void);
int32_t rpt_device_count(int32_t* count);
int32_t rpt_ctx_create(int32_t device, rpt_ctx** out);
int32_t rpt_ctx_destroy(rpt_ctx* ctx);
int32_t rpt_ctx_sync(rpt_ctx* ctx);
/* device buffers released by the library are cached for reuse (multi-GB hipMalloc/hipFree per
 * build is slow); rpt_ctx_trim returns the cache to the driver (rpt_ctx_destroy does too). */
int32_t rpt_ctx_trim(rpt_ctx* ctx);
/* the hipStream_t all work of this ctx is enqueued on (for HIP-event timing by the caller) */
int32_t rpt_ctx_stream(rpt_ctx* ctx, void** hip_stream);

/* Algorithm switches of a context.  They exist for the parity tests of the fallback paths and
 * for A/B timing; every default is the tuned path and results never depend on them (beyond the
 * documented tolerances of the MFMA projections).  rpt_ctx_create seeds them ONCE from the
 * environment (RPT_<NAME>, upper case); no entry point reads the environment afterwards.
 *   no_stream, stream_maxnodes, stream_minper, no_wmid, no_midselect, stream_big_node, no_wsub,
 *   no_wsort, no_wpack, no_csub, no_codes, no_pcodes
 *       median split: which regime handles which level
 *   proj_narrow, proj_bf16_f32     projection: 32 hyperplanes per pass only / bf16 rows on the f32 pipe
 *   proj_bf16_terms (3), proj_bf16_codes, proj_csr_nodense
 *       bf16 rows meet every hyperplane as TWO bf16 terms (|error| <= 2^-17 |x||r| by construction,
 *       inside RPT_PROJ_MFMA's 1e-5); 3 keeps a third term (f32-level agreement, a third more
 *       matrix-pipe work) / codes from the bf16 kernel's epilogue / CSR rows stay on the segmented kernel
 *   knn_wave (-1 auto, 0, 1), knn_kp, knn_kp16, knn_kp8, knn_no_pre32, knn_no_pre16, knn_no_pre8,
 *   knn_csr_pre32, knn_general
 *       query kernels; knn_kp8 > 0 also opts bf16 datasets into the int8 ranking tier
 *   comm_force_exchange            sharded kNN on a ONE-rank communicator still runs record ->
 *                                  ncclAllGather -> merge (set on the communicator's first ctx)
 *   comm_inject_failure            test hook: the device's shard reports a failure (see "Failures"
 *                                  under the multi-GPU entry points)
 *   debug_host, debug_stamps       stderr diagnostics
 * Unknown names: RPT_E_ARG. */
int32_t rpt_ctx_set_option(rpt_ctx* ctx, const char* name, int64_t value);
int32_t rpt_ctx_get_option(rpt_ctx* ctx, const char* name, int64_t* value);

/* ---- kernel timing (bench.py roofline): HIP events recorded on the ctx stream around every
 * launch of a kernel class while enabled.  which: 0 = projection batch kernels (one launch =
 * one pass over the whole point set for up to 96 hyperplanes), 1 = split work,
 * 2 = query plan (query projections + traversal), 3 = distance/top-k kernel, 4 = the wide
 * (> 32 hyperplanes per pass) MFMA projection launches alone (they are also part of class 0).
 * rpt_prof_get synchronises the stream and returns the accumulated ms and launch count. */
int32_t rpt_prof_enable(rpt_ctx* ctx, int32_t on);
int32_t rpt_prof_reset(rpt_ctx* ctx);
int32_t rpt_prof_get(rpt_ctx* ctx, int32_t which, double* total_ms, int64_t* launches);

/* ---- datasets: Embed / DVector / SVector carriers (Internal.hs:56-59,92-93,122) ----
 * dense: row-major X[n][d] (`V.Vector (Embed DVector Double x)` packed once at the boundary).
 * csr:   SVector rows: rowptr[n+1] (int64), col[nnz] (int32, ascending per row, < d), val.
 * *_host variants copy to HBM; *_dev variants borrow device memory that must outlive the
 * handle.  Query batches use the same handle type.
 * Stream order of borrowed memory: the ctx stream is a non-blocking stream of its own, nothing
 * orders it against the stream that PRODUCED the borrowed arrays.  The producer must have
 * finished (synchronise its stream, or make it wait on an event of it via rpt_ctx_stream) before
 * the first rpt_* call that reads the memory; the same holds for output buffers of the *_dev
 * query entry points that another stream initialises.  (rptree_amd.Dataset.from_torch and
 * ShardedForest synchronise torch's current stream for exactly this reason.) */
int32_t rpt_dataset_dense_host(rpt_ctx* ctx, const void* X_host, int64_t n, int32_t d,
                               int32_t dtype, rpt_dataset** out);
int32_t rpt_dataset_dense_dev(rpt_ctx* ctx, const void* X_dev, int64_t n, int32_t d,
                              int32_t dtype, rpt_dataset** out);
int32_t rpt_dataset_csr_host(rpt_ctx* ctx, const int64_t* rowptr_host, const int32_t* col_host,
                             const void* val_host, int64_t n, int32_t d, int32_t dtype,
                             rpt_dataset** out);
/* borrow CSR arrays that already live in HBM (they must outlive the handle; NOT validated:
 * rowptr non-decreasing from 0 to nnz, every column index in [0, d), as the host variant checks) */
int32_t rpt_dataset_csr_dev(rpt_ctx* ctx, const int64_t* rowptr_dev, const int32_t* col_dev,
                            const void* val_dev, int64_t n, int32_t d, int32_t dtype, int64_t nnz,
                            rpt_dataset** out);
int32_t rpt_dataset_free(rpt_dataset* ds);
int32_t rpt_dataset_info(const rpt_dataset* ds, int64_t* n, int32_t* d, int32_t* dtype,
                         int32_t* is_csr, int64_t* nnz);

/* ---- topology: Internal.hs:289 (leaf test), :495,503 (cut at n div 2) ----
 * Enumerates the nodes in DFS (pre-)order.  Each record is 5 int64:
 * {level, heap, offset, size, is_leaf}.  Pass out=NULL to get the count only. */
int32_t rpt_topology(int64_t n, int32_t max_depth, int32_t min_leaf, int64_t* out,
                     int64_t cap_records, int64_t* n_records);

/* ---- projection batch: replaces the N `inner` calls of Internal.hs:504 ----
 * P[c][i] = R[c] `inner` x_i  for C hyperplanes R_host[C][d] (dense-ified SVectors; zeros are
 * skipped exactly like the sparse representation skips them).  Output in the compute type:
 * double for F64 data, float for F32/BF16 data.  P_host / P_dev is [C][n]. */
int32_t rpt_project_host(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t C,
                         int32_t mode, void* P_host);
int32_t rpt_project_dev(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t C,
                        int32_t mode, void* P_dev);

/* ---- forest build: replaces createMulti/create/insert/partitionAtMedian/sortByVG
 * (Internal.hs:217-297,486-512) under forestBatch / treeBatch (Batch.hs:29-63) ----
 * R_host[T][L][d]: the hyperplanes sampled by the HOST (Batch.hs:59-61), dense-ified.
 * L = maxDepth.  flags: RPT_PROJ_* mode. */
int32_t rpt_forest_build(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                         int32_t L, int32_t min_leaf, int32_t flags, rpt_forest** out);
int32_t rpt_forest_free(rpt_forest* f);
int32_t rpt_forest_info(const rpt_forest* f, int64_t* n, int32_t* d, int32_t* T, int32_t* L,
                        int32_t* min_leaf);
/* copy-out accessors so the host can rebuild ordinary `RPT` values (Internal.hs:139-149) */
int32_t rpt_forest_get_perm(rpt_forest* f, int32_t* perm_host /*[T][N]*/);
int32_t rpt_forest_get_nodes(rpt_forest* f, double* thr_host, double* mglo_host,
                             double* mghi_host /* each [T][2^L-1] */);
/* projections computed during the build, [T][L][N] in the compute type (parity tests) */
int32_t rpt_forest_get_proj(rpt_forest* f, void* proj_host);
/* ---- streaming build: `forest` / `tree` (Conduit.hs:58-121) = chunkedAccum (Conduit.hs:169-176)
 * folding insertMulti / insert (Internal.hs:245-297) over chunks of `chunk` points (the last one may
 * be shorter, C.chunksOf).  The reference's semantics, including its quirks: a chunk part that
 * reaches a Bin is split at ITS OWN median and the thresholds are averaged ((thr0 + thr) / 2,
 * Internal.hs:281), margins fold by (max, min) (:280, :86-87); an EMPTY part reaching a Bin replaces
 * the subtree by an empty Tip (:277) — the points stored below it are lost (rpt_forest_get_topology
 * reports how many).  With chunk >= n the result is the batch forest.  Dense rows only.
 * The shape of a streamed tree depends on (n, chunk, minLeaf, maxDepth): such a forest carries an
 * EXPLICIT topology — heap slots 0 .. 2^(maxDepth+1)-2, kind 0 absent / 1 Bin / 2 Tip, the points of
 * Tip h = perm[t][leaf_off[h] .. leaf_off[h] + leaf_len[h]) — the same for every tree.
 * rpt_forest_get_perm / _get_nodes return [T][n] ids (the first `held` of a row are valid) and
 * [T][slots] node arrays; rpt_candidates / rpt_knn_* work on the handle (general query path). */
int32_t rpt_forest_stream_build(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                                int32_t L, int32_t min_leaf, int64_t chunk, int32_t flags,
                                rpt_forest** out);
/* any output pointer may be NULL; kind / leaf_off / leaf_len hold *slots entries */
int32_t rpt_forest_get_topology(rpt_forest* f, int64_t* slots, int8_t* kind_host,
                                int64_t* leaf_off_host, int64_t* leaf_len_host, int64_t* held,
                                int64_t* dropped);
/* import a forest built elsewhere (e.g. deserialiseRPForest, Internal.hs:191-196) */
int32_t rpt_forest_import(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                          int32_t L, int32_t min_leaf, const int32_t* perm_host,
                          const double* thr_host, const double* mglo_host,
                          const double* mghi_host, rpt_forest** out);
/* projection mode the forest's thresholds were computed with; queries project with the same
 * kernels.  A forest imported with rpt_forest_import starts as RPT_PROJ_AUTO: restore the
 * builder's mode with rpt_forest_set_mode (the flat on-disk format stores it). */
int32_t rpt_forest_get_mode(const rpt_forest* f, int32_t* mode);
int32_t rpt_forest_set_mode(rpt_forest* f, int32_t mode);
/* number of nodes whose cut went through the exact tie-resolution path (statistics) */
int32_t rpt_forest_stats(rpt_forest* f, int64_t* tie_nodes, int64_t* big_mid_nodes);

/* ---- split of one level on caller-supplied projections: partitionAtMedian
 * (Internal.hs:486-505) for every node of a segment list, for integer parity on IDENTICAL
 * projection inputs.  key_host[n]: projections indexed by point id.  seg_off/seg_len[S]:
 * disjoint segments of perm_io_host[n] (ids).  On return every segment is stably sorted by
 * key (ties: previous position), thr_mg_host[S][3] = {thr, mglo, mghi}. */
int32_t rpt_split_segments(rpt_ctx* ctx, const double* key_host, int64_t n,
                           int32_t* perm_io_host, const int64_t* seg_off_host,
                           const int64_t* seg_len_host, int32_t S, double* thr_mg_host);

/* ---- queries ----
 * candidates (RPTree.hs:289-314): for every (query, tree) the leaf buckets reached, in
 * left-to-right order.  Output is CSR-like: off_host[nq*T + 1] (int64) into ids_host.
 * Two-call protocol: ids_host = NULL -> only *total is written. */
int32_t rpt_candidates(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* queries,
                       int64_t* off_host, int32_t* ids_host, int64_t cap, int64_t* total);

/* knnH (RPTree.hs:199-217 over candidatesH :318-342) with distf = metricL2: per query the
 * buckets of the leaves with the smallest margin priority, taken while the running count stays
 * <= k (always at least one), the bucket taken last FIRST, every point with its distance — as in
 * the reference the result is neither sorted by distance nor cut to k.  Equal priorities keep
 * (tree, DFS) order (the reference's order among ties depends on its heap's shape).
 * Output is CSR-like: off_host[nq + 1] into ids_host / dist_host.  Two-call protocol:
 * ids_host = NULL -> only off_host and *total are written. */
int32_t rpt_knnh_host(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                      const rpt_dataset* queries, int32_t k, int64_t* off_host, int32_t* ids_host,
                      double* dist_host, int64_t cap, int64_t* total);

/* Dense f64 data: the distances returned are metricDDL2's left fold of (u - v)^2 (Internal.hs:
 * 403-406) with every square correctly rounded, and the results are selected and ordered on those
 * values: candidates are RANKED on a lane-parallel sum (the same value to an ulp), the best
 * k + 8 of them (or the certified k + max(6, k/2) of the f32 prefilter) are evaluated again as the
 * fold and the k best of those by (distance, candidate position) are returned — more than 8
 * DIFFERENT rows within a few ulp of the k-th distance would be needed to change the membership
 * (copies of one row keep their order under both sums).
 * `(** 2)` is libm's pow in a GHC build: where pow(t, 2) is correctly rounded these are its bits;
 * glibc >= 2.28 returns a neighbouring double for about 9 arguments in 10 000, which moves the last
 * bit of about one distance in a thousand (tests/test_oracle_kat.py measures it on the test box).
 * RPT_KNN_DEDUP_DISTANCE collapses entries whose distances are equal under both sums; two rows one
 * ulp apart under one of them and equal under the other may or may not collapse. */
/* knn (RPTree.hs:168-176) with distf = metricL2 (Internal.hs:318, metricDDL2 :403-406 /
 * true Euclidean distance for CSR data, evaluated as |q|^2 + sum over the row's nonzeros of
 * ((x_j - q_j)^2 - q_j^2): absolute error about 1e-8 |q|): per query the k best (distance, id), stable in
 * candidate order (tree ascending, then leaf order).  ids/dist are [nq][k]; count[nq] is the
 * number of valid entries (< k when fewer candidates).  Unused slots: id -1, dist +inf. */
/* Memory note: the first rpt_knn_* call with duplicates kept (flags 0) and k <= 42 on a dense
 * f64 dataset builds an int8 copy of it on the device (+12.5 % of the dataset's size, freed with the
 * dataset; one scale for the whole dataset, rows of a multiple of 16 elements).  Only when that copy
 * cannot rank the call (other row lengths, k >= 40 on small tree shards, no memory, or a forest that
 * has dropped the tier) an f32 copy (+50 %) and an IEEE-half copy (+25 %, elements within the half
 * range) are built as well; f32 datasets likewise get the int8 copy first, the half copy when
 * needed: candidates are ranked on the int8 copy (k + max(48, k) kept; the ranking
 * value is an exact integer, the cut is certified through the triangle inequality with the
 * quantisation errors of the query and of the worst row), the half copy (k + max(8, k/2) kept) or
 * the f32 copy (k + max(6, k/2) kept), exact f64 distances are computed for the kept ones, and a
 * per-query error bound certifies the cut (a query that fails it is tried once more with three
 * times the kept entries, then takes the all-f64 path).
 * Results are identical either way.  A dataset borrowed with rpt_dataset_dense_dev must not be
 * modified while the library holds it. */
int32_t rpt_knn_host(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                     const rpt_dataset* queries, int32_t k, int32_t flags, int32_t* ids_host,
                     double* dist_host, int32_t* count_host);
int32_t rpt_knn_dev(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                    const rpt_dataset* queries, int32_t k, int32_t flags, int32_t* ids_dev,
                    double* dist_dev, int32_t* count_dev);
/* statistics of the last rpt_knn_* call: total candidates visited (sum over queries) */
int32_t rpt_knn_last_candidates(rpt_ctx* ctx, int64_t* total);
/* ... and how many of its queries the f32 prefilter could not certify (equal distances at its
 * cut) and were answered again with all-f64 distances; 0 when the prefilter was not used */
int32_t rpt_knn_last_uncertified(rpt_ctx* ctx, int64_t* total);
/* ... and how many took the in-kernel second attempt (a cut the first, narrower selection could not
 * certify, retried with three times the kept entries by the same workgroup).  Telemetry: a forest whose
 * batches retry often starts its later batches wider. */
int32_t rpt_knn_last_retries(rpt_ctx* ctx, int64_t* total);
/* ... and the shadow its candidates were ranked on: 0 = none (all-f64 distances), 1 = the f32 copy
 * of the dataset, 2 = its IEEE-half copy (round 3: a quarter of the f64 bytes; keeps k + max(8, k / 2)
 * entries for the exact pass, same certificate with the half rounding in the error bound), 3 = its
 * int8 copy (an eighth of the f64 bytes).  A forest on which more than a quarter of a batch cannot
 * be certified drops one tier for its later batches (int8 -> half -> f32 -> none). */
int32_t rpt_knn_last_tier(rpt_ctx* ctx, int32_t* tier);
/* diagnostics of the last rpt_forest_build on this context: split nodes of 1025 .. 8192 points that
 * the packed-code kernel handed back to the general kernels (heavy ties: more than 1024 points of a
 * node share the 16-bit code of its median; the result does not depend on it), and how many of those
 * because a code histogram contradicted the node sizes (a defect if ever non-zero; tested to be 0) */
int32_t rpt_build_last_handed_back(rpt_ctx* ctx, int64_t* nodes, int64_t* inconsistent);

/* multi-GPU merge: G per-shard results (shard g holds trees [g*T/G, (g+1)*T/G)), gathered
 * shard-major as ids_dev[G][nq][k] etc. (e.g. by an RCCL all-gather), merged into the
 * global top-k with the reference's stable order (shard ascending = tree ascending).
 * Any G >= 1 and k <= 1024: up to 4096 entries per query merge in one launch, larger merges
 * fold the shards in one at a time (same order). */
int32_t rpt_knn_merge_dev(rpt_ctx* ctx, const int32_t* ids_dev, const double* dist_dev,
                          const int32_t* count_dev, int32_t G, int64_t nq, int32_t k,
                          int32_t flags, int32_t* out_ids_dev, double* out_dist_dev,
                          int32_t* out_count_dev);
/* the same merge over G packed exchange records (ONE all-gather instead of three): a shard
 * writes its rpt_knn_dev results into one record — distances at off_dist, ids at off_ids,
 * counts at off_count, then ONE int32 status word (0 = the shard answered; anything else = it
 * failed and its lists are void), `bytes` in all (rpt_knn_record_layout; a multiple of 16) — the records
 * of all shards are gathered back to back (record_bytes apart, shard-major), and
 * rpt_knn_merge_records_dev merges them exactly like rpt_knn_merge_dev. */
int32_t rpt_knn_record_layout(int64_t nq, int32_t k, int64_t* bytes, int64_t* off_dist,
                              int64_t* off_ids, int64_t* off_count);
int32_t rpt_knn_merge_records_dev(rpt_ctx* ctx, const void* records_dev, int64_t record_bytes,
                                  int32_t G, int64_t nq, int32_t k, int32_t flags,
                                  int32_t* out_ids_dev, double* out_dist_dev,
                                  int32_t* out_count_dev);

/* ---- multi-GPU: tree shards per device, ONE RCCL all-gather per query batch ----
 * Reference contract: the trees of a forest are independent (createMulti maps create over the
 * IntMap, Internal.hs:234-240) and knn concatenates the per-tree candidates in key order before
 * one stable sort (RPTree.hs:174-176).  Rank r of G holds the contiguous tree block
 * [r*T/G, (r+1)*T/G) and a replica of the point set; the build needs no communication; a query
 * batch is answered per shard into one exchange record (rpt_knn_record_layout), the records are
 * all-gathered over xGMI (ncclAllGather on the ctx streams, librccl) and merged on every device
 * in (distance, shard, rank) order = the reference's order.  The result is identical to
 * rpt_knn_* on the whole forest on one device.
 *
 * A communicator is formed either by ONE process for n devices (rpt_comm_init: ncclCommInitAll,
 * one rpt_ctx and one host worker thread per device; the ctxs are owned by the communicator)
 * or by one process PER device (rpt_comm_init_rank with the caller's ctx; rank 0 makes the id
 * with rpt_comm_unique_id and the host distributes its RPT_COMM_UID_BYTES bytes, e.g. through
 * the launcher's store).  Per-device arguments (ds, data, queries, outputs) are arrays of
 * `nlocal` entries, entry g living on the device of rpt_comm_ctx(comm, g): n entries after
 * rpt_comm_init(n), one after rpt_comm_init_rank.
 *
 * Failures.  With one process per device a rank's error code is invisible to its peers, so a rank
 * whose query kernels fail still joins the all-gather — its record's status word set — and returns
 * its own error; every rank scans the gathered status words after the merge: if one is set, all
 * counts of the answer are -1 and the next rpt_comm_sync (rpt_knn_sharded calls it) returns
 * RPT_E_INTERNAL naming the rank.  A rank that cannot join at all (no memory for its record, the
 * collective cannot be enqueued) aborts its communicator (ncclCommAbort): peers see a failed
 * collective instead of a hang, and every later call on the communicator returns RPT_E_INTERNAL. */
#define RPT_COMM_UID_BYTES 128
typedef struct rpt_comm rpt_comm;
typedef struct rpt_sharded_forest rpt_sharded_forest;
int32_t rpt_comm_init(int32_t n_gpus, rpt_comm** out);
int32_t rpt_comm_unique_id(void* uid_out /*[RPT_COMM_UID_BYTES]*/);
int32_t rpt_comm_init_rank(rpt_ctx* ctx, int32_t nranks, int32_t rank,
                           const void* uid /*[RPT_COMM_UID_BYTES]*/, rpt_comm** out);
int32_t rpt_comm_destroy(rpt_comm* comm);
int32_t rpt_comm_info(const rpt_comm* comm, int32_t* nranks, int32_t* nlocal, int32_t* first_rank);
int32_t rpt_comm_ctx(rpt_comm* comm, int32_t local_index, rpt_ctx** ctx);   /* borrowed */
int32_t rpt_comm_sync(rpt_comm* comm);                    /* rpt_ctx_sync of every local ctx */
/* createMulti (Internal.hs:234-240) sharded: R_host is the WHOLE forest's [T][L][d] block (every
 * rank passes the same); local device g builds the trees of rank first_rank + g.  T >= nranks. */
int32_t rpt_forest_build_sharded(rpt_comm* comm, const rpt_dataset* const* ds,
                                 const double* R_host, int32_t T, int32_t L, int32_t min_leaf,
                                 int32_t flags, rpt_sharded_forest** out);
int32_t rpt_sharded_forest_free(rpt_sharded_forest* sf);
/* the shard of local device g as an ordinary forest handle (borrowed) and its tree block */
int32_t rpt_sharded_forest_local(rpt_sharded_forest* sf, int32_t local_index, rpt_forest** f,
                                 int32_t* first_tree, int32_t* n_trees);
/* knn (RPTree.hs:168-176) over the sharded forest.  _dev: every local device receives the merged
 * answer in its own output buffers ([nq][k] ids / distances, [nq] counts), enqueued on the ctx
 * streams (rpt_comm_sync before reading).  rpt_knn_sharded copies device 0's answer to the host. */
int32_t rpt_knn_sharded_dev(rpt_comm* comm, rpt_sharded_forest* sf,
                            const rpt_dataset* const* data, const rpt_dataset* const* queries,
                            int32_t k, int32_t flags, int32_t* const* ids_dev,
                            double* const* dist_dev, int32_t* const* count_dev);
int32_t rpt_knn_sharded(rpt_comm* comm, rpt_sharded_forest* sf, const rpt_dataset* const* data,
                        const rpt_dataset* const* queries, int32_t k, int32_t flags,
                        int32_t* ids_host, double* dist_host, int32_t* count_host);

/* brute-force exact kNN on the device (evaluation of recall; ties by ascending id) */
int32_t rpt_brute_knn_host(rpt_ctx* ctx, const rpt_dataset* data, const rpt_dataset* queries,
                           int32_t k, int32_t* ids_host, double* dist_host);

#ifdef __cplusplus
}
#endif
#endif
