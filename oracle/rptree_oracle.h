/*
 * rptree_oracle.h — CPU ORACLE for the rp-tree random-projection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (rp-tree_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * What it is: a line-by-line C++ restatement of the Haskell reference ocramz/rp-tree
 * v0.7.1 for the functions on the hot path.  Each function cites the
 * reference file:line it follows (paths relative to the reference checkout).
 *
 * Pinning status:
 *   - vector algebra (inner / ^+^ / ^-^): pinned by the reference's own four known-answer
 *     tests, test/Data/RPTreeSpec.hs:28-45 (see tests/test_oracle_kat.py).
 *   - tree build / candidates / knn / recallWith: the reference holds NO golden vectors and
 *     its tests use an entropy seed (test/Data/RPTreeSpec.hs:48); the reference cannot be
 *     compiled here (no GHC).  These are restated from the source text and checked against
 *     the reference's structural invariants (treeSize == n, knn max dist < 1 on the
 *     two-disc data).  PARITY UNPINNED beyond the 4 KATs.
 *   - RNG (SplitMix64 + splitmix-distributions-0.9.0.0, stack.yaml:46) is third-party and
 *     absent from the reference tree: restated from the published algorithm, self-consistent
 *     only, PARITY UNPINNED.  In a real deployment the Haskell host generates the
 *     hyperplanes and the kernels just consume them.
 *
 * Flat layout shared with the HIP library (include/rptree_hip.h):
 *   perm[T][N]      int32  point ids, concatenation of the leaves in left-to-right DFS order
 *   thr/mglo/mghi   double [T][2^L - 1], heap order (root 0, children 2h+1 / 2h+2), NaN where
 *                   the heap slot is not a Bin node.
 *   Topology (which slots are Bin, every segment's offset and size) is a pure function of
 *   (N, minLeaf, maxDepth): Internal.hs:289 (leaf test) and :495,503 (cut at n div 2).
 */
#ifndef RPTREE_ORACLE_H
#define RPTREE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SplitMix64 + distributions (third-party restatement; parity unpinned) ---- */
typedef struct { uint64_t seed; uint64_t gamma; } rpo_gen;
void     rpo_gen_init(rpo_gen* g, uint64_t seed);           /* mkSMGen */
uint64_t rpo_next_word64(rpo_gen* g);
double   rpo_next_double(rpo_gen* g);                        /* stdUniform */
int      rpo_bernoulli(rpo_gen* g, double p);
double   rpo_normal(rpo_gen* g, double mu, double sig);
double   rpo_uniform_r(rpo_gen* g, double lo, double hi);

/* Gen.hs:148-153,178-195 `sparse p d rand` with rand = normal mu sig.
 * Writes at most d (idx,val) pairs; returns nnz. */
int64_t rpo_sparse_normal(rpo_gen* g, double p, int32_t d, double mu, double sig,
                          int32_t* idx, double* val);
/* same with rand = uniformR lo hi (MNIST-like sparse data, bench/time/Main.hs:124-125) */
int64_t rpo_sparse_uniform(rpo_gen* g, double p, int32_t d, double lo, double hi,
                           int32_t* idx, double* val);

/* Batch.hs:57-61: hyperplanes of a forest.  Draw order: tree outermost, level inner, one
 * generator threaded through.  Output dense-ified R[T][L][d] (zeros where the sparse vector
 * has no entry) and, optionally (may be NULL), per-vector nnz counts nnz[T][L]. */
void rpo_forest_hyperplanes(uint64_t seed, int32_t T, int32_t L, double pnz, int32_t d,
                            double* R, int32_t* nnz);

/* Gen.hs:132-137 normalDense2 via dataBatch (Batch.hs:66-75): n vectors, row-major X[n][d] */
void rpo_data_normal_dense2(uint64_t seed, int64_t n, int32_t d, double* X);
/* test/Data/RPTreeSpec.hs:112-120 circle2d2 (two unit discs in 2-D) */
void rpo_data_circle2d2(uint64_t seed, int64_t n, double* X);
/* Gen.hs:125-130 normalSparse2 via dataBatch -> CSR (rowptr[n+1], col, val); returns nnz.
 * cap = capacity of col/val. */
int64_t rpo_data_normal_sparse2(uint64_t seed, int64_t n, int32_t d, double pnz,
                                int64_t* rowptr, int32_t* col, double* val, int64_t cap);
/* Bernoulli support + U(lo,hi] values -> CSR */
int64_t rpo_data_sparse_uniform(uint64_t seed, int64_t n, int32_t d, double pnz,
                                int64_t* rowptr, int32_t* col, double* val, int64_t cap);

/* ---- vector algebra, Internal.hs:351-470 ---- */
double rpo_inner_ss(int64_t n1, const int32_t* i1, const double* v1,
                    int64_t n2, const int32_t* i2, const double* v2);     /* :351-366 */
double rpo_inner_sd(int64_t n1, const int32_t* i1, const double* v1,
                    int64_t n2, const double* x);                          /* :369-382 */
double rpo_inner_dd(int64_t n, const double* a, const double* b);          /* :384-385 */
double rpo_metric_dd(int64_t n, const double* u, const double* v);         /* :403-406 */
double rpo_metric_sd(int64_t n1, const int32_t* i1, const double* v1,
                     int64_t n2, const double* x);                         /* :396-400 */
double rpo_metric_ss(int64_t n1, const int32_t* i1, const double* v1,
                     int64_t n2, const int32_t* i2, const double* v2);     /* :389-393 */
/* binSDD (+)/(-) 0, Internal.hs:455-470; returns output length (<= n2) */
int64_t rpo_sum_sd(int64_t n1, const int32_t* i1, const double* v1,
                   int64_t n2, const double* x, double* out);
int64_t rpo_diff_sd(int64_t n1, const int32_t* i1, const double* v1,
                    int64_t n2, const double* x, double* out);

/* Conduit.hs:132-141 rpTreeCfg */
void rpo_tree_cfg(int32_t minLeaf, int64_t n, int32_t d,
                  int32_t* maxDepth, int64_t* chunk, double* pnz);

/* ---- partitionAtMedian on precomputed projections, Internal.hs:486-512 ----
 * p[n] projections in the node's current order.  order[n] receives the stable argsort.
 * thr_mg[3] = {thr, mglo, mghi}.  Returns nh (size of the left child), or -1 if n < 1. */
int64_t rpo_partition_at_median(int64_t n, const double* p, int32_t* order, double* thr_mg);

/* ---- forest build, Batch.hs:48-63 -> Internal.hs:217-297,486-512 ----
 * R is the dense-ified hyperplane block [T][L][d]; zeros are skipped exactly as the sparse
 * representation skips them (innerSD / innerSS iterate the hyperplane's nonzeros only).
 * first_tree / n_trees_built allow timing a subset of trees (cpu_baseline sample).
 * proj_out (may be NULL): [T][L][N] projection of every point on the vector of the level,
 * for the levels/nodes actually split (others left untouched) — used for value parity. */
void rpo_forest_build_dense(const double* X, int64_t N, int32_t d,
                            const double* R, int32_t T, int32_t L, int32_t minLeaf,
                            int32_t* perm, double* thr, double* mglo, double* mghi,
                            double* proj_out);
void rpo_forest_build_csr(const int64_t* rowptr, const int32_t* col, const double* val,
                          int64_t N, int32_t d,
                          const double* R, int32_t T, int32_t L, int32_t minLeaf,
                          int32_t* perm, double* thr, double* mglo, double* mghi,
                          double* proj_out);

/* ---- candidates, RPTree.hs:289-314 ----
 * One tree (index t of the flat arrays), one dense query q[d].  Writes the candidate ids
 * (leaf buckets, left-to-right) to out (capacity cap); returns the count (may exceed cap:
 * then only cap were written). */
int64_t rpo_candidates_dense(const double* q, int32_t d,
                             const double* R, int32_t T, int32_t L, int32_t minLeaf, int64_t N,
                             const int32_t* perm, const double* thr, const double* mglo,
                             const double* mghi, int32_t t, int32_t* out, int64_t cap);
int64_t rpo_candidates_sparse(int64_t qn, const int32_t* qi, const double* qv, int32_t d,
                              const double* R, int32_t T, int32_t L, int32_t minLeaf, int64_t N,
                              const int32_t* perm, const double* thr, const double* mglo,
                              const double* mghi, int32_t t, int32_t* out, int64_t cap);

/* ---- knn, RPTree.hs:168-176 with distf = metricL2 ----
 * dense data + dense query -> metricDDL2.  dedup = 0 is the reference (duplicates kept).
 * Returns number of results written (<= k). */
int32_t rpo_knn_dense(const double* X, int64_t N, int32_t d, const double* q,
                      const double* R, int32_t T, int32_t L, int32_t minLeaf,
                      const int32_t* perm, const double* thr, const double* mglo,
                      const double* mghi, int32_t k, int32_t dedup,
                      int32_t* out_ids, double* out_dist);
/* sparse data + sparse query -> metricSSL2 (incl. the binSS truncation quirk) when
 * true_l2 = 0, or the true Euclidean distance when true_l2 = 1. */
int32_t rpo_knn_csr(const int64_t* rowptr, const int32_t* col, const double* val,
                    int64_t N, int32_t d, int64_t qn, const int32_t* qi, const double* qv,
                    const double* R, int32_t T, int32_t L, int32_t minLeaf,
                    const int32_t* perm, const double* thr, const double* mglo,
                    const double* mghi, int32_t k, int32_t dedup, int32_t true_l2,
                    int32_t* out_ids, double* out_dist);

/* ---- knnPQ, RPTree.hs:181-194: rpo_knn_dense / rpo_knn_csr with dedup = 2 — `nub` collapses
 * entries of equal PRIORITY, i.e. equal distance, to one (the first in candidate order here; the
 * reference's pick among ties depends on the heap's shape). */

/* ---- candidatesH / knnH, RPTree.hs:199-217,318-342 (distf = metricL2) ----
 * candidates_h: the leaves one tree contributes, DFS order, with their margin priority.
 * knn_h: buckets of the lowest-priority leaves while the count stays <= k (at least one), the
 * later bucket first, every point with its distance; NOT sorted, NOT cut to k (as the
 * reference).  Returns the number of results (may exceed cap: then only cap were written).
 * Equal priorities keep (tree, DFS) order — the reference's order among ties is unspecified. */
int64_t rpo_candidates_h_dense(const double* q, int32_t d, const double* R, int32_t T, int32_t L,
                               int32_t minLeaf, int64_t N, const double* thr, const double* mglo,
                               const double* mghi, int32_t t, double* prio, int64_t* off,
                               int64_t* len, int64_t cap);
int64_t rpo_knn_h_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                        int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm,
                        const double* thr, const double* mglo, const double* mghi, int32_t k,
                        int32_t* out_ids, double* out_dist, int64_t cap);
int64_t rpo_knn_h_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                      int32_t d, int64_t qn, const int32_t* qi, const double* qv, const double* R,
                      int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm,
                      const double* thr, const double* mglo, const double* mghi, int32_t k,
                      int32_t true_l2, int32_t* out_ids, double* out_dist, int64_t cap);

/* ---- recallWith, RPTree.hs:259-282 (dense data, metricDDL2) ---- */
double rpo_recall_with_dense(const double* X, int64_t N, int32_t d, const double* q,
                             const double* R, int32_t T, int32_t L, int32_t minLeaf,
                             const int32_t* perm, const double* thr, const double* mglo,
                             const double* mghi, int32_t k);

/* brute-force exact kNN (dense, metricDDL2, ties by ascending id) — evaluation helper */
void rpo_brute_knn_dense(const double* X, int64_t N, int32_t d, const double* q, int32_t k,
                         int32_t* out_ids, double* out_dist);


/* ---- typed / threaded variants (same arithmetic, same results) ----
 * xdtype: 0 = double rows (the reference's type), 1 = float rows, converted to double on read
 * (exact): the oracle then IS the reference's arithmetic on the exactly-upcast data — what the
 * f32 / bf16 build extensions are compared with.  threads: trees (or queries) are independent;
 * threads = 1 is the reference (single-threaded), more threads are the all-core
 * courtesy baseline (8d-ii).  The result never depends on the thread count. */
void rpo_forest_build_dense_ex(const void* X, int32_t xdtype, int64_t N, int32_t d,
                               const double* R, int32_t T, int32_t L, int32_t minLeaf,
                               int32_t* perm, double* thr, double* mglo, double* mghi,
                               double* proj_out, int32_t threads);
void rpo_forest_build_csr_ex(const int64_t* rowptr, const int32_t* col, const double* val,
                             int64_t N, int32_t d, const double* R, int32_t T, int32_t L,
                             int32_t minLeaf, int32_t* perm, double* thr, double* mglo,
                             double* mghi, double* proj_out, int32_t threads);
/* knn for a batch Q[nq][d] of dense double queries; out_ids/out_dist [nq][k] (unused: -1/+inf),
 * out_count[nq].  vote_thr > 0 first reduces the candidates with keepCounts (extension). */
void rpo_knn_dense_batch(const void* X, int32_t xdtype, int64_t N, int32_t d, const double* Q,
                         int64_t nq, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                         const int32_t* perm, const double* thr, const double* mglo,
                         const double* mghi, int32_t k, int32_t dedup, int32_t vote_thr,
                         int32_t* out_ids, double* out_dist, int32_t* out_count,
                         int32_t threads);

/* ---- counts / keepCounts, RPTree.hs:464-478 (a commented-out sketch in the reference) ----
 * entries of the id multiset with count >= thr, ascending id (M.foldrWithKey order). */
int64_t rpo_keep_counts(const int32_t* ids, int64_t n, int32_t thr, int32_t* out_ids,
                        int32_t* out_counts);

/* ---- recallWith with Set-of-VALUES semantics, RPTree.hs:276-282: points with equal
 * coordinates (payload `()`) are ONE element of `aa` / `kk`.  rpo_recall_with_dense treats the
 * point id as the payload (all points distinct). */
double rpo_recall_with_dense_values(const double* X, int64_t N, int32_t d, const double* q,
                                    const double* R, int32_t T, int32_t L, int32_t minLeaf,
                                    const int32_t* perm, const double* thr, const double* mglo,
                                    const double* mghi, int32_t k);

/* ---- streaming build: Conduit.hs:147-176 (chunkedAccum / insertMultiC) over
 * Internal.hs:245-297 (insertMulti / insert incl. the Bin branch :272-283) ----
 * Points 0..N-1 arrive in chunks of `chunk`.  The topology is data dependent: per tree the
 * result is a heap array of S = 2^(L+1)-1 slots: kind[T][S] (0 absent, 1 Bin, 2 Tip),
 * thr/mglo/mghi[T][S] (NaN unless Bin), leaf_off/leaf_len[T][S] into leaf_ids[T][N].
 * held[T] = points actually stored; < N when an empty chunk half met a Bin (:277), which
 * replaces the whole subtree by an empty Tip (the data-loss quirk). */
void rpo_stream_forest_dense(const double* X, int64_t N, int32_t d, const double* R, int32_t T,
                             int32_t L, int32_t minLeaf, int64_t chunk, int8_t* kind, double* thr,
                             double* mglo, double* mghi, int64_t* leaf_off, int64_t* leaf_len,
                             int32_t* leaf_ids, int64_t* held);
void rpo_stream_forest_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                           int32_t d, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                           int64_t chunk, int8_t* kind, double* thr, double* mglo, double* mghi,
                           int64_t* leaf_off, int64_t* leaf_len, int32_t* leaf_ids, int64_t* held);


/* `** 2` is the host libm's pow in the reference; see sq() in the .cpp.  rpo_metric_dd_libm folds
 * through this box's pow; rpo_pow2_mismatches counts arguments where pow(t, 2.0) != t * t. */
double rpo_metric_dd_libm(int64_t n, const double* u, const double* v);
int64_t rpo_pow2_mismatches(uint64_t seed, int64_t n);

/* candidates (RPTree.hs:289-314) / knn (:168-176, metricL2) over the heap arrays of a streamed
 * forest: Tip = kind 2 (payload at leaf_off / leaf_len), Bin = kind 1 */
int64_t rpo_stream_candidates_dense(const double* q, int32_t d, const double* R, int32_t T,
                                    int32_t L, int64_t N, const int8_t* kind, const double* thr,
                                    const double* mglo, const double* mghi, const int64_t* leaf_off,
                                    const int64_t* leaf_len, const int32_t* leaf_ids, int32_t t,
                                    int32_t* out, int64_t cap);
int32_t rpo_stream_knn_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                             int32_t T, int32_t L, const int8_t* kind, const double* thr,
                             const double* mglo, const double* mghi, const int64_t* leaf_off,
                             const int64_t* leaf_len, const int32_t* leaf_ids, int32_t k,
                             int32_t dedup, int32_t* out_ids, double* out_dist);
/* knnH (RPTree.hs:199-217) over a streamed forest: whole buckets in increasing margin priority,
 * neither sorted nor cut to k; returns the number of results (may exceed cap: call again). */
int64_t rpo_stream_knn_h_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                               int32_t T, int32_t L, const int8_t* kind, const double* thr,
                               const double* mglo, const double* mghi, const int64_t* leaf_off,
                               const int64_t* leaf_len, const int32_t* leaf_ids, int32_t k,
                               int32_t* out_ids, double* out_dist, int64_t cap);


#ifdef __cplusplus
}
#endif
#endif
