// rptree_oracle.cpp — CPU ORACLE (test infrastructure, see rptree_oracle.h for the rules).
//
// Restates ocramz/rp-tree v0.7.1 for the random-projection hot path.  Citations are
// file:line in the reference checkout.  Compile with -O2 -ffp-contract=off, never
// -ffast-math: GHC emits separate IEEE double multiply and add (no FMA), and the reference's
// summation ORDER (right-nested for innerSD/innerSS, left fold for innerDD) is part of the
// contract that makes leaf assignments reproducible.
#include "rptree_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <thread>
#include <utility>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// SplitMix64 (third-party `splitmix`, restated from the published algorithm)
// ------------------------------------------------------------------------------------------
inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL;
  z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL;
  return z ^ (z >> 33);
}
inline uint64_t mix64v13(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
inline uint64_t mix_gamma(uint64_t z) {
  uint64_t g = mix64v13(z) | 1ULL;
  int n = __builtin_popcountll(g ^ (g >> 1));
  return n >= 24 ? g : g ^ 0xaaaaaaaaaaaaaaaaULL;
}

// Gen.hs:178-195 sparseVG: for i in [0,d): flag <- bernoulli p; if flag: x <- rand; emit (i,x)
template <class Rand>
int64_t sparse_vg(rpo_gen* g, double p, int32_t d, int32_t* idx, double* val, Rand rand) {
  int64_t m = 0;
  for (int32_t i = 0; i < d; ++i) {
    if (rpo_bernoulli(g, p)) {
      double x = rand();
      idx[m] = i;
      val[m] = x;
      ++m;
    }
  }
  return m;
}

// ------------------------------------------------------------------------------------------
// Topology: data independent.  Internal.hs:289 (leaf iff level >= maxDepth || n <= minLeaf),
// :495,503 (left child takes n div 2, right child the rest).
// `x ** 2` (Internal.hs:391,398,404; Gen.hs circle2d): GHC compiles (**) on Double to a call of
// the C library's pow.  Where pow(t, 2.0) is correctly rounded it IS t * t (the exact square,
// rounded once); glibc >= 2.28 is not correctly rounded there — on this image (2.35) 8.5e-4 of
// random arguments come out one ulp off, which moves the last bit of about one distance in a
// thousand (tests/test_oracle_kat.py measures both) — older glibc (IBM accurate pow), other
// platforms' libms and any FMA / non-FMA ifunc variant differ again.  "The reference's bits" of
// a squared term are therefore a property of the HOST's libm, not of rp-tree.  The oracle — and
// the device — take the correctly rounded square, explicitly (g++ -O2 had been folding
// std::pow(t, 2.0) into t * t all along); rpo_metric_dd_libm is the same fold through the real
// pow of this box, for the measurement.
inline double sq(double t) { return t * t; }

// ------------------------------------------------------------------------------------------
inline bool is_leaf(int32_t level, int64_t n, int32_t L, int32_t minLeaf) {
  return level >= L || n <= (int64_t)minLeaf;
}

struct SparseVec {  // one hyperplane (SVector Double): sorted (idx,val) pairs
  std::vector<int32_t> idx;
  std::vector<double> val;
};

std::vector<SparseVec> sparsify(const double* R, int32_t T, int32_t L, int32_t d) {
  std::vector<SparseVec> out((size_t)T * L);
  for (int64_t v = 0; v < (int64_t)T * L; ++v) {
    const double* r = R + v * d;
    for (int32_t i = 0; i < d; ++i)
      if (r[i] != 0.0) {
        out[v].idx.push_back(i);
        out[v].val.push_back(r[i]);
      }
  }
  return out;
}

// A data accessor abstracts DVector rows vs SVector (CSR) rows so that the tree code is
// written once, like the reference's `Inner SVector v` constraint (Internal.hs:316-341).
// E = double is the reference's type; E = float rows (f32 / bf16 datasets, build extensions)
// are converted to double on read, which is exact, so the arithmetic below is the reference's
// arithmetic on the upcast data.
template <class E>
struct DenseDataT {
  const E* X;
  int64_t N;
  int32_t d;
  double inner(const SparseVec& r, int64_t id) const {  // Inner SVector DVector, :332-333
    // innerSD, Internal.hs:369-382 (same loop as rpo_inner_sd)
    const E* x = X + id * d;
    const int64_t n1 = (int64_t)r.idx.size();
    const int64_t m = n1 < d ? n1 : d;
    double acc = 0.0;
    for (int64_t j = m - 1; j >= 0; --j) acc = r.val[j] * (double)x[r.idx[j]] + acc;
    return acc;
  }
  double metric(int64_t id, const double* q) const {  // metricDDL2, Internal.hs:403-406
    const E* x = X + id * d;
    double acc = 0.0;
    for (int32_t j = 0; j < d; ++j) acc = acc + sq((double)x[j] - q[j]);
    return std::sqrt(acc);
  }
};
using DenseData = DenseDataT<double>;
struct CsrData {
  const int64_t* rowptr;
  const int32_t* col;
  const double* val;
  int64_t N;
  int32_t d;
  double inner(const SparseVec& r, int64_t id) const {  // Inner SVector SVector, :322-323
    int64_t a = rowptr[id], b = rowptr[id + 1];
    return rpo_inner_ss((int64_t)r.idx.size(), r.idx.data(), r.val.data(), b - a, col + a,
                        val + a);
  }
};

struct FlatTree {
  int32_t* perm;  // [N]
  double* thr;    // [2^L-1]
  double* mglo;
  double* mghi;
};

// Internal.hs:258-297 `insert` on an empty Tip (= batch `create`, :223-225), recursion
// restated over id lists.  `ids` holds the node's points in the node's current order.
// `par` > 1 = host threads this node may use.  It changes nothing the reference computes: the
// n inner products of :504 are independent of each other (they are split over the threads, each
// one the scalar loop), and the two recursive calls of :296-297 write disjoint subtrees (the
// left one goes to a second thread).  The sort stays the single-threaded stable sort.
template <class Data>
void build_node(const Data& D, const std::vector<SparseVec>& rvs /*L vectors of this tree*/,
                int32_t L, int32_t minLeaf, int32_t level, int64_t heap, int64_t off,
                std::vector<int32_t>& ids, FlatTree& ft, double* proj_tree /*[L][N] or null*/,
                int32_t par = 1) {
  const int64_t n = (int64_t)ids.size();
  if (is_leaf(level, n, L, minLeaf)) {  // :289-290 Tip () xs'
    for (int64_t i = 0; i < n; ++i) ft.perm[off + i] = ids[i];
    return;
  }
  // partitionAtMedian r xs, Internal.hs:491-505
  const SparseVec& r = rvs[level];  // :270 rvs ! ixLev
  std::vector<std::pair<double, int32_t>> projs((size_t)n);
  auto project = [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {  // :504 map (\xe -> (xe, r `inner` eEmbed xe))
      double p = D.inner(r, ids[i]);
      projs[i] = {p, ids[i]};
      if (proj_tree) proj_tree[(int64_t)level * D.N + ids[i]] = p;
    }
  };
  if (par > 1 && n >= 65536) {
    std::vector<std::thread> pool;
    for (int32_t w = 1; w < par; ++w) pool.emplace_back(project, n * w / par, n * (w + 1) / par);
    project(0, n / par);
    for (std::thread& th : pool) th.join();
  } else {
    project(0, n);
  }
  // :504,509-512 sortByVG snd = STABLE merge sort, `comparing` on Double (-0.0 == 0.0)
  std::stable_sort(projs.begin(), projs.end(),
                   [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) {
                     return a.first < b.first;
                   });
  const int64_t nh = n / 2;  // :503
  double mgl, mgr;
  if (n >= 3) {  // :497
    mgl = projs[nh - 1].first;
    mgr = projs[nh + 1].first;
  } else if (n == 2) {  // :498
    mgl = projs[0].first;
    mgr = projs[1].first;
  } else {  // :499
    mgl = mgr = projs[0].first;
  }
  ft.thr[heap] = projs[nh].first;  // :501
  ft.mglo[heap] = mgl;
  ft.mghi[heap] = mgr;
  std::vector<int32_t> ll((size_t)nh), rr((size_t)(n - nh));  // :495 take nh / drop nh
  for (int64_t i = 0; i < nh; ++i) ll[i] = projs[i].second;
  for (int64_t i = nh; i < n; ++i) rr[i - nh] = projs[i].second;
  std::vector<std::pair<double, int32_t>>().swap(projs);
  std::vector<int32_t>().swap(ids);
  if (par > 1) {
    const int32_t pl = par / 2, pr = par - pl;
    std::thread left([&] {
      build_node(D, rvs, L, minLeaf, level + 1, 2 * heap + 1, off, ll, ft, proj_tree, pl);  // :296
    });
    build_node(D, rvs, L, minLeaf, level + 1, 2 * heap + 2, off + nh, rr, ft, proj_tree, pr);  // :297
    left.join();
    return;
  }
  build_node(D, rvs, L, minLeaf, level + 1, 2 * heap + 1, off, ll, ft, proj_tree);       // :296
  build_node(D, rvs, L, minLeaf, level + 1, 2 * heap + 2, off + nh, rr, ft, proj_tree);  // :297
}

// threads > 1: the trees of a forest are independent (createMulti maps `create` over the IntMap,
// Internal.hs:234-240), so they may be built concurrently; every tree is still built by the
// sequential recursion above and the result does not depend on the thread count.  The
// reference itself is single-threaded: threads = 1 is "the reference CPU path", more threads
// are the courtesy all-core baseline of SURVEY 8(d).
template <class Data>
void forest_build(const Data& D, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                  int32_t* perm, double* thr, double* mglo, double* mghi, double* proj_out,
                  int32_t threads = 1) {
  const int64_t nodes = ((int64_t)1 << L) - 1;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  std::vector<SparseVec> all = sparsify(R, T, L, D.d);
  // more threads than trees: the surplus works INSIDE the trees (see build_node's `par`)
  const int32_t par = threads > T ? threads / (T > 0 ? T : 1) : 1;
  auto one_tree = [&](int32_t t) {  // create, Internal.hs:217-225
    for (int64_t h = 0; h < nodes; ++h)
      thr[t * nodes + h] = mglo[t * nodes + h] = mghi[t * nodes + h] = nan;
    std::vector<SparseVec> rvs(all.begin() + (size_t)t * L, all.begin() + (size_t)(t + 1) * L);
    std::vector<int32_t> ids((size_t)D.N);
    for (int64_t i = 0; i < D.N; ++i) ids[i] = (int32_t)i;  // dataset in input order
    FlatTree ft{perm + (int64_t)t * D.N, thr + t * nodes, mglo + t * nodes, mghi + t * nodes};
    build_node(D, rvs, L, minLeaf, 0, 0, 0, ids, ft,
               proj_out ? proj_out + (int64_t)t * L * D.N : nullptr, par);
  };
  if (threads <= 1 || T <= 1) {
    for (int32_t t = 0; t < T; ++t) one_tree(t);  // createMulti, ascending key
    return;
  }
  std::atomic<int32_t> next(0);
  std::vector<std::thread> pool;
  for (int32_t w = 0; w < std::min(threads, T); ++w)
    pool.emplace_back([&] {
      for (int32_t t = next.fetch_add(1); t < T; t = next.fetch_add(1)) one_tree(t);
    });
  for (std::thread& th : pool) th.join();
}

// RPTree.hs:297-314 `candidates`, on the flat layout.  projq[level] = r_level `inner` q.
struct Cand {
  const int32_t* perm;
  const double *thr, *mglo, *mghi;
  int32_t L, minLeaf;
  const double* projq;
  std::vector<int32_t>* out;
  void go(int32_t level, int64_t heap, int64_t off, int64_t n) const {
    if (is_leaf(level, n, L, minLeaf)) {  // :299 Tip
      for (int64_t i = 0; i < n; ++i) out->push_back(perm[off + i]);
      return;
    }
    const double proj = projq[level];              // :303-304
    const double dl = std::fabs(mglo[heap] - proj);  // :306
    const double dr = std::fabs(mghi[heap] - proj);  // :307
    const double th = thr[heap];
    const int64_t nh = n / 2;
    const int64_t hl = 2 * heap + 1, hr = 2 * heap + 2;
    if (proj < th && dl > dr) {  // :309-310 both
      go(level + 1, hl, off, nh);
      go(level + 1, hr, off + nh, n - nh);
    } else if (proj < th) {  // :311
      go(level + 1, hl, off, nh);
    } else if (proj > th && dl < dr) {  // :312-313 both
      go(level + 1, hl, off, nh);
      go(level + 1, hr, off + nh, n - nh);
    } else {  // :314 (includes proj == thr)
      go(level + 1, hr, off + nh, n - nh);
    }
  }
};

template <class InnerQ>
void tree_candidates(InnerQ innerq, const std::vector<SparseVec>& all, int32_t L, int32_t minLeaf,
                     int64_t N, const int32_t* perm, const double* thr, const double* mglo,
                     const double* mghi, int32_t t, std::vector<int32_t>& out) {
  const int64_t nodes = ((int64_t)1 << L) - 1;
  std::vector<double> projq((size_t)L);
  // proj is the same for every node of a level (one vector per level, Internal.hs:171-175);
  // the reference evaluates it lazily per visited Bin — values are identical.
  for (int32_t l = 0; l < L; ++l) projq[l] = innerq(all[(size_t)t * L + l]);
  Cand c{perm + (int64_t)t * N, thr + t * nodes, mglo + t * nodes, mghi + t * nodes,
         L,    minLeaf,           projq.data(),     &out};
  c.go(0, 0, 0, N);
}

// RPTree.hs:318-342 `candidatesH` on the flat layout: one (priority, leaf) entry per leaf
// reached by the same 4-way rule; the priority is the smallest margin distance met on the way
// down (`pl = p min dl` towards the left child, `pr = p min dr` towards the right one), starting
// from +infinity.  Entries are listed in DFS order (left before right), trees in key order.
struct LeafEntry {
  double prio;
  int32_t tree;
  int64_t off, n;
};
struct CandH {
  const double *thr, *mglo, *mghi;
  int32_t L, minLeaf, tree;
  const double* projq;
  std::vector<LeafEntry>* out;
  void go(int32_t level, int64_t heap, int64_t off, int64_t n, double p) const {
    if (is_leaf(level, n, L, minLeaf)) {  // :323 Tip -> insertp p xs
      out->push_back(LeafEntry{p, tree, off, n});
      return;
    }
    const double proj = projq[level];
    const double dl = std::fabs(mglo[heap] - proj);  // :330
    const double dr = std::fabs(mghi[heap] - proj);  // :331
    const double pl = p <= dl ? p : dl;              // :332 p `min` dl
    const double pr = p <= dr ? p : dr;              // :333
    const double th = thr[heap];
    const int64_t nh = n / 2;
    const int64_t hl = 2 * heap + 1, hr = 2 * heap + 2;
    if (proj < th && dl > dr) {  // :335-336
      go(level + 1, hl, off, nh, pl);
      go(level + 1, hr, off + nh, n - nh, pr);
    } else if (proj < th) {  // :337
      go(level + 1, hl, off, nh, pl);
    } else if (proj > th && dl < dr) {  // :338-339
      go(level + 1, hl, off, nh, pl);
      go(level + 1, hr, off + nh, n - nh, pr);
    } else {  // :340
      go(level + 1, hr, off + nh, n - nh, pr);
    }
  }
};

template <class InnerQ>
void forest_leaves_h(InnerQ innerq, const std::vector<SparseVec>& all, int32_t T, int32_t L,
                     int32_t minLeaf, int64_t N, const double* thr, const double* mglo,
                     const double* mghi, std::vector<LeafEntry>& out) {
  const int64_t nodes = ((int64_t)1 << L) - 1;
  std::vector<double> projq((size_t)L);
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t l = 0; l < L; ++l) projq[l] = innerq(all[(size_t)t * L + l]);
    CandH c{thr + t * nodes, mglo + t * nodes, mghi + t * nodes, L, minLeaf, t, projq.data(), &out};
    c.go(0, 0, 0, N, std::numeric_limits<double>::infinity());  // :320 infty = 1 / 0
  }
}

// RPTree.hs:205-217 `knnH`: leaves are taken in increasing priority while the running count
// stays <= k — but always at least one —, each new bucket is PREPENDED (`xsh <> acc`), and the
// result is neither sorted by distance nor cut to k.  Equal priorities: the `heaps` package
// pops them in an order that depends on its internal shape; here ties keep (tree, DFS) order.
// Returns the leaves selected, in result order.
std::vector<LeafEntry> knn_h_select(std::vector<LeafEntry> es, int32_t k) {
  std::stable_sort(es.begin(), es.end(),
                   [](const LeafEntry& a, const LeafEntry& b) { return a.prio < b.prio; });
  std::vector<LeafEntry> acc;
  int64_t n = 0;
  for (const LeafEntry& e : es) {
    const int64_t ntot = e.n + n;
    // `not (null acc)`: acc is the vector of POINTS taken so far (:216) — an empty bucket (a streamed
    // tree can hold one) does not count as "at least one"; batch forests have no empty leaves
    if (ntot > k && n > 0) break;
    acc.insert(acc.begin(), e);
    n = ntot;
  }
  return acc;
}

struct DistId {
  double dist;
  int32_t id;
};

// RPTree.hs:174 `take k $ sortByVG fst cs` (stable), optional de-duplication extension.
int32_t topk_from(std::vector<DistId>& cs, int32_t k, int32_t dedup, int32_t* out_ids,
                  double* out_dist) {
  std::stable_sort(cs.begin(), cs.end(),
                   [](const DistId& a, const DistId& b) { return a.dist < b.dist; });
  int32_t m = 0;
  std::set<int32_t> seen;
  for (size_t i = 0; i < cs.size() && m < k; ++i) {
    if (dedup == 2) {  // knnPQ's `nub`: entries of equal PRIORITY (distance) collapse to one
      if (i > 0 && cs[i].dist == cs[i - 1].dist) continue;
    } else if (dedup) {
      if (seen.count(cs[i].id)) continue;
      seen.insert(cs[i].id);
    }
    out_ids[m] = cs[i].id;
    out_dist[m] = cs[i].dist;
    ++m;
  }
  return m;
}

double true_l2_ss(int64_t n1, const int32_t* i1, const double* v1, int64_t n2, const int32_t* i2,
                  const double* v2) {
  int64_t a = 0, b = 0;
  double s = 0;
  while (a < n1 || b < n2) {
    double x;
    if (a < n1 && (b >= n2 || i1[a] < i2[b])) x = v1[a++];
    else if (b < n2 && (a >= n1 || i2[b] < i1[a])) x = -v2[b++];
    else x = v1[a++] - v2[b++];
    s += x * x;
  }
  return std::sqrt(s);
}

}  // namespace

extern "C" {

// ------------------------------- RNG -------------------------------------------------------
void rpo_gen_init(rpo_gen* g, uint64_t s) {  // mkSMGen
  g->seed = mix64(s);
  g->gamma = mix_gamma(s + 0x9e3779b97f4a7c15ULL);
}
uint64_t rpo_next_word64(rpo_gen* g) {
  g->seed += g->gamma;
  return mix64(g->seed);
}
double rpo_next_double(rpo_gen* g) {
  return (double)(rpo_next_word64(g) >> 11) * 0x1.0p-53;
}
int rpo_bernoulli(rpo_gen* g, double p) { return rpo_next_double(g) < p; }
double rpo_normal(rpo_gen* g, double mu, double sig) {  // Box–Muller, u1 then u2
  double u1 = rpo_next_double(g);
  double u2 = rpo_next_double(g);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2) * sig + mu;
}
double rpo_uniform_r(rpo_gen* g, double lo, double hi) {
  return rpo_next_double(g) * (hi - lo) + lo;
}

int64_t rpo_sparse_normal(rpo_gen* g, double p, int32_t d, double mu, double sig, int32_t* idx,
                          double* val) {
  return sparse_vg(g, p, d, idx, val, [&] { return rpo_normal(g, mu, sig); });
}
int64_t rpo_sparse_uniform(rpo_gen* g, double p, int32_t d, double lo, double hi, int32_t* idx,
                           double* val) {
  return sparse_vg(g, p, d, idx, val, [&] { return rpo_uniform_r(g, lo, hi); });
}

// Batch.hs:57-61: sample seed $ replicateM ntrees $ V.replicateM maxd (sparse pnz dim stdNormal)
void rpo_forest_hyperplanes(uint64_t seed, int32_t T, int32_t L, double pnz, int32_t d, double* R,
                            int32_t* nnz) {
  rpo_gen g;
  rpo_gen_init(&g, seed);
  std::vector<int32_t> idx((size_t)d);
  std::vector<double> val((size_t)d);
  std::memset(R, 0, sizeof(double) * (size_t)T * L * d);
  for (int32_t t = 0; t < T; ++t)
    for (int32_t l = 0; l < L; ++l) {
      int64_t m = rpo_sparse_normal(&g, pnz, d, 0.0, 1.0, idx.data(), val.data());
      double* r = R + ((int64_t)t * L + l) * d;
      for (int64_t j = 0; j < m; ++j) r[idx[j]] = val[j];
      if (nnz) nnz[(int64_t)t * L + l] = (int32_t)m;
    }
}

// Gen.hs:132-137 normalDense2: b <- bernoulli 0.5; dense d (normal 0 0.5) | dense d (normal 2 0.5)
void rpo_data_normal_dense2(uint64_t seed, int64_t n, int32_t d, double* X) {
  rpo_gen g;
  rpo_gen_init(&g, seed);
  for (int64_t i = 0; i < n; ++i) {
    double mu = rpo_bernoulli(&g, 0.5) ? 0.0 : 2.0;
    for (int32_t j = 0; j < d; ++j) X[i * d + j] = rpo_normal(&g, mu, 0.5);
  }
}

// test/Data/RPTreeSpec.hs:112-120 circle2d2 over Gen.hs:115-123 circle2d (note: the test
// `x**2 + y**2 <= r` compares with r, not r^2 — restated as written; r = 1 so no difference)
void rpo_data_circle2d2(uint64_t seed, int64_t n, double* X) {
  rpo_gen g;
  rpo_gen_init(&g, seed);
  const double r = 1.0;
  for (int64_t i = 0; i < n; ++i) {
    int b = rpo_bernoulli(&g, 0.5);
    double x, y;
    for (;;) {
      x = rpo_uniform_r(&g, -r, r);
      y = rpo_uniform_r(&g, -r, r);
      if (sq(x) + sq(y) <= r) break;
    }
    if (b) {
      X[2 * i] = x;
      X[2 * i + 1] = y;
    } else {  // (^+^ d) with d = [2,3]: zipWith (+) d c, Internal.hs:340
      X[2 * i] = 2.0 + x;
      X[2 * i + 1] = 3.0 + y;
    }
  }
}

int64_t rpo_data_normal_sparse2(uint64_t seed, int64_t n, int32_t d, double pnz, int64_t* rowptr,
                                int32_t* col, double* val, int64_t cap) {
  rpo_gen g;
  rpo_gen_init(&g, seed);
  std::vector<int32_t> idx((size_t)d);
  std::vector<double> v((size_t)d);
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {  // Gen.hs:125-130
    double mu = rpo_bernoulli(&g, 0.5) ? 0.0 : 2.0;
    int64_t m = rpo_sparse_normal(&g, pnz, d, mu, 0.5, idx.data(), v.data());
    for (int64_t j = 0; j < m && nnz + j < cap; ++j) {
      col[nnz + j] = idx[j];
      val[nnz + j] = v[j];
    }
    nnz += m;
    rowptr[i + 1] = nnz;
  }
  return nnz;
}

int64_t rpo_data_sparse_uniform(uint64_t seed, int64_t n, int32_t d, double pnz, int64_t* rowptr,
                                int32_t* col, double* val, int64_t cap) {
  rpo_gen g;
  rpo_gen_init(&g, seed);
  std::vector<int32_t> idx((size_t)d);
  std::vector<double> v((size_t)d);
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    // values in (0,1]: 1 - u with u in [0,1)   (toUnitRange-like, bench/time/Main.hs:124-125)
    int64_t m = sparse_vg(&g, pnz, d, idx.data(), v.data(),
                          [&] { return 1.0 - rpo_next_double(&g); });
    for (int64_t j = 0; j < m && nnz + j < cap; ++j) {
      col[nnz + j] = idx[j];
      val[nnz + j] = v[j];
    }
    nnz += m;
    rowptr[i + 1] = nnz;
  }
  return nnz;
}

// ------------------------------- algebra ---------------------------------------------------
// Internal.hs:351-366 innerSS: sorted-index merge join, RIGHT-nested sum with terminal 0:
//   EQ -> (xl * xr +) $ go (succ i1) (succ i2)
double rpo_inner_ss(int64_t n1, const int32_t* i1, const double* v1, int64_t n2, const int32_t* i2,
                    const double* v2) {
  // collect the matched products in visit order, then fold from the right
  static thread_local std::vector<double> prods;
  prods.clear();
  int64_t a = 0, b = 0;
  while (a < n1 && b < n2) {  // :358
    if (i1[a] == i2[b]) {     // :364
      prods.push_back(v1[a] * v2[b]);
      ++a;
      ++b;
    } else if (i1[a] < i2[b]) {  // :365
      ++a;
    } else {  // :366
      ++b;
    }
  }
  const int64_t m = (int64_t)prods.size();
  double acc = 0.0;
  for (int64_t j = m - 1; j >= 0; --j) acc = prods[j] + acc;
  return acc;
}

// Internal.hs:369-382 innerSD: go i | i >= nz1 || i >= nz2 = 0 ; (xl * xr +) $ go (succ i)
// NB the guard compares the NONZERO COUNTER with the dense length (:376) — restated as is.
double rpo_inner_sd(int64_t n1, const int32_t* i1, const double* v1, int64_t n2, const double* x) {
  int64_t m = n1 < n2 ? n1 : n2;
  double acc = 0.0;
  for (int64_t j = m - 1; j >= 0; --j) acc = v1[j] * x[i1[j]] + acc;
  return acc;
}

// Internal.hs:384-385 innerDD = VG.sum $ VG.zipWith (*): left fold from 0
double rpo_inner_dd(int64_t n, const double* a, const double* b) {
  double acc = 0.0;
  for (int64_t j = 0; j < n; ++j) acc = acc + a[j] * b[j];
  return acc;
}

// Internal.hs:403-406 metricDDL2 = sqrt $ VG.sum $ VG.map (** 2) (zipWith (-) u v)
double rpo_metric_dd(int64_t n, const double* u, const double* v) {
  double acc = 0.0;
  for (int64_t j = 0; j < n; ++j) acc = acc + sq(u[j] - v[j]);
  return std::sqrt(acc);
}

// Internal.hs:455-470 binSDD f z: stops when EITHER operand is exhausted (:462)
static int64_t bin_sdd(bool minus, int64_t n1, const int32_t* i1, const double* v1, int64_t n2,
                       const double* x, double* out) {
  int64_t a = 0, b = 0, m = 0;
  while (a < n1 && b < n2) {
    double xl, xr;
    if (i1[a] == b) {  // EQ :468
      xl = v1[a];
      xr = x[b];
      ++a;
      ++b;
    } else if (i1[a] < b) {  // LT :469 f xl z
      xl = v1[a];
      xr = 0.0;
      ++a;
    } else {  // GT :470 f z xr
      xl = 0.0;
      xr = x[b];
      ++b;
    }
    out[m++] = minus ? xl - xr : xl + xr;
  }
  return m;
}
int64_t rpo_sum_sd(int64_t n1, const int32_t* i1, const double* v1, int64_t n2, const double* x,
                   double* out) {
  return bin_sdd(false, n1, i1, v1, n2, x, out);
}
int64_t rpo_diff_sd(int64_t n1, const int32_t* i1, const double* v1, int64_t n2, const double* x,
                    double* out) {
  return bin_sdd(true, n1, i1, v1, n2, x, out);
}

// Internal.hs:396-400 metricSDL2 u v = sqrt $ sum $ map (**2) (u `diffSD` v)
double rpo_metric_sd(int64_t n1, const int32_t* i1, const double* v1, int64_t n2,
                     const double* x) {
  std::vector<double> duv((size_t)n2);
  int64_t m = bin_sdd(true, n1, i1, v1, n2, x, duv.data());
  double acc = 0.0;
  for (int64_t j = 0; j < m; ++j) acc = acc + sq(duv[j]);
  return std::sqrt(acc);
}

// Internal.hs:389-393 metricSSL2 over :435-450 binSS (-) 0 (stops when either is exhausted)
double rpo_metric_ss(int64_t n1, const int32_t* i1, const double* v1, int64_t n2,
                     const int32_t* i2, const double* v2) {
  int64_t a = 0, b = 0;
  double acc = 0.0;
  while (a < n1 && b < n2) {  // :442
    double y;
    if (i1[a] == i2[b]) y = v1[a++] - v2[b++];  // :448
    else if (i1[a] < i2[b]) y = v1[a++] - 0.0;  // :449
    else y = 0.0 - v2[b++];                     // :450
    acc = acc + sq(y);
  }
  return std::sqrt(acc);
}

// Conduit.hs:132-141
void rpo_tree_cfg(int32_t minLeaf, int64_t n, int32_t d, int32_t* maxDepth, int64_t* chunk,
                  double* pnz) {
  double x = std::log((double)n / (double)minLeaf) / std::log(2.0);  // logBase 2 (n / minl)
  *maxDepth = (int32_t)std::ceil(x);
  *chunk = (int64_t)std::ceil((double)n / 100.0);
  double pnzMin = 1.0 / (std::log((double)d) / std::log(10.0));  // 1 / logBase 10 d
  *pnz = pnzMin < 1.0 ? pnzMin : 1.0;
}

// Internal.hs:486-512 on precomputed projections
int64_t rpo_partition_at_median(int64_t n, const double* p, int32_t* order, double* thr_mg) {
  if (n < 1) return -1;  // :492 Nothing
  std::vector<std::pair<double, int32_t>> projs((size_t)n);
  for (int64_t i = 0; i < n; ++i) projs[i] = {p[i], (int32_t)i};
  std::stable_sort(projs.begin(), projs.end(),
                   [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) {
                     return a.first < b.first;
                   });
  const int64_t nh = n / 2;
  for (int64_t i = 0; i < n; ++i) order[i] = projs[i].second;
  thr_mg[0] = projs[nh].first;
  if (n >= 3) {
    thr_mg[1] = projs[nh - 1].first;
    thr_mg[2] = projs[nh + 1].first;
  } else if (n == 2) {
    thr_mg[1] = projs[0].first;
    thr_mg[2] = projs[1].first;
  } else {
    thr_mg[1] = thr_mg[2] = projs[0].first;
  }
  return nh;
}

// ------------------------------- build -----------------------------------------------------
void rpo_forest_build_dense(const double* X, int64_t N, int32_t d, const double* R, int32_t T,
                            int32_t L, int32_t minLeaf, int32_t* perm, double* thr, double* mglo,
                            double* mghi, double* proj_out) {
  DenseData D{X, N, d};
  forest_build(D, R, T, L, minLeaf, perm, thr, mglo, mghi, proj_out);
}
void rpo_forest_build_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                          int32_t d, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                          int32_t* perm, double* thr, double* mglo, double* mghi,
                          double* proj_out) {
  CsrData D{rowptr, col, val, N, d};
  forest_build(D, R, T, L, minLeaf, perm, thr, mglo, mghi, proj_out);
}

// ------------------------------- queries ---------------------------------------------------
int64_t rpo_candidates_dense(const double* q, int32_t d, const double* R, int32_t T, int32_t L,
                             int32_t minLeaf, int64_t N, const int32_t* perm, const double* thr,
                             const double* mglo, const double* mghi, int32_t t, int32_t* out,
                             int64_t cap) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<int32_t> c;
  tree_candidates(
      [&](const SparseVec& r) {
        return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
      },
      all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
  for (int64_t i = 0; i < (int64_t)c.size() && i < cap; ++i) out[i] = c[i];
  return (int64_t)c.size();
}

int64_t rpo_candidates_sparse(int64_t qn, const int32_t* qi, const double* qv, int32_t d,
                              const double* R, int32_t T, int32_t L, int32_t minLeaf, int64_t N,
                              const int32_t* perm, const double* thr, const double* mglo,
                              const double* mghi, int32_t t, int32_t* out, int64_t cap) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<int32_t> c;
  tree_candidates(
      [&](const SparseVec& r) {
        return rpo_inner_ss((int64_t)r.idx.size(), r.idx.data(), r.val.data(), qn, qi, qv);
      },
      all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
  for (int64_t i = 0; i < (int64_t)c.size() && i < cap; ++i) out[i] = c[i];
  return (int64_t)c.size();
}

// RPTree.hs:174-176: cs = map (\xe -> (eEmbed xe `distf` q, xe)) $ fold $ (`candidates` q) <$> tts
int32_t rpo_knn_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                      int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm,
                      const double* thr, const double* mglo, const double* mghi, int32_t k,
                      int32_t dedup, int32_t* out_ids, double* out_dist) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<int32_t> c;
  for (int32_t t = 0; t < T; ++t)  // fold over the IntMap: ascending key, concatenation
    tree_candidates(
        [&](const SparseVec& r) {
          return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
        },
        all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
  std::vector<DistId> cs(c.size());
  for (size_t i = 0; i < c.size(); ++i)
    cs[i] = {rpo_metric_dd(d, X + (int64_t)c[i] * d, q), c[i]};  // metricL2 DVector DVector :339
  return topk_from(cs, k, dedup, out_ids, out_dist);
}

int32_t rpo_knn_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                    int32_t d, int64_t qn, const int32_t* qi, const double* qv, const double* R,
                    int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm, const double* thr,
                    const double* mglo, const double* mghi, int32_t k, int32_t dedup,
                    int32_t true_l2, int32_t* out_ids, double* out_dist) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<int32_t> c;
  for (int32_t t = 0; t < T; ++t)
    tree_candidates(
        [&](const SparseVec& r) {
          return rpo_inner_ss((int64_t)r.idx.size(), r.idx.data(), r.val.data(), qn, qi, qv);
        },
        all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
  std::vector<DistId> cs(c.size());
  for (size_t i = 0; i < c.size(); ++i) {
    int64_t a = rowptr[c[i]], b = rowptr[c[i] + 1];
    double dist = true_l2 ? true_l2_ss(b - a, col + a, val + a, qn, qi, qv)
                          : rpo_metric_ss(b - a, col + a, val + a, qn, qi, qv);  // :324
    cs[i] = {dist, c[i]};
  }
  return topk_from(cs, k, dedup, out_ids, out_dist);
}

// RPTree.hs:199-217 knnH with distf = metricL2 (dense data, dense query)
int64_t rpo_knn_h_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                        int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm,
                        const double* thr, const double* mglo, const double* mghi, int32_t k,
                        int32_t* out_ids, double* out_dist, int64_t cap) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<LeafEntry> es;
  forest_leaves_h(
      [&](const SparseVec& r) {
        return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
      },
      all, T, L, minLeaf, N, thr, mglo, mghi, es);
  int64_t m = 0;
  for (const LeafEntry& e : knn_h_select(es, k))
    for (int64_t i = 0; i < e.n; ++i, ++m)
      if (m < cap) {
        const int32_t id = perm[(int64_t)e.tree * N + e.off + i];
        out_ids[m] = id;
        out_dist[m] = rpo_metric_dd(d, X + (int64_t)id * d, q);
      }
  return m;
}

int64_t rpo_knn_h_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                      int32_t d, int64_t qn, const int32_t* qi, const double* qv, const double* R,
                      int32_t T, int32_t L, int32_t minLeaf, const int32_t* perm,
                      const double* thr, const double* mglo, const double* mghi, int32_t k,
                      int32_t true_l2, int32_t* out_ids, double* out_dist, int64_t cap) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<LeafEntry> es;
  forest_leaves_h(
      [&](const SparseVec& r) {
        return rpo_inner_ss((int64_t)r.idx.size(), r.idx.data(), r.val.data(), qn, qi, qv);
      },
      all, T, L, minLeaf, N, thr, mglo, mghi, es);
  int64_t m = 0;
  for (const LeafEntry& e : knn_h_select(es, k))
    for (int64_t i = 0; i < e.n; ++i, ++m)
      if (m < cap) {
        const int32_t id = perm[(int64_t)e.tree * N + e.off + i];
        const int64_t a = rowptr[id], b = rowptr[id + 1];
        out_ids[m] = id;
        out_dist[m] = true_l2 ? true_l2_ss(b - a, col + a, val + a, qn, qi, qv)
                              : rpo_metric_ss(b - a, col + a, val + a, qn, qi, qv);
      }
  return m;
}

// leaf priorities of one tree (candidatesH), DFS order; returns the number of leaves reached
int64_t rpo_candidates_h_dense(const double* q, int32_t d, const double* R, int32_t T, int32_t L,
                               int32_t minLeaf, int64_t N, const double* thr, const double* mglo,
                               const double* mghi, int32_t t, double* prio, int64_t* off,
                               int64_t* len, int64_t cap) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<LeafEntry> es;
  const int64_t nodes = ((int64_t)1 << L) - 1;
  std::vector<double> projq((size_t)L);
  for (int32_t l = 0; l < L; ++l) {
    const SparseVec& r = all[(size_t)t * L + l];
    projq[l] = rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
  }
  CandH c{thr + t * nodes, mglo + t * nodes, mghi + t * nodes, L, minLeaf, t, projq.data(), &es};
  c.go(0, 0, 0, N, std::numeric_limits<double>::infinity());
  for (size_t i = 0; i < es.size() && (int64_t)i < cap; ++i) {
    prio[i] = es[i].prio;
    off[i] = es[i].off;
    len[i] = es[i].n;
  }
  return (int64_t)es.size();
}

// RPTree.hs:259-282
double rpo_recall_with_dense(const double* X, int64_t N, int32_t d, const double* q,
                             const double* R, int32_t T, int32_t L, int32_t minLeaf,
                             const int32_t* perm, const double* thr, const double* mglo,
                             const double* mghi, int32_t k) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  double sum = 0.0;
  for (int32_t t = 0; t < T; ++t) {  // :265-268 mean over trees
    std::vector<int32_t> c;
    tree_candidates(
        [&](const SparseVec& r) {
          return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
        },
        all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
    std::set<int32_t> aa(c.begin(), c.end());  // :279
    // :280-282 dists = sortBy (comparing snd) over `points tt` (leaf order = perm order)
    std::vector<DistId> ds((size_t)N);
    const int32_t* pt = perm + (int64_t)t * N;
    for (int64_t i = 0; i < N; ++i) ds[i] = {rpo_metric_dd(d, X + (int64_t)pt[i] * d, q), pt[i]};
    std::stable_sort(ds.begin(), ds.end(),
                     [](const DistId& a, const DistId& b) { return a.dist < b.dist; });
    std::set<int32_t> kk;
    for (int64_t i = 0; i < k && i < N; ++i) kk.insert(ds[i].id);
    int64_t inter = 0;
    for (int32_t id : kk) inter += aa.count(id);
    sum += (double)inter / (double)k;  // :276-278
  }
  return sum / (double)T;
}

void rpo_brute_knn_dense(const double* X, int64_t N, int32_t d, const double* q, int32_t k,
                         int32_t* out_ids, double* out_dist) {
  std::vector<DistId> ds((size_t)N);
  for (int64_t i = 0; i < N; ++i) ds[i] = {rpo_metric_dd(d, X + i * d, q), (int32_t)i};
  int64_t kk = k < N ? k : N;
  std::partial_sort(ds.begin(), ds.begin() + kk, ds.end(), [](const DistId& a, const DistId& b) {
    return a.dist < b.dist || (a.dist == b.dist && a.id < b.id);
  });
  for (int64_t i = 0; i < kk; ++i) {
    out_ids[i] = ds[i].id;
    out_dist[i] = ds[i].dist;
  }
}


// ------------------------------- typed / threaded variants ---------------------------------
// xdtype: 0 = double rows (the reference), 1 = float rows (upcast exactly on read).
void rpo_forest_build_dense_ex(const void* X, int32_t xdtype, int64_t N, int32_t d,
                               const double* R, int32_t T, int32_t L, int32_t minLeaf,
                               int32_t* perm, double* thr, double* mglo, double* mghi,
                               double* proj_out, int32_t threads) {
  if (xdtype == 1) {
    DenseDataT<float> D{(const float*)X, N, d};
    forest_build(D, R, T, L, minLeaf, perm, thr, mglo, mghi, proj_out, threads);
  } else {
    DenseDataT<double> D{(const double*)X, N, d};
    forest_build(D, R, T, L, minLeaf, perm, thr, mglo, mghi, proj_out, threads);
  }
}
void rpo_forest_build_csr_ex(const int64_t* rowptr, const int32_t* col, const double* val,
                             int64_t N, int32_t d, const double* R, int32_t T, int32_t L,
                             int32_t minLeaf, int32_t* perm, double* thr, double* mglo,
                             double* mghi, double* proj_out, int32_t threads) {
  CsrData D{rowptr, col, val, N, d};
  forest_build(D, R, T, L, minLeaf, perm, thr, mglo, mghi, proj_out, threads);
}

// knn (RPTree.hs:168-176) for a batch of dense queries Q[nq][d] (double); queries are
// independent, `threads` of them are answered concurrently.  out_ids/out_dist [nq][k], unused
// slots -1 / +inf; out_count[nq].  vote_thr > 0: the candidate ids are first reduced by
// keepCounts (see rpo_keep_counts) — an extension, the reference's knn never does that.
}  // extern "C"
namespace {
template <class E>
void knn_batch(const DenseDataT<E>& D, const double* Q, int64_t nq, const double* R, int32_t T,
               int32_t L, int32_t minLeaf, const int32_t* perm, const double* thr,
               const double* mglo, const double* mghi, int32_t k, int32_t dedup, int32_t vote_thr,
               int32_t* out_ids, double* out_dist, int32_t* out_count, int32_t threads) {
  std::vector<SparseVec> all = sparsify(R, T, L, D.d);
  auto one = [&](int64_t i) {
    const double* q = Q + i * D.d;
    std::vector<int32_t> c;
    for (int32_t t = 0; t < T; ++t)
      tree_candidates(
          [&](const SparseVec& r) {
            return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), D.d, q);
          },
          all, L, minLeaf, D.N, perm, thr, mglo, mghi, t, c);
    if (vote_thr > 0) {  // counts + keepCounts, RPTree.hs:464-478: ascending key order
      std::map<int32_t, int32_t> mm;
      for (int32_t id : c) mm[id] += 1;
      c.clear();
      for (const auto& kv : mm)
        if (kv.second >= vote_thr) c.push_back(kv.first);
    }
    std::vector<DistId> cs(c.size());
    for (size_t j = 0; j < c.size(); ++j) cs[j] = {D.metric(c[j], q), c[j]};
    int32_t* oi = out_ids + i * k;
    double* od = out_dist + i * k;
    const int32_t m = topk_from(cs, k, dedup, oi, od);
    for (int32_t j = m; j < k; ++j) {
      oi[j] = -1;
      od[j] = std::numeric_limits<double>::infinity();
    }
    out_count[i] = m;
  };
  if (threads <= 1 || nq <= 1) {
    for (int64_t i = 0; i < nq; ++i) one(i);
    return;
  }
  std::atomic<int64_t> next(0);
  std::vector<std::thread> pool;
  for (int32_t w = 0; w < threads; ++w)
    pool.emplace_back([&] {
      for (int64_t i = next.fetch_add(1); i < nq; i = next.fetch_add(1)) one(i);
    });
  for (std::thread& th : pool) th.join();
}
}  // namespace
extern "C" {

void rpo_knn_dense_batch(const void* X, int32_t xdtype, int64_t N, int32_t d, const double* Q,
                         int64_t nq, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                         const int32_t* perm, const double* thr, const double* mglo,
                         const double* mghi, int32_t k, int32_t dedup, int32_t vote_thr,
                         int32_t* out_ids, double* out_dist, int32_t* out_count,
                         int32_t threads) {
  if (xdtype == 1) {
    DenseDataT<float> D{(const float*)X, N, d};
    knn_batch(D, Q, nq, R, T, L, minLeaf, perm, thr, mglo, mghi, k, dedup, vote_thr, out_ids,
              out_dist, out_count, threads);
  } else {
    DenseDataT<double> D{(const double*)X, N, d};
    knn_batch(D, Q, nq, R, T, L, minLeaf, perm, thr, mglo, mghi, k, dedup, vote_thr, out_ids,
              out_dist, out_count, threads);
  }
}

// RPTree.hs:464-478 (commented-out sketch in the reference): `counts` folds the candidate list
// into a Map id -> occurrences, `keepCounts thr` keeps the entries with count >= thr via
// M.foldrWithKey — i.e. in ASCENDING key order.  ids[n] in, (out_ids, out_counts) out; returns
// the number kept.
int64_t rpo_keep_counts(const int32_t* ids, int64_t n, int32_t thr, int32_t* out_ids,
                        int32_t* out_counts) {
  std::map<int32_t, int32_t> mm;  // count: M.insertWith mappend x (Sum 1)
  for (int64_t i = 0; i < n; ++i) mm[ids[i]] += 1;
  int64_t m = 0;
  for (const auto& kv : mm)
    if (kv.second >= thr) {  // v >= thr
      out_ids[m] = kv.first;
      out_counts[m] = kv.second;
      ++m;
    }
  return m;
}

// recallWith (RPTree.hs:259-282) with the reference's VALUE semantics: `aa` and `kk` are
// Data.Sets of `Embed` values (derived Ord: the vector's components, then the payload), so
// points with identical coordinates — and, here, no payload (`Embed v ()`) — collapse to one
// element in either set.  rpo_recall_with_dense compares point IDS (payload = the id); this
// variant compares rows with Double's (==) (so -0.0 == 0.0).
}  // extern "C"
namespace {
struct RowLess {
  const double* X;
  int32_t d;
  bool operator()(int32_t a, int32_t b) const {
    const double *x = X + (int64_t)a * d, *y = X + (int64_t)b * d;
    for (int32_t j = 0; j < d; ++j) {
      if (x[j] < y[j]) return true;
      if (y[j] < x[j]) return false;
    }
    return false;
  }
};
}  // namespace
extern "C" {
double rpo_recall_with_dense_values(const double* X, int64_t N, int32_t d, const double* q,
                                    const double* R, int32_t T, int32_t L, int32_t minLeaf,
                                    const int32_t* perm, const double* thr, const double* mglo,
                                    const double* mghi, int32_t k) {
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  double sum = 0.0;
  RowLess less{X, d};
  for (int32_t t = 0; t < T; ++t) {
    std::vector<int32_t> c;
    tree_candidates(
        [&](const SparseVec& r) {
          return rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
        },
        all, L, minLeaf, N, perm, thr, mglo, mghi, t, c);
    std::set<int32_t, RowLess> aa(c.begin(), c.end(), less);
    std::vector<DistId> ds((size_t)N);
    const int32_t* pt = perm + (int64_t)t * N;
    for (int64_t i = 0; i < N; ++i) ds[i] = {rpo_metric_dd(d, X + (int64_t)pt[i] * d, q), pt[i]};
    std::stable_sort(ds.begin(), ds.end(),
                     [](const DistId& a, const DistId& b) { return a.dist < b.dist; });
    std::set<int32_t, RowLess> kk(less);
    for (int64_t i = 0; i < k && i < N; ++i) kk.insert(ds[i].id);
    int64_t inter = 0;
    for (int32_t id : kk) inter += aa.count(id);
    sum += (double)inter / (double)k;
  }
  return sum / (double)T;
}

// ------------------------------- streaming insert ------------------------------------------
// Conduit.hs:147-176 insertMultiC / chunkedAccum over Internal.hs:245-297 insertMulti / insert:
// the source is cut into chunks of `chunk` points (C.chunksOf: the last one may be shorter) and
// every chunk is folded into every tree with `insert`.  Unlike the batch build the topology is
// data dependent, so the tree is returned as heap arrays of 2^(L+1)-1 slots:
//   kind[h]   0 = absent, 1 = Bin, 2 = Tip
//   thr/mglo/mghi[h] for Bins, leaf_off[h]..leaf_off[h]+leaf_len[h] into leaf_ids for Tips.
// Returns, per tree, the number of points held (sum of the Tip sizes): less than N when the
// data-loss branch of `insert` was taken (see stream_insert below).
}  // extern "C"
namespace {
struct StreamNode {
  int kind = 2;  // a fresh accumulator is `Tip () mempty`
  double thr = 0, mglo = 0, mghi = 0;
  std::vector<int32_t> xs;  // Tip payload, in the reference's order
};
struct StreamTree {
  std::vector<StreamNode> nodes;  // heap order
  int64_t dropped = 0;
};

template <class Data>
bool partition_ids(const Data& D, const SparseVec& r, const std::vector<int32_t>& xs, double* thr,
                   double* mgl, double* mgr, std::vector<int32_t>& ll, std::vector<int32_t>& rr) {
  const int64_t n = (int64_t)xs.size();
  if (n < 1) return false;  // Internal.hs:492 Nothing
  std::vector<std::pair<double, int32_t>> projs((size_t)n);
  for (int64_t i = 0; i < n; ++i) projs[i] = {D.inner(r, xs[i]), xs[i]};
  std::stable_sort(projs.begin(), projs.end(),
                   [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) {
                     return a.first < b.first;
                   });
  const int64_t nh = n / 2;
  if (n >= 3) {
    *mgl = projs[nh - 1].first;
    *mgr = projs[nh + 1].first;
  } else if (n == 2) {
    *mgl = projs[0].first;
    *mgr = projs[1].first;
  } else {
    *mgl = *mgr = projs[0].first;
  }
  *thr = projs[nh].first;
  ll.resize((size_t)nh);
  rr.resize((size_t)(n - nh));
  for (int64_t i = 0; i < nh; ++i) ll[i] = projs[i].second;
  for (int64_t i = nh; i < n; ++i) rr[i - nh] = projs[i].second;
  return true;
}

int64_t subtree_points(const StreamTree& st, int64_t h) {
  if (h >= (int64_t)st.nodes.size()) return 0;
  const StreamNode& nd = st.nodes[h];
  if (nd.kind == 2) return (int64_t)nd.xs.size();
  if (nd.kind == 1) return subtree_points(st, 2 * h + 1) + subtree_points(st, 2 * h + 2);
  return 0;
}
void clear_subtree(StreamTree& st, int64_t h) {
  if (h >= (int64_t)st.nodes.size()) return;
  StreamNode& nd = st.nodes[h];
  if (nd.kind == 1) {
    clear_subtree(st, 2 * h + 1);
    clear_subtree(st, 2 * h + 2);
  }
  nd.kind = 0;
  nd.xs.clear();
}

// Internal.hs:258-297 `insert`'s loop, one chunk `xs` (consumed) into the subtree at heap h.
template <class Data>
void stream_insert(const Data& D, const std::vector<SparseVec>& rvs, int32_t maxDepth,
                   int32_t minLeaf, StreamTree& st, int32_t ixLev, int64_t h,
                   std::vector<int32_t>& xs) {
  StreamNode& nd = st.nodes[h];
  if (nd.kind == 1) {                    // :272 Bin _ thr0 margin0 tl0 tr0
    if (ixLev >= maxDepth) return;       // :273-274 (unreachable for a fixed maxDepth)
    double thr, mgl, mgr;
    std::vector<int32_t> ll, rr;
    if (!partition_ids(D, rvs[ixLev], xs, &thr, &mgl, &mgr, ll, rr)) {
      // :277 Nothing -> Tip () mempty : an EMPTY chunk half reaching a Bin REPLACES the whole
      // subtree — every point stored below it is lost.
      st.dropped += subtree_points(st, h);
      clear_subtree(st, h);
      st.nodes[h].kind = 2;
      return;
    }
    nd.mglo = nd.mglo >= mgl ? nd.mglo : mgl;  // :280 margin0 <> margin: (Max, Min), :86-87
    nd.mghi = nd.mghi <= mgr ? nd.mghi : mgr;
    nd.thr = (nd.thr + thr) / 2;               // :281
    stream_insert(D, rvs, maxDepth, minLeaf, st, ixLev + 1, 2 * h + 1, ll);  // :282
    stream_insert(D, rvs, maxDepth, minLeaf, st, ixLev + 1, 2 * h + 2, rr);  // :283
    return;
  }
  // :285 Tip _ xs0   (kind 0 = never touched = the `z` of :268)
  nd.kind = 2;
  std::vector<int32_t> xs1;  // :286 xs' = xs <> xs0 : the new chunk goes FIRST
  xs1.reserve(xs.size() + nd.xs.size());
  xs1.insert(xs1.end(), xs.begin(), xs.end());
  xs1.insert(xs1.end(), nd.xs.begin(), nd.xs.end());
  if (ixLev >= maxDepth || (int64_t)xs1.size() <= (int64_t)minLeaf) {  // :287-288
    nd.xs.swap(xs1);
    return;
  }
  double thr, mgl, mgr;
  std::vector<int32_t> ll, rr;
  partition_ids(D, rvs[ixLev], xs1, &thr, &mgl, &mgr, ll, rr);  // :290 (xs' is not empty here)
  nd.kind = 1;  // :292 Bin () thr margin tl tr
  nd.thr = thr;
  nd.mglo = mgl;
  nd.mghi = mgr;
  nd.xs.clear();
  st.nodes[2 * h + 1] = StreamNode();  // :294-295 loop (ixLev + 1) z ..
  st.nodes[2 * h + 2] = StreamNode();
  stream_insert(D, rvs, maxDepth, minLeaf, st, ixLev + 1, 2 * h + 1, ll);
  stream_insert(D, rvs, maxDepth, minLeaf, st, ixLev + 1, 2 * h + 2, rr);
}

template <class Data>
void stream_forest(const Data& D, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                   int64_t chunk, int8_t* kind, double* thr, double* mglo, double* mghi,
                   int64_t* leaf_off, int64_t* leaf_len, int32_t* leaf_ids, int64_t* held) {
  const int64_t slots = ((int64_t)1 << (L + 1)) - 1;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  std::vector<SparseVec> all = sparsify(R, T, L, D.d);
  for (int32_t t = 0; t < T; ++t) {
    std::vector<SparseVec> rvs(all.begin() + (size_t)t * L, all.begin() + (size_t)(t + 1) * L);
    StreamTree st;
    st.nodes.assign((size_t)slots, StreamNode());
    for (int64_t h = 1; h < slots; ++h) st.nodes[h].kind = 0;
    for (int64_t c0 = 0; c0 < D.N; c0 += chunk) {  // C.chunksOf n .| C.foldl f z
      std::vector<int32_t> xs;
      for (int64_t i = c0; i < D.N && i < c0 + chunk; ++i) xs.push_back((int32_t)i);
      stream_insert(D, rvs, L, minLeaf, st, 0, 0, xs);
    }
    int64_t w = 0;
    for (int64_t h = 0; h < slots; ++h) {
      const StreamNode& nd = st.nodes[h];
      const int64_t o = t * slots + h;
      kind[o] = (int8_t)nd.kind;
      thr[o] = nd.kind == 1 ? nd.thr : nan;
      mglo[o] = nd.kind == 1 ? nd.mglo : nan;
      mghi[o] = nd.kind == 1 ? nd.mghi : nan;
      leaf_off[o] = w;
      leaf_len[o] = nd.kind == 2 ? (int64_t)nd.xs.size() : 0;
      if (nd.kind == 2)
        for (int32_t id : nd.xs) leaf_ids[(int64_t)t * D.N + w++] = id;
    }
    held[t] = w;
  }
}
}  // namespace
extern "C" {

void rpo_stream_forest_dense(const double* X, int64_t N, int32_t d, const double* R, int32_t T,
                             int32_t L, int32_t minLeaf, int64_t chunk, int8_t* kind, double* thr,
                             double* mglo, double* mghi, int64_t* leaf_off, int64_t* leaf_len,
                             int32_t* leaf_ids, int64_t* held) {
  DenseData D{X, N, d};
  stream_forest(D, R, T, L, minLeaf, chunk, kind, thr, mglo, mghi, leaf_off, leaf_len, leaf_ids,
                held);
}


// the same fold over SVector rows (`forest` is polymorphic in Inner SVector v, Conduit.hs:104-113):
// every inner product is innerSS (Internal.hs:351-366)
void rpo_stream_forest_csr(const int64_t* rowptr, const int32_t* col, const double* val, int64_t N,
                           int32_t d, const double* R, int32_t T, int32_t L, int32_t minLeaf,
                           int64_t chunk, int8_t* kind, double* thr, double* mglo, double* mghi,
                           int64_t* leaf_off, int64_t* leaf_len, int32_t* leaf_ids, int64_t* held) {
  CsrData D{rowptr, col, val, N, d};
  stream_forest(D, R, T, L, minLeaf, chunk, kind, thr, mglo, mghi, leaf_off, leaf_len, leaf_ids,
                held);
}

// ---- queries on a streamed tree: RPTree.hs:289-314 `candidates` / :168-176 `knn` walk the RPT
// value whatever built it; on the heap arrays of rpo_stream_forest_dense a Tip is kind 2 with
// its payload at leaf_off / leaf_len, a Bin kind 1; kind 0 never hangs below a Bin.
int64_t rpo_stream_candidates_dense(const double* q, int32_t d, const double* R, int32_t T,
                                    int32_t L, int64_t N, const int8_t* kind, const double* thr,
                                    const double* mglo, const double* mghi, const int64_t* leaf_off,
                                    const int64_t* leaf_len, const int32_t* leaf_ids, int32_t t,
                                    int32_t* out, int64_t cap) {
  const int64_t slots = ((int64_t)1 << (L + 1)) - 1;
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<double> projq((size_t)L);
  for (int32_t l = 0; l < L; ++l) {
    const SparseVec& r = all[(size_t)t * L + l];
    projq[l] = rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
  }
  const int8_t* kd = kind + t * slots;
  const double *th = thr + t * slots, *lo = mglo + t * slots, *hi = mghi + t * slots;
  const int64_t *lo_off = leaf_off + t * slots, *ln = leaf_len + t * slots;
  std::vector<int32_t> res;
  std::function<void(int32_t, int64_t)> go = [&](int32_t level, int64_t h) {
    if (kd[h] != 1) {  // :299 Tip
      if (kd[h] == 2)
        for (int64_t i = 0; i < ln[h]; ++i) res.push_back(leaf_ids[(int64_t)t * N + lo_off[h] + i]);
      return;
    }
    const double proj = projq[level];
    const double dl = std::fabs(lo[h] - proj), dr = std::fabs(hi[h] - proj);
    if (proj < th[h] && dl > dr) {  // :309-310
      go(level + 1, 2 * h + 1);
      go(level + 1, 2 * h + 2);
    } else if (proj < th[h]) {  // :311
      go(level + 1, 2 * h + 1);
    } else if (proj > th[h] && dl < dr) {  // :312-313
      go(level + 1, 2 * h + 1);
      go(level + 1, 2 * h + 2);
    } else {  // :314
      go(level + 1, 2 * h + 2);
    }
  };
  go(0, 0);
  for (int64_t i = 0; i < (int64_t)res.size() && i < cap; ++i) out[i] = res[i];
  return (int64_t)res.size();
}

int32_t rpo_stream_knn_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                             int32_t T, int32_t L, const int8_t* kind, const double* thr,
                             const double* mglo, const double* mghi, const int64_t* leaf_off,
                             const int64_t* leaf_len, const int32_t* leaf_ids, int32_t k,
                             int32_t dedup, int32_t* out_ids, double* out_dist) {
  DenseData D{X, N, d};
  std::vector<DistId> cs;
  std::vector<int32_t> buf((size_t)(N > 0 ? N : 1));
  for (int32_t t = 0; t < T; ++t) {  // :176 fold over the IntMap, ascending key
    const int64_t m = rpo_stream_candidates_dense(q, d, R, T, L, N, kind, thr, mglo, mghi, leaf_off,
                                                  leaf_len, leaf_ids, t, buf.data(), N);
    for (int64_t i = 0; i < m; ++i) cs.push_back(DistId{D.metric(buf[i], q), buf[i]});
  }
  return topk_from(cs, k, dedup, out_ids, out_dist);
}


// knnH (RPTree.hs:199-217) over a streamed forest: candidatesH (:318-342) walks the RPT value whatever
// built it — on the heap arrays a Tip is kind 2 (its bucket at leaf_off / leaf_len, possibly empty),
// a Bin kind 1 — then the same bucket selection as the batch forests' (knn_h_select).
int64_t rpo_stream_knn_h_dense(const double* X, int64_t N, int32_t d, const double* q, const double* R,
                               int32_t T, int32_t L, const int8_t* kind, const double* thr,
                               const double* mglo, const double* mghi, const int64_t* leaf_off,
                               const int64_t* leaf_len, const int32_t* leaf_ids, int32_t k,
                               int32_t* out_ids, double* out_dist, int64_t cap) {
  const int64_t slots = ((int64_t)1 << (L + 1)) - 1;
  std::vector<SparseVec> all = sparsify(R, T, L, d);
  std::vector<LeafEntry> es;
  std::vector<double> projq((size_t)L);
  for (int32_t t = 0; t < T; ++t) {
    for (int32_t l = 0; l < L; ++l) {
      const SparseVec& r = all[(size_t)t * L + l];
      projq[l] = rpo_inner_sd((int64_t)r.idx.size(), r.idx.data(), r.val.data(), d, q);
    }
    const int8_t* kd = kind + t * slots;
    const double *th = thr + t * slots, *lo = mglo + t * slots, *hi = mghi + t * slots;
    const int64_t *lo_off = leaf_off + t * slots, *ln = leaf_len + t * slots;
    std::function<void(int32_t, int64_t, double)> go = [&](int32_t level, int64_t h, double p) {
      if (kd[h] != 1) {  // :323 Tip -> insertp p xs
        if (kd[h] == 2) es.push_back(LeafEntry{p, t, lo_off[h], ln[h]});
        return;
      }
      const double proj = projq[level];
      const double dl = std::fabs(lo[h] - proj), dr = std::fabs(hi[h] - proj);  // :330-331
      const double pl = p <= dl ? p : dl, pr = p <= dr ? p : dr;                // :332-333
      if (proj < th[h] && dl > dr) {  // :335-336
        go(level + 1, 2 * h + 1, pl);
        go(level + 1, 2 * h + 2, pr);
      } else if (proj < th[h]) {  // :337
        go(level + 1, 2 * h + 1, pl);
      } else if (proj > th[h] && dl < dr) {  // :338-339
        go(level + 1, 2 * h + 1, pl);
        go(level + 1, 2 * h + 2, pr);
      } else {  // :340
        go(level + 1, 2 * h + 2, pr);
      }
    };
    go(0, 0, std::numeric_limits<double>::infinity());  // :320
  }
  int64_t m = 0;
  for (const LeafEntry& e : knn_h_select(es, k))
    for (int64_t i = 0; i < e.n; ++i, ++m)
      if (m < cap) {
        const int32_t id = leaf_ids[(int64_t)e.tree * N + e.off + i];
        out_ids[m] = id;
        out_dist[m] = rpo_metric_dd(d, X + (int64_t)id * d, q);
      }
  return m;
}

// metricDDL2 with the squares through THIS box's libm pow (what a GHC-compiled reference calls);
// the exponent is volatile so that the compiler cannot fold the call into a multiplication.
double rpo_metric_dd_libm(int64_t n, const double* u, const double* v) {
  volatile double two = 2.0;
  double acc = 0.0;
  for (int64_t j = 0; j < n; ++j) acc = acc + std::pow(u[j] - v[j], two);
  return std::sqrt(acc);
}
// how many of n pseudo-random doubles t (SplitMix stream `seed`, exponents in [-40, 40]) have
// pow(t, 2.0) != t * t on this box's libm
int64_t rpo_pow2_mismatches(uint64_t seed, int64_t n) {
  volatile double two = 2.0;
  rpo_gen g;
  rpo_gen_init(&g, seed);
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t z = rpo_next_word64(&g);
    const uint64_t bits = (z & 0x000fffffffffffffULL) | ((uint64_t)(1023 - 40 + (z >> 52) % 81) << 52) |
                          (z & 0x8000000000000000ULL);
    double t;
    std::memcpy(&t, &bits, 8);
    bad += std::pow(t, two) != t * t;
  }
  return bad;
}

}  // extern "C"
