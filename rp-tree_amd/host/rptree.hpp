// rptree.hpp — header-only C++ host mirror of the Data.RPTree API surface for the hot path,
// over the C ABI of include/rptree_hip.h.  The reference host language is Haskell (no GHC in
// the build image), so this mirror keeps the reference's names, argument order and error
// behaviour in C++ (the Python mirror is rp-tree_amd/python/rptree_amd).
//
//   forestBatch / treeBatch   Batch.hs:29-63          knn          RPTree.hs:168-176
//   candidates                RPTree.hs:289-314        recallWith   RPTree.hs:259-282
//   knnH / knnPQ              RPTree.hs:181-217,318-342
//   rpTreeCfg / RPTreeConfig  Conduit.hs:123-141       SVector/DVector/Embed  Internal.hs:56-133
//   sparse / stdNormal / sample: host-side hyperplane sampling, Batch.hs:59-61, Gen.hs:148-195
//   (SplitMix64 + Box-Muller restated from the published algorithm: self-consistent, not
//   verified against Hackage's splitmix-distributions — a Haskell host keeps its own generator).
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rptree_hip.h"

namespace rptree {

struct RPTError : std::runtime_error {  // next to Internal.hs:66-72
  int code;
  RPTError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int32_t s) {
  if (s != RPT_OK) throw RPTError(s, rpt_last_error());
}

// ---- vector types, Internal.hs:92-133 ----
struct SVector {
  int svDim;
  std::vector<std::pair<int, double>> svVec;  // ascending indices (unchecked in the reference)
};
struct DVector {
  std::vector<double> dvVec;
};
inline SVector fromListSv(int n, std::vector<std::pair<int, double>> ll) { return {n, std::move(ll)}; }
inline DVector fromListDv(std::vector<double> ll) { return {std::move(ll)}; }
template <class V, class X>
struct Embed {
  V eEmbed;
  X eData;
};

// Inner SVector DVector / metricL2 with the reference's summation order (Internal.hs:369-406)
inline double inner(const SVector& u, const DVector& v) {
  double acc = 0.0;
  size_t m = u.svVec.size() < v.dvVec.size() ? u.svVec.size() : v.dvVec.size();
  for (size_t j = m; j-- > 0;) acc = u.svVec[j].second * v.dvVec[(size_t)u.svVec[j].first] + acc;
  return acc;
}
inline double metricL2(const DVector& u, const DVector& v) {
  double acc = 0.0;
  for (size_t j = 0; j < u.dvVec.size() && j < v.dvVec.size(); ++j)
    acc = acc + std::pow(u.dvVec[j] - v.dvVec[j], 2.0);
  return std::sqrt(acc);
}

// ---- parameters, Conduit.hs:123-141 ----
struct RPTreeConfig {
  int fpMaxTreeDepth;
  int64_t fpDataChunkSize;
  double fpProjNzDensity;
};
inline RPTreeConfig rpTreeCfg(int minl, int64_t n, int d) {
  const double maxd = std::ceil(std::log((double)n / (double)minl) / std::log(2.0));
  const double pnzMin = 1.0 / (std::log((double)d) / std::log(10.0));
  return {(int)maxd, (int64_t)std::ceil((double)n / 100.0), pnzMin < 1.0 ? pnzMin : 1.0};
}

// ---- host-side random generation (stays on the host), Gen.hs:148-195 ----
class SMGen {
  uint64_t seed_, gamma_;
  static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL;
    return z ^ (z >> 33);
  }
  static uint64_t mixGamma(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = (z ^ (z >> 31)) | 1ULL;
    return __builtin_popcountll(z ^ (z >> 1)) >= 24 ? z : z ^ 0xaaaaaaaaaaaaaaaaULL;
  }

 public:
  explicit SMGen(uint64_t s) : seed_(mix64(s)), gamma_(mixGamma(s + 0x9e3779b97f4a7c15ULL)) {}
  uint64_t nextWord64() {
    seed_ += gamma_;
    return mix64(seed_);
  }
  double nextDouble() { return (double)(nextWord64() >> 11) * 0x1.0p-53; }
  bool bernoulli(double p) { return nextDouble() < p; }
  double normal(double mu, double sig) {
    const double u1 = nextDouble(), u2 = nextDouble();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2) * sig + mu;
  }
  double uniformR(double lo, double hi) { return nextDouble() * (hi - lo) + lo; }
};
// `sparse pnz dim stdNormal`
inline SVector sparse(SMGen& g, double pnz, int dim) {
  SVector v{dim, {}};
  for (int i = 0; i < dim; ++i)
    if (g.bernoulli(pnz)) v.svVec.push_back({i, g.normal(0.0, 1.0)});
  return v;
}

// ---- handles ----
class Context {
  rpt_ctx* h_ = nullptr;

 public:
  explicit Context(int device = 0) { check(rpt_ctx_create(device, &h_)); }
  ~Context() { rpt_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  rpt_ctx* get() const { return h_; }
};

class Dataset {
  rpt_dataset* h_ = nullptr;

 public:
  int64_t n = 0;
  int d = 0;
  Dataset(Context& ctx, const std::vector<DVector>& xs) {
    n = (int64_t)xs.size();
    d = n ? (int)xs[0].dvVec.size() : 1;
    std::vector<double> flat((size_t)n * d);
    for (int64_t i = 0; i < n; ++i)
      for (int j = 0; j < d; ++j) flat[(size_t)i * d + j] = xs[(size_t)i].dvVec[(size_t)j];
    check(rpt_dataset_dense_host(ctx.get(), flat.data(), n, d, RPT_F64, &h_));
  }
  ~Dataset() { rpt_dataset_free(h_); }
  Dataset(const Dataset&) = delete;
  Dataset& operator=(const Dataset&) = delete;
  rpt_dataset* get() const { return h_; }
};

// `RPForest d a` (Internal.hs:182): trees keyed 0..T-1, held flat in HBM.
class RPForest {
  rpt_forest* h_ = nullptr;

 public:
  Context* ctx;
  const Dataset* data;
  std::vector<std::vector<SVector>> rpVectors;  // _rpVectors of every tree (one per level)
  int T, L, minLeaf;
  bool streamed = false;  // built by the chunk fold of `forest` (explicit topology)
  // chunk == 0: the batch build (createMulti); chunk > 0: the streaming build, `insert` folded over
  // chunks of `chunk` points (Conduit.hs:147-176 over Internal.hs:245-297)
  RPForest(Context& c, const Dataset& ds, std::vector<std::vector<SVector>> rvss, int maxd,
           int minl, int64_t chunk = 0)
      : ctx(&c), data(&ds), rpVectors(std::move(rvss)), T((int)rpVectors.size()), L(maxd),
        minLeaf(minl) {
    std::vector<double> R((size_t)T * L * ds.d, 0.0);  // dense-ified [T][L][d]
    for (int t = 0; t < T; ++t)
      for (int l = 0; l < L; ++l)
        for (auto& iv : rpVectors[(size_t)t][(size_t)l].svVec)
          R[((size_t)t * L + l) * ds.d + (size_t)iv.first] = iv.second;
    if (chunk > 0) {
      check(rpt_forest_stream_build(c.get(), ds.get(), R.data(), T, L, minl, chunk, RPT_PROJ_AUTO, &h_));
      streamed = true;
    } else {
      check(rpt_forest_build(c.get(), ds.get(), R.data(), T, L, minl, RPT_PROJ_AUTO, &h_));
    }
  }
  ~RPForest() { rpt_forest_free(h_); }
  RPForest(const RPForest&) = delete;
  RPForest& operator=(const RPForest&) = delete;
  rpt_forest* get() const { return h_; }
  std::vector<int32_t> perm() const {
    std::vector<int32_t> p((size_t)T * data->n);
    check(rpt_forest_get_perm(h_, p.data()));
    return p;
  }
  // treeSize (RPTree.hs:362-363): sum of the leaf sizes of tree t
  int64_t treeSize(int) const {
    if (streamed) {  // points held by the Tips (less than n after the data-loss branch, :277)
      int64_t held = 0;
      check(rpt_forest_get_topology(h_, nullptr, nullptr, nullptr, nullptr, &held, nullptr));
      return held;
    }
    int64_t cnt = 0, nrec = 0;
    check(rpt_topology(data->n, L, minLeaf, nullptr, 0, &nrec));
    std::vector<int64_t> rec((size_t)nrec * 5);
    check(rpt_topology(data->n, L, minLeaf, rec.data(), nrec, &nrec));
    for (int64_t i = 0; i < nrec; ++i)
      if (rec[(size_t)i * 5 + 4]) cnt += rec[(size_t)i * 5 + 3];
    return cnt;
  }
};

// forestBatch :: Word64 -> Int -> Int -> Int -> Double -> Int -> data -> RPForest  (Batch.hs:48-63)
inline RPForest forestBatch(Context& ctx, uint64_t seed, int maxd, int minl, int ntrees, double pnz,
                            int dim, const Dataset& src) {
  if (dim != src.d) throw RPTError(RPT_E_ARG, "projection vector dimension != data dimension");
  SMGen g(seed);  // `sample seed`: one generator, trees outermost, levels inner (Batch.hs:59-61)
  std::vector<std::vector<SVector>> rvss((size_t)ntrees);
  for (int t = 0; t < ntrees; ++t)
    for (int l = 0; l < maxd; ++l) rvss[(size_t)t].push_back(sparse(g, pnz, dim));
  return RPForest(ctx, src, std::move(rvss), maxd, minl);
}
inline RPForest treeBatch(Context& ctx, uint64_t seed, int maxDepth, int minLeaf, double pnz, int dim,
                          const Dataset& src) {
  return forestBatch(ctx, seed, maxDepth, minLeaf, 1, pnz, dim, src);
}
// forest :: Word64 -> Int -> Int -> Int -> Int -> Double -> Int -> source -> RPForest
// (Conduit.hs:104-121): seed, max depth, min leaf, trees, CHUNK SIZE, density, dimension, data —
// the reference's streaming semantics (same hyperplane draw order, Conduit.hs:116-118)
inline RPForest forest(Context& ctx, uint64_t seed, int maxd, int minl, int ntrees, int64_t chunksize,
                       double pnz, int dim, const Dataset& src) {
  if (dim != src.d) throw RPTError(RPT_E_ARG, "projection vector dimension != data dimension");
  if (chunksize < 1) throw RPTError(RPT_E_ARG, "chunk size must be >= 1");
  SMGen g(seed);
  std::vector<std::vector<SVector>> rvss((size_t)ntrees);
  for (int t = 0; t < ntrees; ++t)
    for (int l = 0; l < maxd; ++l) rvss[(size_t)t].push_back(sparse(g, pnz, dim));
  return RPForest(ctx, src, std::move(rvss), maxd, minl, chunksize);
}

// knn metricL2 k forest q  (RPTree.hs:168-176): (distance, point id), duplicates kept
inline std::vector<std::pair<double, int32_t>> knn(const RPForest& tts, int k, const DVector& q) {
  std::vector<DVector> qv{q};
  Dataset qs(*tts.ctx, qv);
  std::vector<int32_t> ids((size_t)k);
  std::vector<double> dist((size_t)k);
  int32_t cnt = 0;
  check(rpt_knn_host(tts.ctx->get(), tts.get(), tts.data->get(), qs.get(), k,
                     RPT_KNN_KEEP_DUPLICATES, ids.data(), dist.data(), &cnt));
  std::vector<std::pair<double, int32_t>> out;
  for (int i = 0; i < cnt; ++i) out.push_back({dist[(size_t)i], ids[(size_t)i]});
  return out;
}

// knnPQ metricL2 k forest q  (RPTree.hs:181-194): like knn, one entry per distance (`nub`)
inline std::vector<std::pair<double, int32_t>> knnPQ(const RPForest& tts, int k, const DVector& q) {
  std::vector<DVector> qv{q};
  Dataset qs(*tts.ctx, qv);
  std::vector<int32_t> ids((size_t)k);
  std::vector<double> dist((size_t)k);
  int32_t cnt = 0;
  check(rpt_knn_host(tts.ctx->get(), tts.get(), tts.data->get(), qs.get(), k,
                     RPT_KNN_DEDUP_DISTANCE, ids.data(), dist.data(), &cnt));
  std::vector<std::pair<double, int32_t>> out;
  for (int i = 0; i < cnt; ++i) out.push_back({dist[(size_t)i], ids[(size_t)i]});
  return out;
}

// knnH metricL2 k forest q  (RPTree.hs:199-217): whole buckets of the leaves with the smallest
// margin priority, NOT sorted by distance and NOT cut to k (as the reference)
inline std::vector<std::pair<double, int32_t>> knnH(const RPForest& tts, int k, const DVector& q) {
  std::vector<DVector> qv{q};
  Dataset qs(*tts.ctx, qv);
  int64_t off[2] = {0, 0}, total = 0;
  check(rpt_knnh_host(tts.ctx->get(), tts.get(), tts.data->get(), qs.get(), k, off, nullptr,
                      nullptr, 0, &total));
  std::vector<int32_t> ids((size_t)(total > 0 ? total : 1));
  std::vector<double> dist((size_t)(total > 0 ? total : 1));
  check(rpt_knnh_host(tts.ctx->get(), tts.get(), tts.data->get(), qs.get(), k, off, ids.data(),
                      dist.data(), total, &total));
  std::vector<std::pair<double, int32_t>> out;
  for (int64_t i = 0; i < total; ++i) out.push_back({dist[(size_t)i], ids[(size_t)i]});
  return out;
}

// candidates tree q (RPTree.hs:289-314): ids of the leaf buckets reached in tree t
inline std::vector<int32_t> candidates(const RPForest& tts, int t, const DVector& q) {
  std::vector<DVector> qv{q};
  Dataset qs(*tts.ctx, qv);
  std::vector<int64_t> off((size_t)tts.T + 1);
  int64_t total = 0;
  check(rpt_candidates(tts.ctx->get(), tts.get(), qs.get(), off.data(), nullptr, 0, &total));
  std::vector<int32_t> ids((size_t)(total > 0 ? total : 1));
  check(rpt_candidates(tts.ctx->get(), tts.get(), qs.get(), off.data(), ids.data(), total, &total));
  return std::vector<int32_t>(ids.begin() + off[(size_t)t], ids.begin() + off[(size_t)t + 1]);
}

}  // namespace rptree
