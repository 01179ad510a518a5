// The reference's own integration test (test/Data/RPTreeSpec.hs:50-85) through the C++ host
// mirror: 10 000 points on two unit discs, 10 trees, minLeaf 20, k = 5, query (0,0):
// every tree holds every point; the kNN distances are < 1.
#include <cstdio>

#include "rptree.hpp"
using namespace rptree;

int main() {
  const int n = 10000, ntrees = 10, minLeaf = 20, k = 5, dim = 2;
  SMGen g(42);
  std::vector<DVector> xs;
  for (int i = 0; i < n; ++i) {  // circle2d2, RPTreeSpec.hs:112-120
    const bool b = g.bernoulli(0.5);
    double x, y;
    do {
      x = g.uniformR(-1, 1);
      y = g.uniformR(-1, 1);
    } while (!(x * x + y * y <= 1.0));
    xs.push_back(b ? fromListDv({x, y}) : fromListDv({2.0 + x, 3.0 + y}));
  }
  try {
    Context ctx(0);
    Dataset dats(ctx, xs);
    const RPTreeConfig cfg = rpTreeCfg(minLeaf, n, dim);
    RPForest tts = forestBatch(ctx, 42, cfg.fpMaxTreeDepth, minLeaf, ntrees, 1.0, dim, dats);
    for (int t = 0; t < ntrees; ++t)
      if (tts.treeSize(t) != n) return std::printf("FAIL treeSize\n"), 1;
    auto hits = knn(tts, k, fromListDv({0, 0}));
    double mx = 0;
    for (auto& h : hits) mx = h.first > mx ? h.first : mx;
    if ((int)hits.size() != k || !(mx < 1.0)) return std::printf("FAIL knn max dist %g\n", mx), 1;
    // distances agree with the host-side metricL2 (Internal.hs:403-406)
    for (auto& h : hits)
      if (std::fabs(metricL2(xs[(size_t)h.second], fromListDv({0, 0})) - h.first) > 1e-12)
        return std::printf("FAIL metric\n"), 1;
    std::printf("ok: %d trees hold all %d points; knn max distance %.4f < 1\n", ntrees, n, mx);
    // the streaming build, RPTreeSpec.hs:87-106: `forest` with rpTreeCfg's chunk size (n / 100)
    RPForest tts2 = forest(ctx, 42, cfg.fpMaxTreeDepth, minLeaf, ntrees, cfg.fpDataChunkSize, 1.0, dim, dats);
    for (int t = 0; t < ntrees; ++t)
      if (tts2.treeSize(t) != n) return std::printf("FAIL streaming treeSize %lld\n", (long long)tts2.treeSize(t)), 1;
    auto hits2 = knn(tts2, k, fromListDv({0, 0}));
    double mx2 = 0;
    for (auto& h : hits2) mx2 = h.first > mx2 ? h.first : mx2;
    if ((int)hits2.size() != k || !(mx2 < 1.0)) return std::printf("FAIL streaming knn max dist %g\n", mx2), 1;
    std::printf("ok: streaming forest (chunks of %lld) holds all points; knn max distance %.4f < 1\n",
                (long long)cfg.fpDataChunkSize, mx2);
  } catch (const RPTError& e) {
    std::printf("RPTError %d: %s\n", e.code, e.what());
    return 2;
  }
  return 0;
}
