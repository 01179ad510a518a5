// standalone check of wave_bitonic<double, NR> (debugging aid)
#include "../rp-tree_amd/csrc/split.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace rpt;

template <int NR>
__global__ void tk(const double* keys, const int* ids, int n, int* out_ids, double* out_keys) {
  const int lane = threadIdx.x & 63;
  Keys<double> K{nullptr, 0, 0, nullptr};
  double k[NR];
  int id[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = r * 64 + lane;
    id[r] = i < n ? ids[i] : kPad;
    k[r] = i < n ? keys[i] : pos_inf<double>();
  }
  wave_bitonic<double, NR>(k, id, K);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    out_ids[r * 64 + lane] = id[r];
    out_keys[r * 64 + lane] = k[r];
  }
}

template <int NR>
int run(int n, unsigned seed) {
  std::mt19937 g(seed);
  std::vector<double> keys(n);
  std::vector<int> ids(n);
  for (int i = 0; i < n; ++i) {
    keys[i] = (double)(g() % 7);  // heavy ties
    ids[i] = i;
  }
  std::shuffle(ids.begin(), ids.end(), g);
  double *dk, *ok;
  int *di, *oi;
  hipMalloc(&dk, n * 8); hipMalloc(&di, n * 4); hipMalloc(&ok, NR * 64 * 8); hipMalloc(&oi, NR * 64 * 4);
  hipMemcpy(dk, keys.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(di, ids.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tk<NR>, dim3(1), dim3(64), 0, 0, dk, di, n, oi, ok);
  std::vector<int> hi(NR * 64);
  std::vector<double> hk(NR * 64);
  hipMemcpy(hi.data(), oi, NR * 64 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hk.data(), ok, NR * 64 * 8, hipMemcpyDeviceToHost);
  std::vector<std::pair<double, int>> want(n);
  for (int i = 0; i < n; ++i) want[i] = {keys[i], ids[i]};
  std::sort(want.begin(), want.end());
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += !(hi[i] == want[i].second && hk[i] == want[i].first);
  for (int i = n; i < NR * 64; ++i) bad += hi[i] != kPad;
  printf("NR=%d n=%d bad=%d\n", NR, n, bad);
  if (bad) for (int i = 0; i < n; ++i) if (!(hi[i] == want[i].second && hk[i] == want[i].first)) printf("  %d: got (%g,%d) want (%g,%d)\n", i, hk[i], hi[i], want[i].first, want[i].second);
  return bad;
}

// lexicographic tie-break through earlier levels: keys of level 2 all equal, level 1 heavy
// ties, level 0 distinct-ish
template <int NR>
__global__ void tk_lex(const double* P, int N, const int* ids, int n, int* out_ids) {
  const int lane = threadIdx.x & 63;
  Keys<double> K{P, N, 2, nullptr};
  double k[NR];
  int id[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = r * 64 + lane;
    id[r] = i < n ? ids[i] : kPad;
    k[r] = i < n ? K.key(id[r]) : pos_inf<double>();
  }
  wave_bitonic<double, NR>(k, id, K);
#pragma unroll
  for (int r = 0; r < NR; ++r) out_ids[r * 64 + lane] = id[r];
}

template <int NR>
int run_lex(int n, unsigned seed) {
  std::mt19937 g(seed);
  const int N = 1000;
  std::vector<double> P(3 * N);
  for (int i = 0; i < N; ++i) {
    P[2 * N + i] = 0.0;
    P[1 * N + i] = (double)(g() % 3);
    P[0 * N + i] = (double)(g() % 5);
  }
  std::vector<int> ids(N);
  for (int i = 0; i < N; ++i) ids[i] = i;
  std::shuffle(ids.begin(), ids.end(), g);
  ids.resize(n);
  double* dP;
  int *di, *oi;
  hipMalloc(&dP, 3 * N * 8); hipMalloc(&di, n * 4); hipMalloc(&oi, NR * 64 * 4);
  hipMemcpy(dP, P.data(), 3 * N * 8, hipMemcpyHostToDevice);
  hipMemcpy(di, ids.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tk_lex<NR>, dim3(1), dim3(64), 0, 0, dP, N, di, n, oi);
  std::vector<int> hi(NR * 64);
  hipMemcpy(hi.data(), oi, NR * 64 * 4, hipMemcpyDeviceToHost);
  std::vector<int> want(ids);
  std::sort(want.begin(), want.end(), [&](int a, int b) {
    if (P[N + a] != P[N + b]) return P[N + a] < P[N + b];
    if (P[a] != P[b]) return P[a] < P[b];
    return a < b;
  });
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += hi[i] != want[i];
  for (int i = n; i < NR * 64; ++i) bad += hi[i] != kPad;
  printf("LEX NR=%d n=%d bad=%d\n", NR, n, bad);
  return bad;
}

int main() {
  int badl = run_lex<1>(50, 1) + run_lex<2>(100, 2) + run_lex<4>(175, 3) + run_lex<4>(256, 4) + run_lex<4>(130, 5);

  int bad = 0;
  bad += run<1>(64, 1); bad += run<1>(37, 2); bad += run<2>(128, 3); bad += run<2>(100, 4);
  bad += run<4>(256, 5); bad += run<4>(200, 6); bad += run<4>(129, 7);
  return (bad + badl) != 0;
}
