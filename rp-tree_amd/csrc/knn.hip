// knn.hip — query side of the hot path.
//
//   candidates (RPTree.hs:289-314): per tree a DFS from the root with the query's projection
//       on the level's hyperplane; the 4-way rule may descend both children; the leaf buckets
//       reached are concatenated left to right.
//   knn (RPTree.hs:168-176): concatenate the candidates of all trees in ascending tree key
//       (duplicates kept), distance of every candidate to the query (metricL2), stable sort
//       by distance, take k.  Equivalent total order: (distance, candidate position).
//
// Kernels (wave64, one workgroup per query):
//   count_kernel  : thread = tree; counts the leaf ranges / candidates each tree contributes
//   ranges_kernel : same traversal, writes (perm offset, length, candidate position) ranges
//   topk_kernel   : each wave streams candidate rows (whole-row coalesced gathers from HBM —
//                   this is the bandwidth-bound part, d*sizeof(x) bytes per candidate),
//                   wave-reduces the squared distance, and the block keeps the best k in LDS
//                   by bitonic-merging batches of candidates
//   expand_kernel : materialises candidate ids for rpt_candidates
#include <hip/hip_bf16.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#include "common.h"

namespace rpt {
namespace {

constexpr int kBuf = 2048;  // LDS entries of the top-k merge buffer (best k + batch)

struct Range {
  int64_t poff;  // absolute offset into perm ([T][N] flattened)
  int32_t n;
  int32_t pos;   // position of the range's first candidate in the query's candidate list
};

// ---- traversal ---------------------------------------------------------------------------
// emit(off, n) is called for every leaf reached, left to right.  Depth-first, left child first;
// the right children still to visit lie on the current path at distinct levels, so a 32-bit
// mask of those levels plus the current node's heap index is the whole stack: a pop takes the
// deepest pending level l, the ancestor of the current node at that level, its right child,
// and replays the l+1 halvings along that child's path for (off, n).  (Explicit stacks of
// (level, heap, off, n) cost 128 registers and held the fused kernels to 2 waves per SIMD.)
template <class TK, class Emit>
__device__ inline void traverse(const double* __restrict__ thr, const double* __restrict__ mglo,
                                const double* __restrict__ mghi, const TK* __restrict__ pq,
                                int64_t pq_stride, int L, int min_leaf, int64_t N, Emit emit) {
  unsigned int pending = 0, heap = 0;
  int level = 0, off = 0, n = (int)N;
  // Two levels per memory round trip (round 3): the walk is a chain of dependent node loads — 13 at
  // C2, a fifth of a query's life in the fused kernels — so a node's record is fetched together
  // with the records of BOTH its children (adjacent heap slots), and the child the decision picks
  // is decided on at once.  `last` clamps the children of the deepest Bin level into the arrays
  // (their records are never used: the level test comes first).
  const unsigned int last = L > 0 ? (1u << L) - 2u : 0u;
  auto step = [&](double th, double lo, double hi, double proj) {  // RPTree.hs:303-314
    const double dl = fabs(lo - proj);  // :306
    const double dr = fabs(hi - proj);  // :307
    const int nh = n / 2;
    const bool both = (proj < th && dl > dr) || (proj > th && dl < dr);  // :309-313
    if (both) {  // the right child waits, continue left
      pending |= 1u << level;
      heap = 2 * heap + 1;
      n = nh;
    } else if (proj < th) {  // :311
      heap = 2 * heap + 1;
      n = nh;
    } else {  // :314 (includes proj == thr)
      heap = 2 * heap + 2;
      off += nh;
      n = n - nh;
    }
    ++level;
  };
  for (;;) {
    for (;;) {
      if (level >= L || n <= min_leaf) {  // Tip (RPTree.hs:299)
        emit(off, n);
        break;
      }
      const unsigned int c0 = 2 * heap + 1 < last ? 2 * heap + 1 : last;
      const unsigned int c1 = 2 * heap + 2 < last ? 2 * heap + 2 : last;
      const int l1 = level + 1 < L ? level + 1 : level;
      const double proj = (double)pq[(int64_t)level * pq_stride];
      const double proj1 = (double)pq[(int64_t)l1 * pq_stride];
      const double th = thr[heap], lo = mglo[heap], hi = mghi[heap];
      const double th0 = thr[c0], lo0 = mglo[c0], hi0 = mghi[c0];
      const double th1 = thr[c1], lo1 = mglo[c1], hi1 = mghi[c1];
      step(th, lo, hi, proj);
      if (level >= L || n <= min_leaf) {
        emit(off, n);
        break;
      }
      const bool right = (heap & 1u) == 0u;  // heap = 2 h + 2
      step(right ? th1 : th0, right ? lo1 : lo0, right ? hi1 : hi0, proj1);
    }
    if (!pending) break;
    const int l = 31 - __clz((int)pending);
    pending &= ~(1u << l);
    heap = 2 * (((heap + 1) >> (level - l)) - 1) + 2;
    const unsigned int path = heap + 1;  // 1, then the l+1 decisions from the root down
    off = 0;
    n = (int)N;
    for (int i = l; i >= 0; --i) {
      const int nh = n / 2;
      if ((path >> i) & 1u) {
        off += nh;
        n -= nh;
      } else {
        n = nh;
      }
    }
    level = l + 1;
  }
}

// The same walk over an EXPLICIT topology (forests of the streaming build, common.h rpt_forest:
// kind 0 absent / 1 Bin / 2 Tip per heap slot, a Tip's range = (xoff, xlen)); same stackless
// scheme, nothing to replay on a pop.
template <class TK, class Emit>
__device__ inline void traverse_x(const double* __restrict__ thr, const double* __restrict__ mglo,
                                  const double* __restrict__ mghi, const int8_t* __restrict__ kind,
                                  const int64_t* __restrict__ xoff, const int64_t* __restrict__ xlen,
                                  const TK* __restrict__ pq, int64_t pq_stride, Emit emit) {
  unsigned int pending = 0, heap = 0;
  int level = 0;
  for (;;) {
    for (;;) {
      if (kind[heap] != 1) {  // Tip (RPTree.hs:299); an absent slot holds nothing
        if (kind[heap] == 2) emit((int)xoff[heap], (int)xlen[heap]);
        break;
      }
      const double proj = (double)pq[(int64_t)level * pq_stride];  // RPTree.hs:303-304
      const double th = thr[heap];
      const double dl = fabs(mglo[heap] - proj);
      const double dr = fabs(mghi[heap] - proj);
      const bool both = (proj < th && dl > dr) || (proj > th && dl < dr);
      if (both) {
        pending |= 1u << level;
        heap = 2 * heap + 1;
      } else if (proj < th) {
        heap = 2 * heap + 1;
      } else {
        heap = 2 * heap + 2;
      }
      ++level;
    }
    if (!pending) break;
    const int l = 31 - __clz((int)pending);
    pending &= ~(1u << l);
    heap = 2 * (((heap + 1) >> (level - l)) - 1) + 2;
    level = l + 1;
  }
}

template <class TK>
__global__ void count_x_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                               const double* __restrict__ mghi, int64_t nodes,
                               const int8_t* __restrict__ kind, const int64_t* __restrict__ xoff,
                               const int64_t* __restrict__ xlen, const TK* Pq, int64_t nq, int T, int L,
                               int* __restrict__ cnt_cand, int* __restrict__ cnt_rng) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int nc = 0, nr = 0;
    traverse_x<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, kind, xoff, xlen,
                   Pq + (int64_t)t * L * nq + q, nq, [&](int, int n) {
                     nc += n;
                     ++nr;
                   });
    cnt_cand[q * T + t] = nc;
    cnt_rng[q * T + t] = nr;
  }
}

template <class TK>
__global__ void ranges_x_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                                const double* __restrict__ mghi, int64_t nodes,
                                const int8_t* __restrict__ kind, const int64_t* __restrict__ xoff,
                                const int64_t* __restrict__ xlen, const TK* Pq, int64_t nq, int T, int L,
                                int64_t N, const int64_t* __restrict__ cand_off,
                                const int64_t* __restrict__ rng_off, Range* __restrict__ ranges) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int64_t r = rng_off[q * T + t];
    int64_t pos = cand_off[q * T + t] - cand_off[q * T];
    traverse_x<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, kind, xoff, xlen,
                   Pq + (int64_t)t * L * nq + q, nq, [&](int off, int n) {
                     ranges[r++] = Range{(int64_t)t * N + off, n, (int)pos};
                     pos += n;
                   });
  }
}

// per (query, tree) counts.  grid = nq blocks, blockDim >= T (multiple of 64).
template <class TK>
__global__ void count_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                             const double* __restrict__ mghi, int64_t nodes, const TK* Pq,
                             int64_t nq, int T, int L, int min_leaf, int64_t N,
                             int* __restrict__ cnt_cand /*[nq][T]*/,
                             int* __restrict__ cnt_rng /*[nq][T]*/) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int nc = 0, nr = 0;
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                 nq, L, min_leaf, N, [&](int, int n) {
                   nc += n;
                   ++nr;
                 });
    cnt_cand[q * T + t] = nc;
    cnt_rng[q * T + t] = nr;
  }
}

// exclusive scan of int32 counts into int64 offsets (single block; n is small: nq*T).
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ in, int64_t n,
                                                    int64_t* __restrict__ out /*[n+1]*/) {
  __shared__ long long part[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const long long v = i < n ? in[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      long long add = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n) out[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}

template <class TK>
__global__ void ranges_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                              const double* __restrict__ mghi, int64_t nodes, const TK* Pq,
                              int64_t nq, int T, int L, int min_leaf, int64_t N,
                              const int64_t* __restrict__ cand_off /*[nq*T+1]*/,
                              const int64_t* __restrict__ rng_off /*[nq*T+1]*/,
                              Range* __restrict__ ranges) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int64_t r = rng_off[q * T + t];
    int64_t pos = cand_off[q * T + t] - cand_off[q * T];
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                 nq, L, min_leaf, N, [&](int off, int n) {
                   ranges[r++] = Range{(int64_t)t * N + off, n, (int)pos};
                   pos += n;
                 });
  }
}

// candidatesH (RPTree.hs:318-342): the same descent, every leaf reached carries the smallest
// margin distance met on the way down (`p min dl` towards the left child, `p min dr` towards
// the right one, +infinity at the root).  emit(off, n, priority), left to right.
template <class TK, class Emit>
__device__ inline void traverse_h(const double* __restrict__ thr, const double* __restrict__ mglo,
                                  const double* __restrict__ mghi, const TK* __restrict__ pq,
                                  int64_t pq_stride, int L, int min_leaf, int64_t N, Emit emit) {
  int s_level[32];
  unsigned int s_heap[32];
  int s_off[32], s_n[32];
  double s_p[32];
  int sp = 1;
  s_level[0] = 0;
  s_heap[0] = 0;
  s_off[0] = 0;
  s_n[0] = (int)N;
  s_p[0] = __longlong_as_double(0x7ff0000000000000LL);  // :320 infty = 1 / 0
  while (sp > 0) {
    --sp;
    int level = s_level[sp];
    unsigned int heap = s_heap[sp];
    int off = s_off[sp], n = s_n[sp];
    double p = s_p[sp];
    for (;;) {
      if (level >= L || n <= min_leaf) {  // :323 Tip
        emit(off, n, p);
        break;
      }
      const double proj = (double)pq[(int64_t)level * pq_stride];
      const double th = thr[heap];
      const double dl = fabs(mglo[heap] - proj);  // :330
      const double dr = fabs(mghi[heap] - proj);  // :331
      const double pl = p <= dl ? p : dl;         // :332
      const double pr = p <= dr ? p : dr;         // :333
      const int nh = n / 2;
      const bool both = (proj < th && dl > dr) || (proj > th && dl < dr);  // :335-339
      const bool left = proj < th;
      if (both) {  // push right, continue left
        s_level[sp] = level + 1;
        s_heap[sp] = 2 * heap + 2;
        s_off[sp] = off + nh;
        s_n[sp] = n - nh;
        s_p[sp] = pr;
        ++sp;
        heap = 2 * heap + 1;
        n = nh;
        p = pl;
      } else if (left) {
        heap = 2 * heap + 1;
        n = nh;
        p = pl;
      } else {
        heap = 2 * heap + 2;
        off += nh;
        n = n - nh;
        p = pr;
      }
      ++level;
    }
  }
}

// priorities of the leaf ranges, in the order ranges_kernel lists them
template <class TK>
__global__ void ranges_h_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                                const double* __restrict__ mghi, int64_t nodes, const TK* Pq,
                                int64_t nq, int T, int L, int min_leaf, int64_t N,
                                const int64_t* __restrict__ rng_off /*[nq*T+1]*/,
                                double* __restrict__ prio) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int64_t r = rng_off[q * T + t];
    traverse_h<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes,
                   Pq + (int64_t)t * L * nq + q, nq, L, min_leaf, N,
                   [&](int, int, double p) { prio[r++] = p; });
  }
}

// ... and over an EXPLICIT topology (streamed forests): candidatesH walks the RPT value whatever built
// it (RPTree.hs:318-342); a Tip's bucket may be empty, its entry is inserted all the same (:323)
template <class TK>
__global__ void ranges_hx_kernel(const double* __restrict__ thr, const double* __restrict__ mglo,
                                 const double* __restrict__ mghi, int64_t nodes,
                                 const int8_t* __restrict__ kind, const TK* Pq, int64_t nq, int T, int L,
                                 const int64_t* __restrict__ rng_off /*[nq*T+1]*/, double* __restrict__ prio) {
  const int64_t q = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int64_t r = rng_off[q * T + t];
    const double *th = thr + t * nodes, *lo = mglo + t * nodes, *hi = mghi + t * nodes;
    const TK* pq = Pq + (int64_t)t * L * nq + q;
    int s_level[32];
    unsigned int s_heap[32];
    double s_p[32];
    int sp = 1;
    s_level[0] = 0;
    s_heap[0] = 0;
    s_p[0] = __longlong_as_double(0x7ff0000000000000LL);  // :320 infty = 1 / 0
    while (sp > 0) {
      --sp;
      int level = s_level[sp];
      unsigned int heap = s_heap[sp];
      double p = s_p[sp];
      for (;;) {
        if (kind[heap] != 1) {  // :323 Tip (an absent slot holds nothing: the same test as traverse_x)
          if (kind[heap] == 2) prio[r++] = p;
          break;
        }
        const double proj = (double)pq[(int64_t)level * nq];
        const double dl = fabs(lo[heap] - proj), dr = fabs(hi[heap] - proj);  // :330-331
        const double pl = p <= dl ? p : dl, pr = p <= dr ? p : dr;            // :332-333
        const bool both = (proj < th[heap] && dl > dr) || (proj > th[heap] && dl < dr);  // :335-339
        if (both) {  // push right, continue left
          s_level[sp] = level + 1;
          s_heap[sp] = 2 * heap + 2;
          s_p[sp] = pr;
          ++sp;
          heap = 2 * heap + 1;
          p = pl;
        } else if (proj < th[heap]) {
          heap = 2 * heap + 1;
          p = pl;
        } else {
          heap = 2 * heap + 2;
          p = pr;
        }
        ++level;
      }
    }
  }
}

// candidate ids for rpt_candidates: one block per (query, tree) range list
__global__ void expand_kernel(const int32_t* __restrict__ perm, const Range* __restrict__ ranges,
                              const int64_t* __restrict__ rng_off, const int64_t* __restrict__ cand_off,
                              int T, int32_t* __restrict__ ids) {
  const int64_t q = blockIdx.x;
  const int64_t qbase = cand_off[q * T];
  for (int64_t r = rng_off[q * T]; r < rng_off[(q + 1) * T]; ++r) {
    const Range rg = ranges[r];
    for (int i = threadIdx.x; i < rg.n; i += blockDim.x)
      ids[qbase + rg.pos + i] = perm[rg.poff + i];
  }
}

// ---- distance + top-k --------------------------------------------------------------------
struct Entry {
  double dist;
  int pos;
  int id;
};
__device__ inline bool entry_less(const Entry& a, const Entry& b) {
  return a.dist < b.dist || (a.dist == b.dist && a.pos < b.pos);
}

__device__ inline void bitonic_entries(Entry* e, int np) {
  for (int k = 2; k <= np; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const Entry a = e[lo], b = e[hi];
        if (up ? entry_less(b, a) : entry_less(a, b)) {
          e[lo] = b;
          e[hi] = a;
        }
      }
      __syncthreads();
    }
}

// Merge step: sort buf[0, np) and keep the best k (optionally unique ids) at the front.
// Returns the number of valid best entries.  All threads call; `scratch` is int[kBuf].
// dedup: 0 keep duplicates (knn), 1 each id once, 2 each DISTANCE once (knnPQ's `nub`)
__device__ int merge_best(Entry* buf, int filled, int k, int dedup, int* scratch) {
  int np = 1;
  while (np < filled) np <<= 1;
  for (int i = filled + threadIdx.x; i < np; i += blockDim.x)
    buf[i] = Entry{__longlong_as_double(0x7ff0000000000000LL), 0x7fffffff, -1};
  __syncthreads();
  bitonic_entries(buf, np);
  if (!dedup) return filled < k ? filled : k;
  // same id => same distance (deterministic distance function): a duplicate sits in the run
  // of equal distances before it
  for (int i = threadIdx.x; i < filled; i += blockDim.x) {
    int dup = 0;
    if (dedup == 2) {
      dup = i > 0 && buf[i - 1].dist == buf[i].dist;
    } else {
      for (int j = i - 1; j >= 0 && buf[j].dist == buf[i].dist; --j)
        if (buf[j].id == buf[i].id) {
          dup = 1;
          break;
        }
    }
    scratch[i] = dup;
  }
  __syncthreads();
  // thread 0 compacts the first k unique entries (k is small; filled <= kBuf)
  __shared__ int s_kept;
  if (threadIdx.x == 0) {
    int w = 0;
    for (int i = 0; i < filled && w < k; ++i)
      if (!scratch[i]) {
        if (w != i) buf[w] = buf[i];
        ++w;
      }
    s_kept = w;
  }
  __syncthreads();
  return s_kept;
}

template <class TD>
struct AccOf { typedef float type; };
template <>
struct AccOf<double> { typedef double type; };

template <class TD>
__device__ inline typename AccOf<TD>::type ld(const TD* p) { return (typename AccOf<TD>::type)*p; }
template <>
__device__ inline float ld<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }

template <class TA>
__device__ inline TA wave_sum(TA v) {
  // fixed butterfly: the result does not depend on which wave computes it
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// metricDDL2 exactly as the reference evaluates it (Internal.hs:403-406): a LEFT FOLD of
// (u - v) ** 2 over the coordinates, then the root — one thread per row.  The batched distance
// passes reduce a row by a lane butterfly (same value to ~1 ulp, different last bits); the few
// rows that reach a result are evaluated again this way, so the distances that leave the
// library carry the reference's bits and their order is the order of those values.
__device__ inline double leftfold_distance(const double* __restrict__ x, const double* qs, int d) {
  double acc = 0.0;
  int j = 0;
  // one thread walks one row: the row's loads go out thirty-two elements at a time (sixteen 16-byte
  // loads in flight) so that the walk waits for HBM d / 32 times, not d / 4; the sum itself is the
  // reference's left fold whatever the grouping of the loads
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    typedef double d2 __attribute__((ext_vector_type(2), aligned(16)));
    for (; j + 32 <= d; j += 32) {
      d2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const d2*>(x + j + 2 * u);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double t0 = v[u][0] - qs[j + 2 * u];
        acc = acc + t0 * t0;
        const double t1 = v[u][1] - qs[j + 2 * u + 1];
        acc = acc + t1 * t1;
      }
    }
  }
  for (; j < d; ++j) {
    const double t = x[j] - qs[j];
    acc = acc + t * t;
  }
  return sqrt(acc);
}

// Entries a dense f64 path that RANKS on butterfly sums keeps beyond k before the final selection:
// the k + kLfMargin best by (butterfly distance, position) are re-evaluated as the reference's left
// fold and the k best of THOSE by (fold distance, position) are the answer, so a candidate whose
// two sums round differently near the k-th distance cannot change the membership of the result
// (RPTree.hs:174 ranks ALL candidates on the fold).  It would take more than kLfMargin DIFFERENT
// rows within a few ulp of the k-th distance to defeat this (copies of one row — the same point
// found by several trees — have equal sums of both kinds and keep their order); the prefiltered
// path certifies its cut per query instead.
constexpr int kLfMargin = 8;

// Final stage of those paths: m kept entries (ids / positions in bid / bpos, m <= capacity of lf
// and order) -> left-fold distances, order by (distance, position), the duplicate rule on the
// FINAL values (dedup 2 = knnPQ's nub: one entry per distance; dedup 1: the kept ids are distinct
// already), the first k written.  tid / nthr / sync: the threads that share the arrays.
template <class Sync>
__device__ inline void finalize_leftfold(const double* __restrict__ X, int d, const double* qs, int m,
                                         int k, int dedup, double* lf, int* order, const int* bid,
                                         const int* bpos, int tid, int nthr, Sync sync, int64_t q,
                                         int32_t* __restrict__ out_ids, double* __restrict__ out_dist,
                                         int32_t* __restrict__ out_cnt) {
  for (int i = tid; i < m; i += nthr) lf[i] = leftfold_distance(X + (int64_t)bid[i] * d, qs, d);
  sync();
  for (int i = tid; i < m; i += nthr) {
    const double di = lf[i];
    const int pi = bpos[i];
    int rank = 0;
    // a total order whatever the values are: NaN (a NaN query or row, inf - inf) ranks behind every
    // number, NaNs among themselves by position — the ranks are a permutation of 0 .. m - 1
    if (di == di)
      for (int j = 0; j < m; ++j) rank += lf[j] < di || (lf[j] == di && bpos[j] < pi);
    else
      for (int j = 0; j < m; ++j) rank += lf[j] == lf[j] || bpos[j] < pi;
    order[rank] = i;
  }
  sync();
  if (tid == 0) {
    int w = 0;
    double last = -1.0;
    for (int r = 0; r < m && w < k; ++r) {
      const int i = order[r];
      if (dedup == 2 && w > 0 && lf[i] == last) continue;
      out_ids[q * k + w] = bid[i];
      out_dist[q * k + w] = last = lf[i];
      ++w;
    }
    out_cnt[q] = w;
    for (; w < k; ++w) {
      out_ids[q * k + w] = -1;
      out_dist[q * k + w] = __longlong_as_double(0x7ff0000000000000LL);
    }
  }
}

// dense data.  One block (256 threads) per query.  Candidate list given as ranges.
// `identity`: the candidate list is the whole dataset in id order (brute force), perm unused.
template <class TD>
__global__ __launch_bounds__(256) void topk_dense_kernel(
    const TD* __restrict__ X, int d, const TD* __restrict__ Q, const int32_t* __restrict__ perm,
    const Range* __restrict__ ranges, const int64_t* __restrict__ rng_off, int T, int64_t N,
    int identity, int k, int dedup, int32_t* __restrict__ out_ids, double* __restrict__ out_dist,
    int32_t* __restrict__ out_cnt) {
  typedef typename AccOf<TD>::type TA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Entry* buf = reinterpret_cast<Entry*>(smem);                       // [kBuf]
  int* scratch = reinterpret_cast<int*>(smem + sizeof(Entry) * kBuf);  // [kBuf]
  TA* qs = reinterpret_cast<TA*>(smem + (sizeof(Entry) + 4) * kBuf);   // [d]
  const int64_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < d; j += blockDim.x) qs[j] = ld<TD>(Q + q * d + j);
  __syncthreads();

  int best = 0;    // valid best entries at buf[0, best)
  int filled = 0;  // entries in buf (block-uniform)
  const int cap = kBuf;
  // f64: the running list keeps kLfMargin entries more than k (see kLfMargin)
  const int kk = std::is_same<TD, double>::value ? (k + kLfMargin < kBuf / 2 ? k + kLfMargin : kBuf / 2) : k;
  const int64_t r0 = identity ? 0 : rng_off[q * T];
  const int64_t r1 = identity ? 1 : rng_off[(q + 1) * T];
  for (int64_t r = r0; r < r1; ++r) {
    Range rg;
    if (identity) rg = Range{0, (int32_t)N, 0};
    else rg = ranges[r];
    int done = 0;
    while (done < rg.n) {
      int take = rg.n - done;
      if (take > cap - filled) take = cap - filled;
      // each wave handles candidates wave, wave+4, ... of this slice, four rows in flight
      for (int i0 = wave; i0 < take; i0 += 16) {
        int idv[4];
        TA s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 4 * u;
          const int c = done + (i < take ? i : i0);
          idv[u] = identity ? c : perm[rg.poff + c];
          s[u] = (TA)0;
        }
        for (int j = lane; j < d; j += 64) {
          const TA qj = qs[j];
          TA x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = ld<TD>(X + (int64_t)idv[u] * d + j);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const TA df = x[u] - qj;
            s[u] += df * df;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 4 * u;
          const TA tot = wave_sum(s[u]);
          if (lane == 0 && i < take)
            buf[filled + i] = Entry{(double)sqrt((double)tot), rg.pos + done + i, idv[u]};
        }
      }
      filled += take;
      done += take;
      __syncthreads();
      if (filled == cap) {
        best = merge_best(buf, filled, kk, dedup, scratch);
        filled = best;
        __syncthreads();
      }
    }
  }
  if (filled > best || best == 0) {
    best = merge_best(buf, filled, kk, dedup, scratch);
    __syncthreads();
  }
  if constexpr (std::is_same<TD, double>::value) {
    // the kept entries' distances again as the reference's left fold, the k best of those in the
    // order of those values (ties by candidate position): finalize_leftfold
    double* lf = reinterpret_cast<double*>(scratch);  // [kBuf] ints = kBuf / 2 doubles >= kk
    int* ids2 = reinterpret_cast<int*>(buf + kBuf / 2);  // the upper half of buf is free now:
    int* pos2 = ids2 + kBuf / 2;                          // 3 x kBuf / 2 ints fit its 16 KB
    int* order = pos2 + kBuf / 2;
    for (int i = threadIdx.x; i < best; i += blockDim.x) {
      ids2[i] = buf[i].id;
      pos2[i] = buf[i].pos;
    }
    __syncthreads();
    finalize_leftfold(reinterpret_cast<const double*>(X), d, reinterpret_cast<const double*>(qs), best, k,
                      dedup, lf, order, ids2, pos2, (int)threadIdx.x, (int)blockDim.x,
                      [] { __syncthreads(); }, q, out_ids, out_dist, out_cnt);
    return;
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const bool ok = i < best;
    out_ids[q * k + i] = ok ? buf[i].id : -1;
    out_dist[q * k + i] = ok ? buf[i].dist : __longlong_as_double(0x7ff0000000000000LL);
  }
  if (threadIdx.x == 0) out_cnt[q] = best;
}

// distances of the candidates [first, fill) of an LDS batch (ids in cid) to the query in qs,
// written to cdist.  The waves taking part are numbered slot = 0 .. nslots-1; every row is
// reduced by the same fixed butterfly whichever wave handles it, so the value does not depend
// on the kernel variant.  Each wave keeps EIGHT load instructions in flight.
template <class TD, class TA, int INFL = 8 /* load instructions in flight per wave */,
          bool SQRT = true /* false: leave the squared distance (a TA value) in cdist */,
          class TO = double /* element type of the batch values */>
__device__ __forceinline__ void batch_distances(const TD* __restrict__ X, int d, const int* cid,
                                       TO* cdist, const TA* qs, int first, int fill, int slot,
                                       int nslots, int lane) {
  constexpr int VV = 16 / (int)sizeof(TD);
  const int lpr = (d % VV) == 0 ? d / VV : 0;  // lanes one row needs with 16-byte loads
  if (lpr > 0 && lpr <= 32 && (lpr & (lpr - 1)) == 0) {
    // short rows (<= 512 B): 64 / lpr rows per load instruction, eight instructions in flight
    struct alignas(16) Raw { TD v[VV]; };
    const int rpw = 64 / lpr, sub = lane / lpr, jl = (lane % lpr) * VV;
    for (int i0 = first + slot * INFL * rpw; i0 < fill; i0 += nslots * INFL * rpw) {
      TA s[INFL];
      Raw x[INFL];
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const int i = i0 + u * rpw + sub;
        x[u] = *reinterpret_cast<const Raw*>(X + (int64_t)cid[i < fill ? i : i0] * d + jl);
      }
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        s[u] = (TA)0;
#pragma unroll
        for (int v = 0; v < VV; ++v) {
          const TA df = ld<TD>(&x[u].v[v]) - qs[jl + v];
          s[u] += df * df;
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);  // fixed butterfly
        const int i = i0 + u * rpw + sub;
        if (jl == 0 && i < fill) cdist[i] = SQRT ? (TO)sqrt((double)s[u]) : (TO)s[u];
      }
    }
  } else
  // ---- long rows: one row per load instruction, 8 rows in flight per wave ----
  for (int i0 = first + slot * 8; i0 < fill; i0 += nslots * 8) {
    TA s[8];
    const TD* rows[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u < fill ? i0 + u : i0;
      rows[u] = X + (int64_t)cid[i] * d;
      s[u] = (TA)0;
    }
    constexpr int V = 16 / (int)sizeof(TD);
    if ((d % V) == 0) {
      struct alignas(16) Raw { TD v[V]; };
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      for (int j = lane * V; j < d; j += 64 * V) {
        // All eight row loads are issued before the first is used.  That has to be SAID: at the
        // register cap of several instantiations (168 VGPRs at three workgroups per CU) the scheduler
        // turned this loop into load / wait / load / wait — one or two rows in flight instead of eight
        // — which is the whole of round 3's "66 -> 119 ms" cliff of the bf16 kernel under the
        // consecutive-slot fill (the fill cost 3 registers, the gathers lost their memory
        // parallelism) and had silently hit the f32 / half-tier kernels on rows beyond 512 bytes.
        // The group barrier keeps the eight VMEM reads one scheduling group ahead of the arithmetic;
        // the waits stay progressive (vmcnt 7, 6, ... as the rows are consumed).
        u32x4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const u32x4*>(rows[u] + j);
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);          // eight VMEM reads ...
        __builtin_amdgcn_sched_group_barrier(0x002, 8 * V * 3, 0);  // ... then the VALU work on them
        Raw x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_memcpy(&x[u], &raw[u], 16);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const TA df = ld<TD>(&x[u].v[v]) - qs[j + v];
            s[u] += df * df;
          }
      }
    } else {
      for (int j = lane; j < d; j += 64) {
        const TA qj = qs[j];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const TA df = ld<TD>(rows[u] + j) - qj;
          s[u] += df * df;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const TA tot = wave_sum(s[u]);
      if (lane == 0 && i0 + u < fill) cdist[i0 + u] = SQRT ? (TO)sqrt((double)tot) : (TO)tot;
    }
  }
}

// ---- int8 ranking tier (round 3) ----------------------------------------------------------
// The dataset's int8 shadow holds c = rint(x / s) clamped to [-127, 127] with ONE scale s for the
// whole dataset, stored offset-binary (c + 128), rows of d bytes (d % 16 == 0): an eighth of the f64
// bytes, a 128-element row is one cache line.  The query is quantised SIXTEEN bits deep on the
// same grid, g = rint(256 q / s) clamped to the int16 range, kept as two byte planes (g = 256 H +
// Lo, H + 128 and Lo packed four to a word in LDS), and a candidate is ranked on the INTEGER
//     I = sum (g_j - 256 c_j)^2 = 65536 sum cu^2 - 131072 sum cu hu - 512 sum cu lo + K(query),
// cu = c + 128, hu = H + 128 (the sums of cu alone cancel), K = sum g^2 + 2^24 sum hu + 2^16 sum lo
// - 2^30 d: three v_dot4_u32_u8 per four elements, exact — no rounding anywhere in the ranking
// value (an 8-bit query doubled the uncertainty of the cut: at C2 17 candidates inside it instead of 9).
// By the triangle inequality the exact distance of a row x obeys
//     | |q - x| - (s / 256) sqrt(I) | <= |q - (s / 256) g| + |x - s c| <= eq + emax,
// eq computed per query in f64 (clipped elements included), emax = the dataset's largest
// |x - s c| (measured when the shadow is built), so a cut is certified exactly like the other
// tiers' (knn_fused_kernel), and an uncertified query is answered by the exact kernel.
struct Sh8 {
  double s, emax;  // scale, max row error; s == 0: the tier is not in use
};

// What a ranking tier knows about the exact distance D of a row from its estimate Dh (round 4):
//     lower(Dh) <= D <= upper(Dh),  lower(Dh) = (Dh - A - Bt Dh)(1 - Bf),  upper(Dh) = (Dh + A + Bt Dh)(1 + Bf)
//   f32 shadow : A = 2.1 u (xmax + |q|) + sqrt(d) 4e-23,             Bt = (d + 2) u   (rounded inputs,
//                differences, f32 accumulation, products in the subnormal range; knn_fused_kernel)
//   half shadow: A = 1.05 2^-11 xmax + 2.1 u |q| + sqrt(d) 3.1e-8,   Bt = (d + 2) u
//   int8 shadow: A = eq + emax (triangle inequality, Sh8), Bt = 0;  Dh = (s / 256) sqrt(I) and the
//                value kept is I rounded to f32 (rnd = 1.2e-7 covers it)
//   f32 DATA   : "exact" is the f32 distance the all-f32 kernel ranks on, within Bf = (d + 2) u of D
// The stored ranking value v is a non-negative float (a squared distance, or I); Dh = g sqrt(v).
struct TierBounds {
  double A, Bt, Bf, g, rnd;
  __device__ static TierBounds make(int tier /* 1 f32, 2 half, 3 int8 */, bool f64_data, int d, double xmax,
                                    double qnorm, double q8eq, Sh8 sh8) {
    const double u = 5.9604644775390625e-08;
    TierBounds b;
    b.A = tier == 3   ? (q8eq + sh8.emax) * (1.0 + 1e-12) + 1e-300
          : tier == 2 ? 1.05 * 4.8828125e-4 * xmax + 2.1 * u * qnorm + sqrt((double)d) * 3.1e-8
                      : 2.1 * u * (xmax + qnorm) + sqrt((double)d) * 4e-23;
    b.Bt = tier == 3 ? 0.0 : (double)(d + 2) * u;
    b.Bf = f64_data ? 0.0 : (double)(d + 2) * u * 1.01;
    b.g = tier == 3 ? sh8.s * (1.0 / 256.0) : 1.0;
    b.rnd = tier == 3 ? 1.2e-7 : 0.0;
    return b;
  }
  // the estimate of a stored value (its bits), rounded up
  __device__ double dh_up(unsigned int vb) const {
    return g * sqrt((double)__uint_as_float(vb) * (1.0 + rnd)) * (1.0 + 1e-15);
  }
  __device__ double upper(double dh) const { return (dh + A + Bt * dh) * (1.0 + Bf) * (1.0 + 1e-15); }
  // a row may be dropped against the bound Ulim on the exact k-th distance iff lower(Dh) > Ulim, i.e.
  // iff Dh > (Ulim / (1 - Bf) + A) / (1 - Bt); every stored value at or below the returned bits is KEPT
  __device__ unsigned int keep_bits(double Ulim) const {
    const double dstar = (Ulim / (1.0 - Bf) + A) / (1.0 - Bt) * (1.0 + 1e-15);
    const double t = (dstar / g) * (dstar / g) / (1.0 - rnd) * (1.0 + 1e-15);
    if (!(t < 3.0e38)) return 0x7fc00000u;  // inf / NaN: keep everything that is a number
    float tf = (float)t;
    if ((double)tf < t) tf = __uint_as_float(__float_as_uint(tf) + 1u);  // round up (t >= 0)
    return __float_as_uint(tf);
  }
  // may a row whose stored value is vb be dropped against Ulim?  (the certificate of a fixed cut)
  __device__ bool dropped_ok(unsigned int vb, double Ulim) const { return vb > keep_bits(Ulim); }
};

// quantised query -> planes[0 .. d/4) = hu words, planes[d/4 .. d/2) = lo words; the caller sums
// kq (this thread's share of K without the - 2^30 d term) and e2 (its share of eq^2) over its threads
__device__ inline void quantise_query(const double* qsrc_d, const float* qsrc_f, int d, double s,
                                      unsigned int* planes, int tid, int nthr, double& kq, double& e2) {
  kq = 0.0;
  e2 = 0.0;
  const double inv = 256.0 / s, g2x = s / 256.0;
  const int nw = d / 4;
  for (int w = tid; w < nw; w += nthr) {
    unsigned int wh = 0, wl = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double v = qsrc_d ? qsrc_d[4 * w + b] : (double)qsrc_f[4 * w + b];
      double c = rint(v * inv);
      c = c < -32768.0 ? -32768.0 : (c > 32767.0 ? 32767.0 : c);  // NaN stays NaN: e2 turns NaN, no certificate
      const int g = c == c ? (int)c : 0;
      const double e = v - g2x * (double)g;
      e2 += e * e;
      const unsigned int hu = (unsigned int)((g >> 8) + 128), lo = (unsigned int)(g & 255);
      kq += (double)g * (double)g + 16777216.0 * (double)hu + 65536.0 * (double)lo;  // exact: < 2^53
      wh |= hu << (8 * b);
      wl |= lo << (8 * b);
    }
    planes[w] = wh;
    planes[nw + w] = wl;
  }
}

// number of keys of an LDS list below `mine`: the loads go out eight at a time (one load, one
// compare per turn left every iteration waiting out an LDS round trip: 14 k cycles for 150 keys)
__device__ __forceinline__ int count_below(const unsigned long long* lkey, int n, unsigned long long mine) {
  int rank = 0, j = 0;
  for (; j + 8 <= n; j += 8) {
    unsigned long long a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = lkey[j + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) rank += a[u] < mine;
  }
  for (; j < n; ++j) rank += lkey[j] < mine;
  return rank;
}
// ... and of (distance, position) pairs below (di, pi)
__device__ __forceinline__ int count_below2(const double* dd, const int* pp, int m, double di, int pi) {
  int rank = 0, j = 0;
  if (di != di) {  // NaN ranks behind every number, NaNs by position (a total order: no two equal ranks)
    for (; j < m; ++j) rank += dd[j] == dd[j] || pp[j] < pi;
    return rank;
  }
  for (; j + 8 <= m; j += 8) {
    double a[8];
    int b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = dd[j + u];
      b[u] = pp[j + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) rank += a[u] < di || (a[u] == di && b[u] < pi);
  }
  for (; j < m; ++j) rank += dd[j] < di || (dd[j] == di && pp[j] < pi);
  return rank;
}

// integer squared distances (see above) of the candidates [first, fill) -> cdist (exact integers
// as doubles, in units of (s / 256)^2)
template <int INFL = 16, class TO = double>
__device__ __forceinline__ void batch_distances_i8(const uint8_t* __restrict__ X8, int d, const int* cid,
                                                   TO* cdist, const unsigned int* planes, double kq,
                                                   int first, int fill, int slot, int nslots, int lane) {
  const int lpr = d / 16;  // lanes one row needs with 16-byte loads
  const int nw = d / 4;
  if (lpr <= 32 && (lpr & (lpr - 1)) == 0) {
    const int rpw = 64 / lpr, sub = lane / lpr, piece = lane % lpr;
    const uint4 hv = reinterpret_cast<const uint4*>(planes)[piece];
    const uint4 lv = reinterpret_cast<const uint4*>(planes + nw)[piece];
    const unsigned int hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
    for (int i0 = first + slot * INFL * rpw; i0 < fill; i0 += nslots * INFL * rpw) {
      uint4 x[INFL];
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const int i = i0 + u * rpw + sub;
        x[u] = *reinterpret_cast<const uint4*>(X8 + (int64_t)cid[i < fill ? i : i0] * d + piece * 16);
      }
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const unsigned int xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
        unsigned int a = 0, b = 0, cc = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a = __builtin_amdgcn_udot4(xw[w], hw[w], a, false);
          b = __builtin_amdgcn_udot4(xw[w], lw[w], b, false);
          cc = __builtin_amdgcn_udot4(xw[w], xw[w], cc, false);
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
          cc += __shfl_xor(cc, o);
        }
        const int i = i0 + u * rpw + sub;
        if (piece == 0 && i < fill)
          cdist[i] = (TO)(65536.0 * (double)cc - 131072.0 * (double)a - 512.0 * (double)b + kq);
      }
    }
    return;
  }
  if (lpr <= 64) {
    // rows of up to 1024 bytes: one row per load instruction (lanes beyond the row idle), SIXTEEN rows
    // in flight — with eight a 768-byte row kept half the bytes of the bf16 kernel's rows in flight
    // and the pass was no faster than reading the bf16 rows themselves (C5: 82 against 68 ms)
    constexpr int RF = 16;
    const bool on = lane < lpr;
    uint4 hv = uint4{0u, 0u, 0u, 0u}, lv = hv;
    if (on) {
      hv = reinterpret_cast<const uint4*>(planes)[lane];
      lv = reinterpret_cast<const uint4*>(planes + nw)[lane];
    }
    const unsigned int hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
    for (int i0 = first + slot * RF; i0 < fill; i0 += nslots * RF) {
      uint4 x[RF];
#pragma unroll
      for (int u = 0; u < RF; ++u) {
        x[u] = uint4{0u, 0u, 0u, 0u};
        if (on) x[u] = *reinterpret_cast<const uint4*>(X8 + (int64_t)cid[i0 + u < fill ? i0 + u : i0] * d + lane * 16);
      }
#pragma unroll
      for (int u = 0; u < RF; ++u) {
        const unsigned int xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
        unsigned int a = 0, b = 0, cc = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a = __builtin_amdgcn_udot4(xw[w], hw[w], a, false);
          b = __builtin_amdgcn_udot4(xw[w], lw[w], b, false);
          cc = __builtin_amdgcn_udot4(xw[w], xw[w], cc, false);
        }
        for (int o = 32; o > 0; o >>= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
          cc += __shfl_xor(cc, o);
        }
        if (lane == 0 && i0 + u < fill)
          cdist[i0 + u] = (TO)(65536.0 * (double)cc - 131072.0 * (double)a - 512.0 * (double)b + kq);
      }
    }
    return;
  }
  // longer rows: one row per load instruction and 1024-byte piece, eight rows in flight
  for (int i0 = first + slot * 8; i0 < fill; i0 += nslots * 8) {
    unsigned int a[8], b[8], cc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = b[u] = cc[u] = 0;
    for (int p = lane; p < lpr; p += 64) {
      const uint4 hv = reinterpret_cast<const uint4*>(planes)[p];
      const uint4 lv = reinterpret_cast<const uint4*>(planes + nw)[p];
      const unsigned int hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
      uint4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        x[u] = *reinterpret_cast<const uint4*>(X8 + (int64_t)cid[i0 + u < fill ? i0 + u : i0] * d + p * 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned int xw[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a[u] = __builtin_amdgcn_udot4(xw[w], hw[w], a[u], false);
          b[u] = __builtin_amdgcn_udot4(xw[w], lw[w], b[u], false);
          cc[u] = __builtin_amdgcn_udot4(xw[w], xw[w], cc[u], false);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned int ta = a[u], tb = b[u], tc = cc[u];
      for (int o = 32; o > 0; o >>= 1) {
        ta += __shfl_xor(ta, o);
        tb += __shfl_xor(tb, o);
        tc += __shfl_xor(tc, o);
      }
      if (lane == 0 && i0 + u < fill)
        cdist[i0 + u] = (TO)(65536.0 * (double)tc - 131072.0 * (double)ta - 512.0 * (double)tb + kq);
    }
  }
}

// CSR data for the fused kernel (SVector rows, Internal.hs:92-93): data rows and query rows
struct CsrPtrs {
  const int64_t* rowptr;
  const int32_t* col;
  const void* val;
  int64_t nnz;
  const int64_t* qrowptr;
  const int32_t* qcol;
  const void* qval;
  // f32 prefilter (f64 rows, d <= 65536): the rows again as (u16 column, f32 value), 6 instead of
  // 12 bytes per nonzero; the longest row (bounds the f32 summation chains)
  const uint16_t* col16;
  const float* val32;
  int64_t max_rowlen;
  // half tier: the rows as a fixed-width table of (u16 column | half value << 16) slots
  const uint32_t* ell;
  int ell_w;
};

// Squared-distance sums of U CSR rows per 16-lane group against the dense-ified query qs (LDS):
// s[u] = sum over the row's nonzeros of ((x_j - q_j)^2 - q_j^2); the true Euclidean distance is
// sqrt(|q|^2 + s).  Lane l of the group takes the nonzeros ra + 64 t + 4 l .. + 3 for t = 0, 1, ..
// with one 16-byte column load and one 16/32-byte value load (4-byte aligned addresses), sums
// them in index order and the group adds its sixteen partial sums by a fixed butterfly: a
// row's value does not depend on the kernel or the slot it is computed in (the fused kernel and
// the general path call this same function).  All loads of a step — U rows — are issued before
// the first is used: the walk of a row is a chain of dependent loads (id -> rowptr -> col / val
// -> qs[col]), and issuing them row by row left the kernels latency-bound (C3: 2.2 TB/s).
template <class TD, int U>
__device__ __forceinline__ void csr_rows_dist2(const int32_t* __restrict__ col,
                                               const TD* __restrict__ val, int64_t nnz,
                                               const int64_t (&ra)[U], const int64_t (&rb)[U],
                                               const double* qs, int l16, double (&s)[U]) {
  struct __attribute__((packed, aligned(4))) C4 { int v[4]; };
  struct __attribute__((packed, aligned(4))) V4 { TD v[4]; };
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.0;
  for (int64_t t = 4 * l16;; t += 64) {
    C4 c[U];
    V4 v[U];
    bool more = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = ra[u] + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[u].v[e] = 0;
        v[u].v[e] = (TD)0;
      }
      if (j < rb[u]) {
        if (j + 4 <= nnz) {  // may run past the row's end (masked below), never past the arrays'
          c[u] = *reinterpret_cast<const C4*>(col + j);
          v[u] = *reinterpret_cast<const V4*>(val + j);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < rb[u]) {
              c[u].v[e] = col[j + e];
              v[u].v[e] = val[j + e];
            }
        }
      }
      more = more || j + 64 < rb[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = ra[u] + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = j + e < rb[u];
        const double qj = qs[ok ? c[u].v[e] : 0];
        const double df = (double)v[u].v[e] - qj;
        const double term = df * df - qj * qj;
        s[u] += ok ? term : 0.0;
      }
    }
    if (!__any(more)) break;
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    for (int o = 8; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);  // fixed butterfly inside the group
}

// distances of the CSR candidates [first, fill) of an LDS batch: sixteen rows of a wave in flight
// (four row slots per 16-lane group)
template <class TD>
__device__ __forceinline__ void batch_distances_csr(const int64_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col,
                                                    const TD* __restrict__ val, int64_t nnz,
                                                    const int* cid, double* cdist, const double* qs,
                                                    double qn2, int first, int fill, int wave,
                                                    int lane) {
  constexpr int U = 4;
  const int grp = lane >> 4, l16 = lane & 15;
  for (int i0 = first + wave * 4 * U; i0 < fill; i0 += 4 * 4 * U) {
    int64_t ra[U], rb[U];
    double s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 4 + grp;
      const bool ok = i < fill;
      const int id = cid[ok ? i : first];
      ra[u] = rowptr[id];
      rb[u] = ok ? rowptr[id + 1] : ra[u];
    }
    csr_rows_dist2<TD, U>(col, val, nnz, ra, rb, qs, l16, s);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double t = s[u] + qn2;
      const int i = i0 + u * 4 + grp;
      if (l16 == 0 && i < fill) cdist[i] = sqrt(t > 0 ? t : 0.0);
    }
  }
}

// The same walk over the (u16 column, f32 value) shadow of the rows, in f32 arithmetic: the
// RANKING pass of the CSR prefilter (knn_fused_kernel<.., PRE32, CSR>).  No FMA contraction in
// this translation unit: the error bound of the refine step counts one rounding per operation.
template <int U>
__device__ __forceinline__ void csr_rows_dist2_f32(const uint16_t* __restrict__ col16,
                                                   const float* __restrict__ val32, int64_t nnz,
                                                   const int64_t (&ra)[U], const int64_t (&rb)[U],
                                                   const float* qs32, int l16, float (&s)[U]) {
  struct __attribute__((packed, aligned(2))) C4 { uint16_t v[4]; };
  struct __attribute__((packed, aligned(4))) V4 { float v[4]; };
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.f;
  for (int64_t t = 4 * l16;; t += 64) {
    C4 c[U];
    V4 v[U];
    bool more = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = ra[u] + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[u].v[e] = 0;
        v[u].v[e] = 0.f;
      }
      if (j < rb[u]) {
        if (j + 4 <= nnz) {
          c[u] = *reinterpret_cast<const C4*>(col16 + j);
          v[u] = *reinterpret_cast<const V4*>(val32 + j);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < rb[u]) {
              c[u].v[e] = col16[j + e];
              v[u].v[e] = val32[j + e];
            }
        }
      }
      more = more || j + 64 < rb[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = ra[u] + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = j + e < rb[u];
        const float qj = qs32[ok ? (int)c[u].v[e] : 0];
        const float df = v[u].v[e] - qj;
        const float term = df * df - qj * qj;
        s[u] += ok ? term : 0.f;
      }
    }
    if (!__any(more)) break;
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    for (int o = 8; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
}

// f32 SQUARED distances (clamped at 0) of the CSR candidates [first, fill) from the shadow
__device__ __forceinline__ void batch_distances_csr32(const int64_t* __restrict__ rowptr,
                                                      const uint16_t* __restrict__ col16,
                                                      const float* __restrict__ val32, int64_t nnz,
                                                      const int* cid, double* cdist,
                                                      const float* qs32, float qn2, int first,
                                                      int fill, int wave, int lane) {
  constexpr int U = 4;
  const int grp = lane >> 4, l16 = lane & 15;
  for (int i0 = first + wave * 4 * U; i0 < fill; i0 += 4 * 4 * U) {
    int64_t ra[U], rb[U];
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 4 + grp;
      const bool ok = i < fill;
      const int id = cid[ok ? i : first];
      ra[u] = rowptr[id];
      rb[u] = ok ? rowptr[id + 1] : ra[u];
    }
    csr_rows_dist2_f32<U>(col16, val32, nnz, ra, rb, qs32, l16, s);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float t = qn2 + s[u];
      const int i = i0 + u * 4 + grp;
      if (l16 == 0 && i < fill) cdist[i] = (double)(t > 0.f ? t : 0.f);
    }
  }
}

// f32 SQUARED distances (clamped at 0) of the CSR candidates [first, fill) from the fixed-width
// half shadow (rpt_dataset::shadow_ell): row id -> id * W slots, no rowptr step; sixteen lanes
// per row, a lane takes the 16-byte chunks l16, l16 + 16, .. (four slots each), the chunks of all
// four row slots of the group and of up to four steps are loaded before the first is used (W <=
// 256: ONE round of loads per sixteen rows of the wave).  A slot adds x (x - 2 q_col) = (x - q)^2
// - q^2 with two roundings (explicit FMAs); an absent slot (column 0, x = 0) adds 0.
__device__ __forceinline__ void batch_distances_ell16(const uint32_t* __restrict__ ell, int W,
                                                      const int* cid, double* cdist,
                                                      const float* qs32, float qn2, int first,
                                                      int fill, int wave, int lane) {
  constexpr int U = 4, S = 4;
  const int grp = lane >> 4, l16 = lane & 15;
  const int chunks = W >> 2;
  for (int i0 = first + wave * 4 * U; i0 < fill; i0 += 4 * 4 * U) {
    const uint4* row[U];
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 4 + grp;
      const int id = cid[i < fill ? i : first];
      row[u] = reinterpret_cast<const uint4*>(ell + (int64_t)id * W);
      s[u] = 0.f;
    }
    for (int c0 = 0; c0 < chunks; c0 += 16 * S) {
      uint4 v[U][S];
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const int c = c0 + st * 16 + l16;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u][st] = c < chunks ? row[u][c] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int st = 0; st < S; ++st) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned int w4[4] = {v[u][st].x, v[u][st].y, v[u][st].z, v[u][st].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned short hb = (unsigned short)(w4[e] >> 16);
            _Float16 h;
            __builtin_memcpy(&h, &hb, 2);
            const float x = (float)h;
            const float qj = qs32[w4[e] & 0xffffu];
            s[u] = __builtin_fmaf(x, __builtin_fmaf(-2.f, qj, x), s[u]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      for (int o = 8; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
      const float t = qn2 + s[u];
      const int i = i0 + u * 4 + grp;
      if (l16 == 0 && i < fill) cdist[i] = (double)(t > 0.f ? t : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------
// fused query kernel (dense data): one workgroup per query does everything after the query
// projections — traversal of every tree (thread = tree, twice: count, then emit ranges in
// tree order), candidate ids gathered into LDS, distances (each wave keeps EIGHT whole rows in
// flight, 16-byte loads, butterfly reduction), top-k.  No global scan, no range list in HBM.
// Batches of kFC candidates; the running best k re-enter the next batch at the front with
// their original candidate positions, so the order (distance, position) is global.
// k <= kFK: k rounds of block-wide arg-min; larger k or a range list that does not fit the
// LDS slab -> the query is flagged and the host falls back to the general path.
// ---------------------------------------------------------------------------------------
constexpr int kFC = 2048;     // candidates per batch
constexpr int kFR = 512;      // leaf ranges per query in LDS
constexpr int kFK = 64;       // largest k served by the arg-min selection
constexpr int kFKx = kFK + kLfMargin;  // capacity of the running best list (see kLfMargin)
// the workgroup kernel's prefilter tiers may keep more than that: the threshold selection costs
// the same whatever it keeps, and the int8 tier needs a wide margin at the cut (see Sh8)
constexpr int kBK = 224;
constexpr int kBKx = kBK + kLfMargin;
#ifndef RPT_CONSEC_FILL
#define RPT_CONSEC_FILL 0
#endif
constexpr bool kConsecFill = RPT_CONSEC_FILL != 0;  // consecutive-slot fill in the non-WIDE instantiations too
constexpr int kSelList = 512;  // select_packed's candidate list (in the idle distance slab)
constexpr int kVoteCap = 16384;  // candidates of one query the voting mode can count (64 KB of LDS)

template <class TD, class TK, bool PRE32, bool CSR = false, bool I8 = false /* PRE32 on the int8 shadow */>
// (three workgroups per CU is what the 52 KB slab allows: keep the registers at that occupancy)
__global__ __launch_bounds__(256, 3) void knn_fused_kernel(
    const TD* __restrict__ X, int d, const TD* __restrict__ Q, const int32_t* __restrict__ perm,
    const double* __restrict__ thr, const double* __restrict__ mglo,
    const double* __restrict__ mghi, int64_t nodes, const TK* __restrict__ Pq, int64_t nq, int T,
    int L, int min_leaf, int64_t N, int k, int dedup_vote /* duplicate rule | vote threshold << 8 */,
    int32_t* __restrict__ out_ids,
    double* __restrict__ out_dist, int32_t* __restrict__ out_cnt, unsigned int* ovf_flags,
    unsigned int* ovf_count, unsigned long long* cand_total,
    const void* __restrict__ Xf /* PRE32: f32 (or, sh16, IEEE half) shadow of X */,
    double xmax /* max row norm */,
    int k1 /* PRE32: entries kept by the shadow pass, the last one = the first excluded */,
    CsrPtrs csr /* CSR: X / Q are null, rows and queries are SVectors */, int sh16 = 0,
    Sh8 sh8 = Sh8{0.0, 0.0} /* s > 0: Xf is the int8 shadow (see Sh8) */,
    unsigned long long* dbg = nullptr /* debug_stamps: phase clocks of one workgroup */) {
  int dbgi = 0;
#define KSTAMP() do { if (dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && dbgi < 60) dbg[dbgi++] = clock64(); } while (0)
  KSTAMP();
  typedef typename AccOf<TD>::type TA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Dense prefilter instantiations (round 3): the batch values are f32 (what their packed keys hold
  // anyway) and a candidate's position is computed, not stored, so the same 32 KB hold TWICE the
  // candidates — C2's 3922 per query in ONE batch: one fill, one selection.
  constexpr bool WIDE = PRE32 && !CSR;
  constexpr int FC = WIDE ? 2 * kFC : kFC;   // candidates per batch
  typedef typename std::conditional<WIDE, float, double>::type TB;
  TB* cval = reinterpret_cast<TB*>(smem);                          // [FC] batch values
  double* cdist = reinterpret_cast<double*>(smem);                 // the same slab as doubles: [kFC]
                                                                   // (exact paths, refine, slots, lists)
  int* cid = reinterpret_cast<int*>(cdist + kFC);                  // [FC] (WIDE) / [kFC]
  int* cpos = WIDE ? cid + 1024 : cid + kFC;                       // [kFC]; WIDE: only the refine
                                                                   // stage's <= kBKx positions
  int64_t* rpoff = reinterpret_cast<int64_t*>(cid + 2 * kFC);      // [kFR]
  int* rn = reinterpret_cast<int*>(rpoff + kFR);                   // [kFR]
  int* tcnt = rn + kFR;                                            // [1024] per-tree counts
  int* trng = tcnt + 1024;                                         // [1024]
  double* bdist = reinterpret_cast<double*>(trng + 1024);          // [kBKx]
  int* bid = reinterpret_cast<int*>(bdist + kBKx);                 // [kBKx]
  int* bpos = bid + kBKx;                                          // [kBKx]
  TA* qs = reinterpret_cast<TA*>(bpos + kBKx);                     // [d]
  float* qs32 = reinterpret_cast<float*>(qs + d);                  // [d] (PRE32)
  // voting mode (dedup_vote >> 8 = v > 0, RPTree.hs:464-478 counts / keepCounts): all candidate
  // ids of the query, sorted, so that the ids found in at least v trees can be picked out
  int* vid = reinterpret_cast<int*>(
      (reinterpret_cast<uintptr_t>(qs32 + d) + 15) & ~(uintptr_t)15);  // [kVoteCap]
  __shared__ int s_nr, s_nc;
  __shared__ double s_qn;
  __shared__ double s_red_d[8];
  __shared__ int s_red_p[8], s_red_i[8];
  __shared__ int s_wk[4];
  __shared__ double s_q8e[4], s_q8k[4];  // int8 tier: the waves' shares of eq^2 and of K (see Sh8)
  __shared__ unsigned int s_mm[4], s_lcnt;  // select_packed: the waves' thresholds, the list length
  const int vote = PRE32 ? 0 : (dedup_vote >> 8);
  const int dedup = vote > 0 ? 0 : (dedup_vote & 3);  // kept ids are distinct

  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool rerun = !PRE32 && k1 == -1;  // second pass: only the queries the prefilter gave up on
  if (rerun && ovf_flags[q] != 2u) return;
  // f32 / bf16 data with duplicates kept: the distances are f32 values, so the squared distance is
  // ranked through the packed key as well and the root is taken once, on output (same bits)
  const bool pack32 = !PRE32 && !CSR && sizeof(TA) == 4 && dedup == 0;
  double* qsd = reinterpret_cast<double*>(qs);  // CSR: the dense-ified query, d doubles (the
                                                // slab holds d * (sizeof(TA) + 4) >= 8 d bytes)
  if constexpr (CSR) {
    for (int j = tid; j < d; j += 256) qsd[j] = 0.0;
    __syncthreads();
    const TD* qv = static_cast<const TD*>(csr.qval);
    for (int64_t j = csr.qrowptr[q] + tid; j < csr.qrowptr[q + 1]; j += 256)
      qsd[csr.qcol[j]] = (double)qv[j];
    __syncthreads();
    if (PRE32)  // the slab holds d * 12 bytes: d doubles, then d floats
      for (int j = tid; j < d; j += 256) reinterpret_cast<float*>(qsd + d)[j] = (float)qsd[j];
    if (wave == 0) {
      double sq = 0.0;
      for (int j = lane; j < d; j += 64) sq += qsd[j] * qsd[j];
      sq = wave_sum(sq);
      if (lane == 0) s_qn = sq;  // |q|^2 (the dense variants keep |q| here, PRE32 only)
    }
  } else {
    for (int j = tid; j < d; j += 256) qs[j] = ld<TD>(Q + q * d + j);
    if constexpr (I8) {  // int8 tier: the quantised query's byte planes in the f32 copy's place
      __syncthreads();
      double kqp, e2p;
      if constexpr (sizeof(TA) == 8)
        quantise_query(reinterpret_cast<const double*>(qs), nullptr, d, sh8.s,
                       reinterpret_cast<unsigned int*>(qs32), tid, 256, kqp, e2p);
      else
        quantise_query(nullptr, reinterpret_cast<const float*>(qs), d, sh8.s,
                       reinterpret_cast<unsigned int*>(qs32), tid, 256, kqp, e2p);
      for (int o = 32; o > 0; o >>= 1) {
        kqp += __shfl_xor(kqp, o);  // integer-valued doubles below 2^53: exact in any order
        e2p += __shfl_xor(e2p, o);
      }
      if (lane == 0) {
        s_q8k[wave] = kqp;
        s_q8e[wave] = e2p;
      }
    } else if (PRE32)
      for (int j = tid; j < d; j += 256) qs32[j] = (float)ld<TD>(Q + q * d + j);
  }

  // ---- traversal: ONE pass in the usual case.  Thread = tree; the leaf ranges a tree reaches go
  // to its S slots (in the batch slab, idle until the first fill) and are counted in any case;
  // after the scan over the trees they are compacted in tree order.  A tree that outgrows its
  // slots (or T > kFR) sends the workgroup through the second, emitting traversal.  (Round 3: two
  // traversals of 13 dependent node loads each were a fifth of a C2 query's life.) ----
  const int S = T <= kFR ? kFR / T : 0;
  int64_t* spoff = reinterpret_cast<int64_t*>(cdist);  // [kFR]
  int* sn = reinterpret_cast<int*>(cdist + kFR);       // [kFR]
  int slots_short = 0;
  for (int t = tid; t < T; t += 256) {
    int nc = 0, nr = 0;
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                 nq, L, min_leaf, N, [&](int off, int n) {
                   if (nr < S) {
                     spoff[t * S + nr] = (int64_t)t * N + off;
                     sn[t * S + nr] = n;
                   }
                   nc += n;
                   ++nr;
                 });
    tcnt[t] = nc;
    trng[t] = nr;
    slots_short |= nr > S;
  }
  const int second_pass = __syncthreads_or(slots_short);
  if (tid == 0) {  // exclusive scans over <= 1024 trees
    int c = 0, r = 0;
    for (int t = 0; t < T; ++t) {
      const int a = tcnt[t], b = trng[t];
      tcnt[t] = c;
      trng[t] = r;
      c += a;
      r += b;
    }
    s_nc = c;
    s_nr = r;
  }
  __syncthreads();
  KSTAMP();  // 1: query + traversal + scan
  const int nr_tot = s_nr, nc_tot = s_nc;
  constexpr bool i8 = I8;
  static_assert(!I8 || (PRE32 && !CSR), "the int8 tier is a dense prefilter");
  const double q8k = i8 ? s_q8k[0] + s_q8k[1] + s_q8k[2] + s_q8k[3] - 1073741824.0 * (double)d : 0.0;
  const double q8eq = i8 ? sqrt(s_q8e[0] + s_q8e[1] + s_q8e[2] + s_q8e[3]) : 0.0;
  if (nr_tot > kFR) {  // too many leaf ranges for the slab: general path
    if (tid == 0) {
      ovf_flags[q] = 1u;
      atomicAdd(ovf_count, 1u);
    }
    return;
  }
  if (tid == 0 && !rerun) atomicAdd(cand_total, (unsigned long long)nc_tot);
  // ---- ranges in tree order + rstart[r] = candidates before range r (the batches are filled by
  // position: a binary search over rstart, every load of a batch in flight at once).  rstart takes
  // tcnt's place: a thread reads its trees' scan values before anyone overwrites them ----
  int* rstart = tcnt;  // [nr_tot + 1] <= kFR + 1
  {
    int r0[4], r1[4], c0[4];  // T <= 1024: at most four trees per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tid + 256 * u;
      r0[u] = t < T ? trng[t] : 0;
      r1[u] = t < T ? (t + 1 < T ? trng[t + 1] : nr_tot) : 0;
      c0[u] = t < T ? tcnt[t] : 0;
    }
    __syncthreads();
    for (int u = 0; u < 4; ++u) {
      const int t = tid + 256 * u;
      if (t >= T) break;
      int r = r0[u], c = c0[u];
      if (!second_pass) {
        for (int i2 = 0; r < r1[u]; ++i2, ++r) {
          const int n = sn[t * S + i2];
          rpoff[r] = spoff[t * S + i2];
          rn[r] = n;
          rstart[r] = c;
          c += n;
        }
      } else {
        traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                     nq, L, min_leaf, N, [&](int off, int n) {
                       rpoff[r] = (int64_t)t * N + off;
                       rn[r] = n;
                       rstart[r] = c;
                       c += n;
                       ++r;
                     });
      }
    }
    if (tid == 0) rstart[nr_tot] = nc_tot;
  }
  __syncthreads();

  // ---- voting mode: every candidate id into LDS, ascending (the order of M.foldrWithKey in
  // keepCounts); an id found in c trees is a run of c equal entries ----
  if (vote > 0) {
    if (nc_tot > kVoteCap) {  // more candidates than the slab counts: reported, not answered
      if (tid == 0) {
        ovf_flags[q] = 3u;
        atomicAdd(ovf_count, 1u);
      }
      return;
    }
    int pos = 0;
    for (int r = 0; r < nr_tot; ++r) {  // every thread walks the same range list
      const int n = rn[r];
      for (int i = tid; i < n; i += 256) vid[pos + i] = perm[rpoff[r] + i];
      pos += n;
    }
    int np = 1;
    while (np < nc_tot) np <<= 1;
    for (int i = nc_tot + tid; i < np; i += 256) vid[i] = 0x7fffffff;
    __syncthreads();
    for (int kk = 2; kk <= np; kk <<= 1)
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (np >> 1); i += 256) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
          const bool up = (lo & kk) == 0;
          const int a = vid[lo], b = vid[hi];
          if (up ? b < a : a < b) {
            vid[lo] = b;
            vid[hi] = a;
          }
        }
        __syncthreads();
      }
  }
  int vsrc = 0;  // voting mode: next entry of vid to look at

  // ---- selection: ksel rounds of block-wide arg-min by (distance, position) over the batch
  // entries [0, fill); the winners go to bdist / bid / bpos.  A thread keeps its eight entries
  // in registers (a consumed one gets position -1), a round costs one arg-min over them, a
  // butterfly, four LDS words and ONE barrier (the per-wave results alternate between two
  // buffers) ----
  // The f32 pass ranks SQUARED f32 distances (exact float values): (value bits << 32 | position)
  // is one 64-bit key whose unsigned order is the (distance, position) order — half the
  // cross-lane traffic per round, and the owner of the winner writes it out itself.
  // Round 3: the ksel rounds are WAVE-local (shuffles only, no LDS, no barrier): every wave takes
  // the ksel smallest of its own quarter of the batch, the four lists meet in LDS and the ksel
  // smallest of those 4 x ksel keys — the batch's ksel smallest are among them — are ranked by
  // counting.  Two barriers per batch instead of one per round (C2: 17 rounds per 2048 entries).
  // Round 3 (late): the ksel smallest through a THRESHOLD instead of ksel rounds.  The value half of
  // a key is the bit pattern of a non-negative float, monotone as an unsigned integer.  An upper
  // bound tau of the ksel-th smallest value: once a full list exists, the value of its last entry
  // (the list sits at the front of the batch); before that, every wave takes the ceil(ksel / 4)-th
  // smallest of its 64 per-thread minima (ranked by counting over lane reads) and tau is the largest
  // of the four — at least ksel entries lie at or below it.  The entries <= tau (ksel plus a few)
  // are compacted into a list and ranked by counting on the full keys: three barriers per batch
  // whatever ksel is (the rounds cost 0.03 ms per kept entry and 10 000 queries at C2).  A list of
  // more than 256 entries (values tied by the hundred) falls back to the rounds.
  auto select_packed = [&](int fill, int ksel, int have /* entries of a list from earlier batches */,
                           int pb0 /* WIDE: position of the batch's first NEW candidate */) -> int {
    constexpr int E = FC / 256;
    unsigned long long key[E];
    unsigned int vmin = ~0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid + 256 * e;
      unsigned int pos;
      if constexpr (WIDE) pos = (unsigned int)(i < have ? bpos[i < kBKx ? i : 0] : pb0 + (i - have));
      else pos = (unsigned int)cpos[i < kFC ? i : 0];
      key[e] = i < fill ? ((unsigned long long)__float_as_uint((float)cval[i]) << 32) | pos : ~0ULL;
      const unsigned int v = (unsigned int)(key[e] >> 32);
      if (i < fill) vmin = v < vmin ? v : vmin;
    }
    unsigned int tau;
    if (have == ksel) {
      tau = __float_as_uint((float)bdist[ksel - 1]);
    } else {
      const int r = (ksel + 3) / 4;
      int cnt = 0;
#pragma unroll
      for (int j2 = 0; j2 < 64; ++j2) {  // (constant lane: v_readlane, not an LDS permute)
        const unsigned int o = (unsigned int)__builtin_amdgcn_readlane((int)vmin, j2);
        cnt += (o < vmin || (o == vmin && j2 < lane)) ? 1 : 0;
      }
      const unsigned long long hit = __ballot(cnt == r - 1);
      tau = __shfl(vmin, __ffsll((long long)hit) - 1);  // ~0u when the wave holds fewer than r entries
      if (lane == 0) s_mm[wave] = tau;
    }
    if (tid == 0) s_lcnt = 0u;
    KSTAMP();  // sel: keys + threshold
    __syncthreads();  // (every thread has read its cdist entries: the list below reuses that slab)
    KSTAMP();  // sel: barrier
    if (have != ksel) {
      tau = s_mm[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) tau = s_mm[w] > tau ? s_mm[w] : tau;
    }
    unsigned long long* lkey = reinterpret_cast<unsigned long long*>(cdist);  // [kSelList]
    int* lidx = reinterpret_cast<int*>(cdist + kSelList);                      // [kSelList]
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (key[e] != ~0ULL && (unsigned int)(key[e] >> 32) <= tau) {
        const unsigned int slot = atomicAdd(&s_lcnt, 1u);
        if (slot < (unsigned int)kSelList) {
          lkey[slot] = key[e];
          lidx[slot] = tid + 256 * e;
        }
      }
    KSTAMP();  // sel: compaction
    __syncthreads();
    KSTAMP();  // sel: barrier
    const int n = (int)s_lcnt;
    if (n <= kSelList) {
      for (int t2 = tid; t2 < n; t2 += 256) {
        const unsigned long long mine = lkey[t2];
        const int rank = count_below(lkey, n, mine);
        if (rank < ksel) {
          bdist[rank] = (double)__uint_as_float((unsigned int)(mine >> 32));
          bid[rank] = cid[lidx[t2]];
          bpos[rank] = (int)(unsigned int)mine;
        }
      }
      __syncthreads();
      return n < ksel ? n : ksel;
    }
    // (the winners' lists of the fallback live in the same idle slab: 4 x kBKx keys, then ids)
    unsigned long long (*wb_key)[kBKx] = reinterpret_cast<unsigned long long (*)[kBKx]>(cdist);
    int (*wb_id)[kBKx] = reinterpret_cast<int (*)[kBKx]>(cdist + 4 * kBKx);
    // ---- fallback: ksel wave-local rounds, the four lists merged by rank counting ----
    for (int r = 0; r < ksel; ++r) {
      unsigned long long m = key[0];
#pragma unroll
      for (int e = 1; e < E; ++e) m = key[e] < m ? key[e] : m;
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t2 = __shfl_xor(m, o);
        m = t2 < m ? t2 : m;
      }
      if (m == ~0ULL) {  // this wave's entries are exhausted (wave-uniform)
        if (lane == 0)
          for (int r2 = r; r2 < ksel; ++r2) wb_key[wave][r2] = ~0ULL;
        break;
      }
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (key[e] == m) {  // keys are unique (positions are): exactly one owner
          key[e] = ~0ULL;
          wb_key[wave][r] = m;
          wb_id[wave][r] = cid[tid + 256 * e];
        }
    }
    __syncthreads();
    const int tot = 4 * ksel;
    int myvalid = 0;
    for (int t2 = tid; t2 < tot; t2 += 256) {
      const unsigned long long mine = wb_key[t2 / ksel][t2 % ksel];
      if (mine == ~0ULL) continue;
      ++myvalid;
      int rank = 0;
      for (int w = 0; w < 4; ++w)
        for (int r = 0; r < ksel; ++r) rank += wb_key[w][r] < mine;
      if (rank < ksel) {
        bdist[rank] = (double)__uint_as_float((unsigned int)(mine >> 32));
        bid[rank] = wb_id[t2 / ksel][t2 % ksel];
        bpos[rank] = (int)(unsigned int)mine;
      }
    }
    // entries in all = the valid keys of the four lists, at most ksel are kept
    for (int o = 32; o > 0; o >>= 1) myvalid += __shfl_xor(myvalid, o);
    if (lane == 0) s_mm[wave] = (unsigned int)myvalid;
    __syncthreads();
    const int nvalid = (int)(s_mm[0] + s_mm[1] + s_mm[2] + s_mm[3]);
    __syncthreads();  // s_mm is the next batch's threshold exchange
    return nvalid < ksel ? nvalid : ksel;
  };
  auto select = [&](int fill, int ksel, int dedup) -> int {
    constexpr int E = kFC / 256;
    double dd[E];
    int pp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid + 256 * e;
      dd[e] = i < fill ? cdist[i] : __longlong_as_double(0x7ff0000000000000LL);
      // NaN (a NaN query or row, inf - inf) ranks with +inf, by position: a total order, every entry
      // can be selected (a NaN would lose every comparison and end the rounds with nothing chosen)
      if (dd[e] != dd[e]) dd[e] = __longlong_as_double(0x7ff0000000000000LL);
      pp[e] = i < fill ? cpos[i] : -1;
    }
    int nb = 0, par = 0;
    double last_d = -1.0;
    while (nb < ksel) {
      double bd = __longlong_as_double(0x7ff0000000000000LL);
      int bp = 0x7fffffff, bi = -1;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (pp[e] >= 0 && (dd[e] < bd || (dd[e] == bd && pp[e] < bp))) {
          bd = dd[e];
          bp = pp[e];
          bi = tid + 256 * e;
        }
      for (int o = 32; o > 0; o >>= 1) {
        const double od = __shfl_xor(bd, o);
        const int op = __shfl_xor(bp, o), oi = __shfl_xor(bi, o);
        if (od < bd || (od == bd && op < bp)) {
          bd = od;
          bp = op;
          bi = oi;
        }
      }
      if (lane == 0) {
        s_red_d[par * 4 + wave] = bd;
        s_red_p[par * 4 + wave] = bp;
        s_red_i[par * 4 + wave] = bi;
      }
      __syncthreads();
      bd = s_red_d[par * 4];
      bp = s_red_p[par * 4];
      bi = s_red_i[par * 4];
      for (int w = 1; w < 4; ++w)
        if (s_red_d[par * 4 + w] < bd || (s_red_d[par * 4 + w] == bd && s_red_p[par * 4 + w] < bp)) {
          bd = s_red_d[par * 4 + w];
          bp = s_red_p[par * 4 + w];
          bi = s_red_i[par * 4 + w];
        }
      par ^= 1;
      if (bi < 0) break;  // candidates exhausted
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (bi == tid + 256 * e) pp[e] = -1;  // consumed
      bool keep = true;
      if (dedup == 2 && nb > 0 && bd == last_d) keep = false;  // knnPQ: one per distance
      else if (dedup && bd == last_d)  // same id => same distance: duplicates are among equal dist
        for (int j = nb - 1; j >= 0 && bdist[j] == bd; --j)
          if (bid[j] == cid[bi]) keep = false;
      if (keep) {
        if (tid == 0) {
          bdist[nb] = bd;
          bid[nb] = cid[bi];
          bpos[nb] = bp;
        }
        ++nb;
        last_d = bd;
      }
    }
    __syncthreads();  // the winners are visible; cdist / cid / cpos may be refilled
    return nb;
  };

  // A dense prefilter cut that cannot be certified is tried ONCE more, by the same workgroup, with
  // three times the kept entries (the other workgroups keep the chip busy meanwhile); only then is
  // the query left to the host's second launch of the exact kernel (two such queries of C2's 10 000
  // cost that launch 0.4 ms: they run alone on an empty chip)
  const int k1_retry = (PRE32 && !CSR) ? (3 * k1 < kBK ? 3 * k1 : kBK) : k1;
retry_wider:
  int best = 0;       // entries of the running best list
  int pos_base = 0;   // candidate position of the next unconsumed candidate
  while ((vote > 0 ? vsrc < nc_tot : pos_base < nc_tot) || best == 0) {
    // ---- fill the batch: best list first (keeps its positions), then new candidates ----
    for (int i = tid; i < best; i += 256) {
      cval[i] = (TB)bdist[i];
      cid[i] = bid[i];
      if constexpr (!WIDE) cpos[i] = bpos[i];
    }
    int fill = best;
    const int first_new = fill;
    const int pb0 = pos_base;
    // voting mode: the heads of the runs of at least `vote` equal ids, 256 entries of vid per
    // round, compacted into the batch with their index in vid as position (ascending id order)
    while (!WIDE && vote > 0 && vsrc < nc_tot && fill + 256 <= kFC) {
      const int i = vsrc + tid;
      int keep = 0;
      if (i < nc_tot && (i == 0 || vid[i] != vid[i - 1])) {
        int c = 1;
        while (c < vote && i + c < nc_tot && vid[i + c] == vid[i]) ++c;
        keep = c >= vote;
      }
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) s_wk[wave] = __popcll(bal);
      __syncthreads();
      int base = fill;
      for (int w = 0; w < wave; ++w) base += s_wk[w];
      if (keep) {
        const int slot = base + __popcll(bal & ((1ULL << lane) - 1ULL));
        cid[slot] = vid[i];
        cpos[slot] = i;
      }
      fill += s_wk[0] + s_wk[1] + s_wk[2] + s_wk[3];
      vsrc += 256;
      __syncthreads();  // s_wk is rewritten by the next round
    }
    // the batch's new candidates by position: slot s holds candidate pos_base + s, whose range is
    // found by a fixed nine-step search over rstart; a thread's (up to) eight perm loads are all
    // issued before the first is used (walking the ranges one after the other, a dependent global
    // load per range, was a sixth of a C2 query's life)
    if (vote == 0) {
      int take = nc_tot - pos_base;
      if (take > FC - fill) take = FC - fill;
      constexpr int EF = FC / 256;
      // a thread takes EF CONSECUTIVE slots: one search for the first, then a walk along rstart (a
      // leaf range holds ~100 candidates: a boundary or two per thread) — sixteen independent
      // nine-step searches per thread were 13 % of a C2 query's life
      int64_t addr[EF];
      constexpr bool CONSEC = WIDE || kConsecFill;
      if constexpr (!CONSEC) {
#pragma unroll
        for (int e = 0; e < EF; ++e) {
          const int s2 = tid + 256 * e;
          const int c = pos_base + (s2 < take ? s2 : 0);
          int lo = 0;
#pragma unroll
          for (int step = kFR / 2; step > 0; step >>= 1) {
            const int m = lo + step;
            if (m < nr_tot && rstart[m] <= c) lo = m;
          }
          addr[e] = rpoff[lo] + (c - rstart[lo]);
        }
      } else {
        const int s0 = tid * EF;
        const int c0 = pos_base + (s0 < take ? s0 : 0);
        int lo = 0;
#pragma unroll
        for (int step = kFR / 2; step > 0; step >>= 1) {
          const int m = lo + step;
          if (m < nr_tot && rstart[m] <= c0) lo = m;
        }
        int rs = rstart[lo], re = rstart[lo + 1];
        int64_t rbase = rpoff[lo];
#pragma unroll
        for (int e = 0; e < EF; ++e) {
          const int c = c0 + e;
          if (s0 + e < take) {
            while (c >= re) {  // (c < nc_tot = rstart[nr_tot]: ends; empty ranges are stepped over)
              ++lo;
              rs = re;
              re = rstart[lo + 1];
              rbase = rpoff[lo];
            }
          }
          addr[e] = rbase + (c - rs);
        }
      }
      int32_t idv[EF];
#pragma unroll
      for (int e = 0; e < EF; ++e) idv[e] = (CONSEC ? tid * EF + e : tid + 256 * e) < take ? perm[addr[e]] : 0;
#pragma unroll
      for (int e = 0; e < EF; ++e) {
        const int s2 = CONSEC ? tid * EF + e : tid + 256 * e;
        if (s2 < take) {
          cid[fill + s2] = idv[e];
          if constexpr (!WIDE) cpos[fill + s2] = pos_base + s2;
        }
      }
      fill += take;
      pos_base += take;
    }
    __syncthreads();
    KSTAMP();  // batch fill
    // ---- distances of the new candidates ----
    if constexpr (CSR && PRE32) {
      if (sh16)
        batch_distances_ell16(csr.ell, csr.ell_w, cid, cdist, reinterpret_cast<const float*>(qsd + d),
                              (float)s_qn, first_new, fill, wave, lane);
      else
        batch_distances_csr32(csr.rowptr, csr.col16, csr.val32, csr.nnz, cid, cdist,
                              reinterpret_cast<const float*>(qsd + d), (float)s_qn, first_new, fill,
                              wave, lane);
    } else if constexpr (CSR)
      batch_distances_csr<TD>(csr.rowptr, csr.col, static_cast<const TD*>(csr.val), csr.nnz, cid,
                              cdist, qsd, s_qn, first_new, fill, wave, lane);
    else if constexpr (PRE32) {
      if constexpr (I8)  // int8 rows: an eighth of the f64 bytes, integer ranking values
        batch_distances_i8<16, TB>(static_cast<const uint8_t*>(Xf), d, cid, cval,
                                   reinterpret_cast<const unsigned int*>(qs32), q8k, first_new, fill, wave, 4,
                                   lane);
      else if (sh16)  // half rows: a quarter of the f64 bytes (the query stays f32)
        batch_distances<_Float16, float, 16, false, TB>(static_cast<const _Float16*>(Xf), d, cid, cval, qs32,
                                                        first_new, fill, wave, 4, lane);
      else
        batch_distances<float, float, 16, false, TB>(static_cast<const float*>(Xf), d, cid, cval, qs32,
                                                     first_new, fill, wave, 4, lane);
    } else if (pack32)
      batch_distances<TD, TA, (sizeof(TD) < 8 ? 16 : 8), false>(X, d, cid, cdist, qs, first_new, fill, wave, 4, lane);
    else
      batch_distances<TD, TA, (sizeof(TD) < 8 ? 16 : 8)>(X, d, cid, cdist, qs, first_new, fill, wave, 4, lane);
    __syncthreads();
    KSTAMP();  // distances
    // dense f64 rows ranked on butterfly sums keep kLfMargin entries more (finalize_leftfold)
    const int ksel = (!CSR && std::is_same<TD, double>::value) ? k + kLfMargin : k;
    const int nb = PRE32 ? select_packed(fill, k1, best, pb0) : pack32 ? select_packed(fill, k, best, pb0)
                                                                    : select(fill, ksel, dedup);
    best = nb;
    KSTAMP();  // selection
    if (vote > 0 ? vsrc >= nc_tot : pos_base >= nc_tot) break;
  }
  if constexpr (PRE32 && CSR) {
    // ---- refine (CSR): exact distances of the kept entries; certify the cut on SQUARED distances.
    // Error of the f32 squared distance of a row x (u = 2^-24, one rounding per operation, inputs
    // rounded to f32): a term (x_j - q_j)^2 - q_j^2 is off by at most 6.2 u T_j, T_j = (|x_j| +
    // |q_j|)^2 + q_j^2; the sum of n terms, the butterfly and the |q|^2 term pass through chains of
    // at most m = 4 ceil(n / 64) + 5 additions: (m + 1) u (sum T_j + |q|^2); sum T_j <= 2 |x|^2 +
    // 3 |q|^2.  Together E2 <= (m + 8) u (2 xmax^2 + 4 |q|^2) (clamping at 0 only moves the value
    // towards the true one).  Every dropped candidate therefore has an exact squared distance
    // >= F2 - E2, F2 = the smallest dropped f32 value.
    const bool cut = best == k1;
    const double F2 = cut ? bdist[k1 - 1] : 0.0;
    const int m = cut ? k1 - 1 : best;
    for (int i = tid; i < m; i += 256) {
      cid[i] = bid[i];
      cpos[i] = bpos[i];
    }
    __syncthreads();
    batch_distances_csr<TD>(csr.rowptr, csr.col, static_cast<const TD*>(csr.val), csr.nnz, cid,
                            cdist, qsd, s_qn, 0, m, wave, lane);
    __syncthreads();
    best = select(m, k, 0);
    if (cut && best > 0) {
      const double u = 5.9604644775390625e-08;
      const double chain = (double)(4 * ((csr.max_rowlen + 63) / 64) + 5);
      const double E2 = (chain + 8.0) * u * (2.0 * xmax * xmax + 4.0 * s_qn) * 1.01 + 1e-37;
      const double dk = bdist[best - 1];
      bool certified = (s_qn < 1e36) && (F2 < 1e37) && (F2 - E2 > dk * dk * (1.0 + 1e-14));
      if (sh16) {
        // half shadow: E2 (taken at the norm of the rounded rows, <= 1.001 xmax) bounds the f32
        // value against the squared distance of the ROUNDED row x_h; |x - x_h| <= 2^-11 |x| (+ 2^-25
        // per element in the half subnormal range; 1.05: the f64 -> f32 -> half double rounding), so
        // the exact distance of every dropped row is >= sqrt(F2 - E2') - emax
        const double E2h = E2 * 1.003;
        const double emax = 1.05 * 4.8828125e-4 * xmax + sqrt((double)csr.max_rowlen) * 3.1e-8;
        const double lo = F2 - E2h;
        certified = (s_qn < 1e36) && (F2 < 1e37) && lo > 0.0 && sqrt(lo) - emax > dk * (1.0 + 1e-13);
      }
      if (!certified) {
        if (tid == 0) {  // flag 2: the host re-runs this query with exact distances only
          ovf_flags[q] = 2u;
          atomicAdd(cand_total + 1, 1ULL);
        }
        return;
      }
    }
  } else if constexpr (PRE32) {
    // ---- refine: exact distances of the entries the f32 pass kept; certify the cut ----
    // (Round 4, measured and dropped HERE: the shard kernel's two-round refinement — exact distances
    // first for the entries near the k-th estimate, then for those a bound U on the k-th exact distance
    // cannot exclude: ~ 20 rows instead of 58 at C2.  The 58 rows are one lane-parallel round trip of a
    // workgroup that waits at barriers either way; the second round and its counting passes cost more
    // than the 35 KB they save: 1.288 against 1.274 ms per 10 000 queries on one box.)
    // The f32 distance of a row differs from the exact one by at most
    //   err(x) = 2.1 u (|x| + |q|) + (d + 2) u dist32,  u = 2^-24
    // (inputs rounded to f32, the differences, the f32 accumulation), so every candidate the
    // f32 pass dropped has an exact distance >= F - err(F), F = the smallest dropped f32
    // distance.  If that is not above the exact k-th distance the query goes to the exact path.
    const bool cut = best == k1;                    // something was dropped
    // the f32 pass keeps squared distances; the int8 pass integers, rounded to f32 in the packed
    // key (2^-24 relative, taken off here): F = s sqrt(I) bounds every dropped row from below
    const double F = !cut ? 0.0
                     : i8 ? sh8.s * (1.0 / 256.0) * sqrt(bdist[k1 - 1] * (1.0 - 1.2e-7))
                          : sqrt(bdist[k1 - 1]);
    const int m = cut ? k1 - 1 : best;
    if (wave == 0) {
      double qn = 0.0;
      for (int j = lane; j < d; j += 64) qn += (double)qs[j] * (double)qs[j];
      for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
      if (lane == 0) s_qn = sqrt(qn);
    }
    for (int i = tid; i < m; i += 256) {
      cid[i] = bid[i];
      cpos[i] = bpos[i];
    }
    __syncthreads();
    if constexpr (std::is_same<TD, double>::value) {
      for (int i = tid; i < m; i += 256)  // the reference's own arithmetic for the kept rows
        cdist[i] = leftfold_distance(X + (int64_t)cid[i] * d, qs, d);
    } else {
      batch_distances<TD, TA>(X, d, cid, cdist, qs, 0, m, wave, 4, lane);
    }
    __syncthreads();
    KSTAMP();  // refine: exact distances
    {  // the k best of the m <= 63 refined entries by (distance, position): rank by counting
      const bool mine = tid < m;
      const double di = mine ? cdist[tid] : 0.0;
      const int pi = mine ? cpos[tid] : 0, ii = mine ? cid[tid] : -1;
      int rank = 0;
      if (mine)
        rank = count_below2(cdist, cpos, m, di, pi);
      if (mine && rank < k) {
        bdist[rank] = di;
        bid[rank] = ii;
        bpos[rank] = pi;
      }
      __syncthreads();
      best = m < k ? m : k;
    }
    if (cut && best > 0) {
      const double u = 5.9604644775390625e-08;
      // + sqrt(d) * 4e-23: products in the f32 subnormal range lose relative accuracy
      // (absolute error 2^-150 each; |sqrt a - sqrt b| <= sqrt |a - b|)
      // half shadow: |x~ - x| <= 2^-11 |x| + sqrt(d) 2^-25 (subnormal halves), the query and the
      // arithmetic stay f32
      const double err = sh16 ? 1.05 * 4.8828125e-4 * xmax + 2.1 * u * s_qn + (double)(d + 2) * u * F +
                                    sqrt((double)d) * 3.1e-8
                              : 2.1 * u * (xmax + s_qn) + (double)(d + 2) * u * F + sqrt((double)d) * 4e-23;
      // (norms whose squares leave the f32 range — inf - inf = NaN entries are never ranked —
      // fail the test through s_qn; the dataset side is checked when the shadow is built)
      // F must be finite: an overflowed f32 sum (inf) orders nothing among the dropped
      // f32 DATA ranked on their half shadow: what the cut is compared with is the f32 distance the
      // all-f32 kernel ranks on, itself within (d + 2) u dist of the exact one
      double err2 = std::is_same<TD, double>::value ? err : err + (double)(d + 2) * u * (F + err) * 1.01;
      if (i8) {  // triangle inequality (Sh8): exact distance of a dropped row >= F - eq - emax
        const double e8 = (q8eq + sh8.emax) * (1.0 + 1e-12) + 1e-300;
        err2 = std::is_same<TD, double>::value ? e8 : e8 + (double)(d + 2) * u * F * 1.01;
        if (!(q8eq == q8eq)) err2 = F + 1.0;  // NaN in the query: never certified
      }
      if (!(s_qn < 1e18) || !(F < 1e30) || !(F - err2 > bdist[best - 1])) {
        if (k1 < k1_retry) {  // (uniform: shared values decide)
          k1 = k1_retry;
          // (counted: a forest whose batches retry often gets a wider first attempt, knn_dev)
          if (tid == 0 && !rerun) atomicAdd(reinterpret_cast<unsigned int*>(cand_total) + 4, 1u);
          __syncthreads();
          goto retry_wider;
        }
        if (tid == 0) {  // flag 2: the host re-runs this query with the all-f64 kernel
          ovf_flags[q] = 2u;
          atomicAdd(cand_total + 1, 1ULL);
        }
        return;
      }
    }
  }
  if constexpr (!PRE32 && !CSR && std::is_same<TD, double>::value) {
    // f64 rows ranked on butterfly sums: the k + kLfMargin kept entries again as the reference's
    // left fold, the k best of THOSE in the order of those values (finalize_leftfold)
    __syncthreads();
    finalize_leftfold(reinterpret_cast<const double*>(X), d, reinterpret_cast<const double*>(qs), best, k,
                      dedup, cdist, cpos, bid, bpos, tid, 256, [] { __syncthreads(); }, q, out_ids,
                      out_dist, out_cnt);
    return;
  }
  for (int i = tid; i < k; i += 256) {
    const bool ok = i < best;
    out_ids[q * k + i] = ok ? bid[i] : -1;
    out_dist[q * k + i] = ok ? (pack32 ? (double)sqrt(bdist[i]) : bdist[i])
                             : __longlong_as_double(0x7ff0000000000000LL);
  }
  if (tid == 0) out_cnt[q] = best;
  KSTAMP();  // final ranking, certificate, output
#undef KSTAMP
}

// ---------------------------------------------------------------------------------------
// fused query kernel, wave variant: ONE WAVE per query, four independent queries per
// workgroup and no workgroup barrier anywhere.  For small shards (a few trees per GPU, a few
// hundred candidates per query) the workgroup variant spends most of a query's life in its
// serial phases — traversal by T threads, range scan, k selection rounds with two barriers
// each — while the other waves of the CU can only cover for it if they belong to other
// queries; here every wave is its own query, so 12 queries per CU are in flight and their
// traversal latencies overlap the others' row gathers.  Lane = tree in the traversal (one
// pass: the leaf ranges go to per-tree slots and are compacted after a shuffle scan; a second
// pass only if a tree outgrew its slots), batches of kWC candidates, the wave's own entries of
// the batch in registers during the k arg-min rounds.  Same total order (distance, candidate
// position) and the same distance reduction as the workgroup variant: identical results.
// ---------------------------------------------------------------------------------------
constexpr int kWC = 512;  // candidates per batch of one wave
constexpr int kWR = 128;  // leaf ranges per query
constexpr int kWT = 64;   // trees (lane = tree)
// the variant is chosen when trees x minLeaf (~ candidates per query) is at most this
// (round 3: trees x the leaf size the topology gives, at most 700 — it was trees x minLeaf <= 1024.
// With the threshold selection, the single traversal and the in-kernel retry the workgroup kernel
// answers an 8-tree shard of C2 (8 x 122 candidates) in 0.67 ms, this one in 0.96; the 8 x 76 of a
// C4 shard stay here: 3.2 against 5.9 ms per 100 000 queries)
constexpr int64_t kWaveCandidates = 700;
constexpr int64_t kShardCandidates = 2100;  // ... and the round-4 shard kernels (see launch_fused)

__device__ inline void wave_sync() {  // LDS writes of the wave visible to all its lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ inline size_t fused_wave_bytes(int d, size_t acc_size) {
  const size_t b = (size_t)kWC * 12 + (size_t)kWR * 24 + (size_t)kFKx * 16 + (size_t)d * (acc_size + 4);
  return (b + 15) & ~(size_t)15;
}

template <class TD, class TK, bool PRE32, bool I8 = false /* PRE32 on the int8 shadow */>
__global__ __launch_bounds__(256, 3) void knn_fused_wave_kernel(
    const TD* __restrict__ X, int d, const TD* __restrict__ Q, const int32_t* __restrict__ perm,
    const double* __restrict__ thr, const double* __restrict__ mglo,
    const double* __restrict__ mghi, int64_t nodes, const TK* __restrict__ Pq, int64_t nq, int T,
    int L, int min_leaf, int64_t N, int k, int dedup, int32_t* __restrict__ out_ids,
    double* __restrict__ out_dist, int32_t* __restrict__ out_cnt, unsigned int* ovf_flags,
    unsigned int* ovf_count, unsigned long long* cand_total,
    const void* __restrict__ Xf, double xmax, int k1 /* PRE32: see knn_fused_kernel */,
    unsigned long long* dbg /* debug_stamps: phase clocks of one wave */, int sh16 = 0,
    Sh8 sh8 = Sh8{0.0, 0.0} /* s > 0: Xf is the int8 shadow (see Sh8) */) {
  typedef typename AccOf<TD>::type TA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= nq) return;  // no workgroup barrier below
  int dbgi = 0;
#define KSTAMP() do { if (dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && dbgi < 60) dbg[dbgi++] = clock64(); } while (0)
  KSTAMP();
  const bool rerun = !PRE32 && k1 == -1;  // second pass: only the queries the prefilter gave up on
  if (rerun && ovf_flags[q] != 2u) return;
  const bool pack32 = !PRE32 && sizeof(TA) == 4 && dedup == 0;  // see knn_fused_kernel
  unsigned char* base = smem + (size_t)wave * fused_wave_bytes(d, sizeof(TA));
  double* cdist = reinterpret_cast<double*>(base);                 // [kWC]
  double* bdist = cdist + kWC;                                     // [kFKx]
  int64_t* rpoff = reinterpret_cast<int64_t*>(bdist + kFKx);       // [kWR] compact, tree order
  int64_t* spoff = rpoff + kWR;                                    // [kWR] per-tree slots
  TA* qs = reinterpret_cast<TA*>(spoff + kWR);                     // [d]
  int* cid = reinterpret_cast<int*>(base + (size_t)kWC * 8 + (size_t)kFKx * 8 + (size_t)kWR * 16 +
                                    (((size_t)d * sizeof(TA) + 7) & ~(size_t)7));  // [kWC]
  int* rn = cid + kWC;                                             // [kWR]
  int* sn = rn + kWR;                                              // [kWR]
  int* bid = sn + kWR;                                             // [kFKx]
  int* bpos = bid + kFKx;                                          // [kFKx]
  float* qs32 = reinterpret_cast<float*>(bpos + kFKx);             // [d] (PRE32)

  for (int j = lane; j < d; j += 64) qs[j] = ld<TD>(Q + q * d + j);
  constexpr bool i8 = I8;
  static_assert(!I8 || PRE32, "the int8 tier is a prefilter");
  double q8k = 0.0, q8eq = 0.0;
  if constexpr (I8) {  // int8 tier: the quantised query's byte planes in the f32 copy's place (see Sh8)
    wave_sync();
    if constexpr (sizeof(TA) == 8)
      quantise_query(reinterpret_cast<const double*>(qs), nullptr, d, sh8.s,
                     reinterpret_cast<unsigned int*>(qs32), lane, 64, q8k, q8eq);
    else
      quantise_query(nullptr, reinterpret_cast<const float*>(qs), d, sh8.s,
                     reinterpret_cast<unsigned int*>(qs32), lane, 64, q8k, q8eq);
    for (int o = 32; o > 0; o >>= 1) {
      q8k += __shfl_xor(q8k, o);
      q8eq += __shfl_xor(q8eq, o);
    }
    q8k -= 1073741824.0 * (double)d;
    q8eq = sqrt(q8eq);
    wave_sync();
  } else if (PRE32)
    for (int j = lane; j < d; j += 64) qs32[j] = (float)ld<TD>(Q + q * d + j);

  // ---- traversal: lane = tree; ranges into the tree's S slots, counted in any case ----
  const int S = kWR / T;
  int my_nc = 0, my_nr = 0;
  if (lane < T) {
    const int t = lane;
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                 nq, L, min_leaf, N, [&](int off, int n) {
                   if (my_nr < S) {
                     spoff[t * S + my_nr] = (int64_t)t * N + off;
                     sn[t * S + my_nr] = n;
                   }
                   my_nc += n;
                   ++my_nr;
                 });
  }
  int inc_c = my_nc, inc_r = my_nr;  // inclusive scans over the trees
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(inc_c, o), b = __shfl_up(inc_r, o);
    if (lane >= o) {
      inc_c += a;
      inc_r += b;
    }
  }
  const int nc_tot = __shfl(inc_c, 63), nr_tot = __shfl(inc_r, 63);
  KSTAMP();  // 1: traversal
  if (nr_tot > kWR) {  // too many leaf ranges for the slab: general path
    if (lane == 0) {
      ovf_flags[q] = 1u;
      atomicAdd(ovf_count, 1u);
    }
    return;
  }
  if (lane == 0 && !rerun) atomicAdd(cand_total, (unsigned long long)nc_tot);
  if (__ballot(my_nr > S) == 0ULL) {  // the usual case: compact the slots (a lane reads its own)
    if (lane < T) {
      const int r0 = inc_r - my_nr;
      for (int r = 0; r < my_nr; ++r) {
        rpoff[r0 + r] = spoff[lane * S + r];
        rn[r0 + r] = sn[lane * S + r];
      }
    }
  } else if (lane < T) {  // some tree reached more leaves than its slots hold: second pass
    const int t = lane;
    int r = inc_r - my_nr;
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q,
                 nq, L, min_leaf, N, [&](int off, int n) {
                   rpoff[r] = (int64_t)t * N + off;
                   rn[r] = n;
                   ++r;
                 });
  }
  wave_sync();
  // rstart[r] = candidates before range r (the batches are filled by position, see
  // knn_fused_kernel), in the slot slab the compaction has just emptied: lane l scans ranges 2l, 2l + 1
  int* rstart = reinterpret_cast<int*>(spoff);  // [nr_tot + 1] <= kWR + 1 ints in kWR int64
  {
    const int a = 2 * lane < nr_tot ? rn[2 * lane] : 0, b = 2 * lane + 1 < nr_tot ? rn[2 * lane + 1] : 0;
    int incl = a + b;
    for (int o = 1; o < 64; o <<= 1) {
      const int t2 = __shfl_up(incl, o);
      if (lane >= o) incl += t2;
    }
    const int excl = incl - (a + b);
    if (2 * lane <= nr_tot) rstart[2 * lane] = excl;
    if (2 * lane + 1 <= nr_tot) rstart[2 * lane + 1] = excl + a;
    if (lane == 63) rstart[nr_tot] = incl;  // = nc_tot (the only writer of entry kWR when all slots are in use)
  }
  wave_sync();

  constexpr int E = kWC / 64;  // batch entries a lane owns: lane, lane + 64, ...
  const double kInf = __longlong_as_double(0x7ff0000000000000LL);
  // ---- selection over the batch entries [0, fill): entries below first_new are the running
  // best list (positions in bpos), the others are new candidates at positions pb0, pb0 + 1, ...;
  // ksel rounds of wave-wide arg-min by (distance, position); winners to bdist / bid / bpos ----
  // f32 pass: squared f32 distance bits << 32 | position as one key (see knn_fused_kernel)
  auto wselect_packed = [&](int fill, int first_new, int pb0, int ksel) -> int {
    unsigned long long key[E];
    unsigned int vlo = ~0u;  // the lane's smallest value
#pragma unroll
    for (int s2 = 0; s2 < E; ++s2) {
      const int i = lane + 64 * s2;
      const unsigned int pos = i < first_new ? (unsigned int)bpos[i < kFKx ? i : 0] : (unsigned int)(pb0 + (i - first_new));
      key[s2] = i < fill ? ((unsigned long long)__float_as_uint((float)cdist[i]) << 32) | pos : ~0ULL;
      if (i < fill) {
        const unsigned int v = (unsigned int)(key[s2] >> 32);
        vlo = v < vlo ? v : vlo;
      }
    }
    wave_sync();  // bpos is rewritten below, cdist becomes the list slab
    // a threshold instead of ksel rounds (see knn_fused_kernel's select_packed): the value of the
    // list's last entry once it is full, else the ksel-th smallest of the 64 per-lane minima
    unsigned int tau;
    if (first_new == ksel) {
      tau = __float_as_uint((float)bdist[ksel - 1]);
    } else {
      int cnt = 0;
#pragma unroll
      for (int j2 = 0; j2 < 64; ++j2) {  // (constant lane: v_readlane, not an LDS permute)
        const unsigned int o = (unsigned int)__builtin_amdgcn_readlane((int)vlo, j2);
        cnt += (o < vlo || (o == vlo && j2 < lane)) ? 1 : 0;
      }
      const unsigned long long hit = __ballot(cnt == ksel - 1);
      tau = __shfl(vlo, __ffsll((long long)hit) - 1);  // ~0u with fewer than ksel occupied lanes
    }
    unsigned long long* lkey = reinterpret_cast<unsigned long long*>(cdist);      // [128]
    int* lidx = reinterpret_cast<int*>(cdist + 128);                               // [128]
    unsigned int* lcnt = reinterpret_cast<unsigned int*>(cdist + 192);
    if (lane == 0) *lcnt = 0u;
    wave_sync();
#pragma unroll
    for (int s2 = 0; s2 < E; ++s2)
      if (key[s2] != ~0ULL && (unsigned int)(key[s2] >> 32) <= tau) {
        const unsigned int slot = atomicAdd(lcnt, 1u);
        if (slot < 128u) {
          lkey[slot] = key[s2];
          lidx[slot] = lane + 64 * s2;
        }
      }
    wave_sync();
    const int n = (int)*lcnt;
    if (n <= 128) {
      for (int t = lane; t < n; t += 64) {
        const unsigned long long mine = lkey[t];
        const int rank = count_below(lkey, n, mine);
        if (rank < ksel) {
          bdist[rank] = (double)__uint_as_float((unsigned int)(mine >> 32));
          bid[rank] = cid[lidx[t]];
          bpos[rank] = (int)(unsigned int)mine;
        }
      }
      wave_sync();
      return n < ksel ? n : ksel;
    }
    // fallback (values tied by the hundred): ksel rounds of wave-wide arg-min
    int nb = 0;
    while (nb < ksel) {
      unsigned long long m = key[0];
#pragma unroll
      for (int s2 = 1; s2 < E; ++s2) m = key[s2] < m ? key[s2] : m;
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t2 = __shfl_xor(m, o);
        m = t2 < m ? t2 : m;
      }
      if (m == ~0ULL) break;
#pragma unroll
      for (int s2 = 0; s2 < E; ++s2)
        if (key[s2] == m) {
          key[s2] = ~0ULL;
          bdist[nb] = (double)__uint_as_float((unsigned int)(m >> 32));
          bid[nb] = cid[lane + 64 * s2];
          bpos[nb] = (int)(unsigned int)m;
        }
      ++nb;
    }
    wave_sync();
    return nb;
  };
  auto wselect = [&](int fill, int first_new, int pb0, int ksel, int dedup) -> int {
    // ---- the lane's entries: distance and candidate position (-1 = none / consumed) ----
    double dd[E];
    int pp[E];
#pragma unroll
    for (int s = 0; s < E; ++s) {
      const int i = lane + 64 * s;
      dd[s] = i < fill ? cdist[i] : kInf;
      if (dd[s] != dd[s]) dd[s] = kInf;  // NaN ranks with +inf, by position (see knn_fused_kernel's select)
      pp[s] = i < fill ? (i < first_new ? bpos[i] : pb0 + (i - first_new)) : -1;
    }
    wave_sync();  // bpos is rewritten below
    // ---- selection: k rounds of wave-wide arg-min by (distance, position) ----
    int nb = 0;
    double last_d = -1.0;
    while (nb < ksel) {
      double bd = kInf;
      int bp = 0x7fffffff, bi = -1;
#pragma unroll
      for (int s = 0; s < E; ++s)
        if (pp[s] >= 0 && (dd[s] < bd || (dd[s] == bd && pp[s] < bp))) {
          bd = dd[s];
          bp = pp[s];
          bi = lane + 64 * s;
        }
      for (int o = 32; o > 0; o >>= 1) {
        const double od = __shfl_xor(bd, o);
        const int op = __shfl_xor(bp, o), oi = __shfl_xor(bi, o);
        if (od < bd || (od == bd && op < bp)) {
          bd = od;
          bp = op;
          bi = oi;
        }
      }
      if (bi < 0) break;  // candidates exhausted
#pragma unroll
      for (int s = 0; s < E; ++s)
        if (bi == lane + 64 * s) pp[s] = -1;  // consumed
      bool keep = true;
      if (dedup == 2 && nb > 0 && bd == last_d) keep = false;  // knnPQ: one per distance
      else if (dedup && bd == last_d)  // same id => same distance: duplicates are among equal dist
        for (int j = nb - 1; j >= 0 && bdist[j] == bd; --j)
          if (bid[j] == cid[bi]) keep = false;
      if (keep) {
        if (lane == 0) {
          bdist[nb] = bd;
          bid[nb] = cid[bi];
          bpos[nb] = bp;
        }
        wave_sync();
        ++nb;
        last_d = bd;
      }
    }
    return nb;
  };

  int best = 0, pos_base = 0;
  while (pos_base < nc_tot || best == 0) {
    // ---- fill the batch: best list first (keeps its positions), then new candidates ----
    for (int i = lane; i < best; i += 64) {
      cdist[i] = bdist[i];
      cid[i] = bid[i];
    }
    int fill = best;
    const int first_new = best, pb0 = pos_base;
    {  // by position: a seven-step search over rstart, the lane's perm loads all in flight at once
      int take = nc_tot - pos_base;
      if (take > kWC - fill) take = kWC - fill;
      int64_t addr[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int s2 = lane + 64 * e;
        const int c = pos_base + (s2 < take ? s2 : 0);
        int lo = 0;
#pragma unroll
        for (int step = kWR / 2; step > 0; step >>= 1) {
          const int m = lo + step;
          if (m < nr_tot && rstart[m] <= c) lo = m;
        }
        addr[e] = rpoff[lo] + (c - rstart[lo]);
      }
      int32_t idv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) idv[e] = lane + 64 * e < take ? perm[addr[e]] : 0;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (lane + 64 * e < take) cid[fill + lane + 64 * e] = idv[e];
      fill += take;
      pos_base += take;
    }
    wave_sync();
    KSTAMP();  // batch filled
    if constexpr (PRE32) {
      if constexpr (I8)
        batch_distances_i8<16>(static_cast<const uint8_t*>(Xf), d, cid, cdist,
                              reinterpret_cast<const unsigned int*>(qs32), q8k, first_new, fill, 0, 1, lane);
      else if (sh16)
        batch_distances<_Float16, float, 16, false>(static_cast<const _Float16*>(Xf), d, cid, cdist, qs32,
                                                    first_new, fill, 0, 1, lane);
      else
        batch_distances<float, float, 16, false>(static_cast<const float*>(Xf), d, cid, cdist, qs32,
                                                 first_new, fill, 0, 1, lane);
    } else if (pack32)
      batch_distances<TD, TA, (sizeof(TD) < 8 ? 16 : 8), false>(X, d, cid, cdist, qs, first_new, fill, 0, 1, lane);
    else
      batch_distances<TD, TA, (sizeof(TD) < 8 ? 16 : 8)>(X, d, cid, cdist, qs, first_new, fill, 0, 1, lane);
    wave_sync();
    KSTAMP();  // distances
    const int ksel = std::is_same<TD, double>::value ? k + kLfMargin : k;  // see kLfMargin
    const int nb = PRE32 ? wselect_packed(fill, first_new, pb0, k1)
                 : pack32 ? wselect_packed(fill, first_new, pb0, k)
                          : wselect(fill, first_new, pb0, ksel, dedup);
    best = nb;
    KSTAMP();  // selection
    if (pos_base >= nc_tot) break;
  }
  if constexpr (PRE32) {  // exact distances of the kept entries + certified cut (knn_fused_kernel)
    const bool cut = best == k1;
    // the f32 pass keeps squared distances, the int8 pass integers (knn_fused_kernel)
    const double F = !cut ? 0.0
                     : i8 ? sh8.s * (1.0 / 256.0) * sqrt(bdist[k1 - 1] * (1.0 - 1.2e-7))
                          : sqrt(bdist[k1 - 1]);
    const int m = cut ? k1 - 1 : best;
    double qn = 0.0;
    for (int j = lane; j < d; j += 64) qn += (double)qs[j] * (double)qs[j];
    for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
    for (int i = lane; i < m; i += 64) cid[i] = bid[i];
    wave_sync();
    if constexpr (std::is_same<TD, double>::value) {
      for (int i = lane; i < m; i += 64)  // the reference's own arithmetic (leftfold_distance)
        cdist[i] = leftfold_distance(X + (int64_t)cid[i] * d, qs, d);
    } else {
      batch_distances<TD, TA>(X, d, cid, cdist, qs, 0, m, 0, 1, lane);
    }
    wave_sync();
    KSTAMP();  // refine distances
    {  // the k best of the m <= 63 refined entries by (distance, position): every lane ranks its
       // own entry by counting (no selection rounds), the winners go to bdist / bid / bpos
      const bool mine = lane < m;
      const double di = mine ? cdist[lane] : kInf;
      const int pi = mine ? bpos[lane] : 0x7fffffff, ii = mine ? bid[lane] : -1;
      int rank = 0;
      rank = count_below2(cdist, bpos, m, di, pi);
      wave_sync();
      if (mine && rank < k) {
        bdist[rank] = di;
        bid[rank] = ii;
        bpos[rank] = pi;
      }
      wave_sync();
      best = m < k ? m : k;
    }
    KSTAMP();  // final selection
    if (cut && best > 0) {
      const double u = 5.9604644775390625e-08;
      const double err = sh16 ? 1.05 * 4.8828125e-4 * xmax + 2.1 * u * sqrt(qn) + (double)(d + 2) * u * F +
                                    sqrt((double)d) * 3.1e-8
                              : 2.1 * u * (xmax + sqrt(qn)) + (double)(d + 2) * u * F + sqrt((double)d) * 4e-23;
      double err2 = std::is_same<TD, double>::value ? err : err + (double)(d + 2) * u * (F + err) * 1.01;
      if (i8) {  // triangle inequality (Sh8)
        const double e8 = (q8eq + sh8.emax) * (1.0 + 1e-12) + 1e-300;
        err2 = std::is_same<TD, double>::value ? e8 : e8 + (double)(d + 2) * u * F * 1.01;
        if (!(q8eq == q8eq)) err2 = F + 1.0;  // NaN in the query: never certified
      }
      if (!(sqrt(qn) < 1e18) || !(F < 1e30) || !(F - err2 > bdist[best - 1])) {
        if (lane == 0) {
          ovf_flags[q] = 2u;
          atomicAdd(cand_total + 1, 1ULL);
        }
        return;
      }
    }
  }
  if constexpr (!PRE32 && std::is_same<TD, double>::value) {  // see knn_fused_kernel
    wave_sync();
    finalize_leftfold(reinterpret_cast<const double*>(X), d, reinterpret_cast<const double*>(qs), best, k,
                      dedup, cdist, cid, bid, bpos, lane, 64, [] { wave_sync(); }, q, out_ids, out_dist,
                      out_cnt);
    return;
  }
  for (int i = lane; i < k; i += 64) {
    const bool ok = i < best;
    out_ids[q * k + i] = ok ? bid[i] : -1;
    out_dist[q * k + i] = ok ? (pack32 ? (double)sqrt(bdist[i]) : bdist[i]) : kInf;
  }
  if (lane == 0) out_cnt[q] = best;
  KSTAMP();
#undef KSTAMP
}

// ---------------------------------------------------------------------------------------
// Small tree shards, round 4 (what one of G GPUs holds: a few trees, a few hundred candidates per
// query).  Two changes against knn_fused_wave_kernel's prefiltered instantiations, same answers:
//
// (1) The traversal is its own launch.  In the fused kernel it was 28 % of a query's life (47 k of
//     165 k cycles at a 4-tree C2 shard) with 4 of 64 lanes at work on a chain of dependent loads,
//     inside a wave that holds 168 registers and 12 KB of LDS for the phases that follow.
//     shard_ranges_kernel walks with lane = (query, tree) — 64 / T queries per wave, every lane
//     busy, a dozen registers — and leaves each query's leaf ranges compacted in tree order
//     (RPTree.hs:289-314 per tree, :176 the concatenation).
//
// (2) The rows that get an exact distance are chosen by VALUE, not by count.  The ranking tiers
//     bound a candidate's exact distance D by its ranking value: lower(Dh) <= D <= upper(Dh), Dh the
//     tier's distance estimate (the int8 tier: triangle inequality, Sh8; f32 / half: the rounding
//     analysis in knn_fused_kernel).  So
//       * the k-th smallest exact distance d_k is at most upper(Dh_(k)), Dh_(k) the k-th smallest
//         estimate seen so far: a candidate with lower(Dh) > upper(Dh_(k)) can be dropped at once and
//         for good (the bound only tightens as batches arrive) — the carried list;
//       * once the first entries of the list have their exact distances, U = the k-th smallest of
//         those is an upper bound of d_k too, and only list entries with lower(Dh) <= U need an exact
//         distance at all.
//     Every candidate without an exact distance is strictly farther than d_k, so the k best by
//     (exact distance, position) of the refined ones are the reference's answer (RPTree.hs:174-176) —
//     certified by construction: no fixed k' (the one-wave kernel kept k + 37 rows for every query to
//     certify the worst of ten thousand: 48 KB of f64 rows per query against 63 KB of int8 rows; the
//     band at the cut holds ~ 9), no uncertified re-runs except a list that outgrows its 128 slots.
constexpr int kSKmax = 48;  // largest k the shard kernel serves (the carried list must hold k and its band)

template <class TK>
__global__ __launch_bounds__(256) void shard_ranges_kernel(
    const double* __restrict__ thr, const double* __restrict__ mglo, const double* __restrict__ mghi,
    int64_t nodes, const TK* __restrict__ Pq, int64_t nq, int T, int L, int min_leaf, int64_t N,
    int2* __restrict__ hdr /* [nq] (leaf ranges, candidates) */, int64_t* __restrict__ rng_off /* [nq][kWR] */,
    int* __restrict__ rng_n /* [nq][kWR] */, unsigned int* ovf_flags, unsigned int* ovf_count,
    unsigned long long* cand_total) {
  const int lane = threadIdx.x & 63;
  const int qpw = 64 / T;  // queries per wave (T <= 64)
  const int64_t wave_g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ql = lane / T, t = lane - ql * T;
  const int64_t q = wave_g * qpw + ql;
  const bool on = ql < qpw && q < nq;
  int nr = 0, nc = 0, o0 = 0, n0 = 0, o1 = 0, n1 = 0;
  if (on)
    traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q, nq, L,
                 min_leaf, N, [&](int off, int n) {
                   if (nr == 0) {
                     o0 = off;
                     n0 = n;
                   } else if (nr == 1) {
                     o1 = off;
                     n1 = n;
                   }
                   nc += n;
                   ++nr;
                 });
  int ir = nr, ic = nc;  // inclusive scans over the wave, made segment-relative below
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(ir, o), b = __shfl_up(ic, o);
    if (lane >= o) {
      ir += a;
      ic += b;
    }
  }
  const int seg0 = ql * T, seg1 = seg0 + T - 1 < 63 ? seg0 + T - 1 : 63;
  const int pr = __shfl(ir, seg0 > 0 ? seg0 - 1 : 0), pc = __shfl(ic, seg0 > 0 ? seg0 - 1 : 0);
  const int er = __shfl(ir, seg1), ec = __shfl(ic, seg1);
  const int before_r = seg0 > 0 ? pr : 0, before_c = seg0 > 0 ? pc : 0;
  const int nr_tot = er - before_r, nc_tot = ec - before_c;
  const int base = ir - nr - before_r;  // this tree's first slot in the query's compact list
  unsigned long long visited = 0;
  if (on) {
    if (t == 0) hdr[q] = make_int2(nr_tot, nc_tot);
    if (nr_tot > kWR) {  // too many leaf ranges for the consumer's slab: general path
      if (t == 0) {
        ovf_flags[q] = 1u;
        atomicAdd(ovf_count, 1u);
      }
    } else {
      if (t == 0) visited = (unsigned long long)nc_tot;
      int64_t* ro = rng_off + q * kWR + base;
      int* rn = rng_n + q * kWR + base;
      if (nr >= 1) {
        ro[0] = (int64_t)t * N + o0;
        rn[0] = n0;
      }
      if (nr >= 2) {
        ro[1] = (int64_t)t * N + o1;
        rn[1] = n1;
      }
      if (nr > 2) {  // rare (both children at two nodes of one tree): walk again for the rest
        int i = 0;
        traverse<TK>(thr + t * nodes, mglo + t * nodes, mghi + t * nodes, Pq + (int64_t)t * L * nq + q, nq,
                     L, min_leaf, N, [&](int off, int n) {
                       if (i >= 2) {
                         ro[i] = (int64_t)t * N + off;
                         rn[i] = n;
                       }
                       ++i;
                     });
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
  if (lane == 0 && visited) atomicAdd(cand_total, visited);
}

__host__ __device__ inline size_t shard_wave_bytes(int d, size_t acc_size, int kSL, int WC) {
  const size_t b = (size_t)WC * 8 + (size_t)kWR * 8 + (size_t)(kWR + 4) * 4 + (size_t)kSL * 20 + 16 +
                   (((size_t)d * acc_size + 15) & ~(size_t)15) + (size_t)d * 4;
  return (b + 15) & ~(size_t)15;
}

// TIER: 1 = f32 shadow, 2 = IEEE-half shadow, 3 = int8 shadow (Sh8)
// kSL: slots of the carried list (a multiple of 64); MINB: workgroups per CU the register budget allows.
// Two shapes are instantiated: (128, 4) for shards whose candidates are one batch, (256, 3) beyond
// (the first batch's threshold comes from per-lane minima and lets more of a 980-candidate query
// through than 128 slots hold: 1.6 % of the queries of an 8-tree C2 shard).  WC = candidates per
// batch: 512, and 640 in the second shape — a C4 shard's 610 candidates per query are one batch
// instead of 512 + 98 (a batch costs a fill and a list update whatever it holds).
template <class TD, int TIER, int kSL, int MINB, int WC>
__global__ __launch_bounds__(256, MINB) void knn_shard_wave_kernel(
    const TD* __restrict__ X, int d, const TD* __restrict__ Q, const int32_t* __restrict__ perm, int64_t nq,
    int k, const int2* __restrict__ hdr, const int64_t* __restrict__ rng_off, const int* __restrict__ rng_n,
    int32_t* __restrict__ out_ids, double* __restrict__ out_dist, int32_t* __restrict__ out_cnt,
    unsigned int* ovf_flags, unsigned long long* cand_total, const void* __restrict__ Xf, double xmax, Sh8 sh8,
    int r1_pct /* first refinement round: estimates up to Dh_(k) + r1_pct % of the error bound */,
    unsigned long long* dbg) {
  typedef typename AccOf<TD>::type TA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= nq) return;  // no workgroup barrier below
  int dbgi = 0;
#define KSTAMP() do { if (dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && dbgi < 60) dbg[dbgi++] = clock64(); } while (0)
  KSTAMP();
  const int2 h = hdr[q];
  const int nr_tot = h.x, nc_tot = h.y;
  if (nr_tot > kWR) return;  // flagged by shard_ranges_kernel: the general path answers it
  constexpr int kSLpl = kSL / 64;  // list entries per lane
  unsigned char* base = smem + (size_t)wave * shard_wave_bytes(d, sizeof(TA), kSL, WC);
  unsigned long long* lkey = reinterpret_cast<unsigned long long*>(base);           // [kSL] (value bits << 32 | position)
  double* rdist = reinterpret_cast<double*>(lkey + kSL);                              // [kSL] exact distances
  int64_t* rpoff = reinterpret_cast<int64_t*>(rdist + kSL);                           // [kWR]
  TA* qs = reinterpret_cast<TA*>(rpoff + kWR);                                        // [d]
  float* cval = reinterpret_cast<float*>(base + (size_t)kSL * 16 + (size_t)kWR * 8 +
                                         (((size_t)d * sizeof(TA) + 15) & ~(size_t)15));  // [WC] ranking values
  int* cid = reinterpret_cast<int*>(cval + WC);                                       // [WC]
  int* rstart = cid + WC;                                                            // [kWR + 1]
  int* lid = rstart + kWR + 4;                                                        // [kSL]
  unsigned int* lcnt = reinterpret_cast<unsigned int*>(lid + kSL);                    // [4]
  float* qs32 = reinterpret_cast<float*>(lcnt + 4);                                   // [d] f32 copy / byte planes

  // ---- the query's leaf ranges (tree order) and rstart[r] = candidates before range r ----
  for (int r = lane; r < nr_tot; r += 64) rpoff[r] = rng_off[q * kWR + r];
  {
    const int a = 2 * lane < nr_tot ? rng_n[q * kWR + 2 * lane] : 0;
    const int b = 2 * lane + 1 < nr_tot ? rng_n[q * kWR + 2 * lane + 1] : 0;
    int incl = a + b;
    for (int o = 1; o < 64; o <<= 1) {
      const int t2 = __shfl_up(incl, o);
      if (lane >= o) incl += t2;
    }
    const int excl = incl - (a + b);
    if (2 * lane <= nr_tot) rstart[2 * lane] = excl;
    if (2 * lane + 1 <= nr_tot) rstart[2 * lane + 1] = excl + a;
  }
  // ---- the query: exact copy, ranking copy, norm ----
  double qn = 0.0;
  for (int j = lane; j < d; j += 64) {
    const TA v = ld<TD>(Q + q * d + j);
    qs[j] = v;
    qn += (double)v * (double)v;
    if constexpr (TIER != 3) qs32[j] = (float)v;
  }
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  const double qnorm = sqrt(qn);
  double q8k = 0.0, q8eq = 0.0;
  if constexpr (TIER == 3) {
    wave_sync();
    if constexpr (sizeof(TA) == 8)
      quantise_query(reinterpret_cast<const double*>(qs), nullptr, d, sh8.s, reinterpret_cast<unsigned int*>(qs32),
                     lane, 64, q8k, q8eq);
    else
      quantise_query(nullptr, reinterpret_cast<const float*>(qs), d, sh8.s, reinterpret_cast<unsigned int*>(qs32),
                     lane, 64, q8k, q8eq);
    for (int o = 32; o > 0; o >>= 1) {
      q8k += __shfl_xor(q8k, o);
      q8eq += __shfl_xor(q8eq, o);
    }
    q8k -= 1073741824.0 * (double)d;
    q8eq = sqrt(q8eq);
  }
  wave_sync();
  KSTAMP();  // 1: ranges + query

  // ---- the tier's bounds (TierBounds) ----
  const TierBounds tb = TierBounds::make(TIER, std::is_same<TD, double>::value, d, xmax, qnorm, q8eq, sh8);
  const double A = tb.A, Bt = tb.Bt, Bf = tb.Bf;
  // squares outside the f32 range, NaN: no bound holds — the exact kernel answers (as before)
  if (!(qnorm < 1e18) || !(A == A)) {
    if (lane == 0) {
      ovf_flags[q] = 2u;
      atomicAdd(cand_total + 1, 1ULL);
    }
    return;
  }
  auto dh_up = [&](unsigned int vb) -> double { return tb.dh_up(vb); };
  auto keep_bits = [&](double Ulim) -> unsigned int { return tb.keep_bits(Ulim); };
  auto upper = [&](double dh) -> double { return tb.upper(dh); };

  constexpr int E = WC / 64;
  int n_list = 0, pos_base = 0;
  while (pos_base < nc_tot) {
    // ---- fill: candidate s of the batch by position (a search over rstart, the perm loads in flight together)
    const int take = nc_tot - pos_base < WC ? nc_tot - pos_base : WC;
    {
      int64_t addr[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int s2 = lane + 64 * e;
        const int c = pos_base + (s2 < take ? s2 : 0);
        int lo = 0;
#pragma unroll
        for (int step = kWR / 2; step > 0; step >>= 1) {
          const int m = lo + step;
          if (m < nr_tot && rstart[m] <= c) lo = m;
        }
        addr[e] = rpoff[lo] + (c - rstart[lo]);
      }
      int32_t idv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) idv[e] = lane + 64 * e < take ? perm[addr[e]] : 0;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (lane + 64 * e < take) cid[lane + 64 * e] = idv[e];
    }
    wave_sync();
    KSTAMP();  // batch filled
    if constexpr (TIER == 3)
      batch_distances_i8<16, float>(static_cast<const uint8_t*>(Xf), d, cid, cval,
                                    reinterpret_cast<const unsigned int*>(qs32), q8k, 0, take, 0, 1, lane);
    else if constexpr (TIER == 2)
      batch_distances<_Float16, float, 16, false, float>(static_cast<const _Float16*>(Xf), d, cid, cval, qs32, 0,
                                                         take, 0, 1, lane);
    else
      batch_distances<float, float, 16, false, float>(static_cast<const float*>(Xf), d, cid, cval, qs32, 0, take,
                                                      0, 1, lane);
    wave_sync();
    KSTAMP();  // ranking values
    // ---- the batch against the carried list ----
    unsigned long long key[E];
    unsigned int vlo = ~0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = lane + 64 * e;
      key[e] = i < take ? ((unsigned long long)__float_as_uint(cval[i]) << 32) | (unsigned int)(pos_base + i) : ~0ULL;
      const unsigned int v = (unsigned int)(key[e] >> 32);
      vlo = v < vlo ? v : vlo;
    }
    // an upper bound of the k-th smallest value seen: the list's k-th entry, or (first batch) the
    // k-th smallest of the 64 per-lane minima
    unsigned int tau;
    if (n_list >= k) {
      tau = (unsigned int)(lkey[k - 1] >> 32);
    } else {
      int cnt = 0;
#pragma unroll
      for (int j2 = 0; j2 < 64; ++j2) {
        const unsigned int o = (unsigned int)__builtin_amdgcn_readlane((int)vlo, j2);
        cnt += (o < vlo || (o == vlo && j2 < lane)) ? 1 : 0;
      }
      const unsigned long long hit = __ballot(cnt == k - 1);
      tau = __shfl(vlo, __ffsll((long long)hit) - 1);  // ~0u with fewer than k occupied lanes
    }
    const unsigned int thr0 = tau >= 0x7f800000u ? 0x7fc00000u : keep_bits(upper(dh_up(tau)));
    if (lane == 0) *lcnt = (unsigned int)n_list;
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (key[e] != ~0ULL && (unsigned int)(key[e] >> 32) <= thr0) {
        const unsigned int slot = atomicAdd(lcnt, 1u);
        if (slot < (unsigned int)kSL) {
          lkey[slot] = key[e];
          lid[slot] = cid[lane + 64 * e];
        }
      }
    wave_sync();
    const int n = (int)*lcnt;
    if (n > kSL) {  // more candidates inside the band than the list holds (values tied by the hundred)
      if (lane == 0) {
        ovf_flags[q] = 2u;
        atomicAdd(cand_total + 1, 1ULL);
      }
      return;
    }
    {  // order the list by (value, position): rank by counting, kSLpl entries per lane
      unsigned long long mk[kSLpl];
      int mi[kSLpl], mr[kSLpl];
#pragma unroll
      for (int w = 0; w < kSLpl; ++w) {
        const int i = lane + 64 * w;
        mk[w] = i < n ? lkey[i] : ~0ULL;
        mi[w] = i < n ? lid[i] : 0;
      }
#pragma unroll
      for (int w = 0; w < kSLpl; ++w) mr[w] = lane + 64 * w < n ? count_below(lkey, n, mk[w]) : 0;
      wave_sync();
#pragma unroll
      for (int w = 0; w < kSLpl; ++w)
        if (lane + 64 * w < n) {
          lkey[mr[w]] = mk[w];
          lid[mr[w]] = mi[w];
        }
      wave_sync();
    }
    n_list = n;
    if (n > k) {  // trim to the band of the (now exact) k-th smallest value: a prefix of the ordered list
      const unsigned int vk = (unsigned int)(lkey[k - 1] >> 32);
      const unsigned int thr1 = vk >= 0x7f800000u ? 0x7fc00000u : keep_bits(upper(dh_up(vk)));
      int c = 0;
#pragma unroll
      for (int w = 0; w < kSLpl; ++w)
        c += __popcll(__ballot(lane + 64 * w < n && (unsigned int)(lkey[lane + 64 * w] >> 32) <= thr1));
      n_list = c;
    }
    pos_base += take;
    KSTAMP();  // list updated
  }

  // ---- exact distances, first round: the entries whose estimate is within r1_pct % of the error bound
  // above the k-th estimate (the k-th exact distance is about the k-th estimate: what lies beyond
  // that is what the second round would have to add)
  auto exact = [&](int from, int to) {
    if constexpr (std::is_same<TD, double>::value) {
      for (int i = from + lane; i < to; i += 64)  // the reference's own arithmetic (leftfold_distance)
        rdist[i] = leftfold_distance(reinterpret_cast<const double*>(X) + (int64_t)lid[i] * d,
                                     reinterpret_cast<const double*>(qs), d);
    } else {
      batch_distances<TD, TA>(X, d, lid, rdist, qs, from, to, 0, 1, lane);
    }
  };
  auto prefix_le = [&](unsigned int tb) -> int {
    int c = 0;
#pragma unroll
    for (int w = 0; w < kSLpl; ++w)
      c += __popcll(__ballot(lane + 64 * w < n_list && (unsigned int)(lkey[lane + 64 * w] >> 32) <= tb));
    return c;
  };
  // rank of entry (di, pi) among the refined entries [0, m) by (exact distance, position); NaN behind
  // every number, NaNs by position
  auto rank_of = [&](int m, double di, unsigned int pi) -> int {
    int rank = 0;
    if (di != di) {
      for (int j = 0; j < m; ++j) rank += rdist[j] == rdist[j] || (unsigned int)lkey[j] < pi;
      return rank;
    }
    int j = 0;
    for (; j + 8 <= m; j += 8) {
      double a[8];
      unsigned int b[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a[w] = rdist[j + w];
        b[w] = (unsigned int)lkey[j + w];
      }
#pragma unroll
      for (int w = 0; w < 8; ++w) rank += a[w] < di || (a[w] == di && b[w] < pi);
    }
    for (; j < m; ++j) rank += rdist[j] < di || (rdist[j] == di && (unsigned int)lkey[j] < pi);
    return rank;
  };
  int m = n_list;
  if (n_list > k) {
    const double dk = dh_up((unsigned int)(lkey[k - 1] >> 32));
    int r1 = prefix_le(keep_bits((dk + (A + Bt * dk) * (0.01 * (double)r1_pct) - A) * (1.0 - Bf)));
    // (keep_bits(U) keeps Dh <= (U / (1 - Bf) + A) / (1 - Bt): with the U above that is about
    // Dh <= dk + r1_pct % of the bound)
    r1 = r1 < k ? k : r1;
    exact(0, r1);
    wave_sync();
    KSTAMP();  // first round
    m = r1;
    if (r1 < n_list) {
      // U = the k-th smallest exact distance so far; entries beyond r1 with lower(Dh) <= U follow
      double U = 0.0;
      {
        double dv[kSLpl];
        unsigned long long hit[kSLpl];
#pragma unroll
        for (int w = 0; w < kSLpl; ++w) {
          const int i = lane + 64 * w;
          dv[w] = i < r1 ? rdist[i] : 0.0;
          const int rk = i < r1 ? rank_of(r1, dv[w], (unsigned int)lkey[i]) : -1;
          hit[w] = __ballot(rk == k - 1);
        }
#pragma unroll
        for (int w = 0; w < kSLpl; ++w)
          if (hit[w]) U = __shfl(dv[w], __ffsll((long long)hit[w]) - 1);  // (uniform: exactly one hit)
      }
      const int r2 = U == U ? prefix_le(keep_bits(U)) : n_list;
      if (r2 > r1) {
        exact(r1, r2);
        wave_sync();
        m = r2;
      }
      KSTAMP();  // second round
    }
  } else {
    exact(0, n_list);
    wave_sync();
    KSTAMP();
  }
  // ---- the k best of the m refined entries by (exact distance, position) ----
  {
#pragma unroll
    for (int w = 0; w < kSLpl; ++w) {
      const int i = lane + 64 * w;
      if (i < m) {
        const double di = rdist[i];
        const int rk = rank_of(m, di, (unsigned int)lkey[i]);
        if (rk < k) {
          out_ids[q * k + rk] = lid[i];
          out_dist[q * k + rk] = di;
        }
      }
    }
    const int best = m < k ? m : k;
    for (int i = best + lane; i < k; i += 64) {
      out_ids[q * k + i] = -1;
      out_dist[q * k + i] = __longlong_as_double(0x7ff0000000000000LL);
    }
    if (lane == 0) out_cnt[q] = best;
  }
  KSTAMP();
#undef KSTAMP
}

// CSR data and CSR queries: the query is densified into LDS; distance of a sparse row x:
// d^2 = |q|^2 + sum_{j in nz(x)} ((x_j - q_j)^2 - q_j^2)   (true Euclidean distance)
template <class TD>
__global__ __launch_bounds__(256) void topk_csr_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const TD* __restrict__ val, int64_t nnz, int d, const int64_t* __restrict__ qrowptr,
    const int32_t* __restrict__ qcol, const TD* __restrict__ qval,
    const int32_t* __restrict__ perm, const Range* __restrict__ ranges,
    const int64_t* __restrict__ rng_off, int T, int k, int dedup, int refm /* RPT_KNN_METRIC_REFERENCE */,
    int32_t* __restrict__ out_ids, double* __restrict__ out_dist, int32_t* __restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Entry* buf = reinterpret_cast<Entry*>(smem);
  int* scratch = reinterpret_cast<int*>(smem + sizeof(Entry) * kBuf);
  double* qs = reinterpret_cast<double*>(smem + (sizeof(Entry) + 4) * kBuf);  // [d]
  __shared__ double s_qn2;
  const int64_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < d; j += blockDim.x) qs[j] = 0.0;
  __syncthreads();
  for (int64_t j = qrowptr[q] + threadIdx.x; j < qrowptr[q + 1]; j += blockDim.x)
    qs[qcol[j]] = (double)qval[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    double s = 0;
    for (int j = lane; j < d; j += 64) s += qs[j] * qs[j];
    s = wave_sum(s);
    if (lane == 0) s_qn2 = s;
  }
  __syncthreads();
  const double qn2 = s_qn2;
  int best = 0, filled = 0;
  const int cap = kBuf;
  for (int64_t r = rng_off[q * T]; r < rng_off[(q + 1) * T]; ++r) {
    const Range rg = ranges[r];
    int done = 0;
    while (done < rg.n) {
      int take = rg.n - done;
      if (take > cap - filled) take = cap - filled;
      // RPT_KNN_METRIC_REFERENCE: metricSSL2 (Internal.hs:389-393) as the reference evaluates it —
      // diffSS = binSS (-) 0 (:435-450), a merge of the two index lists that STOPS when either is
      // exhausted (the tail of the longer one is dropped), the squares summed by a left fold in
      // merge order.  One thread per candidate: the walk is sequential by definition.
      if (refm) {
        const int64_t qa = qrowptr[q], qb = qrowptr[q + 1];
        for (int i = threadIdx.x; i < take; i += blockDim.x) {
          const int id = perm[rg.poff + done + i];
          int64_t i1 = rowptr[id];
          const int64_t b1 = rowptr[id + 1];
          int64_t i2 = qa;
          double acc = 0.0;
          while (i1 < b1 && i2 < qb) {
            const int il = col[i1], ir = qcol[i2];
            double df;
            if (il == ir) {
              df = (double)val[i1] - (double)qval[i2];
              ++i1;
              ++i2;
            } else if (il < ir) {
              df = (double)val[i1] - 0.0;
              ++i1;
            } else {
              df = 0.0 - (double)qval[i2];
              ++i2;
            }
            acc = acc + df * df;
          }
          buf[filled + i] = Entry{sqrt(acc), rg.pos + done + i, id};
        }
      } else
      // sixteen lanes per candidate row, sixteen rows per wave in flight (csr_rows_dist2)
      for (int i0 = wave * 16; i0 < take; i0 += 64) {
        constexpr int U = 4;
        int64_t ra[U], rb[U];
        int id[U];
        double s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * 4 + (lane >> 4);
          const bool ok = i < take;
          id[u] = perm[rg.poff + done + (ok ? i : 0)];
          ra[u] = rowptr[id[u]];
          rb[u] = ok ? rowptr[id[u] + 1] : ra[u];
        }
        csr_rows_dist2<TD, U>(col, val, nnz, ra, rb, qs, lane & 15, s);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * 4 + (lane >> 4);
          const double t = s[u] + qn2;
          if ((lane & 15) == 0 && i < take)
            buf[filled + i] = Entry{sqrt(t > 0 ? t : 0.0), rg.pos + done + i, id[u]};
        }
      }
      filled += take;
      done += take;
      __syncthreads();
      if (filled == cap) {
        best = merge_best(buf, filled, k, dedup, scratch);
        filled = best;
        __syncthreads();
      }
    }
  }
  if (filled > best || best == 0) {
    best = merge_best(buf, filled, k, dedup, scratch);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const bool ok = i < best;
    out_ids[q * k + i] = ok ? buf[i].id : -1;
    out_dist[q * k + i] = ok ? buf[i].dist : __longlong_as_double(0x7ff0000000000000LL);
  }
  if (threadIdx.x == 0) out_cnt[q] = best;
}

// knnH: distances of the points of the selected leaf ranges, written in result order
// (sel[r].pos = position of the range's first point in the query's result).  One block per
// query, one wave per point.
template <class TD>
__global__ __launch_bounds__(256) void dist_sel_dense_kernel(
    const TD* __restrict__ X, int d, const TD* __restrict__ Q, const int32_t* __restrict__ perm,
    const Range* __restrict__ sel, const int64_t* __restrict__ sel_off,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_ids,
    double* __restrict__ out_dist) {
  typedef typename AccOf<TD>::type TA;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TA* qs = reinterpret_cast<TA*>(smem);
  const int64_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < d; j += blockDim.x) qs[j] = ld<TD>(Q + q * d + j);
  __syncthreads();
  if constexpr (std::is_same<TD, double>::value) {
    // knnH returns whole buckets: every point's distance as metricDDL2's left fold (one thread per
    // point; see leftfold_distance)
    for (int64_t r = sel_off[q]; r < sel_off[q + 1]; ++r) {
      const Range rg = sel[r];
      for (int i = threadIdx.x; i < rg.n; i += blockDim.x) {
        const int id = perm[rg.poff + i];
        out_ids[out_off[q] + rg.pos + i] = id;
        out_dist[out_off[q] + rg.pos + i] = leftfold_distance(X + (int64_t)id * d, qs, d);
      }
    }
    return;
  }
  for (int64_t r = sel_off[q]; r < sel_off[q + 1]; ++r) {
    const Range rg = sel[r];
    for (int i = wave; i < rg.n; i += 4) {
      const int id = perm[rg.poff + i];
      TA s = (TA)0;
      for (int j = lane; j < d; j += 64) {
        const TA df = ld<TD>(X + (int64_t)id * d + j) - qs[j];
        s += df * df;
      }
      const TA tot = wave_sum(s);
      if (lane == 0) {
        out_ids[out_off[q] + rg.pos + i] = id;
        out_dist[out_off[q] + rg.pos + i] = (double)sqrt((double)tot);
      }
    }
  }
}

template <class TD>
__global__ __launch_bounds__(256) void dist_sel_csr_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const TD* __restrict__ val, int d, const int64_t* __restrict__ qrowptr,
    const int32_t* __restrict__ qcol, const TD* __restrict__ qval,
    const int32_t* __restrict__ perm, const Range* __restrict__ sel,
    const int64_t* __restrict__ sel_off, const int64_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, double* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* qs = reinterpret_cast<double*>(smem);  // [d] densified query
  __shared__ double s_qn2;
  const int64_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < d; j += blockDim.x) qs[j] = 0.0;
  __syncthreads();
  for (int64_t j = qrowptr[q] + threadIdx.x; j < qrowptr[q + 1]; j += blockDim.x)
    qs[qcol[j]] = (double)qval[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    double s = 0;
    for (int j = lane; j < d; j += 64) s += qs[j] * qs[j];
    s = wave_sum(s);
    if (lane == 0) s_qn2 = s;
  }
  __syncthreads();
  const double qn2 = s_qn2;
  for (int64_t r = sel_off[q]; r < sel_off[q + 1]; ++r) {
    const Range rg = sel[r];
    for (int i = wave; i < rg.n; i += 4) {
      const int id = perm[rg.poff + i];
      double s = 0;
      for (int64_t j = rowptr[id] + lane; j < rowptr[id + 1]; j += 64) {
        const double qj = qs[col[j]];
        const double df = (double)val[j] - qj;
        s += df * df - qj * qj;  // same formula as topk_csr_kernel: true Euclidean distance
      }
      s = wave_sum(s) + qn2;
      if (lane == 0) {
        out_ids[out_off[q] + rg.pos + i] = id;
        out_dist[out_off[q] + rg.pos + i] = sqrt(s > 0 ? s : 0.0);
      }
    }
  }
}

// multi-GPU merge: G shard results per query; shard g's [nq][k] ids / distances and [nq] counts
// start g * sstride BYTES after the base pointers (shard-major arrays: sstride = the array's
// own size; one packed exchange record per shard: sstride = the record size)
__global__ __launch_bounds__(256) void merge_kernel(const int32_t* __restrict__ ids,
                                                    const double* __restrict__ dist,
                                                    const int32_t* __restrict__ cnt,
                                                    int64_t ids_stride, int64_t dist_stride,
                                                    int64_t cnt_stride, int G, int64_t nq, int k,
                                                    int dedup, int32_t* __restrict__ out_ids,
                                                    double* __restrict__ out_dist,
                                                    int32_t* __restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int total = G * k;
  int np = 1;
  while (np < total) np <<= 1;
  Entry* buf = reinterpret_cast<Entry*>(smem);
  int* scratch = reinterpret_cast<int*>(smem + sizeof(Entry) * np);
  const int64_t q = blockIdx.x;
  auto shard_cnt = [&](int g) {
    return reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(cnt) + g * cnt_stride)[q];
  };
  // invalid slots sort last: dist = +inf, pos keeps shard order
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int g = i / k, r = i % k;
    const bool ok = r < shard_cnt(g);
    const int64_t src = q * k + r;
    const double* dg =
        reinterpret_cast<const double*>(reinterpret_cast<const char*>(dist) + g * dist_stride);
    const int32_t* ig =
        reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(ids) + g * ids_stride);
    buf[i] = Entry{ok ? dg[src] : __longlong_as_double(0x7ff0000000000000LL), i, ok ? ig[src] : -1};
  }
  __syncthreads();
  int valid = 0;
  for (int g = 0; g < G; ++g) valid += shard_cnt(g);
  int best = merge_best(buf, total, k, dedup, scratch);
  // merge_best counted +inf padding as entries: clamp to the valid ones (dedup can only
  // shrink further; invalid entries have id -1 and sort last)
  __syncthreads();
  int m = 0;
  for (int i = 0; i < best; ++i) m += buf[i].id >= 0;  // every thread computes the same
  best = m < valid ? m : valid;
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const bool ok = i < best;
    out_ids[q * k + i] = ok ? buf[i].id : -1;
    out_dist[q * k + i] = ok ? buf[i].dist : __longlong_as_double(0x7ff0000000000000LL);
  }
  if (threadIdx.x == 0) out_cnt[q] = best;
}

// ---- host side ----------------------------------------------------------------------------
struct QueryPlan {
  DevBuf<char> Pq;  // [T][L][nq] projections of the queries
  DevBuf<int> cnt_cand, cnt_rng;
  DevBuf<int64_t> cand_off, rng_off;
  DevBuf<Range> ranges;
  int64_t total_cand = 0, total_rng = 0;
};

template <class TK>
int32_t make_plan_t(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* q, QueryPlan& pl) {
  const int64_t nq = q->n;
  const int T = f->T, L = f->L;
  hipStream_t st = ctx->stream;
  ProfScope ps(ctx, RPT_PROF_KNN_PLAN);
  RPT_TRY(pl.Pq.alloc((size_t)T * L * nq * sizeof(TK) + 16));
  if (L > 0 && nq > 0) RPT_TRY(project_columns(ctx, q, f->R.p, T * L, f->mode, pl.Pq.p));
  const int64_t m = nq * T;
  RPT_TRY(pl.cnt_cand.alloc((size_t)m));
  RPT_TRY(pl.cnt_rng.alloc((size_t)m));
  RPT_TRY(pl.cand_off.alloc((size_t)m + 1));
  RPT_TRY(pl.rng_off.alloc((size_t)m + 1));
  if (nq == 0) return RPT_OK;
  const int threads = T <= 64 ? 64 : (T <= 128 ? 128 : 256);
  const TK* Pq = reinterpret_cast<const TK*>(pl.Pq.p);
  if (f->xtopo)
    hipLaunchKernelGGL(count_x_kernel<TK>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, (const int8_t*)f->xkind.p,
                       (const int64_t*)f->xoff.p, (const int64_t*)f->xlen.p, Pq, nq, T, L,
                       pl.cnt_cand.p, pl.cnt_rng.p);
  else
    hipLaunchKernelGGL(count_kernel<TK>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, Pq, nq, T, L, f->min_leaf, f->n,
                       pl.cnt_cand.p, pl.cnt_rng.p);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, pl.cnt_cand.p, m, pl.cand_off.p);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, pl.cnt_rng.p, m, pl.rng_off.p);
  RPT_HIP(hipGetLastError());
  int64_t tot[2];
  RPT_HIP(hipMemcpyAsync(&tot[0], pl.cand_off.p + m, 8, hipMemcpyDeviceToHost, st));
  RPT_HIP(hipMemcpyAsync(&tot[1], pl.rng_off.p + m, 8, hipMemcpyDeviceToHost, st));
  RPT_HIP(stream_sync(st));
  pl.total_cand = tot[0];
  pl.total_rng = tot[1];
  RPT_TRY(pl.ranges.alloc((size_t)pl.total_rng));
  if (f->xtopo)
    hipLaunchKernelGGL(ranges_x_kernel<TK>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, (const int8_t*)f->xkind.p,
                       (const int64_t*)f->xoff.p, (const int64_t*)f->xlen.p, Pq, nq, T, L, f->n,
                       pl.cand_off.p, pl.rng_off.p, pl.ranges.p);
  else
    hipLaunchKernelGGL(ranges_kernel<TK>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, Pq, nq, T, L, f->min_leaf, f->n,
                       pl.cand_off.p, pl.rng_off.p, pl.ranges.p);
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

int32_t make_plan(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* q, QueryPlan& pl) {
  RPT_ARG(proj_dtype(q->dtype) == f->pdtype,
          "query dtype must have the forest's projection type (f64 vs f32/bf16)");
  if (f->pdtype == RPT_F64) return make_plan_t<double>(ctx, f, q, pl);
  return make_plan_t<float>(ctx, f, q, pl);
}

size_t topk_smem(int d, size_t acc_size) { return (sizeof(Entry) + 4) * kBuf + (size_t)d * acc_size; }

}  // namespace

int32_t candidates(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* q, int64_t* off_host,
                   int32_t* ids_host, int64_t cap, int64_t* total) {
  QueryPlan pl;
  RPT_TRY(make_plan(ctx, f, q, pl));
  *total = pl.total_cand;
  const int64_t m = q->n * f->T;
  if (off_host) {
    if (m > 0) RPT_HIP(hipMemcpy(off_host, pl.cand_off.p, (size_t)(m + 1) * 8, hipMemcpyDeviceToHost));
    else off_host[0] = 0;
  }
  if (ids_host && pl.total_cand > 0) {
    RPT_ARG(cap >= pl.total_cand, "ids capacity too small");
    DevBuf<int32_t> ids;
    RPT_TRY(ids.alloc((size_t)pl.total_cand));
    hipLaunchKernelGGL(expand_kernel, dim3((unsigned)q->n), dim3(128), 0, ctx->stream,
                       f->perm.p, pl.ranges.p, pl.rng_off.p, pl.cand_off.p, f->T, ids.p);
    RPT_HIP(hipGetLastError());
    RPT_HIP(stream_sync(ctx->stream));
    RPT_HIP(hipMemcpy(ids_host, ids.p, (size_t)pl.total_cand * 4, hipMemcpyDeviceToHost));
  }
  return RPT_OK;
}

template <class TD>
static int32_t launch_topk_dense(rpt_ctx* ctx, const rpt_dataset* data, const rpt_dataset* q,
                                 const int32_t* perm, const Range* ranges, const int64_t* rng_off,
                                 int T, int identity, int k, int dedup, int32_t* ids,
                                 double* dist, int32_t* cnt) {
  typedef typename AccOf<TD>::type TA;
  const size_t smem = topk_smem(data->d, sizeof(TA));
  if (smem > 64 * 1024)
    RPT_HIP(hipFuncSetAttribute((const void*)topk_dense_kernel<TD>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  ProfScope ps(ctx, RPT_PROF_KNN_TOPK);
  hipLaunchKernelGGL(topk_dense_kernel<TD>, dim3((unsigned)q->n), dim3(256), smem, ctx->stream,
                     (const TD*)data->X, data->d, (const TD*)q->X, perm, ranges, rng_off, T,
                     data->n, identity, k, dedup, ids, dist, cnt);
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

constexpr int kMergeMax = 4096;  // entries of one merge launch (LDS)
static inline int prefilter_keep(int k) { return k + (k / 2 > 6 ? k / 2 : 6); }

// f32 shadow of a dense f64 dataset + its largest row norm (one wave per row)
__global__ __launch_bounds__(256) void shadow32_kernel(const double* __restrict__ X, int64_t n, int d,
                                                       float* __restrict__ Xf,
                                                       unsigned long long* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double mx = 0.0;
  for (int64_t r = row0; r < n; r += (int64_t)gridDim.x * 4) {
    double s = 0.0;
    for (int j = lane; j < d; j += 64) {
      const double v = X[r * d + j];
      Xf[r * d + j] = (float)v;
      s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (!(s <= mx)) mx = s == s ? s : __longlong_as_double(0x7ff0000000000000LL);  // NaN row -> +inf
  }
  if (lane == 0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));  // mx >= 0
}

// Builds the f32 shadow once per dataset.  The shadow is an optimisation: whenever it cannot be
// had (no memory for it, a failed launch or copy, rows whose squares leave the f32 range, NaN
// rows) the dataset is marked "tried, unusable" (max_norm = -2) and the caller continues on the
// all-f64 path — never an error.
static int32_t ensure_shadow(rpt_ctx* ctx, const rpt_dataset* data) {
  if (data->shadow32 || data->max_norm == -2.0) return RPT_OK;
  void* p = nullptr;
  DevBuf<unsigned long long> mb;
  unsigned long long bits = 0;
  auto give_up = [&]() {
    if (p) dev_free(p);
    (void)hipGetLastError();       // the exact path starts from a clean error state
    data->max_norm = -2.0;
    return RPT_OK;
  };
  if (dev_alloc(&p, (size_t)data->n * data->d * sizeof(float)) != hipSuccess) {
    p = nullptr;
    return give_up();
  }
  if (mb.alloc(1) != RPT_OK) return give_up();
  if (hipMemsetAsync(mb.p, 0, 8, ctx->stream) != hipSuccess) return give_up();
  int64_t blocks = (data->n + 3) / 4;
  if (blocks > (int64_t)ctx->n_cu * 16) blocks = (int64_t)ctx->n_cu * 16;
  hipLaunchKernelGGL(shadow32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                     (const double*)data->X, data->n, data->d, (float*)p, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(&bits, mb.p, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double m2;
  std::memcpy(&m2, &bits, 8);
  data->max_norm = std::sqrt(m2) * (1.0 + 1e-12);
  // squares would leave the f32 range, or a row holds a NaN (the kernel folds a NaN sum into
  // +inf, see shadow32_kernel): no shadow
  if (!(data->max_norm < 1e18)) return give_up();
  data->shadow32 = (float*)p;
  return RPT_OK;
}

// IEEE-half shadow of a dense f64 dataset (one wave per row) + the largest |element| (bits in
// max_bits[0]); built after the f32 shadow, whose row norm it shares
template <class TIn>
__global__ __launch_bounds__(256) void shadow16_kernel(const TIn* __restrict__ X, int64_t n, int d,
                                                       _Float16* __restrict__ Xh,
                                                       unsigned long long* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double mx = 0.0, mn2 = 0.0;  // largest |element|, largest squared row norm
  for (int64_t r = row0; r < n; r += (int64_t)gridDim.x * 4) {
    double s2 = 0.0;
    for (int j = lane; j < d; j += 64) {
      const double v = (double)X[r * d + j];
      Xh[r * d + j] = (_Float16)(float)v;
      const double a = fabs(v);
      if (!(a <= mx)) mx = a == a ? a : __longlong_as_double(0x7ff0000000000000LL);
      s2 += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    if (!(s2 <= mn2)) mn2 = s2 == s2 ? s2 : __longlong_as_double(0x7ff0000000000000LL);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(mx, o);
    mx = t > mx ? t : mx;
  }
  if (lane == 0) {
    atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));
    atomicMax(max_bits + 1, (unsigned long long)__double_as_longlong(mn2));
  }
}

// The half shadow, once per dataset: allowed to fail like the f32 one (no memory, elements outside
// the half range): the f32 tier then ranks.
static int32_t ensure_shadow16(rpt_ctx* ctx, const rpt_dataset* data) {
  if (data->shadow16_state != 0) return RPT_OK;
  data->shadow16_state = -1;
  void* p = nullptr;
  DevBuf<unsigned long long> mb;
  unsigned long long bits[2] = {0, 0};
  auto give_up = [&]() {
    if (p) dev_free(p);
    (void)hipGetLastError();
    return RPT_OK;
  };
  if (data->dtype == RPT_BF16 || data->csr) return RPT_OK;
  if (dev_alloc(&p, (size_t)data->n * data->d * 2 + 16) != hipSuccess) {
    p = nullptr;
    return give_up();
  }
  if (mb.alloc(2) != RPT_OK) return give_up();
  if (hipMemsetAsync(mb.p, 0, 16, ctx->stream) != hipSuccess) return give_up();
  int64_t blocks = (data->n + 3) / 4;
  if (blocks > (int64_t)ctx->n_cu * 16) blocks = (int64_t)ctx->n_cu * 16;
  if (data->dtype == RPT_F64)
    hipLaunchKernelGGL(shadow16_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const double*)data->X, data->n, data->d, (_Float16*)p, mb.p);
  else
    hipLaunchKernelGGL(shadow16_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const float*)data->X, data->n, data->d, (_Float16*)p, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(bits, mb.p, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double mabs, mn2;
  std::memcpy(&mabs, &bits[0], 8);
  std::memcpy(&mn2, &bits[1], 8);
  if (!(mabs < 6.0e4)) return give_up();  // beyond the half range (or NaN / inf)
  if (data->max_norm < 0.0) {             // f32 data: no f32 shadow pass has measured the norms
    if (data->max_norm == -2.0) return give_up();
    data->max_norm = std::sqrt(mn2) * (1.0 + 1e-12);
    if (!(data->max_norm < 1e18)) return give_up();
  }
  data->shadow16 = (uint16_t*)p;
  data->shadow16_state = 1;
  return RPT_OK;
}

// int8 shadow (Sh8): pass 1 = the largest |element| (bits in max_bits[0]; NaN / inf -> +inf)
template <class TIn>
__global__ __launch_bounds__(256) void maxabs_kernel(const TIn* __restrict__ X, int64_t count,
                                                     unsigned long long* __restrict__ max_bits) {
  double mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const double a = fabs((double)ld<TIn>(X + i));
    if (!(a <= mx)) mx = a == a ? a : __longlong_as_double(0x7ff0000000000000LL);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(mx, o);
    mx = t > mx ? t : mx;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));
}
// pass 2 = the rows (one wave per row) and the largest squared row error |x - s c|^2 (max_bits[1])
template <class TIn>
__global__ __launch_bounds__(256) void shadow8_kernel(const TIn* __restrict__ X, int64_t n, int d, double s,
                                                      int8_t* __restrict__ X8,
                                                      unsigned long long* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const double inv = 1.0 / s;
  double mx = 0.0;
  for (int64_t r = row0; r < n; r += (int64_t)gridDim.x * 4) {
    double e2 = 0.0;
    for (int j = lane; j < d; j += 64) {
      const double v = (double)ld<TIn>(X + r * d + j);
      double c = rint(v * inv);
      c = c < -127.0 ? -127.0 : (c > 127.0 ? 127.0 : c);
      X8[r * d + j] = (int8_t)(uint8_t)((int)c + 128);  // offset binary
      const double e = v - s * c;
      e2 += e * e;
    }
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
    if (!(e2 <= mx)) mx = e2 == e2 ? e2 : __longlong_as_double(0x7ff0000000000000LL);
  }
  if (lane == 0) atomicMax(max_bits + 1, (unsigned long long)__double_as_longlong(mx));
}

// The int8 shadow, once per dataset; allowed to fail like the others (no memory, rows that are not a
// multiple of 16 elements, non-finite or all-zero data): the half tier then ranks.
static int32_t ensure_shadow8(rpt_ctx* ctx, const rpt_dataset* data) {
  if (data->shadow8_state != 0) return RPT_OK;
  data->shadow8_state = -1;
  void* p = nullptr;
  DevBuf<unsigned long long> mb;
  unsigned long long bits[2] = {0, 0};
  auto give_up = [&](const char* why = "") {
    if (ctx->opt.debug_host) fprintf(stderr, "int8 shadow: not built (%s)\n", why);
    if (p) dev_free(p);
    (void)hipGetLastError();
    return RPT_OK;
  };
  if (data->csr || data->n == 0 || (data->d % 16) != 0 || data->d > 16384)
    return give_up("CSR rows, or rows that are not a multiple of 16 elements");
  if (dev_alloc(&p, (size_t)data->n * data->d + 16) != hipSuccess) {
    p = nullptr;
    return give_up("no memory");
  }
  if (mb.alloc(2) != RPT_OK) return give_up();
  if (hipMemsetAsync(mb.p, 0, 16, ctx->stream) != hipSuccess) return give_up();
  const int64_t count = data->n * data->d;
  int64_t b1 = (count + 255) / 256;
  if (b1 > (int64_t)ctx->n_cu * 16) b1 = (int64_t)ctx->n_cu * 16;
  if (data->dtype == RPT_F64)
    hipLaunchKernelGGL(maxabs_kernel<double>, dim3((unsigned)b1), dim3(256), 0, ctx->stream,
                       (const double*)data->X, count, mb.p);
  else if (data->dtype == RPT_F32)
    hipLaunchKernelGGL(maxabs_kernel<float>, dim3((unsigned)b1), dim3(256), 0, ctx->stream,
                       (const float*)data->X, count, mb.p);
  else
    hipLaunchKernelGGL(maxabs_kernel<__hip_bfloat16>, dim3((unsigned)b1), dim3(256), 0, ctx->stream,
                       (const __hip_bfloat16*)data->X, count, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(bits, mb.p, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double mabs;
  std::memcpy(&mabs, &bits[0], 8);
  if (!(mabs > 0.0) || !(mabs < 1e150)) return give_up("max |x| is zero, huge or not finite");  // all zeros, NaN / inf, or no range left
  const double s = mabs / 127.0;
  if (!(s > 0.0) || !(1.0 / s < 1e300)) return give_up();
  int64_t blocks = (data->n + 3) / 4;
  if (blocks > (int64_t)ctx->n_cu * 16) blocks = (int64_t)ctx->n_cu * 16;
  if (data->dtype == RPT_F64)
    hipLaunchKernelGGL(shadow8_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const double*)data->X, data->n, data->d, s, (int8_t*)p, mb.p);
  else if (data->dtype == RPT_F32)
    hipLaunchKernelGGL(shadow8_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const float*)data->X, data->n, data->d, s, (int8_t*)p, mb.p);
  else
    hipLaunchKernelGGL(shadow8_kernel<__hip_bfloat16>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const __hip_bfloat16*)data->X, data->n, data->d, s, (int8_t*)p, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(bits, mb.p, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double e2;
  std::memcpy(&e2, &bits[1], 8);
  if (!(e2 < 1e300)) return give_up("row error not finite");
  data->s8_scale = s;
  data->s8_emax = std::sqrt(e2) * (1.0 + 1e-9);  // (the butterfly sum of squares: a few ulp)
  data->shadow8 = (int8_t*)p;
  data->shadow8_state = 1;
  if (ctx->opt.debug_host)
    fprintf(stderr, "int8 shadow: scale %.6g (max |x| %.6g), max row error %.6g\n", s, mabs, data->s8_emax);
  return RPT_OK;
}

// (u16 column, f32 value) shadow of a CSR f64 dataset, its largest squared row norm and longest row
__global__ __launch_bounds__(256) void shadow_csr_kernel(const int64_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const double* __restrict__ val, int64_t n,
                                                         int64_t nnz, uint16_t* __restrict__ col16,
                                                         float* __restrict__ val32,
                                                         unsigned long long* __restrict__ max_bits) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = gtid; j < nnz; j += gsz) {
    col16[j] = (uint16_t)col[j];
    val32[j] = (float)val[j];
  }
  double mx = 0.0;
  long long ml = 0;
  for (int64_t r = gtid; r < n; r += gsz) {  // rows are short (SVectors): one thread per row
    const int64_t a = rowptr[r], b = rowptr[r + 1];
    double s2 = 0.0;
    for (int64_t j = a; j < b; ++j) s2 += val[j] * val[j];
    if (!(s2 <= mx)) mx = s2 == s2 ? s2 : __longlong_as_double(0x7ff0000000000000LL);  // NaN -> +inf
    if (b - a > ml) ml = b - a;
  }
  atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));  // mx >= 0
  atomicMax(max_bits + 1, (unsigned long long)ml);
}

// Built once per dataset, like the dense shadow: an optimisation that is allowed to fail (no
// memory, NaN rows, squares outside the f32 range): the dataset is then marked and the exact
// kernel answers.
static int32_t ensure_shadow_csr(rpt_ctx* ctx, const rpt_dataset* data) {
  if (data->shadow_col16 || data->max_norm == -2.0) return RPT_OK;
  void *pc = nullptr, *pv = nullptr;
  DevBuf<unsigned long long> mb;
  unsigned long long bits[2] = {0, 0};
  auto give_up = [&]() {
    if (pc) dev_free(pc);
    if (pv) dev_free(pv);
    (void)hipGetLastError();
    data->max_norm = -2.0;
    return RPT_OK;
  };
  const size_t nz = (size_t)(data->nnz > 0 ? data->nnz : 1);
  if (dev_alloc(&pc, nz * 2 + 16) != hipSuccess) {
    pc = nullptr;
    return give_up();
  }
  if (dev_alloc(&pv, nz * 4 + 16) != hipSuccess) {
    pv = nullptr;
    return give_up();
  }
  if (mb.alloc(2) != RPT_OK) return give_up();
  if (hipMemsetAsync(mb.p, 0, 16, ctx->stream) != hipSuccess) return give_up();
  hipLaunchKernelGGL(shadow_csr_kernel, dim3((unsigned)ctx->n_cu * 8), dim3(256), 0, ctx->stream,
                     data->rowptr, data->col, (const double*)data->val, data->n, data->nnz,
                     (uint16_t*)pc, (float*)pv, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(bits, mb.p, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double m2;
  std::memcpy(&m2, &bits[0], 8);
  data->max_norm = std::sqrt(m2) * (1.0 + 1e-12);
  if (!(data->max_norm < 1e18)) return give_up();
  data->max_rowlen = (int64_t)bits[1];
  data->shadow32 = (float*)pv;
  data->shadow_col16 = (uint16_t*)pc;
  return RPT_OK;
}

// CSR f64 rows: largest squared row norm, longest row, largest |value| (one thread per row)
__global__ __launch_bounds__(256) void csr_stats_kernel(const int64_t* __restrict__ rowptr,
                                                        const double* __restrict__ val, int64_t n,
                                                        unsigned long long* __restrict__ max_bits) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  double mx = 0.0, ma = 0.0;
  long long ml = 0;
  for (int64_t r = gtid; r < n; r += gsz) {
    const int64_t a = rowptr[r], b = rowptr[r + 1];
    double s2 = 0.0;
    for (int64_t j = a; j < b; ++j) {
      const double v = val[j], av = fabs(v);
      s2 += v * v;
      if (!(av <= ma)) ma = av == av ? av : inf;
    }
    if (!(s2 <= mx)) mx = s2 == s2 ? s2 : inf;  // NaN -> +inf
    if (b - a > ml) ml = b - a;
  }
  atomicMax(max_bits, (unsigned long long)__double_as_longlong(mx));
  atomicMax(max_bits + 1, (unsigned long long)ml);
  atomicMax(max_bits + 2, (unsigned long long)__double_as_longlong(ma));
}

// the fixed-width half table: sixteen lanes per row, a lane writes whole 16-byte chunks
__global__ __launch_bounds__(256) void ell_fill_kernel(const int64_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const double* __restrict__ val, int64_t n,
                                                       int W, uint32_t* __restrict__ ell) {
  const int l16 = threadIdx.x & 15;
  const int chunks = W >> 2;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += (int64_t)gridDim.x * 16) {
    const int64_t a = rowptr[r], b = rowptr[r + 1];
    uint4* out = reinterpret_cast<uint4*>(ell + r * W);
    for (int c = l16; c < chunks; c += 16) {
      unsigned int w4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t j = a + 4 * (int64_t)c + e;
        w4[e] = 0u;
        if (j < b) {
          const _Float16 h = (_Float16)(float)val[j];
          unsigned short hb;
          __builtin_memcpy(&hb, &h, 2);
          w4[e] = (unsigned int)col[j] | ((unsigned int)hb << 16);
        }
      }
      out[c] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
  }
}

// The half tier's table of a CSR f64 dataset, once per dataset; allowed to fail (no memory, values
// beyond the half range, rows so uneven that the padding would exceed twice the nonzeros).
static int32_t ensure_shadow_ell(rpt_ctx* ctx, const rpt_dataset* data) {
  if (data->ell_state != 0) return RPT_OK;
  data->ell_state = -1;
  if (!data->csr || data->dtype != RPT_F64 || data->d > 65536 || data->n == 0 ||
      data->max_norm == -2.0)
    return RPT_OK;
  void* p = nullptr;
  DevBuf<unsigned long long> mb;
  unsigned long long bits[3] = {0, 0, 0};
  auto give_up = [&]() {
    if (p) dev_free(p);
    (void)hipGetLastError();
    return RPT_OK;
  };
  if (mb.alloc(3) != RPT_OK) return give_up();
  if (hipMemsetAsync(mb.p, 0, 24, ctx->stream) != hipSuccess) return give_up();
  hipLaunchKernelGGL(csr_stats_kernel, dim3((unsigned)ctx->n_cu * 8), dim3(256), 0, ctx->stream,
                     data->rowptr, (const double*)data->val, data->n, mb.p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (hipMemcpyAsync(bits, mb.p, 24, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  double m2, mabs;
  std::memcpy(&m2, &bits[0], 8);
  std::memcpy(&mabs, &bits[2], 8);
  const double mn = std::sqrt(m2) * (1.0 + 1e-12);
  if (!(mn < 1e18) || !(mabs < 6.0e4)) return give_up();
  const int64_t W = (((int64_t)bits[1] + 3) & ~(int64_t)3) > 4 ? (((int64_t)bits[1] + 3) & ~(int64_t)3) : 4;
  if (W > 4096 || (double)data->n * (double)W > 3.0 * (double)data->nnz + 1048576.0) return give_up();
  if (dev_alloc(&p, (size_t)data->n * (size_t)W * 4 + 16) != hipSuccess) {
    p = nullptr;
    return give_up();
  }
  int64_t blocks = (data->n + 15) / 16;
  if (blocks > (int64_t)ctx->n_cu * 32) blocks = (int64_t)ctx->n_cu * 32;
  hipLaunchKernelGGL(ell_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, data->rowptr,
                     data->col, (const double*)data->val, data->n, (int)W, (uint32_t*)p);
  if (hipGetLastError() != hipSuccess) return give_up();
  if (stream_sync(ctx->stream) != hipSuccess) return give_up();
  if (data->max_norm < 0.0) data->max_norm = mn;
  if (data->max_rowlen == 0) data->max_rowlen = (int64_t)bits[1];
  data->shadow_ell = (uint32_t*)p;
  data->ell_w = (int)W;
  data->ell_state = 1;
  return RPT_OK;
}

template <class TD, class TK>
static int32_t launch_fused(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                            const rpt_dataset* q, const void* Pq, int32_t k, int dedup,
                            int32_t* ids, double* dist, int32_t* cnt, unsigned int* ovf,
                            unsigned long long* cand_total, bool rerun = false,
                            int* tier = nullptr /* out: 0 exact, 1 f32 shadow, 2 half shadow */) {
  typedef typename AccOf<TD>::type TA;
  // small shards (few trees => a few hundred candidates per query): one wave per query
  const int64_t force = ctx->opt.knn_wave;  // -1 auto
  const size_t wbytes = fused_wave_bytes(data->d, sizeof(TA));
  const int vote = dedup >> 8;  // voting mode: the workgroup kernel, all-exact distances
  int64_t leaf = f->n;  // the size splitting stops at: the first level's node size <= minLeaf, or depth L
  for (int l = 0; l < f->L && leaf > (int64_t)f->min_leaf; ++l) leaf -= leaf / 2;
  const int64_t est_cand = (int64_t)f->T * (leaf > 0 ? leaf : 1);
  bool wave = f->T >= 1 && f->T <= kWT && wbytes <= 16 * 1024 && est_cand <= kWaveCandidates;
  // the round-4 shard kernels (traversal as its own launch, adaptive certified set) stay ahead of the
  // workgroup kernel for longer: an 8-tree C2 shard (980 candidates) 0.42 against 0.65 ms per 10 000
  // queries, 16 trees (1950) 0.73 against 0.86 (5.7 against 7.9 ms per 100 000); at 32 trees the
  // workgroup kernel is level or ahead
  bool wave_shard = f->T >= 1 && f->T <= kWT && wbytes <= 16 * 1024 && est_cand <= kShardCandidates;
  if (force >= 0) wave = wave_shard = force == 1 && f->T >= 1 && f->T <= kWT && wbytes <= 40 * 1024;
  if (vote > 0 || data->csr) wave = wave_shard = false;
  // debug_stamps: phase clocks of one wave of the wave kernel, printed after the launch
  DevBuf<unsigned long long> dbgdev;
  unsigned long long* dbg = nullptr;
  if (ctx->opt.debug_stamps && dbgdev.alloc(64) == RPT_OK) {
    (void)hipMemsetAsync(dbgdev.p, 0, 64 * 8, ctx->stream);
    dbg = dbgdev.p;
  }
  struct DbgPrint {
    rpt_ctx* ctx;
    unsigned long long* p;
    ~DbgPrint() {
      if (!p) return;
      unsigned long long hs[64];
      (void)stream_sync(ctx->stream);
      if (hipMemcpy(hs, p, sizeof(hs), hipMemcpyDeviceToHost) != hipSuccess) return;
      for (int i = 1; i < 64 && hs[i]; ++i) fprintf(stderr, "knn wave stamp %d: +%llu\n", i, hs[i] - hs[i - 1]);
    }
  } dbgprint{ctx, nullptr};
  ProfScope ps(ctx, RPT_PROF_KNN_TOPK);
  // f64 data, duplicates kept, small k: rank the candidates on the f32 shadow (half the row
  // bytes), exact distances for the best k' only, cut certified per query (see the kernels)
  const int kp_env = (int)ctx->opt.knn_kp;
  // entries the f32 pass keeps: every one costs a selection round per batch (16 of them: 0.75 ms
  // per 10 000 queries at C2), too few and cuts fail their certificate (re-run per query): k + 6
  // certifies 10 000 of 10 000 C2 queries
  int kp = kp_env > k && kp_env < kFK ? kp_env : prefilter_keep(k);
  const bool pre32 = std::is_same<TD, double>::value && dedup == 0 && kp + 1 <= kFK &&
                     !ctx->opt.knn_no_pre32 && data->shadow32 && !rerun && !f->prefilter_off &&
                     (!data->csr || data->shadow_col16);
  // first tier: rank on the HALF shadow (a quarter of the f64 bytes); its rounding is coarser, so it
  // keeps a few more entries for the exact pass: k + max(8, k / 2) (C2: 10 000 of 10 000 queries
  // certified with 18 kept, 2.55 ms per batch against 4.10 ms on the f32 shadow; 26 kept 2.70 ms)
  const int kp16_env = (int)ctx->opt.knn_kp16;
  const int kp16 = kp16_env > k && kp16_env < kFK ? kp16_env : k + (k / 2 > 8 ? k / 2 : 8);
  // (f32 data: the half shadow is the only ranking tier; the kept rows are refined with the very
  // f32 distance the all-f32 kernel ranks on, so the answers are identical)
  const bool base16 = !data->csr && data->shadow16 && !ctx->opt.knn_no_pre16 && !f->pre16_off &&
                      kp16 + 1 <= kFK && dedup == 0 && !rerun && !ctx->opt.knn_no_pre32;
  const bool sh16 = base16 && (pre32 || std::is_same<TD, float>::value);
  // SVector rows: the fixed-width half table is their first tier (the f32 CSR shadow is opt-in)
  const bool ell = std::is_same<TD, double>::value && data->csr && data->shadow_ell &&
                   !ctx->opt.knn_no_pre16 && !f->pre16_off && !f->prefilter_off && kp16 + 1 <= kFK &&
                   dedup == 0 && !rerun && !ctx->opt.knn_no_pre32;
  if (sh16 || ell) kp = kp16;
  // ... or, before that, on the INT8 shadow (an eighth of the f64 bytes; integer ranking values, the
  // cut certified through the triangle inequality, see Sh8): coarser again, k + max(48, k) kept
  const int kp8_env = (int)ctx->opt.knn_kp8;
  // (the one-wave kernel takes its threshold from the 64 per-lane minima: beyond the 48th of them
  // the candidate list outgrows its 128 slots)
  const int kcap8 = wave ? 48 : kBK;
  // (the width of the band at the cut grows with the square root of the row length: sqrt(d / 128)
  // times the margin that certifies C2's 128-element rows)
  // ... and with the number of candidates (the band holds a share of them: 48 certify C2's 3922 and a
  // C4 forest's 3686 per query, not the 4915 of C4's 64 trees, where most queries took the in-kernel
  // retry — a second pass over all candidates: 3.5 ms per 10 000 queries against 2.0 with 100 kept),
  // and with what the forest's earlier batches reported (kp8_boost: retries counted by the kernel)
  const double cand_scale = est_cand > 4000 ? (double)est_cand / 4000.0 : 1.0;
  const int margin8 = (int)((k > 48 ? k : 48) * (data->d > 128 ? std::sqrt((double)data->d / 128.0) : 1.0) *
                            cand_scale * f->kp8_boost);
  int kp8 = kp8_env > k && kp8_env < kcap8 ? kp8_env : k + margin8;
  if (kp8 > kcap8 - 1) kp8 = kcap8 - 1;
  const bool sh8 = !data->csr && data->shadow8 && !ctx->opt.knn_no_pre8 && !f->pre8_off &&
                   kp8 >= k + 8 && dedup == 0 && !rerun && !ctx->opt.knn_no_pre32 &&
                   !ctx->opt.knn_no_pre16 && !f->pre16_off && !f->prefilter_off &&
                   (std::is_same<TD, double>::value || std::is_same<TD, float>::value ||
                    (std::is_same<TD, __hip_bfloat16>::value && !wave));  // (bf16: the workgroup kernel only)
  if (sh8) kp = kp8;
  const Sh8 s8 = sh8 ? Sh8{data->s8_scale, data->s8_emax} : Sh8{0.0, 0.0};
  if (tier) *tier = sh8 ? 3 : (sh16 || ell) ? 2 : pre32 ? 1 : 0;
  const void* shadow = sh8    ? (const void*)data->shadow8
                       : sh16 ? (const void*)data->shadow16
                              : (const void*)data->shadow32;
  if constexpr (!std::is_same<TD, __hip_bfloat16>::value) {
    if (wave_shard && !wave && !((pre32 || sh16 || sh8) && k <= kSKmax && !ctx->opt.knn_shard_old))
      wave_shard = false;  // (only the prefiltered instantiations have the shard kernels)
    // (a forced wave variant on very long rows: the shard kernels' slab must fit the CU's LDS)
    if (wave_shard && 4 * shard_wave_bytes(data->d, sizeof(TA), 256, 640) > 160 * 1024) wave_shard = false;
  } else {
    wave_shard = false;
  }
  if (wave || wave_shard) {
    dbgprint.p = dbg;
    const size_t smem = 4 * wbytes;
    if constexpr (!std::is_same<TD, __hip_bfloat16>::value) {
      // round 4: traversal as its own launch, exact distances for an adaptive certified set
      // (shard_ranges_kernel / knn_shard_wave_kernel); knn_shard_old = 1 keeps the round-3 kernel
      if (wave_shard && (pre32 || sh16 || sh8) && k <= kSKmax && !ctx->opt.knn_shard_old) {
        const int64_t nq = q->n;
        DevBuf<int2> hdr;
        DevBuf<int64_t> roff;
        DevBuf<int> rlen;
        RPT_TRY(hdr.alloc((size_t)nq));
        RPT_TRY(roff.alloc((size_t)nq * kWR));
        RPT_TRY(rlen.alloc((size_t)nq * kWR));
        const int qpw = 64 / f->T;
        hipLaunchKernelGGL(shard_ranges_kernel<TK>, dim3((unsigned)((nq + 4 * qpw - 1) / (4 * qpw))), dim3(256), 0,
                           ctx->stream, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, nq, f->T, f->L,
                           f->min_leaf, f->n, hdr.p, roff.p, rlen.p, ovf + 1, ovf, cand_total);
        RPT_HIP(hipGetLastError());
        const bool one_batch = (int64_t)f->T * (leaf > 0 ? leaf : 1) <= kWC;
        const size_t smem2 = 4 * (one_batch ? shard_wave_bytes(data->d, sizeof(TA), 128, 512)
                                            : shard_wave_bytes(data->d, sizeof(TA), 256, 640));
        auto kern = sh8    ? (one_batch ? knn_shard_wave_kernel<TD, 3, 128, 4, 512> : knn_shard_wave_kernel<TD, 3, 256, 3, 640>)
                    : sh16 ? (one_batch ? knn_shard_wave_kernel<TD, 2, 128, 4, 512> : knn_shard_wave_kernel<TD, 2, 256, 3, 640>)
                           : (one_batch ? knn_shard_wave_kernel<TD, 1, 128, 4, 512> : knn_shard_wave_kernel<TD, 1, 256, 3, 640>);
        if (smem2 > 64 * 1024)
          RPT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
        const int r1_pct = 110;  // (100 / 110 / 130 measured: 0.281 / 0.279 / 0.283 ms per 10 000 queries)
        hipLaunchKernelGGL(kern, dim3((unsigned)((nq + 3) / 4)), dim3(256), smem2, ctx->stream, (const TD*)data->X,
                           data->d, (const TD*)q->X, f->perm.p, nq, k, hdr.p, roff.p, rlen.p, ids, dist, cnt,
                           ovf + 1, cand_total, shadow, data->max_norm, s8, r1_pct, dbg);
        RPT_HIP(hipGetLastError());
        return RPT_OK;
      }
      if (pre32 || sh16 || sh8) {
        auto kern = sh8 ? knn_fused_wave_kernel<TD, TK, true, true> : knn_fused_wave_kernel<TD, TK, true, false>;
        if (smem > 64 * 1024)
          RPT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)smem));
        hipLaunchKernelGGL(kern, dim3((unsigned)((q->n + 3) / 4)),
                           dim3(256), smem, ctx->stream, (const TD*)data->X, data->d,
                           (const TD*)q->X, f->perm.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes,
                           (const TK*)Pq, q->n, f->T, f->L, f->min_leaf, f->n, k, dedup, ids, dist,
                           cnt, ovf + 1, ovf, cand_total, shadow, data->max_norm, kp + 1, dbg,
                           sh16 ? 1 : 0, s8);
        RPT_HIP(hipGetLastError());
        return RPT_OK;
      }
    }
    if (smem > 64 * 1024)
      RPT_HIP(hipFuncSetAttribute((const void*)knn_fused_wave_kernel<TD, TK, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((knn_fused_wave_kernel<TD, TK, false>), dim3((unsigned)((q->n + 3) / 4)),
                       dim3(256), smem, ctx->stream, (const TD*)data->X, data->d,
                       (const TD*)q->X, f->perm.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes,
                       (const TK*)Pq, q->n, f->T, f->L, f->min_leaf, f->n, k, dedup, ids, dist,
                       cnt, ovf + 1, ovf, cand_total, (const void*)nullptr, 0.0, rerun ? -1 : 0, dbg);
    RPT_HIP(hipGetLastError());
    return RPT_OK;
  }
  const size_t smem = (size_t)kFC * 16 + (size_t)kFR * 12 + 2048 * 4 + (size_t)kBKx * 16 +
                      (size_t)data->d * (sizeof(TA) + 4) + 64 +
                      (vote > 0 ? (size_t)kVoteCap * 4 + 16 : 0);
  if constexpr (!std::is_same<TD, double>::value) {
    if (sh16 || sh8) {  // f32 rows ranked on their half / int8 shadow, bf16 rows on their int8 shadow
      auto kern = sh8 ? knn_fused_kernel<TD, TK, true, false, true> : knn_fused_kernel<TD, TK, true, false, false>;
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
      hipLaunchKernelGGL(kern, dim3((unsigned)q->n), dim3(256), smem,
                         ctx->stream, (const TD*)data->X, data->d, (const TD*)q->X, f->perm.p,
                         f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, q->n, f->T, f->L,
                         f->min_leaf, f->n, k, dedup, ids, dist, cnt, ovf + 1, ovf, cand_total,
                         shadow, data->max_norm, kp + 1, CsrPtrs{}, 1, s8, (unsigned long long*)nullptr);
      RPT_HIP(hipGetLastError());
      return RPT_OK;
    }
  }
  if constexpr (std::is_same<TD, double>::value) {
    if ((pre32 || ell) && data->csr) {  // SVector rows ranked on their half table / (u16, f32) shadow
      const CsrPtrs cp{data->rowptr, data->col, data->val, data->nnz, q->rowptr, q->col, q->val,
                       data->shadow_col16, data->shadow32, ell ? (int64_t)data->ell_w : data->max_rowlen,
                       data->shadow_ell, data->ell_w};
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)knn_fused_kernel<TD, TK, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL((knn_fused_kernel<TD, TK, true, true>), dim3((unsigned)q->n), dim3(256),
                         smem, ctx->stream, (const TD*)nullptr, data->d, (const TD*)nullptr,
                         f->perm.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, q->n,
                         f->T, f->L, f->min_leaf, f->n, k, dedup, ids, dist, cnt, ovf + 1, ovf,
                         cand_total, (const void*)nullptr, data->max_norm, kp + 1, cp, ell ? 1 : 0);
      RPT_HIP(hipGetLastError());
      return RPT_OK;
    }
    if (pre32 || sh8) {
      auto kern = sh8 ? knn_fused_kernel<TD, TK, true, false, true> : knn_fused_kernel<TD, TK, true, false, false>;
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
      hipLaunchKernelGGL(kern, dim3((unsigned)q->n), dim3(256), smem,
                         ctx->stream, (const TD*)data->X, data->d, (const TD*)q->X, f->perm.p,
                         f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, q->n, f->T, f->L,
                         f->min_leaf, f->n, k, dedup, ids, dist, cnt, ovf + 1, ovf, cand_total,
                         shadow, data->max_norm, kp + 1, CsrPtrs{}, sh16 ? 1 : 0, s8, dbg);
      dbgprint.p = dbg;
      RPT_HIP(hipGetLastError());
      return RPT_OK;
    }
  }
  if constexpr (!std::is_same<TD, __hip_bfloat16>::value) {
    if (data->csr) {  // SVector rows: the same kernel, distances over CSR rows
      const CsrPtrs cp{data->rowptr, data->col, data->val, data->nnz, q->rowptr, q->col, q->val,
                       nullptr, nullptr, 0, nullptr, 0};
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)knn_fused_kernel<TD, TK, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL((knn_fused_kernel<TD, TK, false, true>), dim3((unsigned)q->n), dim3(256),
                         smem, ctx->stream, (const TD*)nullptr, data->d, (const TD*)nullptr,
                         f->perm.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, q->n,
                         f->T, f->L, f->min_leaf, f->n, k, dedup, ids, dist, cnt, ovf + 1, ovf,
                         cand_total, (const void*)nullptr, 0.0, rerun ? -1 : 0, cp);
      RPT_HIP(hipGetLastError());
      return RPT_OK;
    }
  }
  if (smem > 64 * 1024)
    RPT_HIP(hipFuncSetAttribute((const void*)knn_fused_kernel<TD, TK, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((knn_fused_kernel<TD, TK, false>), dim3((unsigned)q->n), dim3(256), smem,
                     ctx->stream, (const TD*)data->X, data->d, (const TD*)q->X, f->perm.p,
                     f->thr.p, f->mglo.p, f->mghi.p, f->nodes, (const TK*)Pq, q->n, f->T, f->L,
                     f->min_leaf, f->n, k, dedup, ids, dist, cnt, ovf + 1, ovf, cand_total,
                     (const void*)nullptr, 0.0, rerun ? -1 : 0, CsrPtrs{});
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

static int32_t knn_general(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                           const rpt_dataset* q, int32_t k, int32_t flags, int32_t* ids_dev,
                           double* dist_dev, int32_t* count_dev) {
  QueryPlan pl;
  RPT_TRY(make_plan(ctx, f, q, pl));
  ctx->last_candidates = pl.total_cand;
  if (q->n == 0) return RPT_OK;
  const int dedup = flags & 3;
  const int refm = (flags & RPT_KNN_METRIC_REFERENCE) ? 1 : 0;
  if (data->csr) {
    const size_t smem = topk_smem(data->d, 8);
    ProfScope ps(ctx, RPT_PROF_KNN_TOPK);
    if (data->dtype == RPT_F64) {
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)topk_csr_kernel<double>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(topk_csr_kernel<double>, dim3((unsigned)q->n), dim3(256), smem,
                         ctx->stream, data->rowptr, data->col, (const double*)data->val, data->nnz, data->d,
                         q->rowptr, q->col, (const double*)q->val, f->perm.p, pl.ranges.p,
                         pl.rng_off.p, f->T, k, dedup, refm, ids_dev, dist_dev, count_dev);
    } else {
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)topk_csr_kernel<float>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(topk_csr_kernel<float>, dim3((unsigned)q->n), dim3(256), smem,
                         ctx->stream, data->rowptr, data->col, (const float*)data->val, data->nnz, data->d,
                         q->rowptr, q->col, (const float*)q->val, f->perm.p, pl.ranges.p,
                         pl.rng_off.p, f->T, k, dedup, refm, ids_dev, dist_dev, count_dev);
    }
    RPT_HIP(hipGetLastError());
  } else if (data->dtype == RPT_F64) {
    RPT_TRY(launch_topk_dense<double>(ctx, data, q, f->perm.p, pl.ranges.p, pl.rng_off.p, f->T, 0,
                                      k, dedup, ids_dev, dist_dev, count_dev));
  } else if (data->dtype == RPT_F32) {
    RPT_TRY(launch_topk_dense<float>(ctx, data, q, f->perm.p, pl.ranges.p, pl.rng_off.p, f->T, 0,
                                     k, dedup, ids_dev, dist_dev, count_dev));
  } else {
    RPT_TRY(launch_topk_dense<__hip_bfloat16>(ctx, data, q, f->perm.p, pl.ranges.p, pl.rng_off.p,
                                              f->T, 0, k, dedup, ids_dev, dist_dev, count_dev));
  }
  RPT_HIP(stream_sync(ctx->stream));  // the plan's buffers are released on return
  return RPT_OK;
}

int32_t knn_dev(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data, const rpt_dataset* q,
                int32_t k, int32_t flags, int32_t* ids_dev, double* dist_dev, int32_t* count_dev) {
  RPT_ARG(data->dtype == q->dtype, "data and query dtype must match");
  RPT_ARG(k <= kBuf / 2, "k too large for the LDS merge buffer");
  RPT_ARG((size_t)data->d * 8 + (sizeof(Entry) + 4) * kBuf <= 150 * 1024, "d too large");
  RPT_ARG(proj_dtype(q->dtype) == f->pdtype,
          "query dtype must have the forest's projection type (f64 vs f32/bf16)");
  const int vote = (flags >> 8) & 0xffff;  // RPT_KNN_VOTE(v)
  // (the fused kernels walk the implicit batch topology: streamed forests take the general path;
  // so do SVector rows under the reference's own truncating metric)
  const bool refm = data->csr && (flags & RPT_KNN_METRIC_REFERENCE);
  const bool fused = k <= kFK && f->T <= 1024 && (!ctx->opt.knn_general || vote > 0) &&
                     (size_t)data->d * 8 <= 32 * 1024 && !f->xtopo && !refm;
  if (vote > 0 && (!fused || data->csr))
    return fail(RPT_E_UNSUPPORTED, "RPT_KNN_VOTE: dense data, k <= 64 and at most 1024 trees");
  if (!fused || q->n == 0) return knn_general(ctx, f, data, q, k, flags, ids_dev, dist_dev, count_dev);
  const int64_t nq = q->n;
  const int dedup = (flags & 3) | (vote << 8);
  DevBuf<char> Pq;
  // one control block, one memset, one read-back: u64 candidates visited, u64 queries whose
  // prefilter cut was not certified, u32 queries that overflowed the range slab, pad, flags[nq]
  DevBuf<unsigned int> ctl;
  const size_t esz = f->pdtype == RPT_F64 ? 8 : 4;
  RPT_TRY(Pq.alloc((size_t)f->T * f->L * nq * esz + 16));
  RPT_TRY(ctl.alloc((size_t)nq + 6));
  RPT_HIP(hipMemsetAsync(ctl.p, 0, ((size_t)nq + 6) * 4, ctx->stream));
  unsigned int* ovf_p = ctl.p + 5;  // launch_fused: the count at ovf_p[0], the flags from ovf_p + 1
  unsigned long long* ctot_p = reinterpret_cast<unsigned long long*>(ctl.p);
  {
    ProfScope ps(ctx, RPT_PROF_KNN_PLAN);
    if (f->L > 0) RPT_TRY(project_columns(ctx, q, f->R.p, f->T * f->L, f->mode, Pq.p));
  }
  if (f->pdtype == RPT_F64 && dedup == 0 && prefilter_keep(k) < kFK && !ctx->opt.knn_no_pre32 &&
      !f->prefilter_off) {  // (dedup carries the vote threshold too: no prefilter when voting)
    if (!data->csr) {
      // the int8 shadow first; the f32 and half ones (+75 % of the dataset) only when it cannot rank
      // this call: rows that are not a multiple of 16 elements, k beyond what the tier keeps in the
      // one-wave kernel, no memory, or a forest that has dropped the tier (once per dataset each)
      const bool want8 = !ctx->opt.knn_no_pre16 && !f->pre16_off && !ctx->opt.knn_no_pre8 && !f->pre8_off;
      if (want8) RPT_TRY(ensure_shadow8(ctx, data));
      if (!(want8 && data->shadow8 && k + 8 <= 47)) {
        RPT_TRY(ensure_shadow(ctx, data));
        if (data->shadow32 && !ctx->opt.knn_no_pre16 && !f->pre16_off) RPT_TRY(ensure_shadow16(ctx, data));
      }
    }
    // CSR rows: the (u16, f32) shadow halves the bytes of the ranking pass but NOT its time — at C3
    // the exact kernel already gathers rows at 6.4 TB/s and the f32 pass, with its 17 selection
    // rounds per batch, is bound by its serial phases (14.3 ms against 13.8 ms per 10 000 queries):
    // opt-in (knn_csr_pre32), kept for bandwidth-poorer configurations and pinned by a test
    else if (data->dtype == RPT_F64 && data->d <= 65536) {
      if (ctx->opt.knn_csr_pre32) RPT_TRY(ensure_shadow_csr(ctx, data));
      if (!ctx->opt.knn_no_pre16 && !f->pre16_off) RPT_TRY(ensure_shadow_ell(ctx, data));
    }
  }
  int tier = 0;
  if (data->dtype == RPT_F32 && !data->csr && dedup == 0 && !ctx->opt.knn_no_pre16 &&
      !ctx->opt.knn_no_pre32 && !f->pre16_off) {
    const bool want8 = !ctx->opt.knn_no_pre8 && !f->pre8_off;
    if (want8) RPT_TRY(ensure_shadow8(ctx, data));
    if (!(want8 && data->shadow8 && k + 8 <= 47)) RPT_TRY(ensure_shadow16(ctx, data));
  }
  // bf16 rows: the int8 tier exists (identical answers, tested) but is OPT-IN (knn_kp8 > 0): at C5
  // (10 M x 768, k = 50) the cut needs 200 kept rows to certify 99 % of the queries and the pass is
  // then no faster than the plain bf16 kernel (82 against 68 ms per 100 000 queries) — its 768-byte
  // rows keep half the bytes in flight per wave — while the shadow costs 7.7 GB
  if (data->dtype == RPT_BF16 && !data->csr && dedup == 0 && ctx->opt.knn_kp8 > 0 && !ctx->opt.knn_no_pre16 &&
      !ctx->opt.knn_no_pre32 && !ctx->opt.knn_no_pre8 && !f->pre16_off && !f->pre8_off && !f->prefilter_off)
    RPT_TRY(ensure_shadow8(ctx, data));
  auto launch = [&](bool rerun) -> int32_t {
    if (f->pdtype == RPT_F64)
      return launch_fused<double, double>(ctx, f, data, q, Pq.p, k, dedup, ids_dev, dist_dev,
                                          count_dev, ovf_p, ctot_p, rerun, rerun ? nullptr : &tier);
    if (data->dtype == RPT_F32)
      return launch_fused<float, float>(ctx, f, data, q, Pq.p, k, dedup, ids_dev, dist_dev,
                                        count_dev, ovf_p, ctot_p, rerun, rerun ? nullptr : &tier);
    return launch_fused<__hip_bfloat16, float>(ctx, f, data, q, Pq.p, k, dedup, ids_dev, dist_dev,
                                               count_dev, ovf_p, ctot_p, rerun, rerun ? nullptr : &tier);
  };
  RPT_TRY(launch(false));
  // (read back into the pinned arena: a pageable destination makes the copy a staged, blocking one)
  unsigned int hctl_stack[6] = {0, 0, 0, 0, 0, 0};
  unsigned int* hctl = reinterpret_cast<unsigned int*>(pin_alloc(ctx, 32));
  if (!hctl) hctl = hctl_stack;
  RPT_HIP(hipMemcpyAsync(hctl, ctl.p, 24, hipMemcpyDeviceToHost, ctx->stream));
  RPT_HIP(stream_sync(ctx->stream));
  const unsigned int novf = hctl[5];
  unsigned long long tot[2];
  std::memcpy(tot, hctl, 16);
  ctx->last_candidates = (int64_t)tot[0];
  ctx->last_uncertified = (int64_t)tot[1];
  ctx->last_retries = 0;
  if (novf && vote > 0)
    return fail(RPT_E_UNSUPPORTED,
                "RPT_KNN_VOTE: a query reaches more than 16384 candidates or 512 leaves");
  if (novf)  // some query reached more leaf ranges than the LDS slab holds: general path
    return knn_general(ctx, f, data, q, k, flags, ids_dev, dist_dev, count_dev);
  // too many uncertified cuts: one tier down for the later batches on this forest (half -> f32
  // shadow -> all-f64)
  if (tot[1] * 4 > (unsigned long long)nq) {
    if (tier == 3) f->pre8_off = true;
    else if (tier == 2) f->pre16_off = true;
    else f->prefilter_off = true;
  }
  ctx->last_tier = tier;
  // a batch in which more than an eighth of the queries needed the wider second attempt: the next
  // batches on this forest start wider (the int8 tier's kept entries; capped by the kernel's list)
  ctx->last_retries = (int64_t)hctl[4];
  if (tier == 3 && (unsigned long long)hctl[4] * 8 > (unsigned long long)nq && f->kp8_boost < 4.0)
    f->kp8_boost *= 1.5;
  if (tot[1]) {  // queries with equal distances at the prefilter's cut: the all-f64 kernel, them only
    RPT_TRY(launch(true));
    RPT_HIP(stream_sync(ctx->stream));  // Pq / ovf are released on return
  }
  return RPT_OK;
}

// knnH (RPTree.hs:199-217): the union of the trees' candidatesH heaps is popped in increasing
// priority; buckets are taken while the running count stays <= k — but at least one —, each new
// bucket in FRONT of the ones taken before; the result is every point of those buckets with its
// distance, neither sorted nor cut to k.  Equal priorities keep (tree, DFS) order (the reference's
// order among ties depends on the shape of its heap).
// Device: query projections, traversal with priorities, distances.  Host: the per-query
// selection over a few dozen (priority, leaf) entries.
int32_t knn_h(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data, const rpt_dataset* q,
              int32_t k, int64_t* off_host, int32_t* ids_host, double* dist_host, int64_t cap,
              int64_t* total) {
  RPT_ARG(data->dtype == q->dtype && data->csr == q->csr, "data and query must have one layout");
  const int64_t nq = q->n;
  const int T = f->T;
  *total = 0;
  if (off_host) off_host[0] = 0;
  if (nq == 0) return RPT_OK;
  QueryPlan pl;
  RPT_TRY(make_plan(ctx, f, q, pl));
  hipStream_t st = ctx->stream;
  DevBuf<double> prio;
  RPT_TRY(prio.alloc((size_t)pl.total_rng));
  const int threads = T <= 64 ? 64 : (T <= 128 ? 128 : 256);
  if (f->xtopo) {  // streamed forest: the same walk over the stored topology (ranges in traverse_x's order)
    if (f->pdtype == RPT_F64)
      hipLaunchKernelGGL(ranges_hx_kernel<double>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p, f->mglo.p,
                         f->mghi.p, f->nodes, f->xkind.p, (const double*)pl.Pq.p, nq, T, f->L, pl.rng_off.p, prio.p);
    else
      hipLaunchKernelGGL(ranges_hx_kernel<float>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p, f->mglo.p,
                         f->mghi.p, f->nodes, f->xkind.p, (const float*)pl.Pq.p, nq, T, f->L, pl.rng_off.p, prio.p);
  } else if (f->pdtype == RPT_F64)
    hipLaunchKernelGGL(ranges_h_kernel<double>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, (const double*)pl.Pq.p, nq, T, f->L,
                       f->min_leaf, f->n, pl.rng_off.p, prio.p);
  else
    hipLaunchKernelGGL(ranges_h_kernel<float>, dim3((unsigned)nq), dim3(threads), 0, st, f->thr.p,
                       f->mglo.p, f->mghi.p, f->nodes, (const float*)pl.Pq.p, nq, T, f->L,
                       f->min_leaf, f->n, pl.rng_off.p, prio.p);
  RPT_HIP(hipGetLastError());
  RPT_HIP(stream_sync(st));
  std::vector<Range> hr((size_t)pl.total_rng);
  std::vector<double> hp((size_t)pl.total_rng);
  std::vector<int64_t> hoff((size_t)nq * T + 1);
  RPT_HIP(hipMemcpy(hr.data(), pl.ranges.p, hr.size() * sizeof(Range), hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(hp.data(), prio.p, hp.size() * 8, hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(hoff.data(), pl.rng_off.p, hoff.size() * 8, hipMemcpyDeviceToHost));
  std::vector<Range> sel;
  std::vector<int64_t> sel_off((size_t)nq + 1, 0), out_off((size_t)nq + 1, 0);
  std::vector<int64_t> order, taken;
  for (int64_t qi = 0; qi < nq; ++qi) {
    const int64_t r0 = hoff[(size_t)qi * T], r1 = hoff[(size_t)(qi + 1) * T];
    order.resize((size_t)(r1 - r0));
    for (int64_t r = r0; r < r1; ++r) order[(size_t)(r - r0)] = r;
    std::stable_sort(order.begin(), order.end(),
                     [&](int64_t a, int64_t b) { return hp[(size_t)a] < hp[(size_t)b]; });
    taken.clear();
    int64_t n = 0;
    for (int64_t r : order) {  // :209-217
      const int64_t ntot = hr[(size_t)r].n + n;
      // `not (null acc)` (:216) is about the POINTS taken so far: an empty bucket (streamed trees can
      // hold one) does not count as "at least one"
      if (ntot > k && n > 0) break;
      taken.push_back(r);
      n = ntot;
    }
    int64_t pos = 0;
    for (size_t i = taken.size(); i-- > 0;) {  // `xsh <> acc`: the bucket taken last comes first
      Range rg = hr[(size_t)taken[i]];
      rg.pos = (int32_t)pos;
      pos += rg.n;
      sel.push_back(rg);
    }
    sel_off[(size_t)qi + 1] = (int64_t)sel.size();
    out_off[(size_t)qi + 1] = out_off[(size_t)qi] + pos;
  }
  *total = out_off[(size_t)nq];
  if (off_host) std::memcpy(off_host, out_off.data(), out_off.size() * 8);
  if (!ids_host || !dist_host || *total == 0) return RPT_OK;
  RPT_ARG(cap >= *total, "result capacity too small");
  DevBuf<Range> dsel;
  DevBuf<int64_t> dsel_off, dout_off;
  DevBuf<int32_t> ids;
  DevBuf<double> dist;
  RPT_TRY(dsel.alloc(sel.size()));
  RPT_TRY(dsel_off.alloc(sel_off.size()));
  RPT_TRY(dout_off.alloc(out_off.size()));
  RPT_TRY(ids.alloc((size_t)*total));
  RPT_TRY(dist.alloc((size_t)*total));
  RPT_HIP(hipMemcpy(dsel.p, sel.data(), sel.size() * sizeof(Range), hipMemcpyHostToDevice));
  RPT_HIP(hipMemcpy(dsel_off.p, sel_off.data(), sel_off.size() * 8, hipMemcpyHostToDevice));
  RPT_HIP(hipMemcpy(dout_off.p, out_off.data(), out_off.size() * 8, hipMemcpyHostToDevice));
  if (data->csr) {
    const size_t smem = (size_t)data->d * 8;
    RPT_ARG(smem <= 150 * 1024, "d too large");
    if (data->dtype == RPT_F64) {
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)dist_sel_csr_kernel<double>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(dist_sel_csr_kernel<double>, dim3((unsigned)nq), dim3(256), smem, st,
                         data->rowptr, data->col, (const double*)data->val, data->d, q->rowptr,
                         q->col, (const double*)q->val, f->perm.p, dsel.p, dsel_off.p, dout_off.p,
                         ids.p, dist.p);
    } else {
      if (smem > 64 * 1024)
        RPT_HIP(hipFuncSetAttribute((const void*)dist_sel_csr_kernel<float>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(dist_sel_csr_kernel<float>, dim3((unsigned)nq), dim3(256), smem, st,
                         data->rowptr, data->col, (const float*)data->val, data->d, q->rowptr,
                         q->col, (const float*)q->val, f->perm.p, dsel.p, dsel_off.p, dout_off.p,
                         ids.p, dist.p);
    }
  } else {
    const size_t smem = (size_t)data->d * (data->dtype == RPT_F64 ? 8 : 4);
    RPT_ARG(smem <= 150 * 1024, "d too large");
#define RPT_DIST_SEL(TD)                                                                      \
  do {                                                                                        \
    if (smem > 64 * 1024)                                                                     \
      RPT_HIP(hipFuncSetAttribute((const void*)dist_sel_dense_kernel<TD>,                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));    \
    hipLaunchKernelGGL(dist_sel_dense_kernel<TD>, dim3((unsigned)nq), dim3(256), smem, st,    \
                       (const TD*)data->X, data->d, (const TD*)q->X, f->perm.p, dsel.p,       \
                       dsel_off.p, dout_off.p, ids.p, dist.p);                                \
  } while (0)
    if (data->dtype == RPT_F64) RPT_DIST_SEL(double);
    else if (data->dtype == RPT_F32) RPT_DIST_SEL(float);
    else RPT_DIST_SEL(__hip_bfloat16);
#undef RPT_DIST_SEL
  }
  RPT_HIP(hipGetLastError());
  RPT_HIP(stream_sync(st));
  RPT_HIP(hipMemcpy(ids_host, ids.p, (size_t)*total * 4, hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(dist_host, dist.p, (size_t)*total * 8, hipMemcpyDeviceToHost));
  return RPT_OK;
}

static int32_t launch_merge(rpt_ctx* ctx, const int32_t* ids, const double* dist,
                            const int32_t* cnt, int64_t is, int64_t ds, int64_t cs, int32_t G,
                            int64_t nq, int32_t k, int32_t flags, int32_t* out_ids,
                            double* out_dist, int32_t* out_count) {
  int np = 1;
  while (np < G * k) np <<= 1;
  const size_t smem = (sizeof(Entry) + 4) * (size_t)np;
  if (smem > 64 * 1024)
    RPT_HIP(hipFuncSetAttribute((const void*)merge_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(256), smem, ctx->stream, ids, dist,
                     cnt, is, ds, cs, G, nq, k, flags & 3, out_ids, out_dist, out_count);
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

int32_t knn_merge_dev(rpt_ctx* ctx, const int32_t* ids_dev, const double* dist_dev,
                      const int32_t* count_dev, int64_t shard_stride, int32_t G, int64_t nq,
                      int32_t k, int32_t flags, int32_t* out_ids, double* out_dist,
                      int32_t* out_count) {
  if (nq == 0) return RPT_OK;
  // shard_stride 0: three shard-major arrays [G][nq][k], [G][nq][k], [G][nq]
  const int64_t is = shard_stride ? shard_stride : nq * k * 4, ds = shard_stride ? shard_stride : nq * k * 8,
                cs = shard_stride ? shard_stride : nq * 4;
  // one launch holds all G * k entries of a query in LDS (20 bytes each: 4096 entries = 80 KB)
  if ((int64_t)G * k <= kMergeMax)
    return launch_merge(ctx, ids_dev, dist_dev, count_dev, is, ds, cs, G, nq, k, flags, out_ids,
                        out_dist, out_count);
  // larger merges (e.g. 8 shards x k = 1024) fold the shards in one at a time: the running
  // result (shards < g, already in the final order) against shard g is a 2-way merge of 2k
  // entries, and ties still prefer the earlier shard, so the order is that of the G-way merge.
  // The kernel addresses "shard 1" as base + stride, so the pair need not be adjacent.
  DevBuf<int32_t> ti[2], tc[2];
  DevBuf<double> td[2];
  for (int b = 0; b < 2 && G > 2; ++b) {
    RPT_TRY(ti[b].alloc((size_t)nq * k));
    RPT_TRY(td[b].alloc((size_t)nq * k));
    RPT_TRY(tc[b].alloc((size_t)nq));
  }
  const char *ai = (const char*)ids_dev, *ad = (const char*)dist_dev, *ac = (const char*)count_dev;
  for (int g = 1; g < G; ++g) {
    const bool last = g == G - 1;
    int32_t* oi = last ? out_ids : ti[g & 1].p;
    double* od = last ? out_dist : td[g & 1].p;
    int32_t* oc = last ? out_count : tc[g & 1].p;
    const char *gi = (const char*)ids_dev + g * is, *gd = (const char*)dist_dev + g * ds,
               *gc = (const char*)count_dev + g * cs;
    RPT_TRY(launch_merge(ctx, (const int32_t*)ai, (const double*)ad, (const int32_t*)ac, gi - ai,
                         gd - ad, gc - ac, 2, nq, k, flags, oi, od, oc));
    ai = (const char*)oi;
    ad = (const char*)od;
    ac = (const char*)oc;
  }
  return RPT_OK;
}

int32_t brute_knn(rpt_ctx* ctx, const rpt_dataset* data, const rpt_dataset* q, int32_t k,
                  int32_t* ids_host, double* dist_host) {
  RPT_ARG(k <= kBuf / 2, "k too large");
  const int64_t nq = q->n;
  if (nq == 0) return RPT_OK;
  DevBuf<int32_t> ids, cnt;
  DevBuf<double> dist;
  RPT_TRY(ids.alloc((size_t)nq * k));
  RPT_TRY(dist.alloc((size_t)nq * k));
  RPT_TRY(cnt.alloc((size_t)nq));
  if (data->dtype == RPT_F64)
    RPT_TRY(launch_topk_dense<double>(ctx, data, q, nullptr, nullptr, nullptr, 1, 1, k, 0, ids.p,
                                      dist.p, cnt.p));
  else if (data->dtype == RPT_F32)
    RPT_TRY(launch_topk_dense<float>(ctx, data, q, nullptr, nullptr, nullptr, 1, 1, k, 0, ids.p,
                                     dist.p, cnt.p));
  else
    RPT_TRY(launch_topk_dense<__hip_bfloat16>(ctx, data, q, nullptr, nullptr, nullptr, 1, 1, k, 0,
                                              ids.p, dist.p, cnt.p));
  RPT_HIP(stream_sync(ctx->stream));
  RPT_HIP(hipMemcpy(ids_host, ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(dist_host, dist.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
  return RPT_OK;
}

}  // namespace rpt
