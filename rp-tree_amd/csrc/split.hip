// split.hip — forest construction below the projection batch: the median split of every node
// of every tree, level by level.
//
// Replaces partitionAtMedian + sortByVG (Internal.hs:486-512) as driven by insert / create /
// createMulti (Internal.hs:217-297).  Contract restated: a node with n
// points in order o is split by stably sorting o by the projection on the level's hyperplane;
// the left child is the first nh = n div 2 points IN SORTED ORDER, the right child the rest;
// thr = p'[nh], margin = (p'[nh-1], p'[nh+1]) (n>=3; n==2 -> (p'[0],p'[1]); n==1 -> p'[0]).
//
// Because each child inherits the sorted order, the order of a node created by the split at
// level l-1 is the order of the key tuple (p_{l-1}, p_{l-2}, ..., p_0, id): a stable sort by
// p_l of that order is the order of (p_l, p_{l-1}, ..., p_0, id).  The kernels therefore never
// rely on the physical order inside a segment: ties of the primary key are broken by
// comparing the earlier levels' projections (all kept in HBM) and finally the point id.
// That makes leaf assignment identical to the reference for any tie pattern while letting
// the big-node path use an UNSTABLE partition:
//
//   small nodes (n <= kSmallCap): one workgroup per (tree, node) sorts (key, id) in LDS
//       (bitonic) and writes the children in sorted order.
//   big nodes: sample -> splitters; histogram over value bins around the median (one pass that
//       also stashes the gathered keys contiguously); pick the pivot bin; 3-way scatter
//       (left | pivot bin | right, wave-ballot ranking, one atomic per class per block);
//       sort only the pivot bin ("mid", a few hundred points) in LDS.  Leaves produced by
//       this path are ordered afterwards by the same LDS sort.
//   rare: a pivot bin or leaf larger than LDS -> chunked LDS sort + rank-merge passes in HBM.
//
// Algorithmic HBM bytes per point per tree per level (SURVEY.md §8d lower bound: 16):
//   big path: perm 4 + key gather 8 + key stash 8+8 + perm 4+4 = 36; small path: 4 + 8 + 4.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <list>

#include "common.h"
#include "codes.h"

namespace rpt {
namespace {

constexpr int kSmallCap = 4096;   // largest segment sorted by one workgroup in LDS
constexpr int kNB = 1024;         // value bins of the big path (0 and kNB-1 are the tails)
constexpr int kSample = 1024;     // samples per big node
constexpr int kDelta = 80;        // splitter half-width in sample ranks (5 sigma: sd = 16)
constexpr int kChunk = 4096;      // elements per block in hist / scatter
constexpr int kPad = 0x7fffffff;  // id of bitonic padding entries

// Projections of one tree: P[level][N].  tie_less: a precedes b when their primary keys at
// `level` are equal = lexicographic order on the earlier levels, then id (or tb[] when the
// caller defines "previous position" explicitly, rpt_split_segments).
template <class TK>
struct Keys {
  const TK* Pt;  // base of the tree's [L][N] block
  int64_t N;
  int level;
  const int32_t* tb;  // optional final tie-break key indexed by id
  __device__ TK key(int id) const { return Pt[(int64_t)level * N + id]; }
  __device__ bool tie_less(int a, int b) const {
    if (b == kPad) return a != kPad;
    if (a == kPad) return false;
    for (int j = level - 1; j >= 0; --j) {
      const TK ka = Pt[(int64_t)j * N + a], kb = Pt[(int64_t)j * N + b];
      if (ka < kb) return true;
      if (kb < ka) return false;
    }
    return tb ? tb[a] < tb[b] : a < b;
  }
  __device__ bool less(TK ka, int a, TK kb, int b) const {
    if (ka < kb) return true;
    if (kb < ka) return false;
    return tie_less(a, b);
  }
};

// barrier among the lanes of one wave (wave-private LDS regions need no workgroup barrier)
__device__ inline void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In-LDS bitonic sort of np (power of two) (key, id) pairs by Keys::less.  All threads of
// the block must call it; with WAVE the region belongs to the calling wave alone and only its
// 64 lanes take part (no workgroup barrier).
// (key, id) order without the earlier levels' keys: branch-free.  It equals Keys::less unless
// two REAL entries have equal keys; callers that use it check the sorted neighbours for that and
// re-sort with the exact comparator (never on continuous data, always on tie-heavy data).
template <class TK>
__device__ inline bool fast_less(TK ka, int a, TK kb, int b) {
  return (ka < kb) | ((ka == kb) & (a < b));
}

template <class TK, bool WAVE = false, bool FAST = false>
__device__ void lds_bitonic(TK* skey, int* sid, int np, const Keys<TK>& K) {
  const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int nthr = WAVE ? 64 : (int)blockDim.x;
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (np >> 1); i += nthr) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const TK kl = skey[lo], kh = skey[hi];
        const int il = sid[lo], ih = sid[hi];
        const bool h_lt_l = FAST ? fast_less(kh, ih, kl, il) : K.less(kh, ih, kl, il);
        const bool l_lt_h = FAST ? fast_less(kl, il, kh, ih) : K.less(kl, il, kh, ih);
        if (up ? h_lt_l : l_lt_h) {
          skey[lo] = kh;
          skey[hi] = kl;
          sid[lo] = ih;
          sid[hi] = il;
        }
      }
      if (WAVE) wsync();
      else __syncthreads();
    }
  }
}

__device__ inline bool is_leaf_dev(int level, int n, int L, int min_leaf) {
  return level >= L || n <= min_leaf;
}

__host__ __device__ inline int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

template <class TK>
__device__ inline TK pos_inf();
template <>
__device__ inline double pos_inf<double>() { return __longlong_as_double(0x7ff0000000000000LL); }
template <>
__device__ inline float pos_inf<float>() { return __uint_as_float(0x7f800000u); }

// Bitonic sort of NR*64 (key, id) pairs held NR per lane (index = r*64 + lane) by one wave:
// partners at distance >= 64 are other registers of the same lane, closer partners come
// through lane shuffles.  No LDS, no barrier.
template <class TK, int NR, int KK, int J, bool FAST = false>
__device__ inline void wave_bitonic_stage(TK (&k)[NR], int (&id)[NR], const Keys<TK>& K, int lane) {
  if constexpr (J >= 64) {
    constexpr int JR = J >> 6;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if ((r & JR) == 0) {
        const int r2 = r + JR;
        const bool up = ((r * 64) & KK) == 0;  // KK >= 128: the direction bit lives in r
        const bool hi_lt_lo = FAST ? fast_less(k[r2], id[r2], k[r], id[r]) : K.less(k[r2], id[r2], k[r], id[r]);
        if (up ? hi_lt_lo : !hi_lt_lo) {
          const TK tk = k[r];
          k[r] = k[r2];
          k[r2] = tk;
          const int ti = id[r];
          id[r] = id[r2];
          id[r2] = ti;
        }
      }
    }
  } else {
    const bool lower = (lane & J) == 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      TK ok = __shfl_xor(k[r], J);
      int oi = __shfl_xor(id[r], J);
      // pin the exchanged values here: the id is only consumed on key ties, and a cross-lane
      // read must never be sunk into that divergent branch (lanes outside it would not
      // take part in the exchange)
      asm volatile("" : "+v"(ok), "+v"(oi));
      const bool up = ((r * 64 + lane) & KK) == 0;
      const bool o_lt_me = FAST ? fast_less(ok, oi, k[r], id[r]) : K.less(ok, oi, k[r], id[r]);
      if ((lower == up) ? o_lt_me : !o_lt_me) {
        k[r] = ok;
        id[r] = oi;
      }
    }
  }
  if constexpr (J > 1) wave_bitonic_stage<TK, NR, KK, (J >> 1), FAST>(k, id, K, lane);
}

template <class TK, int NR, int KK, bool FAST = false>
__device__ inline void wave_bitonic_phase(TK (&k)[NR], int (&id)[NR], const Keys<TK>& K, int lane) {
  wave_bitonic_stage<TK, NR, KK, (KK >> 1), FAST>(k, id, K, lane);
  if constexpr (KK < 64 * NR) wave_bitonic_phase<TK, NR, (KK << 1), FAST>(k, id, K, lane);
}

template <class TK, int NR, bool FAST = false>
__device__ inline void wave_bitonic(TK (&k)[NR], int (&id)[NR], const Keys<TK>& K) {
  wave_bitonic_phase<TK, NR, 2, FAST>(k, id, K, threadIdx.x & 63);
}

// the same sort, branch-free comparator first; the exact one only if two real neighbours of the
// sorted sequence have equal keys (then the earlier levels' keys decide their order)
template <class TK, int NR>
__device__ inline void wave_bitonic_fast(TK (&k)[NR], int (&id)[NR], const Keys<TK>& K) {
  const int lane = threadIdx.x & 63;
  wave_bitonic<TK, NR, true>(k, id, K);
  bool eq = false;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    TK nk = __shfl_down(k[r], 1);
    int ni = __shfl_down(id[r], 1);
    if (r + 1 < NR) {  // lane 63's successor is lane 0 of the next register
      const TK fk = __shfl(k[r + 1 < NR ? r + 1 : r], 0);
      const int fi = __shfl(id[r + 1 < NR ? r + 1 : r], 0);
      if (lane == 63) {
        nk = fk;
        ni = fi;
      }
    } else if (lane == 63) {
      ni = kPad;
    }
    eq |= id[r] != kPad && ni != kPad && k[r] == nk;
  }
  if (__ballot(eq)) wave_bitonic<TK, NR, false>(k, id, K);
}

// ---------------------------------------------------------------------------------------
// small path: sort one segment per block.  grid = (S, T).
//   src/dst: perm rows [T][N] (may alias for in-place use), segs[S].
//   heap >= 0: write thr/mglo/mghi of the node.  Counts nodes whose cut straddles a tie.
// ---------------------------------------------------------------------------------------
template <class TK>
__global__ __launch_bounds__(256) void small_sort_kernel(
    const int32_t* src, int32_t* dst /* may alias src */, int64_t N, const TK* P, int L,
    int level, const Seg* __restrict__ segs, const int32_t* tb, double* thr, double* mglo,
    double* mghi, int64_t nodes, unsigned long long* tie_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  const int n = sg.n;
  if (n <= 0) return;
  const int np = next_pow2(n);
  TK* skey = reinterpret_cast<TK*>(smem);
  int* sid = reinterpret_cast<int*>(smem + (size_t)np * sizeof(TK));
  Keys<TK> K{P + (int64_t)t * L * N, N, level, tb ? tb + (int64_t)t * N : nullptr};  // tb: [T][N]
  const int32_t* s = src + (int64_t)t * N + sg.off;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    if (i < n) {
      const int id = s[i];
      sid[i] = id;
      skey[i] = K.key(id);
    } else {
      sid[i] = kPad;
      skey[i] = pos_inf<TK>();
    }
  }
  __syncthreads();
  lds_bitonic(skey, sid, np, K);
  int32_t* o = dst + (int64_t)t * N + sg.off;
  for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = sid[i];
  if (sg.heap >= 0 && threadIdx.x == 0) {
    const int nh = n / 2;
    const int64_t h = (int64_t)t * nodes + sg.heap;
    thr[h] = (double)skey[nh];
    mglo[h] = (double)skey[nh > 0 ? nh - 1 : 0];       // Internal.hs:497-499
    mghi[h] = (double)skey[nh + 1 < n ? nh + 1 : n - 1];
    if (nh > 0 && !(skey[nh - 1] < skey[nh])) atomicAdd(tie_count, 1ULL);
  }
}

// ---------------------------------------------------------------------------------------
// fused small-subtree kernel: one workgroup per (tree, top node with n <= kSmallCap) runs up
// to kRmax tree levels without moving a single element:
//   per level: gather the level's key of every element, per-node min/max, per-node value
//   histogram in LDS (kSmallCap/M bins per node, M = nodes at this depth), pivot bin by a
//   wave scan, exact resolution of the pivot bin only (rank counting with the lexicographic
//   tie-break), thr / margins, child = 2*node + side.  O(n) per level, no sort.
//   An element whose child is a leaf (Internal.hs:289) retires with (left-aligned path, the
//   key of its last split).  One bitonic sort by (path, key, tie-break) at the end puts every
//   leaf bucket in the reference's order (children inherit the sorted order,
//   Internal.hs:495,504-505) and groups the still-active nodes in left-to-right order.
//   Retired elements go to the final perm F, active ones to `nxt` for the next launch.
// grid = (S, T), 256 threads.
// ---------------------------------------------------------------------------------------
constexpr int kRmax = 6;                 // levels per launch (<= 64 nodes per block)
constexpr int kSubThreads = 1024;        // 16 waves: latency hiding + small per-thread state
constexpr int kE = kSmallCap / kSubThreads;  // element slots per thread
constexpr int kMaxM = 1 << (kRmax - 1);  // nodes at the deepest processed depth

struct SubNode {  // per relative node of the current depth (LDS)
  int n, nh, pb, cL, cMid, lowb, highb, midoff;
};

__device__ inline int sub_node_size(int n_top, int r, int j) {
  int n = n_top;
  for (int b = r - 1; b >= 0; --b) {
    const int nh = n >> 1;
    n = ((j >> b) & 1) ? n - nh : nh;
  }
  return n;
}

template <class TK>
__global__ __launch_bounds__(kSubThreads) void subtree_kernel(
    const int32_t* __restrict__ src, int32_t* __restrict__ nxt, int32_t* __restrict__ F,
    int64_t N, const TK* __restrict__ P, int L, int level0, int min_leaf,
    const Seg* __restrict__ segs, double* thr, double* mglo, double* mghi, int64_t nodes,
    unsigned long long* tie_count, unsigned long long* dbg, int rmax /* levels to run, <= kRmax */) {
  __shared__ __attribute__((aligned(16))) TK skey[kSmallCap];
  __shared__ int sid[kSmallCap];
  __shared__ unsigned int sx[kSmallCap];          // histogram during the levels, path at the end
  __shared__ unsigned long long nmin[kMaxM], nmax[kMaxM], nmaxL[kMaxM], nminR[kMaxM];
  __shared__ TK nlo[kMaxM], nscale[kMaxM];
  __shared__ SubNode sn[kMaxM];
  __shared__ int midcur[kMaxM];
  __shared__ int s_active;

  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  const int n_top = sg.n;
  if (n_top <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int dbgi = 0;
#define STAMP() do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) dbg[dbgi++] = clock64(); } while (0)
  STAMP();
  const TK* Pt = P + (int64_t)t * L * N;
  const int32_t* s = src + (int64_t)t * N + sg.off;

  int id[kE], node[kE];
  unsigned int dpath[kE];   // (left-aligned path << 8) | level of the retained key; 0 = active
  TK fkey[kE];
#pragma unroll
  for (int e = 0; e < kE; ++e) {
    const int pos = e * kSubThreads + tid;
    id[e] = pos < n_top ? s[pos] : -1;
    node[e] = 0;
    dpath[e] = 0;
    fkey[e] = (TK)0;
  }

  int depth = 0;
  for (; depth < rmax; ++depth) {
    const int level = level0 + depth;
    if (level >= L) break;
    // any element still active?
    if (tid == 0) s_active = 0;
    __syncthreads();
    {
      int a = 0;
#pragma unroll
      for (int e = 0; e < kE; ++e) a |= (id[e] >= 0 && dpath[e] == 0);
      if (a) s_active = 1;
    }
    __syncthreads();
    if (!s_active) break;

    const int M = 1 << depth;
    const int B = (kSmallCap / M) < 1024 ? (kSmallCap / M) : 1024;
    const TK* Pl = Pt + (int64_t)level * N;
    Keys<TK> K{Pt, N, level, nullptr};

    // ---- a. keys ----
    TK key[kE];
#pragma unroll
    for (int e = 0; e < kE; ++e) key[e] = (id[e] >= 0 && dpath[e] == 0) ? Pl[id[e]] : (TK)0;

    STAMP();  // keys issued
    // ---- b. per-node min / max ----
    if (tid < M) {
      nmin[tid] = ~0ULL;
      nmax[tid] = 0ULL;
      nmaxL[tid] = 0ULL;
      nminR[tid] = ~0ULL;
      midcur[tid] = 0;
    }
    for (int i = tid; i < kSmallCap; i += kSubThreads) sx[i] = 0;
    __syncthreads();
    if (M <= 4) {  // few nodes: everybody would hit the same LDS word -> reduce per wave first
      for (int j = 0; j < M; ++j) {
        unsigned long long mn = ~0ULL, mx = 0ULL;
#pragma unroll
        for (int e = 0; e < kE; ++e)
          if (id[e] >= 0 && dpath[e] == 0 && node[e] == j) {
            const unsigned long long o = ord_of(key[e]);
            mn = o < mn ? o : mn;
            mx = o > mx ? o : mx;
          }
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
          mn = a < mn ? a : mn;
          mx = b > mx ? b : mx;
        }
        if (lane == 0) {
          if (mn != ~0ULL) atomicMin(&nmin[j], mn);
          if (mx != 0ULL) atomicMax(&nmax[j], mx);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < kE; ++e)
        if (id[e] >= 0 && dpath[e] == 0) {
          const unsigned long long o = ord_of(key[e]);
          atomicMin(&nmin[node[e]], o);
          atomicMax(&nmax[node[e]], o);
        }
    }
    __syncthreads();
    STAMP();  // minmax
    // ---- c. bin geometry ----
    if (tid < M) {
      SubNode a;
      a.n = sub_node_size(n_top, depth, tid);
      a.nh = a.n >> 1;
      a.pb = a.cL = a.cMid = 0;
      a.lowb = -1;
      a.highb = B;
      a.midoff = 0;
      sn[tid] = a;
      if (nmin[tid] != ~0ULL) {
        const TK lo = ord_to(nmin[tid], TK()), hi = ord_to(nmax[tid], TK());
        nlo[tid] = lo;
        nscale[tid] = lo < hi ? (TK)B / (hi - lo) : (TK)0;
      } else {
        nlo[tid] = (TK)0;
        nscale[tid] = (TK)0;
      }
    }
    __syncthreads();
    // ---- d. histogram ----
    int bin[kE];
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      bin[e] = -1;
      if (id[e] >= 0 && dpath[e] == 0) {
        const int j = node[e];
        int b = (int)((key[e] - nlo[j]) * nscale[j]);
        b = b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
        bin[e] = b;
        atomicAdd(&sx[j * B + b], 1u);
      }
    }
    __syncthreads();
    STAMP();  // hist
    // ---- e. pivot bin per node: one wave per node, wave scan over B bins ----
    for (int j = wave; j < M; j += kSubThreads / 64) {
      const int n = sn[j].n;
      if (n <= 0) continue;
      const unsigned int nh = (unsigned int)sn[j].nh;
      const int per = (B + 63) / 64;
      unsigned int loc = 0;
      for (int i = 0; i < per; ++i) {
        const int b = lane * per + i;
        if (b < B) loc += sx[j * B + b];
      }
      unsigned int inc = loc;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
      }
      unsigned int run = inc - loc;  // elements in bins before this lane's range
      int pb = -1, cL = 0, cMid = 0;
      for (int i = 0; i < per; ++i) {
        const int b = lane * per + i;
        if (b < B) {
          const unsigned int c = sx[j * B + b];
          if (run <= nh && nh < run + c) {
            pb = b;
            cL = (int)run;
            cMid = (int)c;
          }
          run += c;
        }
      }
      // broadcast the owning lane's result
      const unsigned long long own = __ballot(pb >= 0);
      const int src_lane = __ffsll((long long)own) - 1;
      pb = __shfl(pb, src_lane);
      cL = __shfl(cL, src_lane);
      cMid = __shfl(cMid, src_lane);
      // nearest non-empty bins around the pivot bin (for margins outside the pivot bin)
      int lowb = -1, highb = B;
      for (int i = 0; i < per; ++i) {
        const int b = lane * per + i;
        if (b < B && sx[j * B + b]) {
          if (b < pb) lowb = b > lowb ? b : lowb;
          if (b > pb) highb = b < highb ? b : highb;
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(lowb, o), b2 = __shfl_xor(highb, o);
        lowb = a > lowb ? a : lowb;
        highb = b2 < highb ? b2 : highb;
      }
      if (lane == 0) {
        sn[j].pb = pb;
        sn[j].cL = cL;
        sn[j].cMid = cMid;
        sn[j].lowb = lowb;
        sn[j].highb = highb;
      }
    }
    __syncthreads();
    STAMP();  // pick
    // ---- f. collect the pivot bins (mid pool in skey/sid), max(left bin) / min(right bin) ----
    if (tid == 0) {
      int run = 0;
      for (int j = 0; j < M; ++j) {
        sn[j].midoff = run;
        run += sn[j].cMid;
      }
    }
    __syncthreads();
    int mpos[kE];
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      mpos[e] = -1;
      if (bin[e] >= 0) {
        const int j = node[e];
        if (bin[e] == sn[j].pb) {
          const int p = sn[j].midoff + atomicAdd(&midcur[j], 1);
          skey[p] = key[e];
          sid[p] = id[e];
          mpos[e] = p;
        } else if (bin[e] == sn[j].lowb) {
          atomicMax(&nmaxL[j], ord_of(key[e]));
        } else if (bin[e] == sn[j].highb) {
          atomicMin(&nminR[j], ord_of(key[e]));
        }
      }
    }
    __syncthreads();
    STAMP();  // collect
    // ---- g. exact rank inside the pivot bin; node outputs ----
    int side[kE];
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      side[e] = 0;
      if (bin[e] < 0) continue;
      const int j = node[e];
      const SubNode a = sn[j];
      if (mpos[e] < 0) {
        side[e] = bin[e] > a.pb;
        continue;
      }
      int rank = 0;
      for (int q = a.midoff; q < a.midoff + a.cMid; ++q)
        if (q != mpos[e] && K.less(skey[q], sid[q], key[e], id[e])) ++rank;
      const int il = a.nh > 0 ? a.nh - 1 : 0, ih = a.nh + 1 < a.n ? a.nh + 1 : a.n - 1;
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + j);
      if (rank == a.nh - a.cL) thr[h] = (double)key[e];
      if (rank == il - a.cL) mglo[h] = (double)key[e];
      if (rank == ih - a.cL) mghi[h] = (double)key[e];
      side[e] = rank >= a.nh - a.cL;
    }
    __syncthreads();
    if (tid < M && sn[tid].cMid > 0) {  // cMid == 0: phantom slot below a Tip
      const SubNode a = sn[tid];
      const int il = a.nh > 0 ? a.nh - 1 : 0, ih = a.nh + 1 < a.n ? a.nh + 1 : a.n - 1;
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + tid);
      if (il < a.cL) mglo[h] = (double)ord_to(nmaxL[tid], TK());
      if (ih >= a.cL + a.cMid) mghi[h] = (double)ord_to(nminR[tid], TK());
    }
    STAMP();  // rank
    // ---- h. descend ----
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      if (bin[e] < 0) continue;
      const SubNode a = sn[node[e]];
      const int child = 2 * node[e] + side[e];
      const int nc = side[e] ? a.n - a.nh : a.nh;
      if (level + 1 >= L || nc <= min_leaf) {  // the child is a Tip (Internal.hs:289)
        dpath[e] = ((unsigned int)(child << (kRmax - (depth + 1))) << 8) | (unsigned int)(level + 1);
        fkey[e] = key[e];
      } else {
        node[e] = child;
      }
    }
    __syncthreads();
    // ties straddling the cut (statistics): thr element equals its left neighbour
    if (tid < M && sn[tid].n > 1 && sn[tid].cMid > 0) {
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + tid);
      __threadfence_block();
      if (!(mglo[h] < thr[h])) atomicAdd(tie_count, 1ULL);
    }
  }

  // ---- final order ----
  // 1. compact: every terminal node (leaf, or node still active after kRmax levels) becomes a
  //    contiguous slot range at its topological offset; 2. every leaf bucket is sorted by
  //    (retained key, earlier levels, id) inside ONE wave (<= 128 points: two per lane,
  //    bitonic network over shuffles, no block barrier); buckets larger than that fall back
  //    to a block-wide bitonic sort by (path, key, ...).
  __syncthreads();
  STAMP();  // levels done
  __shared__ int tcur[1 << kRmax];
  __shared__ int trec_off[1 << kRmax], trec_n[1 << kRmax];
  __shared__ unsigned int trec_path[1 << kRmax];
  __shared__ int nterm, need_block_sort;
  if (tid < (1 << kRmax)) tcur[tid] = 0;
  if (tid == 0) {
    nterm = 0;
    need_block_sort = 0;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kE; ++e) {
    if (id[e] < 0) continue;
    const bool done = dpath[e] != 0;
    const int d = done ? (int)(dpath[e] & 255u) - level0 : depth;  // depth of the terminal
    const unsigned int pth = done ? (dpath[e] >> 8) : (unsigned int)(node[e] << (kRmax - depth));
    int toff = 0, tn = n_top;
    for (int b = 0; b < d; ++b) {
      const int nh = tn >> 1;
      if ((pth >> (kRmax - 1 - b)) & 1u) {
        toff += nh;
        tn -= nh;
      } else {
        tn = nh;
      }
    }
    const int c = atomicAdd(&tcur[pth], 1);
    const int slot = toff + c;
    skey[slot] = done ? fkey[e] : (TK)0;
    sid[slot] = id[e];
    sx[slot] = done ? dpath[e] : (pth << 8);
    if (c == 0) {
      const int r = atomicAdd(&nterm, 1);
      trec_off[r] = toff;
      trec_n[r] = tn;
      trec_path[r] = done ? dpath[e] : 0u;
      if (done && tn > 128) need_block_sort = 1;
    }
  }
  __syncthreads();
  if (!need_block_sort) {
    for (int r = wave; r < nterm; r += kSubThreads / 64) {
      const unsigned int pl = trec_path[r];
      if (!(pl & 255u)) continue;  // still active: order is irrelevant
      const int toff = trec_off[r], tn = trec_n[r];
      const int lv = (int)(pl & 255u) - 1;
      Keys<TK> K{Pt, N, lv, nullptr};
      TK k0, k1;
      int i0, i1;
      {
        const bool v0 = lane < tn, v1 = lane + 64 < tn;
        k0 = v0 ? skey[toff + lane] : pos_inf<TK>();
        i0 = v0 ? sid[toff + lane] : kPad;
        k1 = v1 ? skey[toff + lane + 64] : pos_inf<TK>();
        i1 = v1 ? sid[toff + lane + 64] : kPad;
      }
      const int npad = tn > 64 ? 128 : 64;
      for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          if (j == 64) {  // partner = the other register of the same lane (index ^ 64)
            const bool up = true;  // k == 128: (idx & 128) == 0 for every index
            const bool one_lt_zero = K.less(k1, i1, k0, i0);
            if (up ? one_lt_zero : !one_lt_zero) {
              const TK tk = k0;
              k0 = k1;
              k1 = tk;
              const int ti = i0;
              i0 = i1;
              i1 = ti;
            }
          } else {
            {  // register 0: index = lane
              const TK ok = __shfl_xor(k0, j);
              const int oi = __shfl_xor(i0, j);
              const bool up = (lane & k) == 0;
              const bool lower = (lane & j) == 0;
              const bool o_lt_me = K.less(ok, oi, k0, i0);
              // lower keeps the min when ascending; upper keeps the max
              const bool take = (lower == up) ? o_lt_me : !o_lt_me;
              if (take) {
                k0 = ok;
                i0 = oi;
              }
            }
            if (npad == 128) {  // register 1: index = 64 + lane
              const TK ok = __shfl_xor(k1, j);
              const int oi = __shfl_xor(i1, j);
              const bool up = ((64 + lane) & k) == 0;
              const bool lower = (lane & j) == 0;
              const bool o_lt_me = K.less(ok, oi, k1, i1);
              const bool take = (lower == up) ? o_lt_me : !o_lt_me;
              if (take) {
                k1 = ok;
                i1 = oi;
              }
            }
          }
        }
      }
      if (lane < tn) sid[toff + lane] = i0;
      if (lane + 64 < tn) sid[toff + lane + 64] = i1;
    }
    __syncthreads();
  } else {
  const int np = next_pow2(n_top);
  for (int i = n_top + tid; i < np; i += kSubThreads) {
    sx[i] = 0xffffffffu;
    skey[i] = pos_inf<TK>();
    sid[i] = kPad;
  }
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (np >> 1); i += kSubThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned int pl = sx[lo], ph = sx[hi];
        const TK kl = skey[lo], kh = skey[hi];
        const int il = sid[lo], ih = sid[hi];
        bool h_lt_l;  // "hi entry precedes lo entry"
        if (pl != ph) h_lt_l = ph < pl;
        else if (kh < kl) h_lt_l = true;
        else if (kl < kh) h_lt_l = false;
        else {
          // same bucket, equal retained key (the projection of level (pl & 255) - 1): the
          // earlier levels decide, then the id
          const int lv = (int)(pl & 255u) - 1;
          Keys<TK> K{Pt, N, lv < 0 ? 0 : lv, nullptr};
          h_lt_l = K.tie_less(ih, il);
        }
        // entries are distinct under the full order, so "lo precedes hi" == !h_lt_l
        if (up ? h_lt_l : !h_lt_l) {
          sx[lo] = ph;
          sx[hi] = pl;
          skey[lo] = kh;
          skey[hi] = kl;
          sid[lo] = ih;
          sid[hi] = il;
        }
      }
      __syncthreads();
    }
  }
  }
  STAMP();  // sorted
  int32_t* of = F + (int64_t)t * N + sg.off;
  int32_t* on = nxt + (int64_t)t * N + sg.off;
  for (int i = tid; i < n_top; i += kSubThreads) {
    if (sx[i] & 255u) of[i] = sid[i];   // retired: final position
    else on[i] = sid[i];                // still active: input of the next launch
  }
  STAMP();
#undef STAMP
}

// ---------------------------------------------------------------------------------------
// wave-centric subtree kernel: ONE WAVE per (tree, top node with n <= kWCap).  Same select as
// subtree_kernel (per level: min/max, value histogram, pivot bin, exact resolution of the
// pivot bin; no element ever moves) but without a single workgroup barrier: all state lives
// in the wave's registers (16 points per lane) and a wave-private LDS slab, so a CU runs many
// independent subtrees at once and their latencies overlap.
// Output: every point is scattered to the slot range of its terminal node — a leaf (final
// perm F; the bucket is written in its final order, phase h) or a node still active after
// kWRmax levels (nxt).
// A wave that meets a pivot bin larger than its LDS slab flags the node and writes nothing;
// the host re-runs those nodes with subtree_kernel.
// grid = ceil(S*T/4) blocks of 256 threads.
// ---------------------------------------------------------------------------------------
// (measured, round 2: feeding the wave kernel ONE gathered word per point — the 16-bit codes of
// its three levels packed by a 0.1 ms pass — instead of a gathered key per point and level changes
// nothing, 1.28 ms either way at C2: the kernel waits on its LDS phases, not on the gathers.)
constexpr int kWCap = 1024;
constexpr int kWE = kWCap / 64;   // points per lane
constexpr int kWRmax = 5;         // levels per launch (<= 16 nodes at the deepest depth)
constexpr int kWHist = 1024;      // histogram entries per wave
constexpr int kWMid = 192;        // pivot-bin pool per wave

constexpr int kWPivot = 1 << 30;  // st flag: the point sits in its node's pivot-bin pool
constexpr int kWHasPos = 1 << 28;  // st flag: bits 18..27 hold the point's position in its node

struct WSlab {
  unsigned int hist[kWHist + 64];  // skewed: bin idx lives at idx + (idx >> 4)
  unsigned long long nmaxL[16], nminR[16];
  int4 pf[16];  // per node: pivot bin, low edge bin, high edge bin, pool offset
  int4 pg[16];  // per node: cL, nh, n, cMid
  int midcur[16];
  double midkey[kWMid];
  int midid[kWMid];
  unsigned char midnode[kWMid], midside[kWMid];
  int toff[32], tcur[32];
  double vthr[16], vlo[16];
  // the points of nodes whose children are leaves, in (node, bin) order: 30-bit monotone image
  // of the key (bin << 20 | position inside the bin)
  unsigned int code[kWCap];
  int noff[16];  // offset of every node of the level inside the wave's segment
};

template <class TK>
__global__ __launch_bounds__(256, 3) void wsub_kernel(
    const int32_t* __restrict__ src, int32_t* __restrict__ nxt, int32_t* __restrict__ F,
    int64_t N, const TK* __restrict__ P, int L, int T, int level0, int min_leaf,
    const Seg* __restrict__ segs, int S, double* thr, double* mglo, double* mghi,
    int64_t nodes, unsigned long long* tie_count, unsigned int* ovf_flags,
    unsigned int* ovf_count, const unsigned int* __restrict__ abort, unsigned long long* dbg) {
  __shared__ WSlab slabs[4];
  int dbgi = 0;
#define WSTAMP() do { if (dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) dbg[dbgi++] = clock64(); } while (0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wg = (int64_t)blockIdx.x * 4 + wave;
  if (wg >= (int64_t)S * T) return;
  if (abort && *abort) return;  // the streaming levels above are being rebuilt by the host
  const int si = (int)(wg % S), t = (int)(wg / S);
  WSlab& W = slabs[wave];
  const Seg sg = segs[si];
  const int n_top = sg.n;
  if (n_top <= 0) return;
  const TK* Pt = P + (int64_t)t * L * N;
  const int32_t* s = src + (int64_t)t * N + sg.off;

  int id[kWE];
  TK key[kWE];   // key of the current level; frozen at the level a point retires
  // active: node index (>= 0); retired: -(ordered << 23 | rank in leaf << 13 |
  // left-aligned path << 8 | leaf level)
  int st[kWE];
#pragma unroll
  for (int e = 0; e < kWE; ++e) {
    const int pos = e * 64 + lane;
    id[e] = pos < n_top ? s[pos] : -1;
    st[e] = 0;
    key[e] = (TK)0;
  }
  int depth = 0;
  bool overflow = false;
  WSTAMP();
  for (; depth < kWRmax; ++depth) {
    const int level = level0 + depth;
    if (level >= L) break;
    WSTAMP();
    int act = 0;
#pragma unroll
    for (int e = 0; e < kWE; ++e) act |= (id[e] >= 0 && st[e] >= 0);
    if (!__any(act)) break;

    const int M = 1 << depth;
    const int B = kWHist / M;
    Keys<TK> K{Pt, N, level, nullptr};
    // ---- a. keys of this level (gathered by id) ----
    {
      const TK* Pl = Pt + (int64_t)level * N;
#pragma unroll
      for (int e = 0; e < kWE; ++e)
        if (id[e] >= 0 && st[e] >= 0) key[e] = Pl[id[e]];
    }
    WSTAMP();
    // ---- b. one value range for all nodes of the level (the children of a node are random
    // halves of it with respect to THIS level's key, so they span about the same range; keys
    // outside the range clamp to the edge bins).  No per-node LDS atomics.
    double kmn = __builtin_huge_val(), kmx = -__builtin_huge_val();
#pragma unroll
    for (int e = 0; e < kWE; ++e)
      if (id[e] >= 0 && st[e] >= 0) {
        const double k = (double)key[e];
        kmn = k < kmn ? k : kmn;
        kmx = k > kmx ? k : kmx;
      }
    for (int o = 32; o > 0; o >>= 1) {
      const double a = __shfl_xor(kmn, o), b = __shfl_xor(kmx, o);
      kmn = a < kmn ? a : kmn;
      kmx = b > kmx ? b : kmx;
    }
    const TK lo = (TK)kmn;
    const TK scale = kmn < kmx ? (TK)((double)B / (kmx - kmn)) : (TK)0;
#pragma unroll
    for (int i = 0; i < (kWHist + 64) / 64; ++i) W.hist[i * 64 + lane] = 0;
    if (lane < 16) {
      W.nmaxL[lane] = 0ULL;
      W.nminR[lane] = ~0ULL;
      W.midcur[lane] = 0;
    }
    wsync();
    WSTAMP();
    // ---- c. bins + histogram.  st keeps (bin << 8 | node) for the later phases; the
    // histogram is skewed by one word per 16 bins so that lane-strided scans are conflict-free
#pragma unroll
    for (int e = 0; e < kWE; ++e)
      if (id[e] >= 0 && st[e] >= 0) {
        const int j = st[e];
        int b = (int)((key[e] - lo) * scale);
        b = b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
        const int idx = j * B + b;
        atomicAdd(&W.hist[idx + (idx >> 4)], 1u);
        st[e] = j | (b << 8);
      }
    wsync();
    WSTAMP();
    // ---- e. pivot bins of ALL nodes in one pass: lane owns 16 consecutive bins, a node owns
    // Wd = 64/M consecutive lanes
    const int Wd = 64 >> depth, lw = 6 - depth;
    const int jl = lane >> lw, rl = lane & (Wd - 1);
    const int nj = sub_node_size(n_top, depth, jl), nhj = nj >> 1;
    unsigned int c[16];
    unsigned int loc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      c[i] = W.hist[lane * 17 + i];
      loc += c[i];
    }
    unsigned int inc = loc;
    for (int o = 1; o < Wd; o <<= 1) {
      const unsigned int v = __shfl_up(inc, o, Wd);
      if (rl >= o) inc += v;
    }
    int pb = -1, cL = 0, cMid = 0;
    {
      unsigned int a = inc - loc;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (c[i] && a <= (unsigned int)nhj && (unsigned int)nhj < a + c[i]) {
          pb = rl * 16 + i;
          cL = (int)a;
          cMid = (int)c[i];
        }
        a += c[i];
      }
    }
    {  // the node-relative exclusive prefix replaces the count of every bin (phase h)
      unsigned int a = inc - loc;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        W.hist[lane * 17 + i] = a;
        a += c[i];
      }
    }
    const unsigned long long found = __ballot(pb >= 0);
    const unsigned long long segm = (Wd == 64 ? ~0ULL : ((1ULL << Wd) - 1ULL)) << (jl * Wd);
    const unsigned long long own = found & segm;
    const int srcl = own ? __ffsll((long long)own) - 1 : lane;  // phantom slot below a Tip: none
    pb = __shfl(pb, srcl);
    cL = __shfl(cL, srcl);
    cMid = __shfl(cMid, srcl);
    const int il = nhj > 0 ? nhj - 1 : 0, ih = nhj + 1 < nj ? nhj + 1 : nj - 1;
    // the margins p'[nh-1] / p'[nh+1] leave the pivot bin only at its edges: only then the
    // points of the nearest non-empty bins are tracked (max of the left one, min of the right)
    const bool need_lo = il < cL, need_hi = ih >= cL + cMid;
    int lowb = -1, highb = B;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int bb = rl * 16 + i;
      if (c[i]) {
        if (bb < pb) lowb = bb;
        if (bb > pb && highb == B) highb = bb;
      }
    }
    for (int o = 1; o < Wd; o <<= 1) {
      const int x = __shfl_xor(lowb, o), y = __shfl_xor(highb, o);
      lowb = x > lowb ? x : lowb;
      highb = y < highb ? y : highb;
    }
    // offsets of the pivot bins in the wave's pool: prefix over the nodes
    const int mine = (rl == 0 && own) ? cMid : 0;
    int pinc = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(pinc, o);
      if (lane >= o) pinc += v;
    }
    const int tot = __shfl(pinc, 63);
    if (rl == 0) {
      W.pf[jl] = make_int4(own ? pb : -1, need_lo ? lowb : -2, need_hi ? highb : B + 1, pinc - mine);
      W.pg[jl] = make_int4(cL, nhj, nj, own ? cMid : 0);
      int off = 0, n = n_top;
      for (int bb = depth - 1; bb >= 0; --bb) {
        const int h = n >> 1;
        if ((jl >> bb) & 1) {
          off += h;
          n -= h;
        } else {
          n = h;
        }
      }
      W.noff[jl] = off;
    }
    if (tot > kWMid) {
      overflow = true;
      break;
    }
    wsync();
    WSTAMP();
    // ---- h. nodes whose children are leaves: exact position of every point inside its node, so
    // that the leaf buckets leave this kernel in their final order.  Counting sort by bin (the
    // histogram now holds the bins' exclusive prefixes), then the few bin-mates are ordered by
    // a 30-bit monotone image of the key kept in LDS; mates with the SAME image (about one pair
    // per 10^5 points, and every real tie) compare their real keys, fetched by id.
    // (st of an active point from here on: node | bin << 8 | position << 18 | kWHasPos)
    bool any_leaf = false;
    for (int j = 0; j < M; ++j) {
      const int4 pgj = W.pg[j];
      any_leaf = any_leaf || (pgj.w > 0 && (level + 1 >= L || pgj.y <= min_leaf ||
                                            pgj.z - pgj.y <= min_leaf));
    }
    if (any_leaf) {
      auto image = [&](TK k) -> unsigned int {
        const TK fq = (k - lo) * scale;
        if (!(fq > (TK)0)) return 0u;
        if (fq >= (TK)B) return ((unsigned int)B << 20) - 1u;
        return (unsigned int)(fq * (TK)1048576);
      };
#pragma unroll
      for (int e = 0; e < kWE; ++e)
        if (id[e] >= 0 && st[e] >= 0) {
          const int j = st[e] & 0xff, b = (st[e] >> 8) & 0x3ff;
          const int4 pgj = W.pg[j];
          if (level + 1 >= L || pgj.y <= min_leaf || pgj.z - pgj.y <= min_leaf) {
            const int idx = j * B + b;
            const int gp = W.noff[j] + (int)atomicAdd(&W.hist[idx + (idx >> 4)], 1u);
            W.code[gp] = image(key[e]);
            st[e] |= (gp << 18) | kWHasPos;
          }
        }
      wsync();
      // the histogram has done its job: it now lists the ids in slot order (for the exact compare)
#pragma unroll
      for (int e = 0; e < kWE; ++e)
        if (id[e] >= 0 && st[e] >= 0 && (st[e] & kWHasPos))
          W.hist[(st[e] >> 18) & 0x3ff] = (unsigned int)id[e];
      wsync();
#pragma unroll
      for (int e = 0; e < kWE; ++e)
        if (id[e] >= 0 && st[e] >= 0 && (st[e] & kWHasPos)) {
          const int j = st[e] & 0xff, gp = (st[e] >> 18) & 0x3ff;
          const unsigned int b = (unsigned int)((st[e] >> 8) & 0x3ff);
          const int lo_j = W.noff[j], hi_j = lo_j + W.pg[j].z;
          const unsigned int mine_w = image(key[e]);
          int nleft = 0, less = 0;
          for (int qq = gp - 1; qq >= lo_j; --qq) {
            const unsigned int w = W.code[qq];
            if ((w >> 20) != b) break;
            ++nleft;
            if (w == mine_w) {
              const int mate = (int)W.hist[qq];
              less += K.less(K.key(mate), mate, key[e], id[e]);
            } else {
              less += w < mine_w;
            }
          }
          for (int qq = gp + 1; qq < hi_j; ++qq) {
            const unsigned int w = W.code[qq];
            if ((w >> 20) != b) break;
            if (w == mine_w) {
              const int mate = (int)W.hist[qq];
              less += K.less(K.key(mate), mate, key[e], id[e]);
            } else {
              less += w < mine_w;
            }
          }
          st[e] = (st[e] & ~(0x3ff << 18)) | ((gp - lo_j - nleft + less) << 18);
        }
    }
    // ---- f. pool the pivot bins; every other point descends right away.  The node records
    // are fetched four points at a time so that the LDS round trips overlap.
#pragma unroll
    for (int e0 = 0; e0 < kWE; e0 += 4) {
      int4 pf4[4], pg4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u;
        const int j = (id[e] >= 0 && st[e] >= 0) ? (st[e] & 0xff) : 0;
        pf4[u] = W.pf[j];
        pg4[u] = W.pg[j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u;
        if (id[e] < 0 || st[e] < 0) continue;
        const int j = st[e] & 0xff, b = (st[e] >> 8) & 0x3ff;
        const int hp = st[e] & (kWHasPos | (0x3ff << 18));  // position in the node, if known
        if (b == pf4[u].x) {
          const int p = pf4[u].w + atomicAdd(&W.midcur[j], 1);
          W.midkey[p] = (double)key[e];
          W.midid[p] = id[e];
          W.midnode[p] = (unsigned char)j;
          st[e] = j | (p << 8) | hp | kWPivot;
          continue;
        }
        if (b == pf4[u].y) atomicMax(&W.nmaxL[j], ord_of(key[e]));
        else if (b == pf4[u].z) atomicMin(&W.nminR[j], ord_of(key[e]));
        const int side = b > pf4[u].x;
        const int child = 2 * j + side;
        const int nc = side ? pg4[u].z - pg4[u].y : pg4[u].y;
        if (level + 1 >= L || nc <= min_leaf) {
          int code = ((child << (kWRmax - (depth + 1))) << 8) | (level + 1);
          const int pos = (hp >> 18) & 0x3ff;
          if (hp & kWHasPos) code |= (1 << 23) | ((side ? pos - pg4[u].y : pos) << 13);
          st[e] = -code;
        } else {
          st[e] = child;
        }
      }
    }
    wsync();
    WSTAMP();
    // ---- g. exact order inside the pivot bins: one lane per pooled point
    for (int p0 = 0; p0 < tot; p0 += 64) {
      const int p = p0 + lane;
      if (p < tot) {
        const int j = W.midnode[p];
        const int4 pf = W.pf[j], pg = W.pg[j];
        const TK kp = (TK)W.midkey[p];
        const int ip = W.midid[p];
        int rank = 0;
        for (int q = pf.w; q < pf.w + pg.w; ++q)
          if (q != p && K.less((TK)W.midkey[q], W.midid[q], kp, ip)) ++rank;
        const int cLj = pg.x, nh = pg.y, n = pg.z;
        const int il2 = nh > 0 ? nh - 1 : 0, ih2 = nh + 1 < n ? nh + 1 : n - 1;
        const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + j);
        if (rank == nh - cLj) {
          thr[h] = (double)kp;
          W.vthr[j] = (double)kp;
        }
        if (rank == il2 - cLj) {
          mglo[h] = (double)kp;
          W.vlo[j] = (double)kp;
        }
        if (rank == ih2 - cLj) mghi[h] = (double)kp;
        W.midside[p] = (unsigned char)(rank >= nh - cLj);
      }
    }
    wsync();
    // the owners of the pooled points descend
#pragma unroll
    for (int e = 0; e < kWE; ++e) {
      const bool piv = id[e] >= 0 && st[e] >= 0 && (st[e] & kWPivot);
      if (!__any(piv)) continue;
      if (piv) {
        const int j = st[e] & 0xff, p = (st[e] >> 8) & 0x3ff;
        const int hp = st[e] & (kWHasPos | (0x3ff << 18));
        const int4 pg = W.pg[j];
        const int side = W.midside[p];
        const int child = 2 * j + side;
        const int nc = side ? pg.z - pg.y : pg.y;
        if (level + 1 >= L || nc <= min_leaf) {
          int code = ((child << (kWRmax - (depth + 1))) << 8) | (level + 1);
          const int pos = (hp >> 18) & 0x3ff;
          if (hp & kWHasPos) code |= (1 << 23) | ((side ? pos - pg.y : pos) << 13);
          st[e] = -code;
        } else {
          st[e] = child;
        }
      }
    }
    if (lane < M && W.pg[lane].w > 0) {
      const int4 pg = W.pg[lane];
      const int cLj = pg.x, nh = pg.y, n = pg.z, cM = pg.w;
      const int il2 = nh > 0 ? nh - 1 : 0, ih2 = nh + 1 < n ? nh + 1 : n - 1;
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + lane);
      double vlo = W.vlo[lane];
      if (il2 < cLj) {
        vlo = (double)ord_to(W.nmaxL[lane], TK());
        mglo[h] = vlo;
      }
      if (ih2 >= cLj + cM) mghi[h] = (double)ord_to(W.nminR[lane], TK());
      if (nh > 0 && !(vlo < W.vthr[lane])) atomicAdd(tie_count, 1ULL);
    }
    wsync();
    WSTAMP();
  }
  if (overflow) {
    if (lane == 0) {
      ovf_flags[si] = 1u;
      atomicAdd(ovf_count, 1u);
    }
    return;
  }

  WSTAMP();
  // ---- scatter every point to the slot range of its terminal node ----
  // lane v < 32: offset of the terminal whose left-aligned path is v
  if (lane < (1 << kWRmax)) {
    int off = 0, n = n_top, d = 0;
    while (d < depth && !is_leaf_dev(level0 + d, n, L, min_leaf)) {
      const int nh = n >> 1;
      if ((lane >> (kWRmax - 1 - d)) & 1) {
        off += nh;
        n -= nh;
      } else {
        n = nh;
      }
      ++d;
    }
    W.toff[lane] = off;
    W.tcur[lane] = 0;
  }
  wsync();
  int32_t* of = F + (int64_t)t * N + sg.off;
  int32_t* on = nxt + (int64_t)t * N + sg.off;
#pragma unroll
  for (int e = 0; e < kWE; ++e) {
    if (id[e] < 0) continue;
    const bool done = st[e] < 0;
    const int code = -st[e];
    const int v = done ? ((code >> 8) & 31) : (st[e] << (kWRmax - depth));
    const bool ordered = done && ((code >> 23) & 1);
    const int slot = W.toff[v] + (ordered ? ((code >> 13) & 1023) : atomicAdd(&W.tcur[v], 1));
    if (done) {
      of[slot] = id[e];
    } else {
      on[slot] = id[e];
    }
  }
  WSTAMP();
#undef WSTAMP
}

// ---------------------------------------------------------------------------------------
// wave-centric subtree kernel, SORT variant (round 3; replaces wsub_kernel's histogram select
// on the default path, same interface, same results).  ONE WAVE per (tree, top node with
// n <= kWCap), up to kWRmax levels.  The wave holds the node's <= 1024 points as 32-bit sort keys,
// 16 CONSECUTIVE positions per lane (position = lane * 16 + r), and every level is ONE bitonic
// sort of all of them:
//     key = path of the point's node (left-aligned, depth + 1 bits) | monotone image of its
//           projection on the level's hyperplane (21 - depth bits) | local index (10 bits)
// so the nodes of the level stay where they are, each is sorted by projection, and the cut of
// Internal.hs:495-505 is a matter of POSITION: left child = the first nh positions of the node,
// thr / margins = the keys at positions nh, nh - 1, nh + 1, and the last level's sort leaves the
// leaf buckets in the reference's order (children inherit the sorted order).  With consecutive
// positions per lane the four closest exchange distances of the network are register-to-register
// and only 21 of its 55 stages cross lanes (DPP quad permutes, ds_swizzle, v_permlane32_swap); a
// lane whose block sorts descending keeps its keys complemented, so every exchange is a plain
// min / max.  tools/micro/wsort_bench.hip: 0.105 ms per 32 768 sorts (C2's 32 x 1024 nodes).
// The image is weakly monotone (subtract, multiply by a positive scale, truncate), so a smaller
// image means a smaller projection; positions whose image equals a neighbour's — about one pair
// every third (wave, level) on continuous data, every real tie — are ordered EXACTLY afterwards:
// their true keys (fetched again by id) go to a small LDS pool, each member counts the run mates
// that precede it by Keys::less (projection, then the earlier levels' projections, then the id)
// and the run's keys are rewritten in that order.  More than kSPool such positions in one level
// (heavy ties) flag the node for subtree_kernel, exactly like wsub_kernel's pool overflow.
// Points are identified by a local index; the ids sit in LDS (4 KB per wave), nothing else moves.
// PK variant (launches that cover <= 3 levels and most of the point set): the images come from
// ONE packed word per point (wpack_kernel below), gathered once, instead of a key gather per level.
// Measured at C2 (32 x 1024 nodes of 977 points, 3 levels; rocprofv3, profiles/r03_*): 0.83 ms +
// 0.19 ms (wpack) + 0.02 (wgeo) against wsub_kernel's 1.29 ms; without the packed words 1.25 ms:
// the three key gathers per point (8.6 GB of sector traffic) bound both.  The PK kernel is VALU
// bound: 9.4 k VALU instructions per wave (SQ_INSTS_VALU), 60 % of all SIMD cycles, 5.4 k of them
// the three sorts.
// grid = ceil(S*T/4) blocks of 256 threads.
// ---------------------------------------------------------------------------------------
constexpr int kSPool = 128;  // positions per level a wave can order exactly
constexpr int kPkBits = 21;  // image bits per level in a packed word (three levels per u64)
constexpr int kPkLevels = 3;

template <bool PK>
struct WSortSlab {
  // !PK: point id by local index (int[kWCap]); PK: the packed images by local index.  At the end
  // both hold the id by position (bit 31 = node still active) for the coalesced stores.
  unsigned long long store[PK ? kWCap : kWCap / 2];
  double tk[kSPool];
  unsigned int rkey[kSPool + 1], rout[kSPool + 1];  // + a dummy slot
  int rid[kSPool];
};

template <int D>
__device__ __forceinline__ unsigned int ws_lane_xor(unsigned int v) {
  if constexpr (D == 1) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
  else if constexpr (D == 2) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
  else if constexpr (D == 4) return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
  else if constexpr (D == 8) return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);
  else return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
}

// ascending compare-exchange with the lane D away; upper = this lane keeps the larger key
template <int D>
__device__ __forceinline__ void ws_cross(unsigned int (&k)[kWE], bool upper) {
#pragma unroll
  for (int r = 0; r < kWE; ++r) {
    unsigned int a, b;
    if constexpr (D == 32) {  // both halves of the wave see (low half's key, high half's key)
      const auto sw = __builtin_amdgcn_permlane32_swap(k[r], k[r], false, false);
      a = sw[0];
      b = sw[1];
    } else {
      a = k[r];
      b = ws_lane_xor<D>(k[r]);
    }
    const unsigned int mn = a < b ? a : b, mx = a < b ? b : a;
    k[r] = upper ? mx : mn;
  }
}

template <int J>
__device__ __forceinline__ void ws_reg(unsigned int (&k)[kWE]) {
#pragma unroll
  for (int r = 0; r < kWE; ++r)
    if ((r & J) == 0) {
      const unsigned int a = k[r], b = k[r + J];
      k[r] = a < b ? a : b;
      k[r + J] = a < b ? b : a;
    }
}

// phase KK of the bitonic network over index = lane * 16 + r: blocks of KK sorted, direction =
// bit KK of the index (descending blocks are kept complemented)
template <int KK>
__device__ __forceinline__ void ws_phase(unsigned int (&k)[kWE], int lane, unsigned int& flip) {
  if constexpr (KK < 16) {
#pragma unroll
    for (int r = 0; r < kWE; ++r)
      if (r & KK) k[r] = ~k[r];
    if constexpr (KK >= 8) ws_reg<4>(k);
    if constexpr (KK >= 4) ws_reg<2>(k);
    ws_reg<1>(k);
#pragma unroll
    for (int r = 0; r < kWE; ++r)
      if (r & KK) k[r] = ~k[r];
  } else {
    const unsigned int want = (KK < 1024 && ((lane * 16) & KK)) ? ~0u : 0u;
    const unsigned int x = want ^ flip;
    flip = want;
#pragma unroll
    for (int r = 0; r < kWE; ++r) k[r] ^= x;
    if constexpr (KK >= 1024) ws_cross<32>(k, (lane & 32) != 0);
    if constexpr (KK >= 512) ws_cross<16>(k, (lane & 16) != 0);
    if constexpr (KK >= 256) ws_cross<8>(k, (lane & 8) != 0);
    if constexpr (KK >= 128) ws_cross<4>(k, (lane & 4) != 0);
    if constexpr (KK >= 64) ws_cross<2>(k, (lane & 2) != 0);
    if constexpr (KK >= 32) ws_cross<1>(k, (lane & 1) != 0);
    ws_reg<8>(k);
    ws_reg<4>(k);
    ws_reg<2>(k);
    ws_reg<1>(k);
  }
}

__device__ __forceinline__ void ws_sort1024(unsigned int (&k)[kWE], int lane) {
  unsigned int flip = 0;
  ws_phase<2>(k, lane, flip);
  ws_phase<4>(k, lane, flip);
  ws_phase<8>(k, lane, flip);
  ws_phase<16>(k, lane, flip);
  ws_phase<32>(k, lane, flip);
  ws_phase<64>(k, lane, flip);
  ws_phase<128>(k, lane, flip);
  ws_phase<256>(k, lane, flip);
  ws_phase<512>(k, lane, flip);
  ws_phase<1024>(k, lane, flip);  // ascending everywhere: the keys leave un-complemented
}

// Node of a position, packed: offset inside the top node (bits 0-10) | size (11-21) | path,
// left-aligned to the current depth (22-26) | leaf (27).  ws_step takes it one level down
// (Internal.hs:289 leaf test, :495,503 halves); branch-free, a leaf only shifts its path.
constexpr unsigned int kWsLeaf = 1u << 27;
__device__ __forceinline__ unsigned int ws_step(unsigned int info, int p, int next_level, int L,
                                                int min_leaf) {
  const int off = (int)(info & 2047u), n = (int)((info >> 11) & 2047u);
  const unsigned int path = (info >> 22) & 31u;
  const bool leaf = (info & kWsLeaf) != 0;
  const int nh = n >> 1;
  const bool right = !leaf && p - off >= nh;
  const int off2 = off + (right ? nh : 0);
  const int n2 = leaf ? n : (right ? n - nh : nh);
  const bool leaf2 = leaf || is_leaf_dev(next_level, n2, L, min_leaf);
  return (unsigned int)off2 | ((unsigned int)n2 << 11) | ((2u * path + (right ? 1u : 0u)) << 22) |
         (leaf2 ? kWsLeaf : 0u);
}

// ---- packed images (PK): one u64 per point with the 21-bit images of the nl <= 3 levels a wave
// launch covers, written in ID order by one coalesced pass over the projections (wpack_kernel) and
// gathered ONCE per point by wsort_kernel — instead of one 8-byte key gather (a 64-byte sector
// of HBM traffic each) per point and LEVEL.  One geometry per (tree, level) from a strided sample
// of the column (wgeo_kernel); values outside the sampled range clamp into the edge images and
// are ordered by the exact fix-up like any other equal images.
template <class TK>
__global__ __launch_bounds__(256) void wgeo_kernel(const TK* __restrict__ P, int64_t N, int L,
                                                   int level0, int nl, double* __restrict__ geo) {
  const int j = blockIdx.x, t = blockIdx.y;
  const TK* col = P + ((int64_t)t * L + level0 + j) * N;
  const int64_t m = N < 4096 ? N : 4096;
  double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
  for (int64_t i = threadIdx.x; i < m; i += 256) {
    const double v = (double)col[i * N / m];
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
  __shared__ double smn[256], smx[256];
  smn[threadIdx.x] = mn;
  smx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double a = smn[threadIdx.x + o], b = smx[threadIdx.x + o];
      if (a < smn[threadIdx.x]) smn[threadIdx.x] = a;
      if (b > smx[threadIdx.x]) smx[threadIdx.x] = b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double lo = smn[0], hi = smx[0];
    const double w = hi - lo;
    const bool ok = lo < hi && w < 1e300;
    geo[((int64_t)t * nl + j) * 2] = ok ? lo - 0.02 * w : 0.0;
    geo[((int64_t)t * nl + j) * 2 + 1] = ok ? (double)(1u << kPkBits) / (1.04 * w) : 0.0;
  }
}

template <class TK>
__global__ __launch_bounds__(256) void wpack_kernel(const TK* __restrict__ P, int64_t N, int L,
                                                    int level0, int nl, const double* __restrict__ geo,
                                                    unsigned long long* __restrict__ packed) {
  const int t = blockIdx.y;
  double lo[kPkLevels], sc[kPkLevels];
#pragma unroll
  for (int j = 0; j < kPkLevels; ++j) {
    lo[j] = j < nl ? geo[((int64_t)t * nl + j) * 2] : 0.0;
    sc[j] = j < nl ? geo[((int64_t)t * nl + j) * 2 + 1] : 0.0;
  }
  const TK* base = P + ((int64_t)t * L + level0) * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    unsigned long long w = 0;
#pragma unroll
    for (int j = 0; j < kPkLevels; ++j)
      if (j < nl) {
        const double fq = ((double)base[(int64_t)j * N + i] - lo[j]) * sc[j];
        unsigned int c;
        asm("v_cvt_u32_f64 %0, %1" : "=v"(c) : "v"(fq));  // saturating; weakly monotone in the key
        c = c < (1u << kPkBits) - 1u ? c : (1u << kPkBits) - 1u;
        w |= (unsigned long long)c << (kPkBits * j);
      }
    packed[(int64_t)t * N + i] = w;
  }
}

template <class TK, bool PK>
__global__ __launch_bounds__(256, 3) void wsort_kernel(
    const int32_t* __restrict__ src, int32_t* __restrict__ nxt, int32_t* __restrict__ F,
    int64_t N, const TK* __restrict__ P, int L, int T, int level0, int min_leaf,
    const Seg* __restrict__ segs, int S, double* thr, double* mglo, double* mghi,
    int64_t nodes, unsigned long long* tie_count, unsigned int* ovf_flags,
    unsigned int* ovf_count, const unsigned int* __restrict__ abort,
    const unsigned long long* __restrict__ packed) {
  __shared__ WSortSlab<PK> slabs[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wg = (int64_t)blockIdx.x * 4 + wave;
  if (wg >= (int64_t)S * T) return;
  if (abort && *abort) return;  // the streaming levels above are being rebuilt by the host
  const int si = (int)(wg % S), t = (int)(wg / S);
  WSortSlab<PK>& W = slabs[wave];
  int* wids = reinterpret_cast<int*>(W.store);
  const Seg sg = segs[si];
  const int n_top = sg.n;
  if (n_top <= 0) return;
  const TK* Pt = P + (int64_t)t * L * N;
  const int32_t* s = src + (int64_t)t * N + sg.off;
  auto id_of = [&](unsigned int label) -> int { return PK ? s[label] : wids[label]; };

  // local index i = e * 64 + lane names point s[i] (coalesced); position p starts with index p
#pragma unroll
  for (int e = 0; e < kWE; ++e) {
    const int i = e * 64 + lane;
    const int id = i < n_top ? s[i] : 0;
    if constexpr (PK) W.store[i] = i < n_top ? packed[(int64_t)t * N + id] : 0ULL;
    else wids[i] = id;
  }
  unsigned int k[kWE], info[kWE];
  {
    const unsigned int top = ((unsigned int)n_top << 11) |
                             (is_leaf_dev(level0, n_top, L, min_leaf) ? kWsLeaf : 0u);
#pragma unroll
    for (int r = 0; r < kWE; ++r) {
      k[r] = (unsigned int)(lane * kWE + r);
      info[r] = top;
    }
  }
  wsync();

  int depth = 0;
  bool overflow = false;
  for (; depth < kWRmax; ++depth) {
    const int level = level0 + depth;
    if (level >= L) break;
    bool act = false;
#pragma unroll
    for (int r = 0; r < kWE; ++r) act = act || (lane * kWE + r < n_top && !(info[r] & kWsLeaf));
    if (!__any(act)) break;

    Keys<TK> K{Pt, N, level, nullptr};
    const TK* Pl = Pt + (int64_t)level * N;
    const int ib = 21 - depth;                  // image bits of this level's sort keys
    const unsigned int imax = (1u << ib) - 1u;
    unsigned int img[kWE];
    if constexpr (PK) {
      // ---- a. images from the packed words ----
#pragma unroll
      for (int r = 0; r < kWE; ++r) {
        const unsigned long long w = W.store[k[r] & 1023u];
        img[r] = ((unsigned int)(w >> (kPkBits * depth)) & ((1u << kPkBits) - 1u)) >> depth;
      }
    } else {
      // ---- a. keys of this level, gathered by id; one value range for all nodes of the level ----
      TK key[kWE];
#pragma unroll
      for (int r = 0; r < kWE; ++r) {
        const bool a = lane * kWE + r < n_top && !(info[r] & kWsLeaf);
        key[r] = a ? Pl[wids[k[r] & 1023u]] : (TK)0;
      }
      double kmn = __builtin_huge_val(), kmx = -__builtin_huge_val();
#pragma unroll
      for (int r = 0; r < kWE; ++r)
        if (lane * kWE + r < n_top && !(info[r] & kWsLeaf)) {
          const double v = (double)key[r];
          kmn = v < kmn ? v : kmn;
          kmx = v > kmx ? v : kmx;
        }
      for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_xor(kmn, o), b = __shfl_xor(kmx, o);
        kmn = a < kmn ? a : kmn;
        kmx = b > kmx ? b : kmx;
      }
      const double scale = kmn < kmx ? (double)(1u << ib) / (kmx - kmn) : 0.0;
#pragma unroll
      for (int r = 0; r < kWE; ++r) {
        const double fq = ((double)key[r] - kmn) * scale;
        unsigned int c;
        asm("v_cvt_u32_f64 %0, %1" : "=v"(c) : "v"(fq));  // saturating
        img[r] = c < imax ? c : imax;
      }
    }
    // ---- b. sort keys: node path | image | local index; a leaf keeps its order ----
#pragma unroll
    for (int r = 0; r < kWE; ++r) {
      const int p = lane * kWE + r;
      const unsigned int path = (info[r] >> 22) & 31u;
      const unsigned int im = (info[r] & kWsLeaf) ? (unsigned int)(p - (int)(info[r] & 2047u)) : img[r];
      k[r] = p < n_top ? (path << (ib + 10)) | (im << 10) | (k[r] & 1023u)
                       : ~0u;  // padding: behind every real key (their top bit is 0)
    }
    ws_sort1024(k, lane);
    // ---- c. positions whose image equals a neighbour's: exact order inside those runs ----
    {
      unsigned int bits = 0;
      const unsigned int up = (unsigned int)__shfl_up((int)k[kWE - 1], 1);
      const unsigned int dn = (unsigned int)__shfl_down((int)k[0], 1);
#pragma unroll
      for (int r = 0; r < kWE; ++r) {
        const unsigned int me = k[r] >> 10;
        const bool eqp = r > 0 ? (k[r > 0 ? r - 1 : 0] >> 10) == me : (lane > 0 && (up >> 10) == me);
        const bool eqn = r + 1 < kWE ? (k[r + 1 < kWE ? r + 1 : r] >> 10) == me
                                     : (lane < 63 && (dn >> 10) == me);
        if (k[r] != ~0u && (eqp || eqn)) bits |= 1u << r;
      }
      if (__any(bits != 0)) {
        const int cnt = __popc(bits);
        int inc = cnt;
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(inc, o);
          if (lane >= o) inc += v;
        }
        const int tot = __shfl(inc, 63);
        if (tot > kSPool) {
          overflow = true;
          break;
        }
        const int base = inc - cnt;
        // (slot kSPool is a dummy: unconditional LDS accesses instead of sixteen divergent regions)
#pragma unroll
        for (int r = 0; r < kWE; ++r)
          W.rkey[(bits & (1u << r)) ? base + __popc(bits & ((1u << r) - 1u)) : kSPool] = k[r];
        wsync();
        // by pool slot from here on: the true key of every member, fetched again by id
        for (int sl = lane; sl < tot; sl += 64) {
          const int id = id_of(W.rkey[sl] & 1023u);
          W.rid[sl] = id;
          W.tk[sl] = (double)Pl[id];
        }
        wsync();
        for (int sl = lane; sl < tot; sl += 64) {
          const unsigned int mk = W.rkey[sl];
          const TK tkm = (TK)W.tk[sl];
          const int idm = W.rid[sl];
          int nleft = 0, less = 0;
          for (int q = sl - 1; q >= 0 && (W.rkey[q] >> 10) == (mk >> 10); --q) {
            ++nleft;
            less += K.less((TK)W.tk[q], W.rid[q], tkm, idm) ? 1 : 0;
          }
          for (int q = sl + 1; q < tot && (W.rkey[q] >> 10) == (mk >> 10); ++q)
            less += K.less((TK)W.tk[q], W.rid[q], tkm, idm) ? 1 : 0;
          W.rout[sl - nleft + less] = mk;
        }
        wsync();
#pragma unroll
        for (int r = 0; r < kWE; ++r) {
          const unsigned int v = W.rout[(bits & (1u << r)) ? base + __popc(bits & ((1u << r) - 1u)) : kSPool];
          k[r] = (bits & (1u << r)) ? v : k[r];
        }
        wsync();
      }
    }
    // ---- d. thresholds and margins: the keys at positions nh, nh - 1, nh + 1 of every node
    // (Internal.hs:496-501; n == 2 -> (p'[0], p'[1]); n == 1 -> p'[0] for all three).  One lane
    // per (node, value): it works out the node's range from (n_top, depth, j) — the nodes of a
    // depth are all Bins or all beyond a Tip of an ancestor, and sizes differ by at most one —,
    // fetches the sort key at the position it needs by sixteen lane reads (its register index is
    // data) and the true key by id ----
    {
      const int j = lane / 3, which = lane - 3 * j;
      int off = 0, n = n_top;
      bool bin = lane < (3 << depth);
      for (int b = depth - 1; b >= 0; --b) {  // the path of node j from the top node
        bin = bin && !is_leaf_dev(level - 1 - b, n, L, min_leaf);
        const int nhb = n >> 1;
        if ((j >> b) & 1) {
          off += nhb;
          n -= nhb;
        } else {
          n = nhb;
        }
      }
      bin = bin && !is_leaf_dev(level, n, L, min_leaf);
      const int nh = n >> 1;
      const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
      const int pt = off + (which == 0 ? nh : which == 1 ? il : ih);
      const int srcl = bin ? pt >> 4 : lane, srcr = pt & 15;
      unsigned int kt = 0;
#pragma unroll
      for (int r = 0; r < kWE; ++r) {
        const unsigned int v = (unsigned int)__shfl((int)k[r], srcl);
        kt = srcr == r ? v : kt;
      }
      double v = 0.0;
      if (bin) {
        v = (double)Pl[id_of(kt & 1023u)];
        const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + j);
        if (which == 0) thr[h] = v;
        else if (which == 1) mglo[h] = v;
        else mghi[h] = v;
      }
      // a cut that straddles a tie (statistics): p'[nh - 1] == p'[nh]
      const double vthr = __shfl(v, j * 3 < 64 ? j * 3 : 0), vlo = __shfl(v, j * 3 + 1 < 64 ? j * 3 + 1 : 0);
      if (bin && which == 0 && nh > 0 && !(vlo < vthr)) atomicAdd(tie_count, 1ULL);
    }
    // ---- e. every position one level down ----
#pragma unroll
    for (int r = 0; r < kWE; ++r) info[r] = ws_step(info[r], lane * kWE + r, level + 1, L, min_leaf);
    wsync();
  }
  if (overflow) {
    if (lane == 0) {
      ovf_flags[si] = 1u;
      atomicAdd(ovf_count, 1u);
    }
    return;
  }
  // ---- every point to its slot: position = slot.  Leaves go to the final perm (in their final
  // order), nodes still active after kWRmax levels to nxt.  Through LDS, so that the stores are
  // coalesced ----
  int idr[kWE];
#pragma unroll
  for (int r = 0; r < kWE; ++r) {
    const bool real = lane * kWE + r < n_top;
    idr[r] = (real ? id_of(k[r] & 1023u) : 0) | ((info[r] & kWsLeaf) ? 0 : (int)0x80000000);
  }
  wsync();
#pragma unroll
  for (int r = 0; r < kWE; ++r) wids[lane * kWE + r] = idr[r];
  wsync();
  int32_t* of = F + (int64_t)t * N + sg.off;
  int32_t* on = nxt + (int64_t)t * N + sg.off;
#pragma unroll
  for (int e = 0; e < kWE; ++e) {
    const int p = e * 64 + lane;
    if (p >= n_top) continue;
    const int v = wids[p];
    if (v < 0) on[p] = v & 0x7fffffff;
    else of[p] = v;
  }
}

// copy segments src -> dst unchanged (leaves that are already in final order). grid=(S,T)
__global__ void copy_segs_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                 int64_t N, const Seg* __restrict__ segs) {
  const Seg sg = segs[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.y * N + sg.off;
  for (int i = threadIdx.x; i < sg.n; i += blockDim.x) dst[base + i] = src[base + i];
}

__global__ void iota_kernel(int32_t* perm, int64_t N, int T) {
  const int64_t total = N * T;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    perm[i] = (int32_t)(i % N);
}

__global__ void fill_f64_kernel(double* p, int64_t n, double v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// ---------------------------------------------------------------------------------------
// big path
// ---------------------------------------------------------------------------------------
template <class TK>
struct BinInfo {  // per (tree, node): value bins around the median
  TK lo, hi, scale;
};
struct NodeAux {  // per (tree, node)
  int pivot_bin;
  int cL;    // elements in bins below the pivot bin
  int cMid;  // elements in the pivot bin
  unsigned int curL, curM, curR;        // scatter cursors
  unsigned long long maxL, minR;        // ordered images of max(left) / min(right)
};

template <class TK>
__device__ inline int bin_of(TK key, const BinInfo<TK>& b) {
  if (key < b.lo) return 0;
  if (!(key < b.hi)) return kNB - 1;
  const int v = 1 + (int)((key - b.lo) * b.scale);
  return v < kNB - 2 ? v : kNB - 2;
}

// sample kSample keys of the node at equidistant positions, sort them, pick the splitters.
// grid = (S, T), 256 threads.
template <class TK>
__global__ __launch_bounds__(256) void sample_kernel(const int32_t* __restrict__ src, int64_t N,
                                                     const TK* P, int L, int level,
                                                     const Seg* __restrict__ segs, int S,
                                                     BinInfo<TK>* bins) {
  __shared__ TK skey[kSample];
  __shared__ int sid[kSample];
  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  Keys<TK> K{P + (int64_t)t * L * N, N, level, nullptr};
  const int32_t* s = src + (int64_t)t * N + sg.off;
  for (int j = threadIdx.x; j < kSample; j += blockDim.x) {
    const int64_t pos = ((int64_t)j * sg.n) / kSample;
    const int id = s[pos];
    sid[j] = j;  // ties between samples do not matter: any order gives valid splitters
    skey[j] = K.key(id);
  }
  __syncthreads();
  Keys<TK> K0{nullptr, 0, 0, nullptr};  // level 0, no tb: tie -> index order, no HBM access
  lds_bitonic(skey, sid, kSample, K0);
  if (threadIdx.x == 0) {
    const int nh = sg.n / 2;
    int rs = (int)(((int64_t)nh * kSample) / sg.n);
    int a = rs - kDelta, b = rs + kDelta;
    if (a < 0) a = 0;
    if (b > kSample - 1) b = kSample - 1;
    BinInfo<TK> bi;
    bi.lo = skey[a];
    bi.hi = skey[b];
    bi.scale = bi.lo < bi.hi ? (TK)(kNB - 2) / (bi.hi - bi.lo) : (TK)0;
    bins[(int64_t)t * S + blockIdx.x] = bi;
  }
}

// histogram pass: gathers the keys of the node (perm order), stashes them contiguously and
// counts value bins.  grid = (chunks, S, T), 256 threads, kChunk elements per block.
template <class TK>
__global__ __launch_bounds__(256) void hist_kernel(const int32_t* __restrict__ src, int64_t N,
                                                   const TK* P, int L, int level,
                                                   const Seg* __restrict__ segs, int S,
                                                   const BinInfo<TK>* __restrict__ bins,
                                                   TK* __restrict__ Kst, unsigned int* hist) {
  __shared__ unsigned int sh[kNB];
  const Seg sg = segs[blockIdx.y];
  const int t = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.x * kChunk;
  if (c0 >= sg.n) return;
  for (int i = threadIdx.x; i < kNB; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const BinInfo<TK> bi = bins[(int64_t)t * S + blockIdx.y];
  const TK* Pl = P + ((int64_t)t * L + level) * N;
  const int64_t base = (int64_t)t * N + sg.off;
  unsigned int c_lo = 0, c_hi = 0;
  for (int j = 0; j < kChunk / 256; ++j) {
    const int64_t i = c0 + j * 256 + threadIdx.x;
    if (i < sg.n) {
      const int id = src[base + i];
      const TK key = Pl[id];
      Kst[base + i] = key;
      const int b = bin_of(key, bi);
      if (b == 0) ++c_lo;
      else if (b == kNB - 1) ++c_hi;
      else atomicAdd(&sh[b], 1u);
    }
  }
  // the two tails take ~90% of the points: reduce them per wave, not through LDS atomics
  for (int o = 32; o > 0; o >>= 1) {
    c_lo += __shfl_down(c_lo, o);
    c_hi += __shfl_down(c_hi, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sh[0], c_lo);
    atomicAdd(&sh[kNB - 1], c_hi);
  }
  __syncthreads();
  unsigned int* gh = hist + ((int64_t)t * S + blockIdx.y) * kNB;
  for (int i = threadIdx.x; i < kNB; i += blockDim.x)
    if (sh[i]) atomicAdd(&gh[i], sh[i]);
}

// pick the bin holding rank nh; reset the histogram for the next level. grid=(S,T), 256 thr
__global__ __launch_bounds__(256) void pick_kernel(const Seg* __restrict__ segs, int S,
                                                   unsigned int* hist, NodeAux* aux) {
  __shared__ unsigned int part[256];
  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  unsigned int* gh = hist + ((int64_t)t * S + blockIdx.x) * kNB;
  constexpr int PER = kNB / 256;
  unsigned int v[PER], s = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    v[j] = gh[threadIdx.x * PER + j];
    s += v[j];
    gh[threadIdx.x * PER + j] = 0;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of 256 partial sums (Hillis-Steele)
  for (int o = 1; o < 256; o <<= 1) {
    unsigned int add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned int run = part[threadIdx.x] - s;  // exclusive prefix
  const unsigned int nh = (unsigned int)(sg.n / 2);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (run <= nh && nh < run + v[j]) {
      NodeAux a;
      a.pivot_bin = threadIdx.x * PER + j;
      a.cL = (int)run;
      a.cMid = (int)v[j];
      a.curL = a.curM = a.curR = 0;
      a.maxL = 0ULL;
      a.minR = ~0ULL;
      aux[(int64_t)t * S + blockIdx.x] = a;
    }
    run += v[j];
  }
}

// 3-way scatter. grid = (chunks, S, T), 256 threads, kChunk elements per block.
template <class TK>
__global__ __launch_bounds__(256) void scatter_kernel(const int32_t* __restrict__ src,
                                                      int32_t* __restrict__ dst, int64_t N,
                                                      const Seg* __restrict__ segs, int S,
                                                      const BinInfo<TK>* __restrict__ bins,
                                                      const TK* __restrict__ Kst, NodeAux* aux) {
  constexpr int IT = kChunk / 256;
  __shared__ int cnt[3][IT][4];
  __shared__ unsigned int gbase[3];
  const Seg sg = segs[blockIdx.y];
  const int t = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.x * kChunk;
  if (c0 >= sg.n) return;
  NodeAux* A = &aux[(int64_t)t * S + blockIdx.y];
  const int pb = A->pivot_bin, cL = A->cL, cMid = A->cMid;
  const BinInfo<TK> bi = bins[(int64_t)t * S + blockIdx.y];
  const int64_t base = (int64_t)t * N + sg.off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ULL << lane) - 1ULL;

  int ids[IT];
  unsigned int cls_bits = 0;  // 2 bits per iteration: 0 left, 1 mid, 2 right, 3 invalid
  unsigned long long mxL = 0ULL, mnR = ~0ULL;
#pragma unroll
  for (int j = 0; j < IT; ++j) {
    const int64_t i = c0 + j * 256 + threadIdx.x;
    int cls = 3;
    ids[j] = 0;
    if (i < sg.n) {
      ids[j] = src[base + i];
      const TK key = Kst[base + i];
      const int b = bin_of(key, bi);
      cls = b < pb ? 0 : (b > pb ? 2 : 1);
      const unsigned long long o = ord_of(key);
      if (cls == 0) mxL = o > mxL ? o : mxL;
      if (cls == 2) mnR = o < mnR ? o : mnR;
    }
    cls_bits |= (unsigned int)cls << (2 * j);
    const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1),
                             m2 = __ballot(cls == 2);
    if (lane == 0) {
      cnt[0][j][wave] = __popcll(m0);
      cnt[1][j][wave] = __popcll(m1);
      cnt[2][j][wave] = __popcll(m2);
    }
  }
  // max(left) / min(right): wave reduce, one atomic per wave
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_down(mxL, o), b = __shfl_down(mnR, o);
    mxL = a > mxL ? a : mxL;
    mnR = b < mnR ? b : mnR;
  }
  if (lane == 0) {
    if (mxL != 0ULL) atomicMax(&A->maxL, mxL);
    if (mnR != ~0ULL) atomicMin(&A->minR, mnR);
  }
  __syncthreads();
  if (threadIdx.x < 3) {  // exclusive scan over (iteration, wave) + one reservation per class
    const int c = threadIdx.x;
    int run = 0;
    for (int j = 0; j < IT; ++j)
      for (int w = 0; w < 4; ++w) {
        const int v = cnt[c][j][w];
        cnt[c][j][w] = run;
        run += v;
      }
    unsigned int* cur = c == 0 ? &A->curL : (c == 1 ? &A->curM : &A->curR);
    gbase[c] = run ? atomicAdd(cur, (unsigned int)run) : 0u;
  }
  __syncthreads();
  const unsigned int start[3] = {0u, (unsigned int)cL, (unsigned int)(cL + cMid)};
#pragma unroll
  for (int j = 0; j < IT; ++j) {
    const int cls = (cls_bits >> (2 * j)) & 3;
    const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1),
                             m2 = __ballot(cls == 2);
    if (cls < 3) {
      const unsigned long long m = cls == 0 ? m0 : (cls == 1 ? m1 : m2);
      const unsigned int pos = start[cls] + gbase[cls] + (unsigned int)cnt[cls][j][wave] +
                               (unsigned int)__popcll(m & lt);
      dst[base + pos] = ids[j];
    }
  }
}

// sort the pivot bin in place and emit thr / margins. grid = (S, T), 256 threads.
// A pivot bin larger than kSmallCap is left to the HBM merge sort (flagged in big_flags).
template <class TK>
__global__ __launch_bounds__(256) void mid_kernel(int32_t* __restrict__ dst, int64_t N,
                                                  const TK* P, int L, int level,
                                                  const Seg* __restrict__ segs, int S,
                                                  const NodeAux* __restrict__ aux, double* thr,
                                                  double* mglo, double* mghi, int64_t nodes,
                                                  unsigned long long* tie_count,
                                                  unsigned int* big_flags /*[T][S]*/,
                                                  unsigned int* big_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  const NodeAux A = aux[(int64_t)t * S + blockIdx.x];
  const int cL = A.cL, cMid = A.cMid, n = sg.n;
  if (cMid > kSmallCap) {
    if (threadIdx.x == 0) {
      big_flags[(int64_t)t * S + blockIdx.x] = 1u;
      atomicAdd(big_count, 1u);
    }
    return;
  }
  const int np = next_pow2(cMid);
  TK* skey = reinterpret_cast<TK*>(smem);
  int* sid = reinterpret_cast<int*>(smem + (size_t)np * sizeof(TK));
  Keys<TK> K{P + (int64_t)t * L * N, N, level, nullptr};
  int32_t* m = dst + (int64_t)t * N + sg.off + cL;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    if (i < cMid) {
      const int id = m[i];
      sid[i] = id;
      skey[i] = K.key(id);
    } else {
      sid[i] = kPad;
      skey[i] = pos_inf<TK>();
    }
  }
  __syncthreads();
  lds_bitonic(skey, sid, np, K);
  for (int i = threadIdx.x; i < cMid; i += blockDim.x) m[i] = sid[i];
  if (threadIdx.x == 0) {
    const int nh = n / 2;
    const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
    const int64_t h = (int64_t)t * nodes + sg.heap;
    const TK vthr = skey[nh - cL];
    const TK vlo = il >= cL ? skey[il - cL] : ord_to(A.maxL, TK());
    const TK vhi = ih < cL + cMid ? skey[ih - cL] : ord_to(A.minR, TK());
    thr[h] = (double)vthr;
    mglo[h] = (double)vlo;
    mghi[h] = (double)vhi;
    if (nh > 0 && !(vlo < vthr)) atomicAdd(tie_count, 1ULL);
  }
}

// ---------------------------------------------------------------------------------------
// Mid-size nodes on packed codes (round 3): one workgroup per (tree, node of kWCap < n <= kCsCap
// points) runs the r <= 4 levels that bring its children under the wave kernel's size, none of
// which produces a Tip.  Where subtree_kernel gathers an exact key per point and level (a 64-byte
// sector for 4 or 8 bytes), this kernel gathers ONE 8-byte word per point — the 16-bit codes
// (codes.h) of the r levels, packed in id order by cpack_kernel — and selects every level's
// median on the codes: code(x) < code(y) implies x < y, so a two-pass radix histogram (high byte,
// low byte) finds the code of the median element, the points of smaller / larger codes go left /
// right, and exact keys are fetched only for the points that SHARE the pivot code (ranked with
// the lexicographic tie-break, Keys::less) and for those of the nearest occupied codes below and
// above it (the margins, Internal.hs:497-499).  No element moves until the end, where the
// points are written grouped by child (their order inside a child is irrelevant: the wave kernel
// that takes the children orders by keys and ids).  A node whose pivot codes hold more points
// than the pool (heavy ties) is flagged and run again by the general kernels.
// grid = (S, T), kCsThreads threads.
// ---------------------------------------------------------------------------------------
constexpr int kCsCap = 8192;
constexpr int kCsThreads = 512;
constexpr int kCsMaxR = 4;                 // levels per launch = 16-bit codes per packed word
constexpr int kCsMaxM = 1 << (kCsMaxR - 1);
constexpr int kCsPool = 1024;              // points sharing their node's pivot code, per level

// the codes of levels level0 .. level0 + r - 1 of every point, packed: one coalesced pass
__global__ __launch_bounds__(256) void cpack_kernel(const uint16_t* __restrict__ Cd, int64_t N, int L,
                                                    int level0, int r,
                                                    unsigned long long* __restrict__ pk) {
  const int t = blockIdx.y;
  const uint16_t* c = Cd + ((int64_t)t * L + level0) * N;
  unsigned long long* o = pk + (int64_t)t * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    unsigned long long w = 0;
    for (int dd = 0; dd < r; ++dd) w |= (unsigned long long)c[(int64_t)dd * N + i] << (16 * dd);
    o[i] = w;
  }
}

struct CsNode {
  int n, nh, hb, cL, cMid, pc, poff;
};

template <class TK, int E_>
__global__ __launch_bounds__(kCsThreads, (E_ <= 8 ? 6 : 4)) void csub_kernel(
    const int32_t* __restrict__ src, int32_t* __restrict__ nxt, int64_t N, const TK* __restrict__ P,
    const unsigned long long* __restrict__ pk, int L, int level0, int r,
    const Seg* __restrict__ segs, double* thr, double* mglo, double* mghi, int64_t nodes,
    unsigned long long* tie_count, unsigned int* flags /* [0] count, [1 + seg] */) {
  __shared__ unsigned int hist[kCsMaxM * 256], hist2[kCsMaxM * 256];
  __shared__ CsNode sn[kCsMaxM];
  __shared__ int s_lowc[kCsMaxM], s_highc[kCsMaxM], pcur[kCsMaxM];
  __shared__ unsigned long long s_maxL[kCsMaxM], s_minR[kCsMaxM];
  __shared__ __attribute__((aligned(16))) TK pkey[kCsPool];
  __shared__ int pid[kCsPool];
  __shared__ int pown[kCsPool];               // (owner thread << 8) | (slot << 4) | node
  __shared__ unsigned int sidew[kCsThreads];  // per thread: the sides of its pooled slots
  __shared__ int ccur[1 << kCsMaxR];
  __shared__ int s_fail, s_total;
  __shared__ unsigned int s_ties;  // added to the global statistic only if the node completes

  const Seg sg = segs[blockIdx.x];
  const int t = blockIdx.y;
  const int n_top = sg.n;
  if (n_top <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TK* Pt = P + (int64_t)t * L * N;
  const int32_t* s = src + (int64_t)t * N + sg.off;
  const unsigned long long* pkt = pk + (int64_t)t * N;

  int id[E_];
  unsigned long long w[E_];
  unsigned long long node4 = 0;  // 4 bits per slot: the slot's node at the current depth
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    const int pos = e * kCsThreads + tid;
    id[e] = pos < n_top ? s[pos] : -1;
  }
#pragma unroll
  for (int e = 0; e < E_; ++e) w[e] = id[e] >= 0 ? pkt[id[e]] : 0ULL;
  if (tid == 0) {
    s_fail = 0;
    s_ties = 0;
  }

  for (int depth = 0; depth < r; ++depth) {
    const int level = level0 + depth;
    const int M = 1 << depth;
    Keys<TK> K{Pt, N, level, nullptr};
    const TK* Pl = Pt + (int64_t)level * N;
    // ---- a. high-byte histogram ----
    for (int i = tid; i < M * 256; i += kCsThreads) {
      hist[i] = 0;
      hist2[i] = 0;
    }
    if (tid < M) {
      s_lowc[tid] = -1;
      s_highc[tid] = 65536;
      pcur[tid] = 0;
      s_maxL[tid] = 0ULL;
      s_minR[tid] = ~0ULL;
    }
    __syncthreads();
    auto code_of_slot = [&](int e) { return (int)((w[e] >> (16 * depth)) & 0xffffULL); };
#pragma unroll
    for (int e = 0; e < E_; ++e)
      if (id[e] >= 0) atomicAdd(&hist[(int)((node4 >> (4 * e)) & 15ULL) * 256 + (code_of_slot(e) >> 8)], 1u);
    __syncthreads();
    // ---- b. the high byte of the median's code: one wave per node ----
    auto pick = [&](const unsigned int* h, unsigned int target, int& bin, int& before, int& cnt) {
      unsigned int c4[4], loc = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c4[i] = h[lane * 4 + i];
        loc += c4[i];
      }
      unsigned int inc = loc;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
      }
      unsigned int run = inc - loc;
      int pb = -1, cb = 0, cc = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (run <= target && target < run + c4[i]) {
          pb = lane * 4 + i;
          cb = (int)run;
          cc = (int)c4[i];
        }
        run += c4[i];
      }
      const unsigned long long own = __ballot(pb >= 0);
      const int sl = own ? __ffsll((long long)own) - 1 : 0;
      bin = __shfl(pb, sl);
      before = __shfl(cb, sl);
      cnt = __shfl(cc, sl);
    };
    if (wave < M) {
      const int j = wave;
      CsNode a;
      a.n = sub_node_size(n_top, depth, j);
      a.nh = a.n >> 1;
      int hb, before, cnt;
      pick(hist + j * 256, (unsigned int)a.nh, hb, before, cnt);
      a.hb = hb;
      a.cL = before;
      a.cMid = cnt;
      a.pc = 0;
      a.poff = 0;
      if (lane == 0) sn[j] = a;
    }
    __syncthreads();
    // ---- c. low-byte histogram of the points in that high byte ----
#pragma unroll
    for (int e = 0; e < E_; ++e)
      if (id[e] >= 0) {
        const int j = (int)((node4 >> (4 * e)) & 15ULL);
        const int c = code_of_slot(e);
        if ((c >> 8) == sn[j].hb) atomicAdd(&hist2[j * 256 + (c & 255)], 1u);
      }
    __syncthreads();
    if (wave < M) {
      const int j = wave;
      const CsNode a = sn[j];
      int lb, before, cnt;
      pick(hist2 + j * 256, (unsigned int)(a.nh - a.cL), lb, before, cnt);
      if (lane == 0) {
        sn[j].pc = (a.hb << 8) | lb;
        sn[j].cL = a.cL + before;
        sn[j].cMid = cnt;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int j = 0; j < M; ++j) {
        sn[j].poff = run;
        run += sn[j].cMid;
        if (sn[j].hb < 0 || sn[j].cMid <= 0) s_fail = 2;  // cannot happen on a consistent histogram
      }
      if (run > kCsPool && !s_fail) s_fail = 1;
      s_total = run;
    }
    // ---- d. nearest occupied codes around the pivot code: first among the points that share its
    // high byte (a few dozen), all points only where that finds nothing ----
#pragma unroll
    for (int e = 0; e < E_; ++e)
      if (id[e] >= 0) {
        const int j = (int)((node4 >> (4 * e)) & 15ULL);
        const int pc = sn[j].pc, c = code_of_slot(e);
        if ((c >> 8) == (pc >> 8)) {
          if (c < pc) atomicMax(&s_lowc[j], c);
          if (c > pc) atomicMin(&s_highc[j], c);
        }
      }
    __syncthreads();
    if (s_fail) {  // uniform: heavy ties -> the general kernels run this node again
      if (tid == 0) {
        flags[1 + blockIdx.x] = (unsigned int)s_fail | ((unsigned int)depth << 8) | ((unsigned int)s_total << 12);
        atomicAdd(flags, 1u);
      }
      return;
    }
    {
      bool need = false;
      for (int j = 0; j < M; ++j) {
        const CsNode a = sn[j];
        need = need || (s_lowc[j] < 0 && a.cL > 0) || (s_highc[j] > 65535 && a.cL + a.cMid < a.n);
      }
      __syncthreads();  // every thread has read s_lowc / s_highc before any of them changes
      if (need) {       // uniform
#pragma unroll
        for (int e = 0; e < E_; ++e)
          if (id[e] >= 0) {
            const int j = (int)((node4 >> (4 * e)) & 15ULL);
            const int pc = sn[j].pc, c = code_of_slot(e);
            if ((c >> 8) != (pc >> 8)) {
              if (c < pc) atomicMax(&s_lowc[j], c);
              if (c > pc) atomicMin(&s_highc[j], c);
            }
          }
        __syncthreads();
      }
    }
    // ---- e. exact keys: the pivot-code points into the pool (with their owner slot), the
    // neighbours' extremes ----
    sidew[tid] = 0u;
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      if (id[e] < 0) continue;
      const int j = (int)((node4 >> (4 * e)) & 15ULL);
      const int pc = sn[j].pc;
      const int c = (int)((w[e] >> (16 * depth)) & 0xffffULL);
      if (c == pc) {
        const int p = sn[j].poff + atomicAdd(&pcur[j], 1);
        pkey[p] = Pl[id[e]];
        pid[p] = id[e];
        pown[p] = (tid << 8) | (e << 4) | j;
      } else if (c == s_lowc[j]) {
        atomicMax(&s_maxL[j], ord_of(Pl[id[e]]));
      } else if (c == s_highc[j]) {
        atomicMin(&s_minR[j], ord_of(Pl[id[e]]));
      }
    }
    __syncthreads();
    // ---- f. exact rank inside the pivot code (one thread per pooled point); thresholds and
    // margins; the side of a pooled point goes to its owner's word ----
    for (int q = tid; q < s_total; q += kCsThreads) {
      const int ow = pown[q];
      const int j = ow & 15;
      const CsNode a = sn[j];
      const TK kq = pkey[q];
      const int iq = pid[q];
      int rank = 0;
      for (int o = a.poff; o < a.poff + a.cMid; ++o)
        if (o != q && K.less(pkey[o], pid[o], kq, iq)) ++rank;
      const int il = a.nh > 0 ? a.nh - 1 : 0, ih = a.nh + 1 < a.n ? a.nh + 1 : a.n - 1;
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + j);
      if (rank == a.nh - a.cL) thr[h] = (double)kq;
      if (rank == il - a.cL) mglo[h] = (double)kq;
      if (rank == ih - a.cL) mghi[h] = (double)kq;
      if (rank >= a.nh - a.cL) atomicOr(&sidew[ow >> 8], 1u << ((ow >> 4) & 15));
    }
    if (tid < M) {
      const CsNode a = sn[tid];
      const int il = a.nh > 0 ? a.nh - 1 : 0, ih = a.nh + 1 < a.n ? a.nh + 1 : a.n - 1;
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + tid);
      if (il < a.cL) mglo[h] = (double)ord_to(s_maxL[tid], TK());
      if (ih >= a.cL + a.cMid) mghi[h] = (double)ord_to(s_minR[tid], TK());
    }
    __syncthreads();
    unsigned int sidebits = sidew[tid];
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      const int j = (int)((node4 >> (4 * e)) & 15ULL);
      const int c = (int)((w[e] >> (16 * depth)) & 0xffffULL);
      sidebits |= (id[e] >= 0 && c > sn[j].pc ? 1u : 0u) << e;
    }
    // ---- g. descend ----
    {
      unsigned long long nn = 0;
#pragma unroll
      for (int e = 0; e < E_; ++e) {
        const unsigned long long j = (node4 >> (4 * e)) & 15ULL;
        nn |= ((2 * j + ((sidebits >> e) & 1u)) & 15ULL) << (4 * e);
      }
      node4 = nn;
    }
    __syncthreads();
    if (tid < M && sn[tid].n > 1) {  // ties straddling the cut (statistics)
      const int64_t h = (int64_t)t * nodes + ((((int64_t)sg.heap + 1) << depth) - 1 + tid);
      __threadfence_block();
      if (!(mglo[h] < thr[h])) atomicAdd(&s_ties, 1u);
    }
  }

  // ---- the points grouped by child, any order inside a child ----
  const int C = 1 << r;
  if (tid < C) ccur[tid] = 0;
  __syncthreads();
  if (tid == 0 && s_ties) atomicAdd(tie_count, (unsigned long long)s_ties);
  int32_t* on = nxt + (int64_t)t * N + sg.off;
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    const int c = (int)((node4 >> (4 * e)) & 15ULL);
    const bool live = id[e] >= 0;
    int slot = 0;
    for (int j = 0; j < C; ++j) {
      const unsigned long long m = __ballot(live && c == j);
      if (!m) continue;
      int base = 0;
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&ccur[j], __popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1);
      if (live && c == j) slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
    }
    if (live) {
      int toff = 0, tn = n_top;
      for (int b = r - 1; b >= 0; --b) {
        const int nh = tn >> 1;
        if ((c >> b) & 1) {
          toff += nh;
          tn -= nh;
        } else {
          tn = nh;
        }
      }
      on[toff + slot] = id[e];
    }
  }
}

// ---------------------------------------------------------------------------------------
// streaming path for the top levels (every node of the level is a Bin, <= kStreamMaxNodes
// nodes per tree): elements never move.  node_of[t][id] holds the in-level node index of
// every point; one level =
//   stream_hist   : coalesced pass over (key_l[id], node_of[id]) -> per-node value histogram
//                   in LDS (kStreamBins bins split over the nodes), flushed by atomics
//   stream_pick   : one wave per (tree, node): pivot bin, counts, nearest non-empty bins
//   stream_assign : coalesced pass: bin < pivot -> left child, > pivot -> right child, pivot
//                   bin -> appended to the node's mid list; min/max of the NEXT level's key per
//                   child (bin geometry of the next level)
//   stream_mid    : exact order of the pivot bin (LDS sort with the lexicographic tie-break),
//                   thr / margins, children of the mid points
// HBM bytes per point per tree per level: 10 (hist) + 20 (assign) — all coalesced — against
// 36 + a random 64-B sector for the gather-based path.  stream_to_perm finally counting-sorts
// the points by node into the permutation the deeper levels work on.
// A pivot bin larger than LDS (heavy ties / extreme outliers) makes the host leave the
// streaming path at that level (the gather path has the general fallbacks).
// ---------------------------------------------------------------------------------------
constexpr int kStreamMaxNodes = 1024;
// (measured, round 2: 65536 bins — 128 KB of LDS, one block per CU — halve the pivot bins, and what
// stream_assign / stream_mid gain (-0.2 ms per C2 build) stream_hist / stream_pick lose)
constexpr int kStreamBins = 32768;      // histogram entries per block: 16-bit counters packed
                                        // two per LDS word (64 KB); a block sees < 65536 points
constexpr int kStreamThreads = 1024;

template <class TK>
struct SNode {  // per (tree, node) of the current streaming level
  TK lo, scale;
  int n, nh, pb, cL, cMid, lowb, highb, midoff;
  unsigned int midcur;
  unsigned long long maxL, minR;
  unsigned long long cthr;  // code mode: the pivot and margin bins as code thresholds (code_thr)
};

// ---- CODE mode (codes.h): hist bins (TK)code with the ordinary geometry (stream_geom /
// stream_bin); stream_assign gets the pivot and margin bins as CODE THRESHOLDS, found by
// inverting the monotone bin function with a binary search over the 65536 codes — exactly
// consistent with the histogram by construction.  Packed into 4 x 16 bits:
//   [0] T_lo: codes below go left        [1] T_hi: codes above go right   (pivot bin = [T_lo, T_hi])
//   [2] E_lo: codes in [E_lo, T_lo) lie in the nearest non-empty bin below the pivot bin (the
//       bins in between are empty) — tracked for the low margin; 0xffff: not needed
//   [3] E_hi: codes in (T_hi, E_hi] lie in the nearest non-empty bin above; 0: not needed
template <class TK>
__device__ inline int stream_bin(TK key, TK lo, TK scale, int B);
template <class TK>
__device__ inline unsigned long long code_thr(TK lo, TK scale, int B, int pb, int lowb /* < 0: none */,
                                              int highb /* >= B: none */) {
  auto first_ge = [&](int p) -> int {  // smallest code whose bin is >= p (65536: none)
    if (p <= 0) return 0;
    // the bin function is monotone: start from its algebraic inverse and step to the exact
    // boundary (0-2 steps); the bisection below is the answer to any geometry that defeats that
    {
      const TK x = lo + (TK)p / scale;
      int m = x >= (TK)65536 ? 65536 : (x > (TK)0 ? (int)x : 0);  // NaN -> 0
      int steps = 0;
      while (m > 0 && steps < 6 && stream_bin((TK)(m - 1), lo, scale, B) >= p) {
        --m;
        ++steps;
      }
      while (m < 65536 && steps < 6 && stream_bin((TK)m, lo, scale, B) < p) {
        ++m;
        ++steps;
      }
      if (steps < 6) return m;
    }
    int a = 0, b = 65536;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (stream_bin((TK)m, lo, scale, B) >= p) b = m;
      else a = m + 1;
    }
    return a;
  };
  const int f0 = first_ge(pb);
  const unsigned long long tlo = (unsigned long long)(f0 > 65535 ? 65535 : f0);
  const unsigned long long thi = pb >= B - 1 ? 65535ULL : (unsigned long long)(first_ge(pb + 1) - 1);
  const unsigned long long elo = lowb < 0 ? 0xffffULL : (unsigned long long)first_ge(lowb);
  const unsigned long long ehi =
      highb >= B ? 0ULL : (highb >= B - 1 ? 65535ULL : (unsigned long long)(first_ge(highb + 1) - 1));
  return tlo | (thi << 16) | ((elo > 65535 ? 65535ULL : elo) << 32) | (ehi << 48);
}

// value bins per node of a streaming level: the 32768 LDS counters are split over the M nodes.
// Nodes of up to ~2M points keep at most 4096 bins (one stream_pick block scans them); larger
// nodes take every bin they can get, or a pivot bin would outgrow the LDS sort (kSmallCap)
// and force the whole build onto the general path.
inline int stream_bins(int M, int64_t node_points, int64_t big_node) {
  const int b = kStreamBins / M;
  return (b > 4096 && node_points <= big_node) ? 4096 : b;
}

// min / max of one level's keys over the whole tree (root geometry). grid = (nblk, T)
// In CODE mode (Cd != null, codes.h) the streaming kernels histogram and classify the 16-bit
// codes of the keys, converted to TK: every formula below (geometry, bins, min/max) works on
// those integer-valued keys unchanged, and exact keys are fetched only for the points that need
// them (pivot bins, margins).
template <class TK>
__global__ __launch_bounds__(kStreamThreads) void stream_minmax0(const TK* __restrict__ P,
                                                                 const uint16_t* __restrict__ Cd,
                                                                 int64_t N, int L, int64_t per,
                                                                 unsigned long long* cmin,
                                                                 unsigned long long* cmax) {
  const int t = blockIdx.y;
  const TK* Pl = P + (int64_t)t * L * N;
  const uint16_t* Cl = Cd ? Cd + (int64_t)t * L * N : nullptr;
  const int64_t i0 = (int64_t)blockIdx.x * per;
  int64_t i1 = i0 + per < N ? i0 + per : N;
  // the range only shapes the root's bins (keys outside clamp to the edge bins): an eighth of
  // every block's points is plenty when there are many
  if (N >= 65536) i1 = i0 + ((i1 - i0 + 7) >> 3);
  unsigned long long mn = ~0ULL, mx = 0ULL;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += kStreamThreads) {
    const unsigned long long o = ord_of(Cl ? (TK)Cl[i] : Pl[i]);
    mn = o < mn ? o : mn;
    mx = o > mx ? o : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  __shared__ unsigned long long smn[kStreamThreads / 64], smx[kStreamThreads / 64];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic pair per block: all waves hit the same two words
    for (int w = 1; w < kStreamThreads / 64; ++w) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
    }
    if (mn != ~0ULL) {
      atomicMin(&cmin[t], mn);
      atomicMax(&cmax[t], mx);
    }
  }
}

template <class TK>
struct AGeom {  // per node, one 16-byte LDS read
  TK lo, sc;
};
struct ABins {  // per node, one 16-byte LDS read
  int pb, lowb, highb, pad;
};

// bin geometry of one node from the min / max of its keys (hist and pick must agree bit for bit)
template <class TK>
__device__ inline void stream_geom(unsigned long long mn, unsigned long long mx, int B, TK& lo,
                                   TK& scale) {
  lo = (TK)0;
  scale = (TK)0;
  if (mn != ~0ULL) {
    const TK l = ord_to(mn, TK()), h = ord_to(mx, TK());
    lo = l;
    scale = l < h ? (TK)B / (h - l) : (TK)0;
  }
}

// Uniform bins over the node's [min, max].  (Measured: folding 2B cells into B bins with a finer
// centre halves the pivot bins of bell-shaped keys but the two-cluster C2 data puts medians into
// the coarse tails — 0.40 -> 0.74 ms of stream_mid; any monotone map would keep the split exact.)
template <class TK>
__device__ inline int stream_bin(TK key, TK lo, TK scale, int B) {
  int b = (int)((key - lo) * scale);
  return b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
}

// per-block value histogram of one level (16-bit counters packed two per word), written as a
// PARTIAL to part[t][block][kStreamBins/2] with plain coalesced stores: stream_pick adds the
// partials up (an atomic flush costs one L2 atomic per bin per block — as many as there are
// points when a rank holds few trees).  grid = (nblk, T)
template <class TK>
__global__ __launch_bounds__(kStreamThreads) void stream_hist(
    const TK* __restrict__ P, const uint16_t* __restrict__ Cd, const uint16_t* __restrict__ node_of,
    int64_t N, int L, int level, int M, int B, int64_t per,
    const unsigned long long* __restrict__ cmin, const unsigned long long* __restrict__ cmax,
    unsigned int* __restrict__ part) {
  __shared__ unsigned int hist[kStreamBins / 2];  // two 16-bit counters per word
  __shared__ __attribute__((aligned(16))) AGeom<TK> ngeo[kStreamMaxNodes];
  const int t = blockIdx.y;
  for (int i = threadIdx.x; i < (M * B) / 2; i += kStreamThreads) hist[i] = 0;
  for (int j = threadIdx.x; j < M; j += kStreamThreads) {
    AGeom<TK> g;
    stream_geom<TK>(cmin[(int64_t)t * M + j], cmax[(int64_t)t * M + j], B, g.lo, g.sc);
    ngeo[j] = g;
  }
  __syncthreads();
  const TK* Pl = P + ((int64_t)t * L + level) * N;
  const uint16_t* no = node_of + (int64_t)t * N;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
  auto one = [&](int j, TK key) {
    const AGeom<TK> g = ngeo[j];
    const int e = j * B + stream_bin(key, g.lo, g.sc, B);
    atomicAdd(&hist[e >> 1], 1u << ((e & 1) * 16));  // per < 65536: no carry between halves
  };
  typedef TK key2_t __attribute__((ext_vector_type(2)));
  if (Cd) {  // code mode: 4 bytes per point; eight points per thread and step (16-byte loads)
    const uint16_t* Cl = Cd + ((int64_t)t * L + level) * N;
    auto onec = [&](unsigned int j, unsigned int code) { one((int)j, (TK)code); };
    if (((N | i0 | per) & 7) == 0) {
      const int64_t ie = i0 + ((i1 - i0) & ~(int64_t)7);
      for (int64_t i = i0 + 8 * (int64_t)threadIdx.x; i < ie; i += 8 * kStreamThreads) {
        const uint4 jj = *reinterpret_cast<const uint4*>(no + i);
        const uint4 cc = *reinterpret_cast<const uint4*>(Cl + i);
        onec(jj.x & 0xffffu, cc.x & 0xffffu);
        onec(jj.x >> 16, cc.x >> 16);
        onec(jj.y & 0xffffu, cc.y & 0xffffu);
        onec(jj.y >> 16, cc.y >> 16);
        onec(jj.z & 0xffffu, cc.z & 0xffffu);
        onec(jj.z >> 16, cc.z >> 16);
        onec(jj.w & 0xffffu, cc.w & 0xffffu);
        onec(jj.w >> 16, cc.w >> 16);
      }
      for (int64_t i = ie + threadIdx.x; i < i1; i += kStreamThreads) onec(no[i], Cl[i]);
    } else {
      for (int64_t i = i0 + threadIdx.x; i < i1; i += kStreamThreads) onec(no[i], Cl[i]);
    }
  } else if (((N | i0 | per) & 1) == 0 && sizeof(TK) == 8) {  // 16-byte key loads, 4-byte node loads
    const int64_t ie = i0 + ((i1 - i0) & ~(int64_t)1);
    for (int64_t i = i0 + 2 * (int64_t)threadIdx.x; i < ie; i += 2 * kStreamThreads) {
      const unsigned int jj = *reinterpret_cast<const unsigned int*>(no + i);
      const key2_t kk = *reinterpret_cast<const key2_t*>(Pl + i);
      one((int)(jj & 0xffffu), kk[0]);
      one((int)(jj >> 16), kk[1]);
    }
    if (threadIdx.x == 0 && ie < i1) one(no[ie], Pl[ie]);
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kStreamThreads) one(no[i], Pl[i]);
  }
  __syncthreads();
  unsigned int* gp = part + ((int64_t)t * gridDim.x + blockIdx.x) * (kStreamBins / 2);
  for (int i = threadIdx.x; i < (M * B) / 2; i += kStreamThreads) gp[i] = hist[i];
}

// BPT consecutive 16-bit counters added into c[]
template <int BPT>
__device__ inline void add_u16(const uint16_t* p, unsigned int (&c)[BPT]) {
  if constexpr (BPT == 1) {
    c[0] += *p;
  } else if constexpr (BPT == 2) {
    const unsigned int w = *reinterpret_cast<const unsigned int*>(p);
    c[0] += w & 0xffffu;
    c[1] += w >> 16;
  } else if constexpr (BPT == 4) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    c[0] += w.x & 0xffffu;
    c[1] += w.x >> 16;
    c[2] += w.y & 0xffffu;
    c[3] += w.y >> 16;
  } else {
#pragma unroll
    for (int q = 0; q < BPT / 8; ++q) {
      const uint4 w = *reinterpret_cast<const uint4*>(p + 8 * q);
      c[8 * q + 0] += w.x & 0xffffu;
      c[8 * q + 1] += w.x >> 16;
      c[8 * q + 2] += w.y & 0xffffu;
      c[8 * q + 3] += w.y >> 16;
      c[8 * q + 4] += w.z & 0xffffu;
      c[8 * q + 5] += w.z >> 16;
      c[8 * q + 6] += w.w & 0xffffu;
      c[8 * q + 7] += w.w >> 16;
    }
  }
}

// per (tree, node): sum of the histogram partials, pivot bin, counts, nearest non-empty bins;
// also (re)initialises the node record and the children's min/max cells of the next level.
// G threads per node (8 .. 256), BPT consecutive bins per thread (wide loads of the 16-bit
// partial counters), B = G * BPT bins per node.  grid = (ceil(M / (256/G)), T), 256 threads
template <class TK, int BPT, int G>
__global__ __launch_bounds__(256) void stream_pick(int64_t N, int level, int M, int nblk,
                                                   const unsigned int* __restrict__ part,
                                                   const unsigned long long* __restrict__ cmin,
                                                   const unsigned long long* __restrict__ cmax,
                                                   SNode<TK>* __restrict__ nd,
                                                   unsigned int* __restrict__ poolcur,
                                                   unsigned long long* __restrict__ cmin_next,
                                                   unsigned long long* __restrict__ cmax_next,
                                                   unsigned int* __restrict__ bigmid,
                                                   int code_mode) {
  constexpr int NPB = 256 / G, B = G * BPT, W = G < 64 ? G : 64, WPG = G / W;  // waves per group
  __shared__ unsigned int wtot[4];
  __shared__ int s_pb[NPB], s_cL[NPB], s_cMid[NPB], s_low[NPB], s_high[NPB];
  const int g = threadIdx.x / G, r = threadIdx.x % G, lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int j = blockIdx.x * NPB + g, t = blockIdx.y;
  const bool live = j < M;
  unsigned int c[BPT];
#pragma unroll
  for (int q = 0; q < BPT; ++q) c[q] = 0;
  if (live) {
    const uint16_t* p16 =
        reinterpret_cast<const uint16_t*>(part + (int64_t)t * nblk * (kStreamBins / 2)) +
        (int64_t)j * B + r * BPT;
#pragma unroll 4
    for (int p = 0; p < nblk; ++p) add_u16<BPT>(p16 + (int64_t)p * kStreamBins, c);
  }
  int64_t n = N;
  for (int b = level - 1; b >= 0; --b) {
    const int64_t h = n >> 1;
    n = ((j >> b) & 1) ? n - h : h;
  }
  const unsigned int nh = (unsigned int)(n >> 1);
  unsigned int tot = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) tot += c[q];
  unsigned int inc = tot;  // inclusive scan inside the group's lanes of this wave
#pragma unroll
  for (int o = 1; o < W; o <<= 1) {
    const unsigned int v = __shfl_up(inc, o, W);
    if ((lane & (W - 1)) >= o) inc += v;
  }
  if (threadIdx.x < NPB) {
    s_pb[threadIdx.x] = -1;
    s_cL[threadIdx.x] = 0;
    s_cMid[threadIdx.x] = 0;
    s_low[threadIdx.x] = -1;
    s_high[threadIdx.x] = B;
  }
  if (WPG > 1 && lane == 63) wtot[wv] = inc;
  __syncthreads();
  unsigned int run = inc - tot;
  if (WPG > 1)
    for (int w = g * WPG; w < wv; ++w) run += wtot[w];
  if (live && run <= nh && nh < run + tot) {  // exactly one thread of the group
    unsigned int a = run;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      if (a <= nh && nh < a + c[q]) {
        s_pb[g] = r * BPT + q;
        s_cL[g] = (int)a;
        s_cMid[g] = (int)c[q];
      }
      a += c[q];
    }
  }
  __syncthreads();
  const int pb = s_pb[g], cL = s_cL[g], cMid = s_cMid[g];
  // the margins p'[nh-1] / p'[nh+1] fall outside the pivot bin only at its edges: only then
  // the assign pass has to track max(left) / min(right) (in the nearest non-empty bins)
  const int il = nh > 0 ? (int)nh - 1 : 0, ih = (int)nh + 1 < (int)n ? (int)nh + 1 : (int)n - 1;
  const bool need_lo = il < cL, need_hi = ih >= cL + cMid;
  if (live && (need_lo || need_hi)) {
    int lowb = -1, highb = B;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int b = r * BPT + q;
      if (c[q]) {
        if (b < pb) lowb = b;                 // ascending: the last one wins
        if (b > pb && highb == B) highb = b;  // the first one wins
      }
    }
    if (lowb >= 0) atomicMax(&s_low[g], lowb);
    if (highb < B) atomicMin(&s_high[g], highb);
  }
  __syncthreads();
  if (live && r == 0) {
    SNode<TK> a;
    stream_geom<TK>(cmin[(int64_t)t * M + j], cmax[(int64_t)t * M + j], B, a.lo, a.scale);
    a.n = (int)n;
    a.nh = (int)nh;
    a.pb = pb;
    a.cL = cL;
    a.cMid = cMid;
    a.lowb = need_lo ? s_low[g] : -2;
    a.highb = need_hi ? s_high[g] : B + 1;
    a.cthr = code_mode ? code_thr<TK>(a.lo, a.scale, B, pb, need_lo ? s_low[g] : -1, need_hi ? s_high[g] : B)
                       : 0ULL;
    a.midoff = (int)atomicAdd(&poolcur[t], (unsigned int)cMid);
    a.midcur = 0;
    a.maxL = 0ULL;
    a.minR = ~0ULL;
    nd[(int64_t)t * M + j] = a;
    cmin_next[(int64_t)t * 2 * M + 2 * j] = ~0ULL;  // children of this node
    cmin_next[(int64_t)t * 2 * M + 2 * j + 1] = ~0ULL;
    cmax_next[(int64_t)t * 2 * M + 2 * j] = 0ULL;
    cmax_next[(int64_t)t * 2 * M + 2 * j + 1] = 0ULL;
    if (cMid > kSmallCap) atomicAdd(bigmid, 1u);
  }
}

// ---- stream_pick for nodes with more than 4096 bins (the first levels of very large point
// sets): the partials are summed into totals[t][M * B] first, then one block per (tree, node)
// scans them, every thread owning B / 256 consecutive bins.
__global__ __launch_bounds__(256) void stream_pick_sum(int MB2 /* M * B / 2 */, int nblk,
                                                       const unsigned int* __restrict__ part,
                                                       unsigned int* __restrict__ totals) {
  const int i = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  if (i >= MB2) return;
  const unsigned int* p = part + (int64_t)t * nblk * (kStreamBins / 2) + i;
  unsigned int lo = 0, hi = 0;
#pragma unroll 4
  for (int b = 0; b < nblk; ++b) {
    const unsigned int w = p[(int64_t)b * (kStreamBins / 2)];
    lo += w & 0xffffu;
    hi += w >> 16;
  }
  totals[(int64_t)t * kStreamBins + 2 * i] = lo;
  totals[(int64_t)t * kStreamBins + 2 * i + 1] = hi;
}

template <class TK>
__global__ __launch_bounds__(256) void stream_pick_big(int64_t N, int level, int M, int B,
                                                       const unsigned int* __restrict__ totals,
                                                       const unsigned long long* __restrict__ cmin,
                                                       const unsigned long long* __restrict__ cmax,
                                                       SNode<TK>* __restrict__ nd,
                                                       unsigned int* __restrict__ poolcur,
                                                       unsigned long long* __restrict__ cmin_next,
                                                       unsigned long long* __restrict__ cmax_next,
                                                       unsigned int* __restrict__ bigmid,
                                                       int code_mode) {
  __shared__ unsigned int wtot[4];
  __shared__ int s_owner, s_pb, s_cL, s_cMid, s_low, s_high;
  const int j = blockIdx.x, t = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int bpt = B / 256;  // 32 .. 128 consecutive bins per thread
  const unsigned int* h = totals + (int64_t)t * kStreamBins + (int64_t)j * B;
  int64_t n = N;
  for (int b = level - 1; b >= 0; --b) {
    const int64_t hh = n >> 1;
    n = ((j >> b) & 1) ? n - hh : hh;
  }
  const unsigned int nh = (unsigned int)(n >> 1);
  unsigned int tot = 0;
  int firstne = B, lastne = -1;  // non-empty bins of this thread's range
  for (int q = 0; q < bpt; ++q) {
    const unsigned int c = h[threadIdx.x * bpt + q];
    tot += c;
    if (c) {
      if (firstne == B) firstne = threadIdx.x * bpt + q;
      lastne = threadIdx.x * bpt + q;
    }
  }
  unsigned int inc = tot;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (threadIdx.x == 0) {
    s_owner = -1;
    s_pb = -1;
    s_cL = 0;
    s_cMid = 0;
    s_low = -1;
    s_high = B;
  }
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  unsigned int run = inc - tot;
  for (int w = 0; w < wv; ++w) run += wtot[w];
  if (run <= nh && nh < run + tot) {  // exactly one thread
    unsigned int a = run;
    for (int q = 0; q < bpt; ++q) {
      const unsigned int c = h[threadIdx.x * bpt + q];
      if (a <= nh && nh < a + c) {
        s_pb = threadIdx.x * bpt + q;
        s_cL = (int)a;
        s_cMid = (int)c;
      }
      a += c;
    }
    s_owner = (int)threadIdx.x;
  }
  __syncthreads();
  const int pb = s_pb, cL = s_cL, cMid = s_cMid, owner = s_owner;
  const int il = nh > 0 ? (int)nh - 1 : 0, ih = (int)nh + 1 < (int)n ? (int)nh + 1 : (int)n - 1;
  const bool need_lo = il < cL, need_hi = ih >= cL + cMid;
  if (owner >= 0 && (need_lo || need_hi)) {
    if ((int)threadIdx.x < owner && lastne >= 0) atomicMax(&s_low, lastne);
    if ((int)threadIdx.x > owner && firstne < B) atomicMin(&s_high, firstne);
    if ((int)threadIdx.x == owner) {
      int lowb = -1, highb = B;
      for (int q = 0; q < bpt; ++q) {
        const int b = threadIdx.x * bpt + q;
        if (h[b]) {
          if (b < pb) lowb = b;
          if (b > pb && highb == B) highb = b;
        }
      }
      if (lowb >= 0) atomicMax(&s_low, lowb);
      if (highb < B) atomicMin(&s_high, highb);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    SNode<TK> a;
    stream_geom<TK>(cmin[(int64_t)t * M + j], cmax[(int64_t)t * M + j], B, a.lo, a.scale);
    a.n = (int)n;
    a.nh = (int)nh;
    a.pb = pb;
    a.cL = cL;
    a.cMid = cMid;
    a.lowb = need_lo ? s_low : -2;
    a.highb = need_hi ? s_high : B + 1;
    a.cthr = code_mode ? code_thr<TK>(a.lo, a.scale, B, pb, need_lo ? s_low : -1, need_hi ? s_high : B)
                       : 0ULL;
    a.midoff = (int)atomicAdd(&poolcur[t], (unsigned int)cMid);
    a.midcur = 0;
    a.maxL = 0ULL;
    a.minR = ~0ULL;
    nd[(int64_t)t * M + j] = a;
    cmin_next[(int64_t)t * 2 * M + 2 * j] = ~0ULL;
    cmin_next[(int64_t)t * 2 * M + 2 * j + 1] = ~0ULL;
    cmax_next[(int64_t)t * 2 * M + 2 * j] = 0ULL;
    cmax_next[(int64_t)t * 2 * M + 2 * j + 1] = 0ULL;
    if (cMid > kSmallCap) atomicAdd(bigmid, 1u);
  }
}

// CODE = the code-mode instantiation (Cd != null): its tables are 8-byte thresholds and 32-bit range cells,
// 68 KB of LDS instead of 100 — TWO blocks per CU, twice the waves to hide the pass's load latency behind
template <class TK, bool CODE>
__global__ __launch_bounds__(kStreamThreads) void stream_assign(
    const TK* __restrict__ P, const uint16_t* __restrict__ Cd, uint16_t* __restrict__ node_of,
    int64_t N, int L, int level, int M, int B, int64_t per, int has_next, SNode<TK>* nd,
    int32_t* __restrict__ pool, TK* __restrict__ poolkey, unsigned long long* cmin_next,
    unsigned long long* cmax_next) {
  // code mode: ngeo holds kStreamMaxNodes 8-byte words (cth), nbin nothing, smin / smax 2 M 32-bit cells in
  // their first kStreamMaxNodes words and the pivot lists' per-node counters behind them
  __shared__ __attribute__((aligned(16))) AGeom<TK> ngeo[CODE ? kStreamMaxNodes * 8 / sizeof(AGeom<TK>) : kStreamMaxNodes];
  __shared__ __attribute__((aligned(16))) ABins nbin[CODE ? 1 : kStreamMaxNodes];
  __shared__ int nmidoff[kStreamMaxNodes];
  __shared__ unsigned long long smin[CODE ? kStreamMaxNodes + kStreamMaxNodes / 2 : 2 * kStreamMaxNodes],
      smax[CODE ? kStreamMaxNodes + kStreamMaxNodes / 2 : 2 * kStreamMaxNodes];
  const int t = blockIdx.y;
  SNode<TK>* ndt = nd + (int64_t)t * M;
  unsigned long long* cth = reinterpret_cast<unsigned long long*>(ngeo);  // code mode: code_thr
  for (int j = threadIdx.x; j < M; j += kStreamThreads) {
    if (Cd) {
      cth[j] = ndt[j].cthr;
    } else {
      ngeo[j] = AGeom<TK>{ndt[j].lo, ndt[j].scale};
      nbin[j] = ABins{ndt[j].pb, ndt[j].lowb, ndt[j].highb, 0};
    }
    nmidoff[j] = ndt[j].midoff;
  }
  for (int c = threadIdx.x; c < (CODE ? M : 2 * M); c += kStreamThreads) {  // (code mode: 2 M 32-bit cells)
    smin[c] = ~0ULL;
    smax[c] = 0ULL;
  }
  __syncthreads();
  const TK* Pl = P + ((int64_t)t * L + level) * N;
  const TK* Pn = has_next ? Pl + N : Pl;
  uint16_t* no = node_of + (int64_t)t * N;
  int32_t* pl = pool + (int64_t)t * N;
  TK* pk = poolkey + (int64_t)t * N;  // the pivot-bin lists carry the key: stream_mid reads it
                                      // coalesced instead of gathering it by id
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
  // The children's min/max of the NEXT level's key only shape that level's bins (keys outside
  // the range are clamped into the edge bins), so a sample is enough: the first quarter of the
  // block's points — every point while nodes are small.
  // (measured: the sampling — LDS min/max atomics — is the expensive part of this kernel; an
  // eighth of the points, and 32-bit atomics on the codes themselves in code mode)
  const int64_t isamp = (N >> level) < 1024 ? i1 : (i0 + ((i1 - i0 + 7) >> 3) + 7) & ~(int64_t)7;
  // few nodes: a per-thread running min/max per child avoids hammering one LDS word
  const bool few = M <= 4;
  unsigned long long tmn[8], tmx[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    tmn[c] = ~0ULL;
    tmx[c] = 0ULL;
  }
  // 8-byte-key mode with at most 512 nodes: the upper halves of nbin / smin / smax are unused and
  // hold the block's pivot-bin list (one global atomic per node and block instead of one per point)
  // (the list itself: 4096 entries of its own — a block of the one-block-per-CU grid sees ~2000
  // pivot-bin points per level at C2)
  constexpr int kKeyList = 4096;
  __shared__ int32_t list_id[kKeyList];
  __shared__ unsigned int list_meta[kKeyList];
  const bool klist = !CODE && M <= kStreamMaxNodes / 2;
  int32_t* kid = list_id;
  unsigned int* kmeta = list_meta;
  unsigned int* kcnt = reinterpret_cast<unsigned int*>(nbin + kStreamMaxNodes / 2);  // [512]
  unsigned int* kbase = kcnt + kStreamMaxNodes / 2;                                  // [512]
  __shared__ unsigned int kfill[1];
  if (klist) {
    for (int j = threadIdx.x; j < M; j += kStreamThreads) kcnt[j] = 0u;
    if (threadIdx.x == 0) kfill[0] = 0u;
    __syncthreads();
  }
  // classify one point: child node index, or -1 when it went to the node's pivot-bin list.
  // kb = the value that is binned: the key itself, or (code mode) the key's 16-bit code; the
  // exact key is then read only where it decides something — pivot bin, margin bins.
  auto classify = [&](int64_t i, int j, TK kb) -> int {
    const AGeom<TK> g = ngeo[j];
    const ABins nb = nbin[j];
    const int b = stream_bin(kb, g.lo, g.sc, B);
    const int pb = nb.pb;
    if (b == pb) {
      if (klist) {  // parked in the LDS list, copied out after the pass (see the code-mode branch)
        const unsigned int e = atomicAdd(&kfill[0], 1u);
        if (e < (unsigned int)kKeyList) {
          const unsigned int loc = atomicAdd(&kcnt[j], 1u);
          kid[e] = (int32_t)i;
          kmeta[e] = (loc << 16) | (unsigned int)j;
          return -1;
        }
      }
      const unsigned int p = atomicAdd(&ndt[j].midcur, 1u);
      pl[nmidoff[j] + p] = (int32_t)i;
      pk[nmidoff[j] + p] = Cd ? Pl[i] : kb;
      return -1;
    }
    if (b == nb.lowb) atomicMax(&ndt[j].maxL, ord_of(Cd ? Pl[i] : kb));
    if (b == nb.highb) atomicMin(&ndt[j].minR, ord_of(Cd ? Pl[i] : kb));
    return 2 * j + (b > pb);
  };
  auto sample = [&](int child, TK knext) {
    const unsigned long long o = ord_of(knext);
    if (few) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c == child) {
          tmn[c] = o < tmn[c] ? o : tmn[c];
          tmx[c] = o > tmx[c] ? o : tmx[c];
        }
    } else {
      atomicMin(&smin[child], o);
      atomicMax(&smax[child], o);
    }
  };
  // two consecutive points per thread and step: 16-byte key loads, 4-byte node loads/stores
  // (the key rows are 16-byte aligned when N is even and the block ranges start on even offsets)
  typedef TK key2_t __attribute__((ext_vector_type(2)));
  if constexpr (CODE) {  // code mode: eight points per thread and step, 16-byte code and node loads / stores
    const uint16_t* Cl = Cd + ((int64_t)t * L + level) * N;
    const uint16_t* Cn = has_next ? Cl + N : Cl;
    // the pivot list's per-node counters live in LDS that code mode leaves unused: the upper halves
    // of smin / smax (the sampling cells are 32-bit here)
    constexpr int kAssignList = kKeyList;
    int32_t* lid = list_id;              // [kAssignList]
    unsigned int* lmeta = list_meta;
    unsigned int* lcnt = reinterpret_cast<unsigned int*>(smin + kStreamMaxNodes);   // [kStreamMaxNodes]
    unsigned int* lbase = reinterpret_cast<unsigned int*>(smax + kStreamMaxNodes);  // [kStreamMaxNodes]
    __shared__ unsigned int lfill[1];
    for (int j = threadIdx.x; j < M; j += kStreamThreads) lcnt[j] = 0u;
    if (threadIdx.x == 0) lfill[0] = 0u;
    __syncthreads();
    // one 8-byte LDS word per point (code_thr) and integer compares; the exact key is read only
    // for the pivot bin and the two margin bins
    auto classc = [&](int64_t i, unsigned int j, unsigned int code) -> int {
      const unsigned long long th = cth[j];
      const unsigned int tlo = (unsigned int)th & 0xffffu, thi = (unsigned int)(th >> 16) & 0xffffu;
      if (code < tlo) {
        if (code >= ((unsigned int)(th >> 32) & 0xffffu)) atomicMax(&ndt[j].maxL, ord_of(Pl[i]));
        return (int)(2 * j);
      }
      if (code > thi) {
        if (code <= (unsigned int)(th >> 48)) atomicMin(&ndt[j].minR, ord_of(Pl[i]));
        return (int)(2 * j + 1);
      }
      // pivot-bin point: parked in an LDS list (slot inside the block's share of its node's list
      // from an LDS counter); the block reserves its share with ONE global atomic per node after
      // the pass and copies the list out then.  A returning global atomic per pivot point in the
      // middle of the streaming pass cost a third of this kernel (ablation: 80 -> 54 us per level).
      const unsigned int e = atomicAdd(&lfill[0], 1u);
      if (e < (unsigned int)kAssignList) {
        const unsigned int loc = atomicAdd(&lcnt[j], 1u);
        lid[e] = (int32_t)i;
        lmeta[e] = (loc << 16) | j;   // loc < kAssignList <= 65536, j < kStreamMaxNodes
        return -1;
      }
      const unsigned int p = atomicAdd(&ndt[j].midcur, 1u);  // list full: the direct way
      pl[nmidoff[j] + p] = (int32_t)i;
      pk[nmidoff[j] + p] = Pl[i];
      return -1;
    };
    unsigned int* smin32 = reinterpret_cast<unsigned int*>(smin);  // [2M], all ones / zero like
    unsigned int* smax32 = reinterpret_cast<unsigned int*>(smax);  // the 64-bit cells they overlay
    unsigned int cmn[8], cmx[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      cmn[c] = 0xffffffffu;
      cmx[c] = 0u;
    }
    auto samplec = [&](int child, unsigned int code) {
      if (few) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c == child) {
            cmn[c] = code < cmn[c] ? code : cmn[c];
            cmx[c] = code > cmx[c] ? code : cmx[c];
          }
      } else {
        atomicMin(&smin32[child], code);
        atomicMax(&smax32[child], code);
      }
    };
    auto one = [&](int64_t i) {
      const int c = classc(i, no[i], Cl[i]);
      if (c >= 0) {
        no[i] = (uint16_t)c;
        if (has_next && i < isamp) samplec(c, Cn[i]);
      }
    };
    if (((N | i0 | per) & 7) == 0) {
      const int64_t ie = i0 + ((i1 - i0) & ~(int64_t)7);
      for (int64_t i = i0 + 8 * (int64_t)threadIdx.x; i < ie; i += 8 * kStreamThreads) {
        const uint4 jj = *reinterpret_cast<const uint4*>(no + i);
        const uint4 cc = *reinterpret_cast<const uint4*>(Cl + i);
        const unsigned int jw[4] = {jj.x, jj.y, jj.z, jj.w}, cw[4] = {cc.x, cc.y, cc.z, cc.w};
        int ch[8];
        unsigned int ow[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned int ja = jw[w] & 0xffffu, jb = jw[w] >> 16;
          ch[2 * w] = classc(i + 2 * w, ja, cw[w] & 0xffffu);
          ch[2 * w + 1] = classc(i + 2 * w + 1, jb, cw[w] >> 16);
          // a pivot-bin point keeps its node until stream_mid has ordered the bin
          ow[w] = (ch[2 * w] >= 0 ? (unsigned int)ch[2 * w] : ja) |
                  ((ch[2 * w + 1] >= 0 ? (unsigned int)ch[2 * w + 1] : jb) << 16);
        }
        *reinterpret_cast<uint4*>(no + i) = uint4{ow[0], ow[1], ow[2], ow[3]};
        if (has_next && i < isamp) {
          const uint4 nn = *reinterpret_cast<const uint4*>(Cn + i);
          const unsigned int nw[4] = {nn.x, nn.y, nn.z, nn.w};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (ch[2 * w] >= 0) samplec(ch[2 * w], nw[w] & 0xffffu);
            if (ch[2 * w + 1] >= 0) samplec(ch[2 * w + 1], nw[w] >> 16);
          }
        }
      }
      for (int64_t i = ie + threadIdx.x; i < i1; i += kStreamThreads) one(i);
    } else {
      for (int64_t i = i0 + threadIdx.x; i < i1; i += kStreamThreads) one(i);
    }
    // the block's pivot-bin points -> the nodes' global lists
    __syncthreads();
    for (int j = threadIdx.x; j < M; j += kStreamThreads)
      if (lcnt[j]) lbase[j] = atomicAdd(&ndt[j].midcur, lcnt[j]);
    __syncthreads();
    {
      const unsigned int ne = lfill[0] < (unsigned int)kAssignList ? lfill[0] : (unsigned int)kAssignList;
      for (unsigned int e = threadIdx.x; e < ne; e += kStreamThreads) {
        const unsigned int j = lmeta[e] & 0xffffu, loc = lmeta[e] >> 16;
        const int32_t i = lid[e];
        const unsigned int p = lbase[j] + loc;
        pl[nmidoff[j] + p] = i;
        pk[nmidoff[j] + p] = Pl[i];
      }
    }
    // flush: children this block sampled -> the next level's code range (as keys: ord_of((TK)code))
    if (has_next && few) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c >= 2 * M) break;
        unsigned int mn = cmn[c], mx = cmx[c];
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned int a = __shfl_xor(mn, o), b2 = __shfl_xor(mx, o);
          mn = a < mn ? a : mn;
          mx = b2 > mx ? b2 : mx;
        }
        if ((threadIdx.x & 63) == 0 && mn != 0xffffffffu) {
          atomicMin(&smin32[c], mn);
          atomicMax(&smax32[c], mx);
        }
      }
    }
    __syncthreads();
    if (has_next)
      for (int c = threadIdx.x; c < 2 * M; c += kStreamThreads)
        if (smin32[c] != 0xffffffffu) {
          atomicMin(&cmin_next[(int64_t)t * 2 * M + c], ord_of((TK)smin32[c]));
          atomicMax(&cmax_next[(int64_t)t * 2 * M + c], ord_of((TK)smax32[c]));
        }
    return;
  } else if (((N | i0 | per) & 1) == 0 && sizeof(TK) == 8) {
    const int64_t ie = i0 + ((i1 - i0) & ~(int64_t)1);
    for (int64_t i = i0 + 2 * (int64_t)threadIdx.x; i < ie; i += 2 * kStreamThreads) {
      const unsigned int jj = *reinterpret_cast<const unsigned int*>(no + i);
      const key2_t kk = *reinterpret_cast<const key2_t*>(Pl + i);
      const int c0 = classify(i, (int)(jj & 0xffffu), kk[0]);
      const int c1 = classify(i + 1, (int)(jj >> 16), kk[1]);
      if (c0 >= 0 && c1 >= 0) {
        *reinterpret_cast<unsigned int*>(no + i) = (unsigned int)c0 | ((unsigned int)c1 << 16);
      } else {
        if (c0 >= 0) no[i] = (uint16_t)c0;
        if (c1 >= 0) no[i + 1] = (uint16_t)c1;
      }
      if (has_next && i < isamp) {
        const key2_t kn = *reinterpret_cast<const key2_t*>(Pn + i);
        if (c0 >= 0) sample(c0, kn[0]);
        if (c1 >= 0) sample(c1, kn[1]);
      }
    }
    if (threadIdx.x == 0 && ie < i1) {
      const int c = classify(ie, no[ie], Pl[ie]);
      if (c >= 0) {
        no[ie] = (uint16_t)c;
        if (has_next) sample(c, Pn[ie]);
      }
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kStreamThreads) {
      const int c = classify(i, no[i], Pl[i]);
      if (c >= 0) {
        no[i] = (uint16_t)c;
        if (has_next && i < isamp) sample(c, Pn[i]);
      }
    }
  }
  if (klist) {  // the block's pivot-bin points -> the nodes' global lists
    __syncthreads();
    for (int j = threadIdx.x; j < M; j += kStreamThreads)
      if (kcnt[j]) kbase[j] = atomicAdd(&ndt[j].midcur, kcnt[j]);
    __syncthreads();
    const unsigned int ne = kfill[0] < (unsigned int)kKeyList ? kfill[0] : (unsigned int)kKeyList;
    for (unsigned int e = threadIdx.x; e < ne; e += kStreamThreads) {
      const unsigned int j = kmeta[e] & 0xffffu, loc = kmeta[e] >> 16;
      const int32_t i = kid[e];
      const unsigned int p = kbase[j] + loc;
      pl[nmidoff[j] + p] = i;
      pk[nmidoff[j] + p] = Pl[i];
    }
  }
  if (has_next && few) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c >= 2 * M) break;
      unsigned long long mn = tmn[c], mx = tmx[c];
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(mn, o), b2 = __shfl_xor(mx, o);
        mn = a < mn ? a : mn;
        mx = b2 > mx ? b2 : mx;
      }
      if ((threadIdx.x & 63) == 0) {
        if (mn != ~0ULL) atomicMin(&smin[c], mn);
        if (mx != 0ULL) atomicMax(&smax[c], mx);
      }
    }
  }
  __syncthreads();
  if (has_next)
    for (int c = threadIdx.x; c < 2 * M; c += kStreamThreads) {
      if (smin[c] != ~0ULL) atomicMin(&cmin_next[(int64_t)t * 2 * M + c], smin[c]);
      if (smax[c] != 0ULL) atomicMax(&cmax_next[(int64_t)t * 2 * M + c], smax[c]);
    }
}

// exact order of the pivot bins of one level: children of the pivot-bin points, min/max of the
// next level's key, thr / margins.  Pivot bins of <= 128 points take a wave each (bitonic
// network over lane shuffles, no LDS); larger ones (the first levels, or ties) are sorted by
// the whole block in LDS afterwards.  grid = (ceil(M/npb), T), 256 threads, npb = 1 or 4.
template <class TK>
struct MidArgs {
  const TK* P;
  const uint16_t* Cd;  // code mode: the children's range of the next level is a range of CODES
  uint16_t* node_of;
  int64_t N;
  int L, level, M, has_next;
  const SNode<TK>* nd;
  const int32_t* pool;
  const TK* poolkey;
  unsigned long long *cmin_next, *cmax_next;
  int64_t heap0, nodes;
  double *thr, *mglo, *mghi;
  unsigned long long* tie_count;
};

// the value whose range shapes the next level's bins: the next level's key, or its code
template <class TK>
__device__ inline TK mid_next_key(const MidArgs<TK>& A, int t, int id) {
  const int64_t o = ((int64_t)t * A.L + A.level + 1) * A.N + id;
  return A.Cd ? (TK)A.Cd[o] : A.P[o];
}

template <class TK>
__device__ inline void mid_wave_path(const MidArgs<TK>& A, const SNode<TK>& a, int t, int j,
                                     int lane) {
  const int cMid = a.cMid, M = A.M;
  const int64_t N = A.N;
  Keys<TK> K{A.P + (int64_t)t * A.L * N, N, A.level, nullptr};
  const int32_t* m = A.pool + (int64_t)t * N + a.midoff;
  const TK* mk = A.poolkey + (int64_t)t * N + a.midoff;
  // at most 128 points: two per lane (the 4-register instantiation of wave_bitonic proved
  // codegen-sensitive on ROCm 7.2 and is not used; larger bins take the block-level path)
  TK k[2];
  int id[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r * 64 + lane;
    id[r] = i < cMid ? m[i] : kPad;
    k[r] = i < cMid ? mk[i] : pos_inf<TK>();
  }
  if (cMid <= 64) {
    TK k1[1] = {k[0]};
    int i1[1] = {id[0]};
    wave_bitonic_fast<TK, 1>(k1, i1, K);
    k[0] = k1[0];
    id[0] = i1[0];
  } else {
    wave_bitonic_fast<TK, 2>(k, id, K);
  }
  const int kk = a.nh - a.cL;  // the first kk points of the sorted pivot bin go left
  uint16_t* no = A.node_of + (int64_t)t * N;
  unsigned long long mn[2] = {~0ULL, ~0ULL}, mx[2] = {0ULL, 0ULL};
  const int n = a.n, nh = a.nh;
  // the children's range of the next level's key comes from stream_assign's sample; the
  // pivot-bin points only add to it when they are a large part of the node (ties)
  const bool feed_next = A.has_next && cMid * 4 >= n;
  const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
  TK vthr = (TK)0, vlo = (TK)0, vhi = (TK)0;
  int have = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r * 64 + lane;
    if (i < cMid) {
      const int side = i >= kk;
      no[id[r]] = (uint16_t)(2 * j + side);
      if (feed_next) {
        const unsigned long long o = ord_of(mid_next_key(A, t, id[r]));
        mn[side] = o < mn[side] ? o : mn[side];
        mx[side] = o > mx[side] ? o : mx[side];
      }
      if (i == nh - a.cL) {
        vthr = k[r];
        have |= 1;
      }
      if (i == il - a.cL) {
        vlo = k[r];
        have |= 2;
      }
      if (i == ih - a.cL) {
        vhi = k[r];
        have |= 4;
      }
    }
  }
  if (feed_next) {
    for (int sd = 0; sd < 2; ++sd) {
      unsigned long long x = mn[sd], y = mx[sd];
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long p = __shfl_xor(x, o), q = __shfl_xor(y, o);
        x = p < x ? p : x;
        y = q > y ? q : y;
      }
      if (lane == 0) {
        if (x != ~0ULL) atomicMin(&A.cmin_next[(int64_t)t * 2 * M + 2 * j + sd], x);
        if (y != 0ULL) atomicMax(&A.cmax_next[(int64_t)t * 2 * M + 2 * j + sd], y);
      }
    }
  }
  // gather the three order statistics on lane 0
  const unsigned long long b1 = __ballot(have & 1), b2 = __ballot(have & 2), b4 = __ballot(have & 4);
  const TK tthr = __shfl(vthr, __ffsll((long long)b1) - 1);
  const TK tlo = b2 ? __shfl(vlo, __ffsll((long long)b2) - 1) : ord_to(a.maxL, TK());
  const TK thi = b4 ? __shfl(vhi, __ffsll((long long)b4) - 1) : ord_to(a.minR, TK());
  if (lane == 0) {
    const int64_t h = (int64_t)t * A.nodes + A.heap0 + j;
    A.thr[h] = (double)tthr;
    A.mglo[h] = (double)tlo;
    A.mghi[h] = (double)thi;
    if (nh > 0 && !(tlo < tthr)) atomicAdd(A.tie_count, 1ULL);
  }
}

// LDS sort of one pivot bin by the whole block, or (WAVE) by one wave in its private region
template <class TK, bool WAVE>
__device__ inline void mid_lds_path(const MidArgs<TK>& A, const SNode<TK>& a, int t, int j,
                                      unsigned char* smem) {
  const int cMid = a.cMid, M = A.M;
  const int64_t N = A.N;
  const int np = next_pow2(cMid);
  const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int nthr = WAVE ? 64 : (int)blockDim.x;
  TK* skey = reinterpret_cast<TK*>(smem);
  int* sid = reinterpret_cast<int*>(smem + (size_t)np * sizeof(TK));
  Keys<TK> K{A.P + (int64_t)t * A.L * N, N, A.level, nullptr};
  const int32_t* m = A.pool + (int64_t)t * N + a.midoff;
  const TK* mk = A.poolkey + (int64_t)t * N + a.midoff;
  for (int i = tid; i < np; i += nthr) {
    if (i < cMid) {
      sid[i] = m[i];
      skey[i] = mk[i];
    } else {
      sid[i] = kPad;
      skey[i] = pos_inf<TK>();
    }
  }
  if (WAVE) wsync();
  else __syncthreads();
  lds_bitonic<TK, WAVE, true>(skey, sid, np, K);
  {  // equal keys among real neighbours: the exact comparator decides (see fast_less)
    int eq = 0;
    for (int i = tid; i + 1 < cMid; i += nthr) eq |= skey[i] == skey[i + 1];
    const bool any = WAVE ? __ballot(eq) != 0ULL : __syncthreads_or(eq) != 0;
    if (any) lds_bitonic<TK, WAVE, false>(skey, sid, np, K);
  }
  const int kk = a.nh - a.cL;  // the first kk points of the sorted pivot bin go left
  uint16_t* no = A.node_of + (int64_t)t * N;
  unsigned long long mn[2] = {~0ULL, ~0ULL}, mx[2] = {0ULL, 0ULL};
  const bool feed_next = A.has_next && cMid * 4 >= a.n;  // see mid_wave_path
  for (int i = tid; i < cMid; i += nthr) {
    const int side = i >= kk;
    no[sid[i]] = (uint16_t)(2 * j + side);
    if (feed_next) {
      const unsigned long long o = ord_of(mid_next_key(A, t, sid[i]));
      mn[side] = o < mn[side] ? o : mn[side];
      mx[side] = o > mx[side] ? o : mx[side];
    }
  }
  if (feed_next) {
    for (int sd = 0; sd < 2; ++sd) {
      unsigned long long x = mn[sd], y = mx[sd];
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long p = __shfl_xor(x, o), q = __shfl_xor(y, o);
        x = p < x ? p : x;
        y = q > y ? q : y;
      }
      if ((threadIdx.x & 63) == 0) {
        if (x != ~0ULL) atomicMin(&A.cmin_next[(int64_t)t * 2 * M + 2 * j + sd], x);
        if (y != 0ULL) atomicMax(&A.cmax_next[(int64_t)t * 2 * M + 2 * j + sd], y);
      }
    }
  }
  if (tid == 0) {
    const int n = a.n, nh = a.nh;
    const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
    const int64_t h = (int64_t)t * A.nodes + A.heap0 + j;
    const TK vthr = skey[nh - a.cL];
    const TK vlo = il >= a.cL ? skey[il - a.cL] : ord_to(a.maxL, TK());
    const TK vhi = ih < a.cL + cMid ? skey[ih - a.cL] : ord_to(a.minR, TK());
    A.thr[h] = (double)vthr;
    A.mglo[h] = (double)vlo;
    A.mghi[h] = (double)vhi;
    if (nh > 0 && !(vlo < vthr)) atomicAdd(A.tie_count, 1ULL);
  }
}

// Large pivot bins (the first levels; every level of a 10 M-point shard): SELECTION instead of
// a sort.  The split needs the element of rank kk in the pivot bin, its two neighbours' keys and
// every element's side — not the order.  One linear pass bins the keys into 1024 sub-bins of
// the bin's own [min, max] (any monotone map keeps the split exact), a scan finds the sub-bin
// holding rank kk, and only that sub-bin's elements — which include every key tied with the
// threshold — are sorted with the exact comparator (one wave, <= 128 elements; more, i.e.
// heavy ties: the caller falls back to the full LDS sort).  Elements of lower / higher sub-bins
// go left / right unsorted; the neighbours outside the candidates are the max below / min above.
// Whole block (256 threads); smem holds the list (keys, ids).  Returns false for the fallback.
template <class TK>
__device__ inline bool mid_select_path(const MidArgs<TK>& A, const SNode<TK>& a, int t, int j,
                                       unsigned char* smem) {
  __shared__ int sel_hist[1024];
  __shared__ unsigned long long sel_red[8];
  __shared__ int sel_misc[4];   // [0] sub-bin, [1] below, [2] m, [3] candidate cursor
  __shared__ int sel_cand[128];
  const int cMid = a.cMid, M = A.M, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t N = A.N;
  TK* skey = reinterpret_cast<TK*>(smem);
  int* sid = reinterpret_cast<int*>(smem + (size_t)cMid * sizeof(TK));
  const int32_t* m = A.pool + (int64_t)t * N + a.midoff;
  const TK* mk = A.poolkey + (int64_t)t * N + a.midoff;
  unsigned long long omn = ~0ULL, omx = 0ULL;
  for (int i = tid; i < cMid; i += 256) {
    const TK k = mk[i];
    skey[i] = k;
    sid[i] = m[i];
    const unsigned long long o = ord_of(k);
    omn = o < omn ? o : omn;
    omx = o > omx ? o : omx;
  }
  for (int i = tid; i < 1024; i += 256) sel_hist[i] = 0;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long p = __shfl_xor(omn, o), q = __shfl_xor(omx, o);
    omn = p < omn ? p : omn;
    omx = q > omx ? q : omx;
  }
  if (lane == 0) {
    sel_red[wave] = omn;
    sel_red[4 + wave] = omx;
  }
  if (tid == 0) sel_misc[3] = 0;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    omn = sel_red[w] < omn ? sel_red[w] : omn;
    omx = sel_red[4 + w] > omx ? sel_red[4 + w] : omx;
  }
  const TK kmin = ord_to(omn, TK()), kmax = ord_to(omx, TK());
  if (!(kmin < kmax)) return false;  // one value: the exact comparator has to order all of it
  const double inv = 1024.0 / ((double)kmax - (double)kmin);
  auto sub_of = [&](TK k) -> int {
    const int b = (int)(((double)k - (double)kmin) * inv);
    return b > 1023 ? 1023 : b;
  };
  for (int i = tid; i < cMid; i += 256) atomicAdd(&sel_hist[sub_of(skey[i])], 1);
  __syncthreads();
  const int kk = a.nh - a.cL;  // rank (in the sorted pivot bin) of the threshold element
  if (wave == 0) {  // scan of 1024 counters, 16 per lane
    int c[16], tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      c[q] = sel_hist[lane * 16 + q];
      tot += c[q];
    }
    int inc = tot;
    for (int o = 1; o < 64; o <<= 1) {
      const int p = __shfl_up(inc, o);
      if (lane >= o) inc += p;
    }
    int run = inc - tot;  // elements before this lane's 16 sub-bins
    if (kk >= run && kk < inc) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (kk >= run && kk < run + c[q]) {
          sel_misc[0] = lane * 16 + q;
          sel_misc[1] = run;
          sel_misc[2] = c[q];
        }
        run += c[q];
      }
    }
  }
  __syncthreads();
  const int sstar = sel_misc[0], below = sel_misc[1], mc = sel_misc[2];
  if (mc > 128) return false;  // heavy ties around the threshold: full sort
  // candidates; max key below / min key above them
  unsigned long long lowmax = 0ULL, highmin = ~0ULL;
  for (int i = tid; i < cMid; i += 256) {
    const TK k = skey[i];
    const int sb = sub_of(k);
    if (sb == sstar) {
      sel_cand[atomicAdd(&sel_misc[3], 1)] = i;
    } else {
      const unsigned long long o = ord_of(k);
      if (sb < sstar) lowmax = o > lowmax ? o : lowmax;
      else highmin = o < highmin ? o : highmin;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long p = __shfl_xor(lowmax, o), q = __shfl_xor(highmin, o);
    lowmax = p > lowmax ? p : lowmax;
    highmin = q < highmin ? q : highmin;
  }
  __syncthreads();  // sel_red is read above by every thread: rewrite only now
  if (lane == 0) {
    sel_red[wave] = lowmax;
    sel_red[4 + wave] = highmin;
  }
  __syncthreads();
  uint16_t* no = A.node_of + (int64_t)t * N;
  unsigned long long mn[2] = {~0ULL, ~0ULL}, mx[2] = {0ULL, 0ULL};
  const bool feed_next = A.has_next && cMid * 4 >= a.n;  // see mid_wave_path
  auto place = [&](int id, int side) {
    no[id] = (uint16_t)(2 * j + side);
    if (feed_next) {
      const unsigned long long o = ord_of(mid_next_key(A, t, id));
      mn[side] = o < mn[side] ? o : mn[side];
      mx[side] = o > mx[side] ? o : mx[side];
    }
  };
  if (wave == 0) {
    for (int w = 0; w < 4; ++w) {
      lowmax = sel_red[w] > lowmax ? sel_red[w] : lowmax;
      highmin = sel_red[4 + w] < highmin ? sel_red[4 + w] : highmin;
    }
    Keys<TK> K{A.P + (int64_t)t * A.L * N, N, A.level, nullptr};
    TK k[2];
    int id[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int q = r * 64 + lane;
      const int i = q < mc ? sel_cand[q] : -1;
      id[r] = i >= 0 ? sid[i] : kPad;
      k[r] = i >= 0 ? skey[i] : pos_inf<TK>();
    }
    wave_bitonic_fast<TK, 2>(k, id, K);
    // key at rank q of the sorted pivot bin, for q in [below - 1, below + mc]
    const int n = a.n, nh = a.nh;
    const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
    const int ql = il - a.cL, qh = ih - a.cL;  // may leave the pivot bin: maxL / minR
    TK vthr = (TK)0, vlo = (TK)0, vhi = (TK)0;
    int have = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int q = below + r * 64 + lane;  // rank of this lane's element
      if (r * 64 + lane < mc) {
        place(id[r], q >= kk);
        if (q == kk) {
          vthr = k[r];
          have |= 1;
        }
        if (q == ql) {
          vlo = k[r];
          have |= 2;
        }
        if (q == qh) {
          vhi = k[r];
          have |= 4;
        }
      }
    }
    const unsigned long long b1 = __ballot(have & 1), b2 = __ballot(have & 2), b4 = __ballot(have & 4);
    const TK tthr = __shfl(vthr, __ffsll((long long)b1) - 1);
    TK tlo, thi;
    if (b2) tlo = __shfl(vlo, __ffsll((long long)b2) - 1);
    else if (ql < 0) tlo = ord_to(a.maxL, TK());
    else tlo = ord_to(lowmax, TK());   // ql == below - 1
    if (b4) thi = __shfl(vhi, __ffsll((long long)b4) - 1);
    else if (qh >= cMid) thi = ord_to(a.minR, TK());
    else thi = ord_to(highmin, TK());  // qh == below + mc
    if (lane == 0) {
      const int64_t h = (int64_t)t * A.nodes + A.heap0 + j;
      A.thr[h] = (double)tthr;
      A.mglo[h] = (double)tlo;
      A.mghi[h] = (double)thi;
      if (nh > 0 && !(tlo < tthr)) atomicAdd(A.tie_count, 1ULL);
    }
  }
  for (int i = tid; i < cMid; i += 256) {
    const int sb = sub_of(skey[i]);
    if (sb != sstar) place(sid[i], sb > sstar);
  }
  if (feed_next) {
    for (int sd = 0; sd < 2; ++sd) {
      unsigned long long x = mn[sd], y = mx[sd];
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long p = __shfl_xor(x, o), q = __shfl_xor(y, o);
        x = p < x ? p : x;
        y = q > y ? q : y;
      }
      if (lane == 0) {
        if (x != ~0ULL) atomicMin(&A.cmin_next[(int64_t)t * 2 * M + 2 * j + sd], x);
        if (y != 0ULL) atomicMax(&A.cmax_next[(int64_t)t * 2 * M + 2 * j + sd], y);
      }
    }
  }
  return true;
}

constexpr int kMidWaveLds = 512;  // pivot bins up to this size: LDS sort by one wave

template <class TK>
__global__ __launch_bounds__(256) void stream_mid(const MidArgs<TK> A, int npb, int wave_max,
                                                  int use_select /* 0: always the full sort */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int sbig[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.y;
  const int jw = blockIdx.x * npb + w;
  int big = 0;
  if (w < npb && jw < A.M) {
    const SNode<TK> a = A.nd[(int64_t)t * A.M + jw];
    if (a.cMid > wave_max) {
      if (npb > 1 && a.cMid <= kMidWaveLds)  // the waves of the block work on four nodes at once
        mid_lds_path<TK, true>(A, a, t, jw, smem + (size_t)w * kMidWaveLds * (sizeof(TK) + 4));
      else
        big = a.cMid <= kSmallCap;  // larger: the host rebuilds (sflags)
    } else if (a.cMid > 0) {
      mid_wave_path<TK>(A, a, t, jw, lane);
    }
  }
  if (lane == 0) sbig[w] = big;
  __syncthreads();
  for (int q = 0; q < npb; ++q) {
    if (!sbig[q]) continue;  // block-uniform
    const int j = blockIdx.x * npb + q;
    const SNode<TK> a = A.nd[(int64_t)t * A.M + j];
    // selection first (linear); the full LDS sort when ties make its candidate set too large
    if (use_select && mid_select_path<TK>(A, a, t, j, smem)) {
      __syncthreads();
      continue;
    }
    __syncthreads();
    mid_lds_path<TK, false>(A, a, t, j, smem);
    __syncthreads();
  }
}

// counting sort of the points by node -> perm segments, staged through LDS: a block sorts
// chunks of kPermChunk points by node in LDS and copies them out in sorted order, so a wave
// writes a handful of contiguous runs instead of 64 scattered 4-byte stores (measured: the
// direct scatter wrote 955 MB of HBM traffic for 128 MB of permutation).  Does nothing once
// *abort is set (a pivot bin outgrew LDS: the host rebuilds with the general path).
// grid = (nblk, T)
constexpr int kPermChunk = 8192;
template <class TK>
__global__ __launch_bounds__(kStreamThreads) void stream_to_perm(
    const uint16_t* __restrict__ node_of, int64_t N, int levels, int64_t per,
    unsigned int* __restrict__ gcur, int32_t* __restrict__ perm,
    const unsigned int* __restrict__ abort) {
  constexpr int EPT = kPermChunk / kStreamThreads;  // points per thread per chunk
  __shared__ unsigned int cnt[2 * kStreamMaxNodes];  // counts, then the chunk-local prefixes
  __shared__ int gb[2 * kStreamMaxNodes], noff[2 * kStreamMaxNodes];
  __shared__ int sid[kPermChunk];
  __shared__ unsigned short sj[kPermChunk];
  __shared__ unsigned int wsum[kStreamThreads / 64];  // 72 KB in all: two blocks per CU
  if (*abort) return;
  const int t = blockIdx.y;
  const int M = 1 << levels;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint16_t* no = node_of + (int64_t)t * N;
  int32_t* pm = perm + (int64_t)t * N;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
  for (int j = threadIdx.x; j < M; j += kStreamThreads) {
    int64_t off = 0, n = N;
    for (int b = levels - 1; b >= 0; --b) {
      const int64_t nh = n >> 1;
      if ((j >> b) & 1) {
        off += nh;
        n -= nh;
      } else {
        n = nh;
      }
    }
    noff[j] = (int)off;
  }
  for (int64_t c0 = i0; c0 < i1; c0 += kPermChunk) {
    const int cn = (int)(i1 - c0 < kPermChunk ? i1 - c0 : kPermChunk);
    for (int j = threadIdx.x; j < M; j += kStreamThreads) cnt[j] = 0;
    __syncthreads();
    unsigned int pk[EPT];  // (node << 16) | rank inside this chunk
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = e * kStreamThreads + threadIdx.x;
      pk[e] = 0xffffffffu;
      if (k < cn) {
        const unsigned int j = no[c0 + k];
        pk[e] = (j << 16) | atomicAdd(&cnt[j], 1u);
      }
    }
    __syncthreads();
    // exclusive prefix of cnt over the nodes (M <= 2 * kStreamMaxNodes: two per thread)
    unsigned int v0 = 0, v1 = 0;
    const int j0 = 2 * threadIdx.x, j1 = j0 + 1;
    if (j0 < M) v0 = cnt[j0];
    if (j1 < M) v1 = cnt[j1];
    unsigned int inc = v0 + v1;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int pre = inc - (v0 + v1);
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (j0 < M) {  // gb: where the chunk's first point of the node goes, minus its chunk slot
      cnt[j0] = pre;
      gb[j0] = v0 ? noff[j0] + (int)atomicAdd(&gcur[(int64_t)t * M + j0], v0) - (int)pre : 0;
    }
    if (j1 < M) {
      cnt[j1] = pre + v0;
      gb[j1] = v1 ? noff[j1] + (int)atomicAdd(&gcur[(int64_t)t * M + j1], v1) - (int)(pre + v0) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e)
      if (pk[e] != 0xffffffffu) {
        const unsigned int j = pk[e] >> 16, r = pk[e] & 0xffffu;
        const int sl = (int)(cnt[j] + r);
        sid[sl] = (int)(c0 + e * kStreamThreads + threadIdx.x);
        sj[sl] = (unsigned short)j;
      }
    __syncthreads();
    for (int k = threadIdx.x; k < cn; k += kStreamThreads) pm[gb[sj[k]] + k] = sid[k];
    __syncthreads();
  }
}

// ---- code geometry (codes.h): mm[0..1] = (ord(min), ord(max)) of the projections of a strided
// sample of the rows on the columns that the streaming levels will histogram (level c % L below
// Lc) — ONE geometry for the whole build; mm starts as (~0, 0).  Straight from the data: S rows,
// `stride` apart, against the C dense-ified hyperplanes R[C][d], sixteen columns and 256 sample
// rows per block, plain FMA dot products — the range only shapes the code bins (values outside
// it clamp into the edge codes), so it needs neither the build's projection kernel nor its
// rounding.  (Round 2's first version gathered the rows, ran the MFMA kernels over them and
// reduced the result: seven launches, 0.12 ms per build.)
// grid = (ceil(C / 16), ceil(S / 256)), 256 threads.
__device__ inline double sample_value(double v) { return v; }
__device__ inline double sample_value(float v) { return (double)v; }
__device__ inline double sample_value(uint16_t v) {  // bf16 bits
  return (double)__uint_as_float((unsigned int)v << 16);
}
template <class TIn, class TK>
__global__ __launch_bounds__(256) void sample_range_kernel(const TIn* __restrict__ X, int64_t stride,
                                                           int S, int d, const double* __restrict__ R,
                                                           int C, int L, int Lc,
                                                           unsigned long long* __restrict__ mm) {
  constexpr int CG = 16, KCH = 128;
  __shared__ __attribute__((aligned(16))) double rl[KCH * CG];
  const int c0 = blockIdx.x * CG;
  const int si = blockIdx.y * 256 + threadIdx.x;
  const bool live = si < S;
  const TIn* xr = X + (int64_t)(live ? si : 0) * stride * d;
  double acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.0;
  for (int k0 = 0; k0 < d; k0 += KCH) {
    const int kn = d - k0 < KCH ? d - k0 : KCH;
    __syncthreads();
    for (int i = threadIdx.x; i < kn * CG; i += 256) {
      const int c = i / kn, k = i % kn;  // consecutive threads read consecutive k of one column
      rl[k * CG + c] = c0 + c < C ? R[(int64_t)(c0 + c) * d + k0 + k] : 0.0;
    }
    __syncthreads();
    // a thread walks its own sample row: sixteen elements of it in flight at a time (one load per
    // multiply-add round left the walk waiting for HBM d times: 81 us of a 4.6 ms C2 build)
    int k = 0;
    for (; k + 16 <= kn; k += 16) {
      TIn xv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) xv[u] = xr[k0 + k + u];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double x = sample_value(xv[u]);
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = __builtin_fma(x, rl[(k + u) * CG + c], acc[c]);
      }
    }
    for (; k < kn; ++k) {
      const double x = sample_value(xr[k0 + k]);
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_fma(x, rl[k * CG + c], acc[c]);
    }
  }
  unsigned long long mn = ~0ULL, mx = 0ULL;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const TK v = (TK)acc[c];
    if (live && c0 + c < C && (c0 + c) % L < Lc && v == v) {  // NaN samples do not shape anything
      const unsigned long long o = ord_of(v);
      mn = o < mn ? o : mn;
      mx = o > mx ? o : mx;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) {
    atomicMin(&mm[0], mn);
    atomicMax(&mm[1], mx);
  }
}

// start-of-build initialisation in ONE launch: NaN node records (not a Bin), zero counters and
// flags.  (Each separate fill / memset costs a ~4 us dispatch; a 4-tree shard builds in 1.3 ms.)
// ---- codes AFTER the projection (round 3), for the projection kernels that have no code epilogue
// (bf16 rows on the bf16 matrix pipe, CSR rows): the geometry from a strided sample of the
// streamed P columns themselves, then one coalesced pass P -> 16-bit codes.  The codes are a
// function of the stored keys, so "code(x) < code(y) => key(x) < key(y)" holds by construction.
template <class TK>
__global__ __launch_bounds__(256) void pcode_range_kernel(const TK* __restrict__ P, int64_t N, int L,
                                                          int Lc, unsigned long long* mm) {
  const int c = blockIdx.x;
  if (c % L >= Lc) return;
  const TK* col = P + (int64_t)c * N;
  const int64_t m = N < 4096 ? N : 4096;
  unsigned long long mn = ~0ULL, mx = 0ULL;
  for (int64_t i = threadIdx.x; i < m; i += 256) {
    const unsigned long long o = ord_of(col[i * N / m]);
    mn = o < mn ? o : mn;
    mx = o > mx ? o : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(mm, mn);
    atomicMax(mm + 1, mx);
  }
}

template <class TK>
__global__ __launch_bounds__(256) void pcode_kernel(const TK* __restrict__ P, int64_t N, int L, int Lc,
                                                    const unsigned long long* __restrict__ mm,
                                                    uint16_t* __restrict__ codes) {
  const int c = blockIdx.y;
  if (c % L >= Lc) return;
  const CodeGeo<TK> g = code_geo<TK>(mm[0], mm[1]);
  const TK* col = P + (int64_t)c * N;
  uint16_t* out = codes + (int64_t)c * N;
  const int64_t n8 = N / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    struct alignas(16) V8 { uint16_t v[8]; } o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = code_of(col[i * 8 + e], g);
    *reinterpret_cast<V8*>(out + i * 8) = o;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n8 * 8 + threadIdx.x; i < N; i += 256) out[i] = code_of(col[i], g);
}

__global__ void build_init_kernel(double* thr, double* mglo, double* mghi, int64_t n,
                                  unsigned long long* counters /*[2]*/, unsigned int* sflags /*[4]*/) {
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    thr[i] = nan;
    mglo[i] = nan;
    mghi[i] = nan;
  }
  if (blockIdx.x == 0 && threadIdx.x < 2) counters[threadIdx.x] = 0ULL;
  if (blockIdx.x == 0 && threadIdx.x < 4) sflags[threadIdx.x] = 0u;
}

// start of the streaming phase: empty root ranges, zero pivot-list cursors
__global__ void stream_init_kernel(unsigned long long* cmin, unsigned long long* cmax, int T,
                                   unsigned int* poolcur, int npool) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
    cmin[i] = ~0ULL;
    cmax[i] = 0ULL;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npool; i += gridDim.x * blockDim.x)
    poolcur[i] = 0u;
}

__global__ void fill_u64_kernel(unsigned long long* p, int64_t n, unsigned long long v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// ---------------------------------------------------------------------------------------
// HBM merge sort of arbitrary segments (rare path: pivot bins / leaves larger than LDS).
// GSeg list lives in device memory; buf holds ids, sorted in place (tmp = scratch).
// ---------------------------------------------------------------------------------------
struct GSeg {
  int64_t off;   // absolute offset into the perm buffer ([T][N] flattened)
  int32_t n;
  int32_t t;     // tree (selects the projection block)
  int32_t heap;  // node to emit (or -1)
  int32_t nh_rel, il_rel, ih_rel;  // positions (relative to off) of thr / mglo / mghi or -1
};

template <class TK>
__global__ __launch_bounds__(256) void gsort_chunk_kernel(int32_t* buf, int64_t N, const TK* P,
                                                          int L, int level,
                                                          const GSeg* __restrict__ gs,
                                                          const int32_t* tb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const GSeg g = gs[blockIdx.y];
  const int64_t c0 = (int64_t)blockIdx.x * kSmallCap;
  if (c0 >= g.n) return;
  const int n = (int)((g.n - c0) < kSmallCap ? (g.n - c0) : kSmallCap);
  const int np = next_pow2(n);
  TK* skey = reinterpret_cast<TK*>(smem);
  int* sid = reinterpret_cast<int*>(smem + (size_t)np * sizeof(TK));
  Keys<TK> K{P + (int64_t)g.t * L * N, N, level, tb ? tb + (int64_t)g.t * N : nullptr};
  int32_t* s = buf + g.off + c0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    if (i < n) {
      const int id = s[i];
      sid[i] = id;
      skey[i] = K.key(id);
    } else {
      sid[i] = kPad;
      skey[i] = pos_inf<TK>();
    }
  }
  __syncthreads();
  lds_bitonic(skey, sid, np, K);
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = sid[i];
}

// one rank-merge pass: runs of width w -> runs of width 2w, in -> out. grid=(ceil(nmax/256),G)
template <class TK>
__global__ __launch_bounds__(256) void gsort_merge_kernel(const int32_t* __restrict__ in,
                                                          int32_t* __restrict__ out, int64_t N,
                                                          const TK* P, int L, int level,
                                                          const GSeg* __restrict__ gs, int64_t w,
                                                          const int32_t* tb) {
  const GSeg g = gs[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  Keys<TK> K{P + (int64_t)g.t * L * N, N, level, tb ? tb + (int64_t)g.t * N : nullptr};
  const int32_t* s = in + g.off;
  const int id = s[i];
  const int64_t run = i / w, pair0 = (run & ~1LL) * w;
  const bool in_a = (run & 1) == 0;
  const int64_t o0 = in_a ? pair0 + w : pair0;                     // sibling run start
  int64_t olen = g.n - o0;
  if (olen > w) olen = w;
  int64_t rank = 0;
  if (olen > 0) {
    // number of sibling elements that precede this element (keys are unique under the full
    // order, so lower and upper bound coincide)
    const TK key = K.key(id);
    int64_t lo = 0, hi = olen;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      const int oid = s[o0 + mid];
      if (K.less(K.key(oid), oid, key, id)) lo = mid + 1;
      else hi = mid;
    }
    rank = lo;
  }
  const int64_t own = in_a ? i - pair0 : i - (pair0 + w);
  out[g.off + pair0 + own + rank] = id;
}

__global__ void gsort_copy_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                  const GSeg* __restrict__ gs) {
  const GSeg g = gs[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < g.n) out[g.off + i] = in[g.off + i];
}

// emit thr/margins of nodes whose segment (or pivot bin) was sorted by the HBM path
template <class TK>
__global__ void gsort_emit_kernel(const int32_t* __restrict__ buf, int64_t N, const TK* P, int L,
                                  int level, const GSeg* __restrict__ gs, int G,
                                  const NodeAux* aux_of /*or null*/, const int* aux_idx,
                                  double* thr, double* mglo, double* mghi, int64_t nodes,
                                  unsigned long long* tie_count) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= G) return;
  const GSeg g = gs[gi];
  if (g.heap < 0) return;
  const TK* Pl = P + ((int64_t)g.t * L + level) * N;
  const int64_t h = (int64_t)g.t * nodes + g.heap;
  const TK vthr = Pl[buf[g.off + g.nh_rel]];
  TK vlo, vhi;
  if (g.il_rel >= 0) vlo = Pl[buf[g.off + g.il_rel]];
  else vlo = ord_to(aux_of[aux_idx[gi]].maxL, TK());
  if (g.ih_rel >= 0) vhi = Pl[buf[g.off + g.ih_rel]];
  else vhi = ord_to(aux_of[aux_idx[gi]].minR, TK());
  thr[h] = (double)vthr;
  mglo[h] = (double)vlo;
  mghi[h] = (double)vhi;
  if (!(vlo < vthr)) atomicAdd(tie_count, 1ULL);
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
template <class TK>
int32_t gsort(rpt_ctx* ctx, int32_t* buf, int32_t* tmp, int64_t N, const TK* P, int L, int level,
              const std::vector<GSeg>& list, DevBuf<GSeg>& dlist, const int32_t* tb) {
  if (list.empty()) return RPT_OK;
  RPT_TRY(dlist.alloc(list.size()));
  RPT_HIP(hipMemcpyAsync(dlist.p, list.data(), list.size() * sizeof(GSeg), hipMemcpyHostToDevice,
                         ctx->stream));
  RPT_HIP(stream_sync(ctx->stream));  // list is host stack memory
  int64_t nmax = 0;
  for (const GSeg& g : list) nmax = g.n > nmax ? g.n : nmax;
  const unsigned G = (unsigned)list.size();
  const size_t smem = (size_t)kSmallCap * (sizeof(TK) + 4);
  hipLaunchKernelGGL(gsort_chunk_kernel<TK>, dim3((unsigned)((nmax + kSmallCap - 1) / kSmallCap), G),
                     dim3(256), smem, ctx->stream, buf, N, P, L, level, dlist.p, tb);
  int32_t* a = buf;
  int32_t* b = tmp;
  for (int64_t w = kSmallCap; w < nmax; w *= 2) {
    hipLaunchKernelGGL(gsort_merge_kernel<TK>, dim3((unsigned)((nmax + 255) / 256), G), dim3(256),
                       0, ctx->stream, a, b, N, P, L, level, dlist.p, w, tb);
    std::swap(a, b);
  }
  if (a != buf)
    hipLaunchKernelGGL(gsort_copy_kernel, dim3((unsigned)((nmax + 255) / 256), G), dim3(256), 0,
                       ctx->stream, a, buf, dlist.p);
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define HT(label) do { if (ctx->opt.debug_host) { (void)stream_sync(ctx->stream); double t__ = now_ms(); fprintf(stderr, "host %-28s %8.3f ms\n", label, t__ - ht_last); ht_last = t__; } } while (0)

template <class TK>
int32_t build_forest_t(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int32_t mode,
                       bool no_stream = false) {
  double ht_last = now_ms();
  const int64_t N = f->n;
  const int T = f->T, L = f->L;
  hipStream_t st = ctx->stream;
  f->mode = mode;
  ctx->last_csub_redo = ctx->last_csub_bad = 0;

  // deepest level that still splits a node: the largest node of level l has ceil(N / 2^l) points
  int Lused = 0;
  {
    int64_t n = N;
    for (int l = 0; l < L; ++l) {
      if (n > (int64_t)f->min_leaf) Lused = l + 1;
      n -= n / 2;
    }
  }
  DevBuf<unsigned long long> counters;  // [0] tie nodes, [1] (uint) big-mid count
  DevBuf<unsigned int> sflags;          // see below
  RPT_TRY(counters.alloc(2));
  RPT_TRY(sflags.alloc(4));
  hipLaunchKernelGGL(build_init_kernel, dim3(256), dim3(256), 0, st, f->thr.p, f->mglo.p, f->mghi.p,
                     (int64_t)T * f->nodes, counters.p, sflags.p);
  if (N == 0) return RPT_OK;
  if (Lused == 0) {  // the root is a Tip: data in input order (Internal.hs:289-290)
    hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, st, f->perm.p, N, T);
    RPT_HIP(hipGetLastError());
    return RPT_OK;
  }

  // ---- projection batch: all (tree, level) hyperplanes; enqueued first, the host prepares the
  // split while the device projects ----
  RPT_TRY(f->proj.alloc((size_t)T * L * N * sizeof(TK)));
  TK* P = reinterpret_cast<TK*>(f->proj.p);
  // 16-bit codes of the keys of the levels the streaming path will handle (codes.h): worth
  // their sample pass from a few hundred thousand points on; dense rows only.  The number of
  // streamed levels is a function of (N, minLeaf, L): every node of level l is a Bin while the
  // smallest one, floor-halved l times, is above minLeaf (the topology loop below agrees).
  const int stream_cap = ctx->opt.stream_maxnodes > 0
                             ? (int)(ctx->opt.stream_maxnodes < kStreamMaxNodes ? ctx->opt.stream_maxnodes
                                                                                : kStreamMaxNodes)
                             : (N / 1024 > kSmallCap ? 1024 : 512);
  int Lc = 0;
  if (N >= 2048 && !no_stream && !ctx->opt.no_stream) {
    int64_t nmin = N;
    while (Lc < Lused && nmin > (int64_t)f->min_leaf && (1 << Lc) <= stream_cap) {
      ++Lc;
      nmin /= 2;
    }
  }
  // ... and of the levels csub_kernel will run right below them: after Lc levels the nodes hold
  // N / 2^Lc points; between the wave kernel's 1024 and kCsCap they take r more levels on codes
  int Lcodes = Lc;
  if (Lc > 0 && !ctx->opt.no_csub) {
    int64_t nmax = N, nmin = N;
    for (int l = 0; l < Lc; ++l) {
      nmax -= nmax / 2;
      nmin /= 2;
    }
    if (nmax > kWCap && nmax <= kCsCap) {
      int r = 0;
      while (nmax > kWCap) {
        nmax -= nmax / 2;
        nmin /= 2;
        ++r;
      }
      if (r <= kCsMaxR && Lc + r < L && nmin > (int64_t)f->min_leaf) Lcodes = Lc + r;
    }
  }
  DevBuf<uint16_t> codes;
  DevBuf<unsigned long long> code_mm;
  uint16_t* Cd = nullptr;  // [T][L][N], the columns of levels < Lcodes only
  const bool codes_wanted = Lc > 0 && Lused == L && N >= ((int64_t)1 << 17) && !ctx->opt.no_codes;
  if (codes_wanted && !project_writes_codes(ctx, ds, mode) && !ctx->opt.no_pcodes &&
      (N % 8) == 0) {
    // no code epilogue in this projection kernel (bf16 rows on the bf16 pipe, CSR rows): project,
    // then derive the codes from the stored keys in one coalesced pass (pcode_kernel)
    RPT_TRY(project_columns(ctx, ds, f->R.p, T * L, mode, P));
    RPT_TRY(code_mm.alloc(2));
    RPT_TRY(codes.alloc((size_t)T * L * N));
    ProfScope ps(ctx, RPT_PROF_SPLIT);  // the pass is split work: it exists for the streamed levels
    hipLaunchKernelGGL(stream_init_kernel, dim3(1), dim3(64), 0, st, code_mm.p, code_mm.p + 1, 1,
                       (unsigned int*)nullptr, 0);  // (~0, 0)
    hipLaunchKernelGGL(pcode_range_kernel<TK>, dim3((unsigned)(T * L)), dim3(256), 0, st,
                       (const TK*)P, N, L, Lcodes, code_mm.p);
    const unsigned pb = (unsigned)std::min<int64_t>((N / 8 + 255) / 256, (int64_t)ctx->n_cu * 4);
    hipLaunchKernelGGL(pcode_kernel<TK>, dim3(pb > 0 ? pb : 1, (unsigned)(T * L)), dim3(256), 0, st,
                       (const TK*)P, N, L, Lcodes, (const unsigned long long*)code_mm.p, codes.p);
    RPT_HIP(hipGetLastError());
    Cd = codes.p;
  } else if (codes_wanted && project_writes_codes(ctx, ds, mode)) {
    // geometry: the range of ALL streamed columns over a strided sample of the rows (a few
    // columns are not enough: on clustered data a hyperplane's range depends on how it separates
    // the clusters, and a column that leaves the range clamps half its points into one code),
    // through the MFMA kernels whatever the build's mode (the range only shapes bins)
    constexpr int S = 4096;
    const int C = T * L;
    RPT_TRY(code_mm.alloc(2));
    hipLaunchKernelGGL(stream_init_kernel, dim3(1), dim3(64), 0, st, code_mm.p, code_mm.p + 1, 1,
                       (unsigned int*)nullptr, 0);  // (~0, 0)
    const int64_t stride = N / S;
    const dim3 sgrid((unsigned)((C + 15) / 16), (unsigned)((S + 255) / 256));
    if (ds->dtype == RPT_F64)
      hipLaunchKernelGGL((sample_range_kernel<double, TK>), sgrid, dim3(256), 0, st,
                         (const double*)ds->X, stride, S, f->d, f->R.p, C, L, Lcodes, code_mm.p);
    else if (ds->dtype == RPT_F32)
      hipLaunchKernelGGL((sample_range_kernel<float, TK>), sgrid, dim3(256), 0, st,
                         (const float*)ds->X, stride, S, f->d, f->R.p, C, L, Lcodes, code_mm.p);
    else
      hipLaunchKernelGGL((sample_range_kernel<uint16_t, TK>), sgrid, dim3(256), 0, st,
                         (const uint16_t*)ds->X, stride, S, f->d, f->R.p, C, L, Lcodes, code_mm.p);
    RPT_TRY(codes.alloc((size_t)T * L * N));
    CodeOut co;
    co.codes = codes.p;
    co.mm = code_mm.p;
    co.ld = N;
    co.L = L;
    co.Lc = Lcodes;
    bool written = false;
    RPT_TRY(project_columns(ctx, ds, f->R.p, C, mode, P, &co, &written));
    if (written) Cd = codes.p;
    else codes.release();
  } else if (Lused == L) {
    RPT_TRY(project_columns(ctx, ds, f->R.p, T * L, mode, P));
  } else {  // levels >= Lused are never reached (Internal.hs:270: rvs ! ixLev is lazy)
    for (int t = 0; t < T; ++t)
      RPT_TRY(project_columns(ctx, ds, f->R.p + (int64_t)t * L * f->d, Lused, mode,
                              P + (int64_t)t * L * N));
  }

  // ---- topology: split nodes and leaves per level ----
  std::vector<Node> topo;
  enumerate_topology(N, L, f->min_leaf, topo);
  std::vector<std::vector<Seg>> splits((size_t)Lused), leaves((size_t)Lused + 1);
  for (const Node& nd : topo) {
    Seg s{nd.off, (int32_t)nd.n, nd.leaf ? -1 : (int32_t)nd.heap};
    if (nd.leaf) leaves[(size_t)nd.level].push_back(s);
    else splits[(size_t)nd.level].push_back(s);
  }
  // DFS order is not offset order per level; both are fine (segments are disjoint)

  HT("projection");
  // ---- work buffers ----
  DevBuf<int32_t> bufA, bufB;
  DevBuf<TK> Kst;
  RPT_TRY(bufA.alloc((size_t)T * N));
  RPT_TRY(bufB.alloc((size_t)T * N));
  size_t Smax = 1, SbigMax = 0;
  for (auto& v : splits) {
    Smax = v.size() > Smax ? v.size() : Smax;
    if (!v.empty() && v[0].n > kSmallCap) SbigMax = v.size() > SbigMax ? v.size() : SbigMax;
  }
  // a level can mix big and small nodes only around n == kSmallCap (+-1): handle per node
  DevBuf<Seg> dsegs, dsegs2;
  DevBuf<BinInfo<TK>> bins;
  DevBuf<NodeAux> aux;
  DevBuf<unsigned int> hist, bigflags;
  DevBuf<GSeg> dglist;
  DevBuf<unsigned int> ovf;
  bool have_big = false;
  for (auto& v : splits)
    for (const Seg& s : v) have_big = have_big || s.n > kSmallCap;
  if (have_big) {
    RPT_TRY(Kst.alloc((size_t)T * N));
    size_t sb = 0;
    for (auto& v : splits) {
      size_t c = 0;
      for (const Seg& s : v) c += s.n > kSmallCap;
      sb = c > sb ? c : sb;
    }
    RPT_TRY(bins.alloc((size_t)T * sb));
    RPT_TRY(aux.alloc((size_t)T * sb));
    RPT_TRY(hist.alloc((size_t)T * sb * kNB));
    RPT_TRY(bigflags.alloc((size_t)T * sb));
    RPT_HIP(hipMemsetAsync(hist.p, 0, (size_t)T * sb * kNB * 4, st));
  }

  int32_t* F = f->perm.p;
  unsigned long long* dbgbuf = nullptr;
  DevBuf<unsigned long long> dbgdev;
  if (ctx->opt.debug_stamps) {
    RPT_TRY(dbgdev.alloc(256));
    RPT_HIP(hipMemsetAsync(dbgdev.p, 0, 256 * 8, st));
    dbgbuf = dbgdev.p;
  }
  unsigned long long* tie_count = counters.p;
  unsigned int* big_count = reinterpret_cast<unsigned int*>(counters.p + 1);

  auto upload = [&](const std::vector<Seg>& v, DevBuf<Seg>& d) -> int32_t {
    RPT_TRY(d.ensure(v.size()));
    return upload_async(ctx, d.p, v.data(), v.size() * sizeof(Seg));  // staged: no wait
  };

  // pending split nodes per level, each remembering which ping-pong buffer holds its points
  struct PNode {
    Seg seg;
    int buf;
  };
  std::vector<std::vector<PNode>> pending((size_t)Lused + kRmax + 1);
  int32_t* bufs[2] = {bufA.p, bufB.p};
  // sflags[0]: a streaming level met a pivot bin larger than LDS (heavy ties / extreme
  // outliers).  The levels after it work on inconsistent node ids (harmless: every access
  // stays in bounds), the kernels that consume the streamed permutation do nothing once it is
  // set, and the host — which looks at it at its next synchronisation point, normally the one
  // at the very end — rebuilds the forest with the general path, which has the fallbacks.
  // (sflags: allocated and zeroed with the node records above)
  bool stream_unchecked = false;
  auto stream_aborted = [&](bool* aborted) -> int32_t {  // call right after a sync point
    *aborted = false;
    if (!stream_unchecked) return RPT_OK;
    unsigned int v = 0;
    RPT_HIP(hipMemcpy(&v, sflags.p, 4, hipMemcpyDeviceToHost));
    stream_unchecked = false;
    *aborted = v != 0;
    return RPT_OK;
  };
  HT("alloc work buffers");
  // ---- streaming path for the leading levels ----
  int Lstream = 0;
  if (N >= 2048 && !no_stream && !ctx->opt.no_stream)
  {
    // measured at C2: with 32 bins per node (1024 nodes) an eighth of all points lands in pivot
    // bins and the exact resolution eats what the shorter wave phase saves -> 512 by default
    // (a level whose nodes would still be above the block kernel's 4096 points afterwards is
    // cheaper streamed with coarse bins — large pivot bins are split by selection — than on the
    // general path: 10 M-point shards stream one more level)
    int max_nodes = ctx->opt.stream_maxnodes > 0 ? (int)ctx->opt.stream_maxnodes
                                                 : (N / 1024 > kSmallCap ? 1024 : 512);
    if (max_nodes > kStreamMaxNodes) max_nodes = kStreamMaxNodes;
    while (Lstream < Lused && splits[(size_t)Lstream].size() == ((size_t)1 << Lstream) &&
           (1 << Lstream) <= max_nodes)
      ++Lstream;
  }
  int streamed = 0;  // levels completed by the streaming path
  if (Lstream > 0) {
    DevBuf<uint16_t> node_of;
    DevBuf<unsigned int> part, poolcur, gcur, totals;
    DevBuf<SNode<TK>> snodes;
    DevBuf<unsigned long long> mm[4];  // cmin/cmax ping-pong
    // histogram pass: every block leaves a 64 KB partial that stream_pick reads back, so keep
    // >= min_per points per block; 16-bit LDS counters cap a block at 65535 points
    const int64_t min_per = ctx->opt.stream_minper > 0 ? ctx->opt.stream_minper : 32768;
    int64_t nblk = (2 * (int64_t)ctx->n_cu + T - 1) / T;
    if (nblk > (N + min_per - 1) / min_per) nblk = (N + min_per - 1) / min_per;
    if (nblk < (N + 65519) / 65520) nblk = (N + 65519) / 65520;
    if (nblk < 1) nblk = 1;
    const int64_t per = (((N + nblk - 1) / nblk) + 7) & ~(int64_t)7;  // multiple of 8, <= 65534
    const dim3 sgrid((unsigned)nblk, (unsigned)T);
    // assign pass: no per-block table to flush, fill the chip
    // blocks per CU of stream_assign: ONE since the pivot-bin points go through the LDS list (every
    // block flushes 2 M range cells with global atomics and loads M node records: fewer blocks, less
    // of both — 4-tree shard 1.28 -> 1.23 ms, 8 trees 1.79 -> 1.74, 32 trees unchanged; two were
    // better while the pass waited on a returning atomic per pivot point)
    // ... and TWO again in code mode since round 4: its instantiation needs 68 KB of LDS, both blocks are
    // resident and the pass has twice the waves to hide its loads behind (C2 build 4.55 -> 4.46 ms, C4 shard
    // split 9.55 -> 9.2 ms, 4-tree shard unchanged; key mode: 100 KB, one block)
    const int64_t afac = ctx->opt.tune0 > 0 ? ctx->opt.tune0 : ((Cd && Lstream <= Lc) ? 2 : 1);
    int64_t nblkA = (afac * (int64_t)ctx->n_cu + T - 1) / T;
    if (nblkA > (N + 8191) / 8192) nblkA = (N + 8191) / 8192;
    if (nblkA < 1) nblkA = 1;
    const int64_t perA = (((N + nblkA - 1) / nblkA) + 7) & ~(int64_t)7;  // 16-byte key / code loads
    const dim3 agrid((unsigned)nblkA, (unsigned)T);
    RPT_TRY(node_of.alloc((size_t)T * N));
    RPT_TRY(part.alloc((size_t)T * nblk * (kStreamBins / 2)));
    RPT_TRY(poolcur.alloc((size_t)T * Lstream));
    RPT_TRY(snodes.alloc((size_t)T * kStreamMaxNodes));
    for (auto& b : mm) RPT_TRY(b.alloc((size_t)T * 2 * kStreamMaxNodes));
    RPT_HIP(hipMemsetAsync(node_of.p, 0, (size_t)T * N * 2, st));
    unsigned long long *cmin = mm[0].p, *cmax = mm[1].p, *cminN = mm[2].p, *cmaxN = mm[3].p;
    hipLaunchKernelGGL(stream_init_kernel, dim3(4), dim3(256), 0, st, cmin, cmax, T, poolcur.p,
                       T * Lstream);
    // code mode only for the levels that have codes (the estimate above covers them all unless
    // an option changed the plan in between)
    const uint16_t* Cs = (Cd && Lstream <= Lc) ? Cd : nullptr;
    hipLaunchKernelGGL(stream_minmax0<TK>, sgrid, dim3(kStreamThreads), 0, st, P, Cs, N, L, per, cmin,
                       cmax);
    HT("stream alloc+minmax0");
    int32_t* pool = bufB.p;  // the ping-pong buffers are idle while nothing moves
    DevBuf<TK> poolkey;      // keys of the pivot-bin lists, same indexing as pool
    RPT_TRY(poolkey.alloc((size_t)T * N));
    const int wave_max = ctx->opt.no_wmid ? 0 : 128;
    const int use_select = ctx->opt.no_midselect ? 0 : 1;
    // nodes above this size get more than 4096 bins (option stream_big_node: test hook)
    const int64_t big_node = ctx->opt.stream_big_node > 0 ? ctx->opt.stream_big_node
                                                          : ((int64_t)1 << 21);
    {
      ProfScope ps(ctx, RPT_PROF_SPLIT);
      for (int level = 0; level < Lstream; ++level) {
        const int M = 1 << level;
        const int has_next = level + 1 < Lstream ? 1 : 0;
        const int B = stream_bins(M, N >> level, big_node);
        hipLaunchKernelGGL(stream_hist<TK>, sgrid, dim3(kStreamThreads), 0, st, P, Cs, node_of.p, N,
                           L, level, M, B, per, cmin, cmax, part.p);
        unsigned int* pc = poolcur.p + (size_t)level * T;
#define RPT_PICK(BPT, G)                                                                       \
  hipLaunchKernelGGL((stream_pick<TK, BPT, G>), dim3((unsigned)((M + 256 / G - 1) / (256 / G)), \
                                                      (unsigned)T),                            \
                     dim3(256), 0, st, N, level, M, (int)nblk, part.p, cmin, cmax, snodes.p, pc, \
                     cminN, cmaxN, sflags.p, Cs ? 1 : 0)
        switch (B) {
          case 4096: RPT_PICK(16, 256); break;
          case 2048: RPT_PICK(8, 256); break;
          case 1024: RPT_PICK(8, 128); break;
          case 512: RPT_PICK(8, 64); break;
          case 256: RPT_PICK(8, 32); break;
          case 128: RPT_PICK(8, 16); break;
          case 64: RPT_PICK(8, 8); break;
          case 32: RPT_PICK(8, 4); break;  // 1024 nodes
          default: {  // > 4096 bins per node (very large nodes): sum the partials, then scan
            RPT_TRY(totals.ensure((size_t)T * kStreamBins));
            hipLaunchKernelGGL(stream_pick_sum, dim3((unsigned)((M * B / 2 + 255) / 256), (unsigned)T),
                               dim3(256), 0, st, M * B / 2, (int)nblk, part.p, totals.p);
            hipLaunchKernelGGL(stream_pick_big<TK>, dim3((unsigned)M, (unsigned)T), dim3(256), 0, st,
                               N, level, M, B, totals.p, cmin, cmax, snodes.p, pc, cminN, cmaxN,
                               sflags.p, Cs ? 1 : 0);
          }
        }
#undef RPT_PICK
        if (Cs)
          hipLaunchKernelGGL((stream_assign<TK, true>), agrid, dim3(kStreamThreads), 0, st, P, Cs, node_of.p, N,
                             L, level, M, B, perA, has_next, snodes.p, pool, poolkey.p, cminN, cmaxN);
        else
          hipLaunchKernelGGL((stream_assign<TK, false>), agrid, dim3(kStreamThreads), 0, st, P, Cs, node_of.p, N,
                             L, level, M, B, perA, has_next, snodes.p, pool, poolkey.p, cminN, cmaxN);
        const size_t smem = (size_t)kSmallCap * (sizeof(TK) + 4);
        const int npb = M >= 16 ? 4 : 1;
        MidArgs<TK> ma{P,     Cs,   node_of.p, N,     L,        level,    M,
                       has_next, snodes.p, pool, poolkey.p, cminN, cmaxN, (int64_t)M - 1, f->nodes,
                       f->thr.p, f->mglo.p, f->mghi.p, tie_count};
        hipLaunchKernelGGL(stream_mid<TK>, dim3((unsigned)((M + npb - 1) / npb), (unsigned)T),
                           dim3(256), smem, st, ma, npb, wave_max, use_select);
        std::swap(cmin, cminN);
        std::swap(cmax, cmaxN);
      }
      RPT_HIP(hipGetLastError());
      streamed = Lstream;
      stream_unchecked = true;
    }
    HT("stream levels");
    {
      ProfScope ps(ctx, RPT_PROF_SPLIT);
      const int M = 1 << streamed;
      std::vector<Seg> lvl((size_t)M);
      for (int j = 0; j < M; ++j) {
        int64_t off = 0, n = N;
        for (int b = streamed - 1; b >= 0; --b) {
          const int64_t nh = n / 2;
          if ((j >> b) & 1) {
            off += nh;
            n -= nh;
          } else {
            n = nh;
          }
        }
        lvl[(size_t)j] = Seg{off, (int32_t)n, (int32_t)(M - 1 + j)};
      }
      RPT_TRY(gcur.alloc((size_t)T * M));
      RPT_HIP(hipMemsetAsync(gcur.p, 0, (size_t)T * M * 4, st));
      hipLaunchKernelGGL(stream_to_perm<TK>, sgrid, dim3(kStreamThreads), 0, st, node_of.p, N,
                         streamed, per, gcur.p, bufA.p, sflags.p);
      RPT_HIP(hipGetLastError());
      // nodes of level `streamed`: Bins stay pending, Tips get their final order now
      std::vector<Seg> lsmall;
      std::vector<GSeg> lbig;
      std::vector<Seg> lbig_copy;
      for (const Seg& c : lvl) {
        if (!is_leaf(streamed, c.n, L, f->min_leaf)) {
          pending[(size_t)streamed].push_back(PNode{c, 0});
        } else if (c.n <= kSmallCap) {
          lsmall.push_back(Seg{c.off, c.n, -1});
        } else {
          lbig_copy.push_back(Seg{c.off, c.n, -1});
          for (int t = 0; t < T; ++t)
            lbig.push_back(GSeg{(int64_t)t * N + c.off, c.n, t, -1, -1, -1, -1});
        }
      }
      if (!lsmall.empty() || !lbig.empty()) {
        // these consumers do not look at the abort flag: settle it first
        RPT_HIP(ctx_sync(ctx));
        bool aborted = false;
        RPT_TRY(stream_aborted(&aborted));
        if (aborted) return build_forest_t<TK>(ctx, ds, f, mode, true);
      }
      if (!lsmall.empty()) {
        RPT_TRY(upload(lsmall, dsegs));
        int nm = 0;
        for (const Seg& sgm : lsmall) nm = sgm.n > nm ? sgm.n : nm;
        const size_t sm2 = (size_t)next_pow2(nm) * (sizeof(TK) + 4);
        hipLaunchKernelGGL(small_sort_kernel<TK>, dim3((unsigned)lsmall.size(), T), dim3(256), sm2,
                           st, bufA.p, F, N, P, L, streamed - 1, dsegs.p, (const int32_t*)nullptr,
                           f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count);
      }
      if (!lbig.empty()) {
        RPT_TRY(gsort<TK>(ctx, bufA.p, bufB.p, N, P, L, streamed - 1, lbig, dglist, nullptr));
        RPT_TRY(upload(lbig_copy, dsegs));
        hipLaunchKernelGGL(copy_segs_kernel, dim3((unsigned)lbig_copy.size(), T), dim3(256), 0, st,
                           bufA.p, F, N, dsegs.p);
      }
      RPT_HIP(hipGetLastError());
      // node_of & co go back to the allocator here; it hands them out again only after the
      // stream has been synchronised
    }
  }
  HT("stream to_perm + free");
  if (stream_unchecked) {
    // only wsub_kernel honours the abort flag: anything else waits for it
    bool all_wave = !ctx->opt.no_wsub;
    for (const PNode& pn : pending[(size_t)streamed]) all_wave = all_wave && pn.seg.n <= kWCap;
    if (!all_wave) {
      RPT_HIP(ctx_sync(ctx));
      bool aborted = false;
      RPT_TRY(stream_aborted(&aborted));
      if (aborted) return build_forest_t<TK>(ctx, ds, f, mode, true);
    }
  }
  if (streamed == 0) {
    hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, st, bufA.p, N, T);
    if (!splits[0].empty()) pending[0].push_back(PNode{splits[0][0], 0});
  }
  // descendants of a node `depth` levels below it that are still split nodes
  std::function<void(const Seg&, int, int, int, std::vector<Seg>&)> descend =
      [&](const Seg& sgm, int level, int depth, int want, std::vector<Seg>& out) {
        if (is_leaf(level, sgm.n, L, f->min_leaf)) return;
        if (depth == want) {
          out.push_back(sgm);
          return;
        }
        const int nh = sgm.n / 2;
        descend(Seg{sgm.off, nh, 2 * sgm.heap + 1}, level + 1, depth + 1, want, out);
        descend(Seg{sgm.off + nh, sgm.n - nh, 2 * sgm.heap + 2}, level + 1, depth + 1, want, out);
      };

  DevBuf<double> wgeo;                 // wsort_kernel's packed images: geometry and words
  DevBuf<unsigned long long> wpacked;
  DevBuf<unsigned long long> cpacked;  // csub_kernel's packed codes
  DevBuf<unsigned int> csflags;
  DevBuf<Seg> dsegs_cs;
  // wsub_kernel launches whose overflow flags have not been looked at yet
  struct Deferred {
    int level = 0, b = 0;
    std::vector<Seg> nodes;
    // [0] count, [1 + i] flag of node i: pivot bins larger than the wave's pool (heavy ties):
    // those nodes are run again by subtree_kernel
    DevBuf<unsigned int> flags;
  };
  std::list<Deferred> deferred;

  for (int level = 0; level < Lused; ++level) {
    for (int b = 0; b < 2; ++b) {
      std::vector<Seg> small, big, mid;
      // nodes between the wave kernel's size and kCsCap whose next r levels have codes and end in
      // split nodes: csub_kernel
      int cs_r = 0;
      if (Cd && !ctx->opt.no_csub) {
        int nmax = 0, nmin = 0x7fffffff;
        for (const PNode& pn : pending[(size_t)level])
          if (pn.buf == b && pn.seg.n > kWCap && pn.seg.n <= kCsCap) {
            nmax = pn.seg.n > nmax ? pn.seg.n : nmax;
            nmin = pn.seg.n < nmin ? pn.seg.n : nmin;
          }
        if (nmax > 0) {
          int r = 0;
          while (nmax > kWCap) {
            nmax -= nmax / 2;
            nmin /= 2;
            ++r;
          }
          if (r <= kCsMaxR && level + r <= Lcodes && !is_leaf(level + r, nmin, L, f->min_leaf)) cs_r = r;
        }
      }
      for (const PNode& pn : pending[(size_t)level])
        if (pn.buf == b) {
          if (cs_r > 0 && pn.seg.n > kWCap && pn.seg.n <= kCsCap) mid.push_back(pn.seg);
          else (pn.seg.n > kSmallCap ? big : small).push_back(pn.seg);
        }
      if (small.empty() && big.empty() && mid.empty()) continue;
      ProfScope ps(ctx, RPT_PROF_SPLIT);
      int32_t* cur = bufs[b];
      int32_t* nxt = bufs[1 - b];
      if (!mid.empty()) {
        const unsigned S = (unsigned)mid.size();
        RPT_TRY(upload(mid, dsegs_cs));
        RPT_TRY(csflags.ensure((size_t)S + 1));
        RPT_HIP(hipMemsetAsync(csflags.p, 0, ((size_t)S + 1) * 4, st));
        RPT_TRY(cpacked.ensure((size_t)T * N));
        const unsigned pb = (unsigned)std::min<int64_t>((N + 1023) / 1024, (int64_t)ctx->n_cu * 8);
        hipLaunchKernelGGL(cpack_kernel, dim3(pb, T), dim3(256), 0, st, (const uint16_t*)Cd, N, L, level,
                           cs_r, cpacked.p);
        int csmax = 0;
        for (const Seg& sgm : mid) csmax = sgm.n > csmax ? sgm.n : csmax;
#define RPT_CSUB(E)                                                                              \
  hipLaunchKernelGGL((csub_kernel<TK, E>), dim3(S, T), dim3(kCsThreads), 0, st, cur, nxt, N, P,  \
                     (const unsigned long long*)cpacked.p, L, level, cs_r, dsegs_cs.p, f->thr.p, \
                     f->mglo.p, f->mghi.p, f->nodes, tie_count, csflags.p)
        // point slots per thread: fewer slots, fewer registers, more workgroups per CU
        if (csmax <= 8 * kCsThreads) RPT_CSUB(8);
        else if (csmax <= 12 * kCsThreads) RPT_CSUB(12);
        else RPT_CSUB(16);
#undef RPT_CSUB
        RPT_HIP(hipGetLastError());
        unsigned int nfl = 0;
        RPT_HIP(hipMemcpyAsync(&nfl, csflags.p, 4, hipMemcpyDeviceToHost, st));
        RPT_HIP(ctx_sync(ctx));
        std::vector<unsigned int> fl;
        if (nfl) {  // pivot codes shared by more points than the pool holds: the general kernels
          fl.resize((size_t)S + 1);
          RPT_HIP(hipMemcpy(fl.data(), csflags.p, fl.size() * 4, hipMemcpyDeviceToHost));
          for (unsigned i = 0; i < S; ++i) {
            ctx->last_csub_redo += fl[1 + i] != 0;
            ctx->last_csub_bad += (fl[1 + i] & 255u) == 2u;
          }
        }
        for (unsigned i = 0; i < S; ++i) {
          if (nfl && fl[1 + i]) {
            (mid[i].n > kSmallCap ? big : small).push_back(mid[i]);
            continue;
          }
          std::vector<Seg> rest;
          descend(mid[i], level, 0, cs_r, rest);
          for (const Seg& sgm : rest) pending[(size_t)level + cs_r].push_back(PNode{sgm, 1 - b});
        }
      }

      // small nodes: n <= kWCap -> one wave per subtree; kWCap < n <= kSmallCap -> one block
      std::vector<Seg> wsmall, bsmall;
      const bool no_wsub = ctx->opt.no_wsub != 0;
      for (const Seg& sgm : small) ((sgm.n <= kWCap && !no_wsub) ? wsmall : bsmall).push_back(sgm);
      if (!wsmall.empty()) {
        RPT_TRY(upload(wsmall, dsegs));
        const unsigned S = (unsigned)wsmall.size();
        deferred.emplace_back();
        Deferred& df = deferred.back();
        df.level = level;
        df.b = b;
        RPT_TRY(df.flags.alloc((size_t)S + 1));
        RPT_HIP(hipMemsetAsync(df.flags.p, 0, ((size_t)S + 1) * 4, st));
        if (!ctx->opt.no_wsort) {  // one bitonic sort of the node's sort keys per level
          // packed images (one gather per point instead of one per point and level) when the
          // launch covers at most three levels and most of the point set
          const int nl = std::min(L - level, kWRmax);
          int64_t cover = 0;
          for (const Seg& sgm : wsmall) cover += sgm.n;
          const bool pk = !ctx->opt.no_wpack && nl <= kPkLevels && cover * 2 >= N && N * T >= (1 << 16);
          if (pk) {
            RPT_TRY(wgeo.ensure((size_t)T * kPkLevels * 2));
            RPT_TRY(wpacked.ensure((size_t)T * N));
            hipLaunchKernelGGL(wgeo_kernel<TK>, dim3((unsigned)nl, T), dim3(256), 0, st, P, N, L, level,
                               nl, wgeo.p);
            const unsigned pb = (unsigned)std::min<int64_t>((N + 1023) / 1024, (int64_t)ctx->n_cu * 8);
            hipLaunchKernelGGL(wpack_kernel<TK>, dim3(pb, T), dim3(256), 0, st, P, N, L, level, nl,
                               wgeo.p, wpacked.p);
            hipLaunchKernelGGL((wsort_kernel<TK, true>), dim3((unsigned)(((int64_t)S * T + 3) / 4)),
                               dim3(256), 0, st, cur, nxt, F, N, P, L, T, level, f->min_leaf, dsegs.p,
                               (int)S, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count,
                               df.flags.p + 1, df.flags.p, (const unsigned int*)sflags.p,
                               (const unsigned long long*)wpacked.p);
          } else {
            hipLaunchKernelGGL((wsort_kernel<TK, false>), dim3((unsigned)(((int64_t)S * T + 3) / 4)),
                               dim3(256), 0, st, cur, nxt, F, N, P, L, T, level, f->min_leaf, dsegs.p,
                               (int)S, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count,
                               df.flags.p + 1, df.flags.p, (const unsigned int*)sflags.p,
                               (const unsigned long long*)nullptr);
          }
        }
        else                     // round 1/2: histogram select in the wave's registers
          hipLaunchKernelGGL(wsub_kernel<TK>, dim3((unsigned)(((int64_t)S * T + 3) / 4)), dim3(256), 0,
                             st, cur, nxt, F, N, P, L, T, level, f->min_leaf, dsegs.p, (int)S,
                             f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count, df.flags.p + 1,
                             df.flags.p, (const unsigned int*)sflags.p, dbgbuf);
        // nodes the wave kernel left active (deeper than kWRmax levels) stay pending
        std::vector<Seg> rest;
        for (const Seg& sgm : wsmall) descend(sgm, level, 0, kWRmax, rest);
        // The overflow flags matter on the host only when something consumes this launch's
        // active nodes; otherwise they are looked at once, at the end of the build.
        const bool defer = rest.empty();
        if (defer) {
          df.nodes = wsmall;
        } else {
          unsigned int novf = 0;
          RPT_HIP(hipMemcpyAsync(&novf, df.flags.p, 4, hipMemcpyDeviceToHost, st));
          RPT_HIP(ctx_sync(ctx));
          bool aborted = false;
          RPT_TRY(stream_aborted(&aborted));
          if (aborted) return build_forest_t<TK>(ctx, ds, f, mode, true);
          if (novf) {  // run again by the block-level kernel: drop their descendants
            std::vector<unsigned int> fl((size_t)S + 1);
            RPT_HIP(hipMemcpy(fl.data(), df.flags.p, fl.size() * 4, hipMemcpyDeviceToHost));
            std::vector<Seg> redo;
            for (unsigned i = 0; i < S; ++i)
              if (fl[1 + i]) redo.push_back(wsmall[i]);
            auto drop = [&](const Seg& x) {
              for (const Seg& r : redo)
                if (x.off >= r.off && x.off < r.off + r.n) return true;
              return false;
            };
            rest.erase(std::remove_if(rest.begin(), rest.end(), drop), rest.end());
            for (const Seg& r : redo) bsmall.push_back(r);
          }
          deferred.pop_back();
          for (const Seg& sgm : rest) pending[(size_t)level + kWRmax].push_back(PNode{sgm, 1 - b});
        }
      }
      if (!bsmall.empty()) {
        // whole subtrees in LDS.  Nodes above the wave kernel's size run only the levels that
        // bring them under it (the wave kernel is the faster one); nodes that kernel handed
        // back (pool overflow: heavy ties) run kRmax levels, which finishes their subtrees.
        std::vector<Seg> grp[2];  // [0] above kWCap, [1] handed back
        for (const Seg& sgm : bsmall) grp[(sgm.n <= kWCap && !no_wsub) ? 1 : 0].push_back(sgm);
        for (int gi = 0; gi < 2; ++gi) {
          if (grp[gi].empty()) continue;
          int r = kRmax;
          if (gi == 0 && !no_wsub) {
            int nmax = 0;
            for (const Seg& sgm : grp[gi]) nmax = sgm.n > nmax ? sgm.n : nmax;
            r = 0;
            while (nmax > kWCap) {  // the larger child of n points has n - n/2
              nmax -= nmax / 2;
              ++r;
            }
          }
          RPT_TRY(upload(grp[gi], gi == 0 ? dsegs : dsegs2));
          hipLaunchKernelGGL(subtree_kernel<TK>, dim3((unsigned)grp[gi].size(), T),
                             dim3(kSubThreads), 0, st, cur, nxt, F, N, P, L, level, f->min_leaf,
                             (gi == 0 ? dsegs : dsegs2).p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes,
                             tie_count, dbgbuf, r);
          if (level + r < Lused) {
            std::vector<Seg> rest;
            for (const Seg& sgm : grp[gi]) descend(sgm, level, 0, r, rest);
            for (const Seg& sgm : rest) pending[(size_t)level + r].push_back(PNode{sgm, 1 - b});
          }
        }
      }
      if (!big.empty()) {
        RPT_TRY(upload(big, dsegs2));
        const unsigned S = (unsigned)big.size();
        int nmax = 0;
        for (const Seg& sgm : big) nmax = sgm.n > nmax ? sgm.n : nmax;
        const unsigned chunks = (unsigned)((nmax + kChunk - 1) / kChunk);
        hipLaunchKernelGGL(sample_kernel<TK>, dim3(S, T), dim3(256), 0, st, cur, N, P, L, level,
                           dsegs2.p, (int)S, bins.p);
        hipLaunchKernelGGL(hist_kernel<TK>, dim3(chunks, S, T), dim3(256), 0, st, cur, N, P, L,
                           level, dsegs2.p, (int)S, bins.p, Kst.p, hist.p);
        hipLaunchKernelGGL(pick_kernel, dim3(S, T), dim3(256), 0, st, dsegs2.p, (int)S, hist.p,
                           aux.p);
        hipLaunchKernelGGL(scatter_kernel<TK>, dim3(chunks, S, T), dim3(256), 0, st, cur, nxt, N,
                           dsegs2.p, (int)S, bins.p, Kst.p, aux.p);
        RPT_HIP(hipMemsetAsync(bigflags.p, 0, (size_t)T * S * 4, st));
        RPT_HIP(hipMemsetAsync(big_count, 0, 4, st));
        const size_t smem = (size_t)kSmallCap * (sizeof(TK) + 4);
        hipLaunchKernelGGL(mid_kernel<TK>, dim3(S, T), dim3(256), smem, st, nxt, N, P, L, level,
                           dsegs2.p, (int)S, aux.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes,
                           tie_count, bigflags.p, big_count);
        RPT_HIP(hipGetLastError());
        unsigned int nbig = 0;
        RPT_HIP(hipMemcpyAsync(&nbig, big_count, 4, hipMemcpyDeviceToHost, st));
        RPT_HIP(stream_sync(st));
        if (nbig) {  // rare: pivot bins larger than LDS -> HBM merge sort of those bins
          std::vector<unsigned int> flags((size_t)T * S);
          std::vector<NodeAux> haux((size_t)T * S);
          RPT_HIP(hipMemcpy(flags.data(), bigflags.p, flags.size() * 4, hipMemcpyDeviceToHost));
          RPT_HIP(hipMemcpy(haux.data(), aux.p, haux.size() * sizeof(NodeAux),
                            hipMemcpyDeviceToHost));
          std::vector<GSeg> gl;
          std::vector<int> gidx;
          for (int t = 0; t < T; ++t)
            for (unsigned sI = 0; sI < S; ++sI)
              if (flags[(size_t)t * S + sI]) {
                const NodeAux& a = haux[(size_t)t * S + sI];
                const Seg& sgm = big[sI];
                const int n = sgm.n, nh = n / 2;
                const int il = nh > 0 ? nh - 1 : 0, ih = nh + 1 < n ? nh + 1 : n - 1;
                GSeg g;
                g.off = (int64_t)t * N + sgm.off + a.cL;
                g.n = a.cMid;
                g.t = t;
                g.heap = sgm.heap;
                g.nh_rel = nh - a.cL;
                g.il_rel = il >= a.cL ? il - a.cL : -1;
                g.ih_rel = ih < a.cL + a.cMid ? ih - a.cL : -1;
                gl.push_back(g);
                gidx.push_back(t * (int)S + (int)sI);
              }
          f->big_mid_nodes += (int64_t)gl.size();
          RPT_TRY(gsort<TK>(ctx, nxt, cur /*scratch: cur is dead for these nodes*/, N, P, L,
                            level, gl, dglist, nullptr));
          DevBuf<int> didx;
          RPT_TRY(didx.alloc(gidx.size()));
          RPT_HIP(hipMemcpy(didx.p, gidx.data(), gidx.size() * 4, hipMemcpyHostToDevice));
          hipLaunchKernelGGL(gsort_emit_kernel<TK>, dim3((unsigned)((gl.size() + 63) / 64)),
                             dim3(64), 0, st, nxt, N, P, L, level, dglist.p, (int)gl.size(),
                             aux.p, didx.p, f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count);
          RPT_HIP(stream_sync(st));
        }
        // children: leaves get their final (sorted) order now, the rest stays pending
        std::vector<Seg> lsmall;
        std::vector<GSeg> lbig;
        std::vector<Seg> lbig_copy;
        for (const Seg& sgm : big) {
          const int nh = sgm.n / 2;
          const Seg ch[2] = {Seg{sgm.off, nh, 2 * sgm.heap + 1},
                             Seg{sgm.off + nh, sgm.n - nh, 2 * sgm.heap + 2}};
          for (const Seg& c : ch) {
            if (!is_leaf(level + 1, c.n, L, f->min_leaf)) {
              pending[(size_t)level + 1].push_back(PNode{c, 1 - b});
            } else if (c.n <= kSmallCap) {
              lsmall.push_back(Seg{c.off, c.n, -1});
            } else {
              lbig_copy.push_back(Seg{c.off, c.n, -1});
              for (int t = 0; t < T; ++t)
                lbig.push_back(GSeg{(int64_t)t * N + c.off, c.n, t, -1, -1, -1, -1});
            }
          }
        }
        if (!lsmall.empty()) {
          RPT_TRY(upload(lsmall, dsegs));
          int nm = 0;
          for (const Seg& sgm : lsmall) nm = sgm.n > nm ? sgm.n : nm;
          const size_t sm2 = (size_t)next_pow2(nm) * (sizeof(TK) + 4);
          hipLaunchKernelGGL(small_sort_kernel<TK>, dim3((unsigned)lsmall.size(), T), dim3(256),
                             sm2, st, nxt, F, N, P, L, level, dsegs.p, (const int32_t*)nullptr,
                             f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count);
        }
        if (!lbig.empty()) {
          RPT_TRY(gsort<TK>(ctx, nxt, cur, N, P, L, level, lbig, dglist, nullptr));
          RPT_TRY(upload(lbig_copy, dsegs));
          hipLaunchKernelGGL(copy_segs_kernel, dim3((unsigned)lbig_copy.size(), T), dim3(256), 0,
                             st, nxt, F, N, dsegs.p);
        }
      }
      RPT_HIP(hipGetLastError());
    }
  }
  HT("pending loop");
  if (dbgbuf) {
    unsigned long long hs[256];
    RPT_HIP(stream_sync(st));
    RPT_HIP(hipMemcpy(hs, dbgbuf, sizeof(hs), hipMemcpyDeviceToHost));
    for (int i = 1; i < 256 && hs[i]; ++i) fprintf(stderr, "stamp %d: +%llu\n", i, hs[i] - hs[i - 1]);
  }
  // ---- the one synchronisation point of the common path ----
  {
    const size_t nd = deferred.size();
    unsigned long long* hres = reinterpret_cast<unsigned long long*>(pin_alloc(ctx, (2 + nd) * 8));
    std::vector<unsigned long long> hfallback;
    if (!hres) {
      hfallback.resize(2 + nd);
      hres = hfallback.data();
    }
    for (size_t i = 0; i < 2 + nd; ++i) hres[i] = 0;
    RPT_HIP(hipMemcpyAsync(&hres[0], tie_count, 8, hipMemcpyDeviceToHost, st));
    RPT_HIP(hipMemcpyAsync(&hres[1], sflags.p, 4, hipMemcpyDeviceToHost, st));
    size_t k = 2;
    for (Deferred& df : deferred)
      RPT_HIP(hipMemcpyAsync(&hres[k++], df.flags.p, 4, hipMemcpyDeviceToHost, st));
    RPT_HIP(stream_sync(st));
    const unsigned long long ties0 = hres[0];
    const bool aborted = stream_unchecked && (unsigned int)hres[1] != 0;
    std::vector<unsigned int> novf(nd);
    for (size_t i = 0; i < nd; ++i) novf[i] = (unsigned int)hres[2 + i];
    ctx->pin_off = 0;
    if (aborted) return build_forest_t<TK>(ctx, ds, f, mode, true);
    f->tie_nodes = (int64_t)ties0;
    bool redone = false;
    k = 0;
    for (Deferred& df : deferred) {
      if (novf[k++] == 0) continue;
      // pivot bins larger than a wave's pool: those nodes again with the block-level kernel
      // (a deferred launch covers the rest of its subtrees: kRmax > kWRmax)
      const size_t S = df.nodes.size();
      std::vector<unsigned int> fl(S + 1);
      RPT_HIP(hipMemcpy(fl.data(), df.flags.p, fl.size() * 4, hipMemcpyDeviceToHost));
      std::vector<Seg> redo;
      for (size_t i = 0; i < S; ++i)
        if (fl[1 + i]) redo.push_back(df.nodes[i]);
      if (redo.empty()) continue;
      ProfScope ps(ctx, RPT_PROF_SPLIT);
      RPT_TRY(upload(redo, dsegs));
      hipLaunchKernelGGL(subtree_kernel<TK>, dim3((unsigned)redo.size(), T), dim3(kSubThreads), 0,
                         st, bufs[df.b], bufs[1 - df.b], F, N, P, L, df.level, f->min_leaf, dsegs.p,
                         f->thr.p, f->mglo.p, f->mghi.p, f->nodes, tie_count, dbgbuf, kRmax);
      RPT_HIP(hipGetLastError());
      redone = true;
    }
    if (redone) {
      unsigned long long ties = 0;
      RPT_HIP(hipMemcpyAsync(&ties, tie_count, 8, hipMemcpyDeviceToHost, st));
      RPT_HIP(ctx_sync(ctx));
      f->tie_nodes = (int64_t)ties;
    }
  }
  return RPT_OK;
}

// ---------------------------------------------------------------------------------------
// Streaming insert: `forest` / `tree` of Conduit.hs:58-121 = chunkedAccum
// (Conduit.hs:169-176: C.chunksOf n, the last chunk may be shorter) folding `insert`
// (Internal.hs:258-297) over the chunks, every tree with the same chunks (insertMulti :245-255).
//   chunk part reaching a Bin (:272-283)  split at ITS OWN median (stable sort by the level's
//       projection, cut at n div 2); thr' = (thr0 + thr) / 2, margin' = (max lows, min highs); an
//       EMPTY part answers `Tip () mempty` (:277): the subtree built so far is dropped
//   chunk part reaching a Tip (:285-297)  xs' = xs <> xs0 (the new points FIRST); stays a Tip at
//       maxDepth or with <= minLeaf points, else it is split and both halves start from empty Tips
// Every size in this recursion is n div 2 of a known size, so WHICH nodes are Bins / Tips, every
// part's and every Tip's length — the whole evolution of the tree's shape — is a function of
// (N, chunk, minLeaf, maxDepth) alone and identical for all trees: the host plans a chunk (which
// segments are sorted at which level, what is copied where), the device does the data-dependent
// part for all trees at once:
//   per level of a chunk   gather the level's segments into a working row (chunk parts from the
//                          previous level's sorted row, Tip contents from the Tip store), stable
//                          segmented sort by P[t][level][id] with the POSITION in the segment as
//                          tie-break (small_sort_kernel / gsort with tb = position), fold
//                          thr / margins into the node arrays (set or average)
//   per chunk              the Tip store is rewritten in heap order (untouched Tips copied, the
//                          others = new part ++ old contents)
// The fold over chunks is sequential by definition (chunk c's medians move the thresholds chunk
// c + 1 is compared with... in fact only the node arrays and the Tip store carry over); the cost is
// ~5 launches per (chunk, level).  Projections: all T x L columns for all points up front, the
// batch kernels.
// ---------------------------------------------------------------------------------------
struct XCopy {
  int32_t src;  // 0 = Tip store A, 1 = previous working row, 2 = iota (value soff + i)
  int32_t dst;  // 0 = current working row, 1 = new Tip store B
  int64_t soff, doff, len;
};

__global__ __launch_bounds__(256) void xs_copy_kernel(const XCopy* __restrict__ descs,
                                                      const int32_t* __restrict__ A,
                                                      const int32_t* __restrict__ Wp,
                                                      int32_t* __restrict__ Wc, int32_t* __restrict__ B,
                                                      int64_t N) {
  const XCopy c = descs[blockIdx.x];
  const int64_t row = (int64_t)blockIdx.y * N;
  const int32_t* s = (c.src == 0 ? A : Wp) + row + c.soff;
  int32_t* d = (c.dst == 0 ? Wc : B) + row + c.doff;
  for (int64_t i = threadIdx.x; i < c.len; i += blockDim.x) d[i] = c.src == 2 ? (int32_t)(c.soff + i) : s[i];
}

// tb[t][id] = position of the point inside its segment (the stable sort's tie-break)
__global__ __launch_bounds__(256) void xs_pos_kernel(const Seg* __restrict__ segs,
                                                     const int32_t* __restrict__ W,
                                                     int32_t* __restrict__ tb, int64_t N) {
  const Seg sg = segs[blockIdx.x];
  const int64_t row = (int64_t)blockIdx.y * N;
  for (int i = threadIdx.x; i < sg.n; i += blockDim.x) tb[row + W[row + sg.off + i]] = i;
}

struct XNode {
  int32_t heap;
  int32_t avg;  // 0: a new Bin (Internal.hs:292), 1: fold into the Bin that is there (:280-283)
};

__global__ void xs_fold_kernel(const XNode* __restrict__ xn, int S, const double* __restrict__ tthr,
                               const double* __restrict__ tlo, const double* __restrict__ thi,
                               double* thr, double* mglo, double* mghi, int64_t slots) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int64_t t = blockIdx.y, h = t * slots + xn[s].heap;
  const double a = tthr[t * S + s], lo = tlo[t * S + s], hi = thi[t * S + s];
  if (!xn[s].avg) {
    thr[h] = a;
    mglo[h] = lo;
    mghi[h] = hi;
  } else {
    thr[h] = (thr[h] + a) / 2;               // :281 thr' = (thr0 + thr) / 2
    mglo[h] = mglo[h] >= lo ? mglo[h] : lo;  // :280 margin0 <> margin = (Max, Min), :86-87
    mghi[h] = mghi[h] <= hi ? mghi[h] : hi;
  }
}

// slots that are not Bins carry no numbers
__global__ void xs_mask_kernel(const int8_t* __restrict__ kind, int64_t slots, int T, double* thr,
                               double* mglo, double* mghi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots * T) return;
  if (kind[i % slots] != 1) {
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    thr[i] = mglo[i] = mghi[i] = nan;
  }
}

template <class TK>
int32_t stream_build_t(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int64_t chunk,
                       int32_t mode) {
  const int64_t N = f->n;
  const int T = f->T, L = f->L, min_leaf = f->min_leaf;
  const int64_t slots = f->nodes;  // 2^(L+1) - 1
  hipStream_t st = ctx->stream;
  f->mode = mode;
  f->xtopo = true;
  std::vector<int8_t>& kind = f->xkind_h;
  kind.assign((size_t)slots, 0);
  kind[0] = 2;  // `Tip () mempty` (Conduit.hs:160)
  std::vector<int64_t> cnt((size_t)slots, 0), aoff((size_t)slots, 0);
  f->dropped = 0;
  {
    const int64_t tot = (int64_t)T * slots;
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->thr.p, tot,
                       std::numeric_limits<double>::quiet_NaN());
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->mglo.p, tot,
                       std::numeric_limits<double>::quiet_NaN());
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->mghi.p, tot,
                       std::numeric_limits<double>::quiet_NaN());
  }
  if (N > 0 && L > 0) {
    RPT_TRY(f->proj.alloc((size_t)T * L * N * sizeof(TK)));
    RPT_TRY(project_columns(ctx, ds, f->R.p, T * L, mode, f->proj.p));
  }
  const TK* P = reinterpret_cast<const TK*>(f->proj.p);
  DevBuf<int32_t> bufA, bufB, W0, W1, tb, tmp;
  const size_t rows = (size_t)T * (size_t)(N > 0 ? N : 1);
  RPT_TRY(bufB.alloc(rows));
  RPT_TRY(W0.alloc(rows));
  RPT_TRY(W1.alloc(rows));
  RPT_TRY(tb.alloc(rows));
  RPT_TRY(tmp.alloc(rows));
  int32_t* A = f->perm.p;  // Tip store, ping-pong with bufB; ends up in f->perm
  int32_t* B = bufB.p;
  DevBuf<XCopy> dcopy;
  DevBuf<Seg> dsegs;
  DevBuf<GSeg> dglist;
  DevBuf<XNode> dxn;
  DevBuf<double> tthr, tlo, thi;
  DevBuf<unsigned long long> tie;
  RPT_TRY(tie.alloc(1));
  RPT_HIP(hipMemsetAsync(tie.p, 0, 8, st));

  auto subtree_points = [&](int64_t h) {
    int64_t tot = 0;
    std::vector<int64_t> stk{h};
    while (!stk.empty()) {
      const int64_t x = stk.back();
      stk.pop_back();
      if (x >= slots) continue;
      if (kind[(size_t)x] == 2) tot += cnt[(size_t)x];
      else if (kind[(size_t)x] == 1) {
        stk.push_back(2 * x + 1);
        stk.push_back(2 * x + 2);
      }
    }
    return tot;
  };
  auto clear_subtree = [&](int64_t h) {
    std::vector<int64_t> stk{h};
    while (!stk.empty()) {
      const int64_t x = stk.back();
      stk.pop_back();
      if (x >= slots) continue;
      if (kind[(size_t)x] == 1) {
        stk.push_back(2 * x + 1);
        stk.push_back(2 * x + 2);
      }
      kind[(size_t)x] = 0;
      cnt[(size_t)x] = 0;
    }
  };

  struct Item {
    int64_t h, n, soff;
    int src;  // where the part lies: 1 = the previous level's working row, 2 = iota
  };
  struct LevelPlan {
    std::vector<XCopy> copies;
    std::vector<Seg> segs;     // heap = index into xn
    std::vector<XNode> xn;
  };
  struct Fin {  // a Tip that ends the chunk as `part ++ old contents`
    int64_t h, part_n, old_off, old_n;
    size_t level, copy_part, copy_old;  // indices of its two copies in plans[level].copies
  };

  for (int64_t c0 = 0; c0 < N; c0 += chunk) {  // C.chunksOf chunk .| C.foldl
    const int64_t nc = std::min(chunk, N - c0);
    // ---- plan the chunk (host; sizes only) ----
    std::vector<LevelPlan> plans;
    std::vector<Fin> fins;
    std::vector<char> touched((size_t)slots, 0);
    const std::vector<int64_t> aoff_old = aoff, cnt_old = cnt;
    const std::vector<int8_t> kind_old = kind;
    std::vector<Item> frontier{Item{0, nc, c0, 2}};
    for (int level = 0; !frontier.empty(); ++level) {
      plans.emplace_back();
      LevelPlan& pl = plans.back();
      std::vector<Item> next;
      int64_t woff = 0;
      for (const Item& it : frontier) {
        const size_t h = (size_t)it.h;
        if (kind[h] == 1) {  // Bin (:272)
          if (level >= L) continue;  // :273-274
          if (it.n < 1) {            // :277 Nothing -> Tip () mempty: the subtree is lost
            f->dropped += subtree_points(it.h);
            clear_subtree(it.h);
            kind[h] = 2;
            touched[h] = 1;
            continue;
          }
          pl.copies.push_back(XCopy{it.src, 0, it.soff, woff, it.n});
          pl.segs.push_back(Seg{woff, (int32_t)it.n, (int32_t)pl.xn.size()});
          pl.xn.push_back(XNode{(int32_t)it.h, 1});
          const int64_t nh = it.n / 2;
          next.push_back(Item{2 * it.h + 1, nh, woff, 1});
          next.push_back(Item{2 * it.h + 2, it.n - nh, woff + nh, 1});
          woff += it.n;
        } else {  // Tip, or never touched = the `z` of :268
          kind[h] = 2;
          const int64_t total = it.n + cnt[h];
          if (level >= L || total <= (int64_t)min_leaf) {  // :287-288 Tip () xs'
            Fin fn{it.h, it.n, 0, 0, (size_t)level, pl.copies.size(), pl.copies.size() + 1};
            if (kind_old[h] == 2 && !touched[h]) {
              fn.old_off = aoff_old[h];
              fn.old_n = cnt_old[h];
            }
            pl.copies.push_back(XCopy{it.src, 1, it.soff, 0, it.n});       // doff: once B is laid out
            pl.copies.push_back(XCopy{0, 1, fn.old_off, 0, fn.old_n});
            fins.push_back(fn);
            cnt[h] = total;
            touched[h] = 1;
          } else {  // :290-297 a new Bin over xs' = xs <> xs0, children from empty Tips
            const int64_t old_n = (kind_old[h] == 2 && !touched[h]) ? cnt_old[h] : 0;
            pl.copies.push_back(XCopy{it.src, 0, it.soff, woff, it.n});
            if (old_n) pl.copies.push_back(XCopy{0, 0, aoff_old[h], woff + it.n, old_n});
            pl.segs.push_back(Seg{woff, (int32_t)total, (int32_t)pl.xn.size()});
            pl.xn.push_back(XNode{(int32_t)it.h, 0});
            kind[h] = 1;
            cnt[h] = 0;
            touched[h] = 1;
            const int64_t nh = total / 2;
            for (int64_t ch = 2 * it.h + 1; ch <= 2 * it.h + 2; ++ch) {
              kind[(size_t)ch] = 2;
              cnt[(size_t)ch] = 0;
              touched[(size_t)ch] = 1;
            }
            next.push_back(Item{2 * it.h + 1, nh, woff, 1});
            next.push_back(Item{2 * it.h + 2, total - nh, woff + nh, 1});
            woff += total;
          }
        }
      }
      frontier.swap(next);
    }
    // the new Tip store, heap order
    {
      int64_t w = 0;
      for (int64_t h = 0; h < slots; ++h)
        if (kind[(size_t)h] == 2) {
          aoff[(size_t)h] = w;
          w += cnt[(size_t)h];
        } else {
          aoff[(size_t)h] = w;
        }
    }
    for (const Fin& fn : fins) {
      if (kind[(size_t)fn.h] != 2) continue;  // (dropped again later in the same chunk: cannot happen)
      LevelPlan& pl = plans[fn.level];
      pl.copies[fn.copy_part].doff = aoff[(size_t)fn.h];
      pl.copies[fn.copy_old].doff = aoff[(size_t)fn.h] + fn.part_n;
    }
    // Tips this chunk did not reach keep their contents
    if (plans.empty()) plans.emplace_back();
    for (int64_t h = 0; h < slots; ++h)
      if (kind[(size_t)h] == 2 && !touched[(size_t)h] && cnt[(size_t)h] > 0)
        plans[0].copies.push_back(XCopy{0, 1, aoff_old[(size_t)h], aoff[(size_t)h], cnt[(size_t)h]});

    // ---- run it (all trees at once) ----
    int32_t* Wp = W0.p;
    int32_t* Wc = W1.p;
    for (size_t level = 0; level < plans.size(); ++level) {
      LevelPlan& pl = plans[level];
      {
        std::vector<XCopy> cp;
        for (const XCopy& c : pl.copies)
          if (c.len > 0) cp.push_back(c);
        if (!cp.empty()) {
          RPT_TRY(dcopy.ensure(cp.size()));
          RPT_TRY(upload_async(ctx, dcopy.p, cp.data(), cp.size() * sizeof(XCopy)));
          hipLaunchKernelGGL(xs_copy_kernel, dim3((unsigned)cp.size(), T), dim3(256), 0, st, dcopy.p,
                             (const int32_t*)A, (const int32_t*)Wp, Wc, B, N);
        }
      }
      const int S = (int)pl.segs.size();
      if (S > 0) {
        const TK* Pl = P + (int64_t)level * N;  // (t, level) is column t * L of this base
        RPT_TRY(dsegs.ensure((size_t)S));
        RPT_TRY(upload_async(ctx, dsegs.p, pl.segs.data(), (size_t)S * sizeof(Seg)));
        RPT_TRY(dxn.ensure((size_t)S));
        RPT_TRY(upload_async(ctx, dxn.p, pl.xn.data(), (size_t)S * sizeof(XNode)));
        RPT_TRY(tthr.ensure((size_t)S * T));
        RPT_TRY(tlo.ensure((size_t)S * T));
        RPT_TRY(thi.ensure((size_t)S * T));
        hipLaunchKernelGGL(xs_pos_kernel, dim3((unsigned)S, T), dim3(256), 0, st, dsegs.p,
                           (const int32_t*)Wc, tb.p, N);
        std::vector<Seg> small;
        std::vector<GSeg> big;
        for (const Seg& sg : pl.segs) {
          if (sg.n <= kSmallCap) {
            small.push_back(sg);
          } else {
            const int nn = sg.n, nh = nn / 2;
            for (int t = 0; t < T; ++t)
              big.push_back(GSeg{(int64_t)t * N + sg.off, nn, t, sg.heap, nh, nh > 0 ? nh - 1 : 0,
                                 nh + 1 < nn ? nh + 1 : nn - 1});
          }
        }
        if (!small.empty()) {
          DevBuf<Seg>& ds2 = dsegs;  // the small ones, re-uploaded behind the position pass
          int nmax = 0;
          for (const Seg& sg : small) nmax = sg.n > nmax ? sg.n : nmax;
          if (small.size() != pl.segs.size())
            RPT_TRY(upload_async(ctx, ds2.p, small.data(), small.size() * sizeof(Seg)));
          const size_t smem = (size_t)next_pow2(nmax) * (sizeof(TK) + 4);
          if (smem > 64 * 1024)
            RPT_HIP(hipFuncSetAttribute((const void*)small_sort_kernel<TK>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          hipLaunchKernelGGL(small_sort_kernel<TK>, dim3((unsigned)small.size(), T), dim3(256), smem, st,
                             (const int32_t*)Wc, Wc, N, Pl, L, 0, ds2.p, (const int32_t*)tb.p, tthr.p,
                             tlo.p, thi.p, (int64_t)S, tie.p);
        }
        if (!big.empty()) {
          RPT_TRY(gsort<TK>(ctx, Wc, tmp.p, N, Pl, L, 0, big, dglist, tb.p));
          hipLaunchKernelGGL(gsort_emit_kernel<TK>, dim3((unsigned)((big.size() + 63) / 64)), dim3(64),
                             0, st, (const int32_t*)Wc, N, Pl, L, 0, dglist.p, (int)big.size(),
                             (const NodeAux*)nullptr, (const int*)nullptr, tthr.p, tlo.p, thi.p,
                             (int64_t)S, tie.p);
        }
        hipLaunchKernelGGL(xs_fold_kernel, dim3((unsigned)((S + 255) / 256), T), dim3(256), 0, st,
                           dxn.p, S, tthr.p, tlo.p, thi.p, f->thr.p, f->mglo.p, f->mghi.p, slots);
      }
      RPT_HIP(hipGetLastError());
      std::swap(Wp, Wc);
    }
    std::swap(A, B);
  }
  if (A != f->perm.p && N > 0)
    RPT_HIP(hipMemcpyAsync(f->perm.p, A, (size_t)T * N * 4, hipMemcpyDeviceToDevice, st));
  // the final shape, as rpo_stream_forest_dense lists it
  f->xoff_h.assign((size_t)slots, 0);
  f->xlen_h.assign((size_t)slots, 0);
  {
    int64_t w = 0;
    for (int64_t h = 0; h < slots; ++h) {
      f->xoff_h[(size_t)h] = w;
      if (kind[(size_t)h] == 2) {
        f->xlen_h[(size_t)h] = cnt[(size_t)h];
        w += cnt[(size_t)h];
      }
    }
    f->held = w;
  }
  RPT_TRY(f->xkind.alloc((size_t)slots));
  RPT_TRY(f->xoff.alloc((size_t)slots));
  RPT_TRY(f->xlen.alloc((size_t)slots));
  RPT_HIP(hipMemcpyAsync(f->xkind.p, kind.data(), (size_t)slots, hipMemcpyHostToDevice, st));
  RPT_HIP(hipMemcpyAsync(f->xoff.p, f->xoff_h.data(), (size_t)slots * 8, hipMemcpyHostToDevice, st));
  RPT_HIP(hipMemcpyAsync(f->xlen.p, f->xlen_h.data(), (size_t)slots * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(xs_mask_kernel, dim3((unsigned)((slots * T + 255) / 256)), dim3(256), 0, st,
                     (const int8_t*)f->xkind.p, slots, T, f->thr.p, f->mglo.p, f->mghi.p);
  RPT_HIP(hipGetLastError());
  RPT_HIP(ctx_sync(ctx));  // work buffers and the host vectors behind the copies are released
  return RPT_OK;
}

}  // namespace

int32_t stream_build_forest(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int64_t chunk,
                            int32_t mode) {
  if (f->pdtype == RPT_F64) return stream_build_t<double>(ctx, ds, f, chunk, mode);
  return stream_build_t<float>(ctx, ds, f, chunk, mode);
}

int32_t build_forest(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int32_t mode) {
  if (f->pdtype == RPT_F64) return build_forest_t<double>(ctx, ds, f, mode);
  return build_forest_t<float>(ctx, ds, f, mode);
}

// partitionAtMedian on caller-supplied keys: every segment fully (stably) sorted.
int32_t split_segments(rpt_ctx* ctx, const double* key_host, int64_t n, int32_t* perm_io_host,
                       const int64_t* seg_off, const int64_t* seg_len, int32_t S,
                       double* thr_mg_host) {
  hipStream_t st = ctx->stream;
  std::vector<char> seen((size_t)n, 0);
  std::vector<int32_t> pos((size_t)n, 0);
  for (int32_t s = 0; s < S; ++s) {
    RPT_ARG(seg_off[s] >= 0 && seg_len[s] >= 0 && seg_off[s] + seg_len[s] <= n,
            "segment out of range");
    for (int64_t i = seg_off[s]; i < seg_off[s] + seg_len[s]; ++i) {
      const int32_t id = perm_io_host[i];
      RPT_ARG(id >= 0 && id < n && !seen[(size_t)id], "perm entries must be distinct ids < n");
      seen[(size_t)id] = 1;
      pos[(size_t)id] = (int32_t)i;  // "previous position" = the stable tie-break
    }
  }
  DevBuf<double> dkey, dthr, dlo, dhi;
  DevBuf<int32_t> dperm, dtmp, dtb;
  DevBuf<Seg> dsegs;
  DevBuf<GSeg> dglist;
  DevBuf<unsigned long long> tie;
  RPT_TRY(dkey.alloc((size_t)n));
  RPT_TRY(dperm.alloc((size_t)n));
  RPT_TRY(dtmp.alloc((size_t)n));
  RPT_TRY(dtb.alloc((size_t)n));
  RPT_TRY(dthr.alloc((size_t)S));
  RPT_TRY(dlo.alloc((size_t)S));
  RPT_TRY(dhi.alloc((size_t)S));
  RPT_TRY(tie.alloc(1));
  RPT_HIP(hipMemcpy(dkey.p, key_host, (size_t)n * 8, hipMemcpyHostToDevice));
  RPT_HIP(hipMemcpy(dperm.p, perm_io_host, (size_t)n * 4, hipMemcpyHostToDevice));
  RPT_HIP(hipMemcpy(dtb.p, pos.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  RPT_HIP(hipMemset(tie.p, 0, 8));
  hipLaunchKernelGGL(fill_f64_kernel, dim3(16), dim3(256), 0, st, dthr.p, (int64_t)S,
                     std::numeric_limits<double>::quiet_NaN());
  hipLaunchKernelGGL(fill_f64_kernel, dim3(16), dim3(256), 0, st, dlo.p, (int64_t)S,
                     std::numeric_limits<double>::quiet_NaN());
  hipLaunchKernelGGL(fill_f64_kernel, dim3(16), dim3(256), 0, st, dhi.p, (int64_t)S,
                     std::numeric_limits<double>::quiet_NaN());
  std::vector<Seg> small;
  std::vector<GSeg> big;
  for (int32_t s = 0; s < S; ++s) {
    if (seg_len[s] <= 0) continue;
    if (seg_len[s] <= kSmallCap) small.push_back(Seg{seg_off[s], (int32_t)seg_len[s], s});
    else {
      const int nn = (int)seg_len[s], nh = nn / 2;
      big.push_back(GSeg{seg_off[s], nn, 0, s, nh, nh > 0 ? nh - 1 : 0,
                         nh + 1 < nn ? nh + 1 : nn - 1});
    }
  }
  if (!small.empty()) {
    RPT_TRY(dsegs.alloc(small.size()));
    RPT_HIP(hipMemcpy(dsegs.p, small.data(), small.size() * sizeof(Seg), hipMemcpyHostToDevice));
    int nmax = 0;
    for (const Seg& s : small) nmax = s.n > nmax ? s.n : nmax;
    const size_t smem = (size_t)next_pow2(nmax) * (8 + 4);
    hipLaunchKernelGGL(small_sort_kernel<double>, dim3((unsigned)small.size(), 1), dim3(256), smem,
                       st, dperm.p, dperm.p, n, dkey.p, 1, 0, dsegs.p, dtb.p, dthr.p, dlo.p,
                       dhi.p, (int64_t)S, tie.p);
  }
  if (!big.empty()) {
    RPT_TRY(gsort<double>(ctx, dperm.p, dtmp.p, n, dkey.p, 1, 0, big, dglist, dtb.p));
    hipLaunchKernelGGL(gsort_emit_kernel<double>, dim3((unsigned)((big.size() + 63) / 64)),
                       dim3(64), 0, st, dperm.p, n, dkey.p, 1, 0, dglist.p, (int)big.size(),
                       (const NodeAux*)nullptr, (const int*)nullptr, dthr.p, dlo.p, dhi.p,
                       (int64_t)S, tie.p);
  }
  RPT_HIP(hipGetLastError());
  RPT_HIP(stream_sync(st));
  RPT_HIP(hipMemcpy(perm_io_host, dperm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  std::vector<double> a((size_t)S), b((size_t)S), c((size_t)S);
  RPT_HIP(hipMemcpy(a.data(), dthr.p, (size_t)S * 8, hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(b.data(), dlo.p, (size_t)S * 8, hipMemcpyDeviceToHost));
  RPT_HIP(hipMemcpy(c.data(), dhi.p, (size_t)S * 8, hipMemcpyDeviceToHost));
  for (int32_t s = 0; s < S; ++s) {
    thr_mg_host[3 * s + 0] = a[(size_t)s];
    thr_mg_host[3 * s + 1] = b[(size_t)s];
    thr_mg_host[3 * s + 2] = c[(size_t)s];
  }
  return RPT_OK;
}

}  // namespace rpt
