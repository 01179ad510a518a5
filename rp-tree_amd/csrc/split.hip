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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>

#include "common.h"

namespace rpt {
namespace {

constexpr int kSmallCap = 4096;   // largest segment sorted by one workgroup in LDS
constexpr int kNB = 1024;         // value bins of the big path (0 and kNB-1 are the tails)
constexpr int kSample = 1024;     // samples per big node
constexpr int kDelta = 80;        // splitter half-width in sample ranks (5 sigma: sd = 16)
constexpr int kChunk = 4096;      // elements per block in hist / scatter
constexpr int kPad = 0x7fffffff;  // id of bitonic padding entries

// ---- ordered-integer image of a floating key (for atomicMax / atomicMin) ----------------
__device__ inline unsigned long long ord_of(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ inline double ord_to(unsigned long long u, double) {
  u = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u;
  return __longlong_as_double((long long)u);
}
__device__ inline unsigned long long ord_of(float v) {
  unsigned int u = __float_as_uint(v);
  u = (u >> 31) ? ~u : (u | 0x80000000u);
  return (unsigned long long)u;
}
__device__ inline float ord_to(unsigned long long w, float) {
  unsigned int u = (unsigned int)w;
  u = (u >> 31) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

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

// In-LDS bitonic sort of np (power of two) (key, id) pairs by Keys::less.  All threads of
// the block must call it.
template <class TK>
__device__ void lds_bitonic(TK* skey, int* sid, int np, const Keys<TK>& K) {
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const TK kl = skey[lo], kh = skey[hi];
        const int il = sid[lo], ih = sid[hi];
        const bool h_lt_l = K.less(kh, ih, kl, il);
        const bool l_lt_h = K.less(kl, il, kh, ih);
        if (up ? h_lt_l : l_lt_h) {
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
  Keys<TK> K{P + (int64_t)t * L * N, N, level, tb};
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
    unsigned long long* tie_count, unsigned long long* dbg) {
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
  for (; depth < kRmax; ++depth) {
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
  Keys<TK> K{P + (int64_t)g.t * L * N, N, level, tb};
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
  Keys<TK> K{P + (int64_t)g.t * L * N, N, level, tb};
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
  RPT_HIP(hipStreamSynchronize(ctx->stream));  // list is host stack memory
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

template <class TK>
int32_t build_forest_t(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int32_t mode) {
  const int64_t N = f->n;
  const int T = f->T, L = f->L;
  hipStream_t st = ctx->stream;
  f->mode = mode;

  // ---- topology: split nodes and leaves per level ----
  std::vector<Node> topo;
  enumerate_topology(N, L, f->min_leaf, topo);
  int Lused = 0;
  for (const Node& nd : topo)
    if (!nd.leaf && nd.level + 1 > Lused) Lused = nd.level + 1;
  std::vector<std::vector<Seg>> splits((size_t)Lused), leaves((size_t)Lused + 1);
  for (const Node& nd : topo) {
    Seg s{nd.off, (int32_t)nd.n, nd.leaf ? -1 : (int32_t)nd.heap};
    if (nd.leaf) leaves[(size_t)nd.level].push_back(s);
    else splits[(size_t)nd.level].push_back(s);
  }
  // DFS order is not offset order per level; both are fine (segments are disjoint)

  if ((int64_t)T * f->nodes > 0)
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->thr.p,
                       (int64_t)T * f->nodes, std::numeric_limits<double>::quiet_NaN());
  if ((int64_t)T * f->nodes > 0) {
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->mglo.p,
                       (int64_t)T * f->nodes, std::numeric_limits<double>::quiet_NaN());
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, f->mghi.p,
                       (int64_t)T * f->nodes, std::numeric_limits<double>::quiet_NaN());
  }
  if (N == 0) return RPT_OK;
  if (Lused == 0) {  // the root is a Tip: data in input order (Internal.hs:289-290)
    hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, st, f->perm.p, N, T);
    RPT_HIP(hipGetLastError());
    return RPT_OK;
  }

  // ---- projection batch: all (tree, level) hyperplanes, X read ceil(T*L/32) times ----
  RPT_TRY(f->proj.alloc((size_t)T * L * N * sizeof(TK)));
  TK* P = reinterpret_cast<TK*>(f->proj.p);
  if (Lused == L) {
    RPT_TRY(project_columns(ctx, ds, f->R.p, T * L, mode, P));
  } else {  // levels >= Lused are never reached (Internal.hs:270: rvs ! ixLev is lazy)
    for (int t = 0; t < T; ++t)
      RPT_TRY(project_columns(ctx, ds, f->R.p + (int64_t)t * L * f->d, Lused, mode,
                              P + (int64_t)t * L * N));
  }

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
  DevBuf<unsigned long long> counters;  // [0] tie nodes, [1] (uint) big-mid count
  DevBuf<GSeg> dglist;
  RPT_TRY(counters.alloc(2));
  RPT_HIP(hipMemsetAsync(counters.p, 0, 16, st));
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
  hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, st, bufA.p, N, T);

  int32_t* F = f->perm.p;
  unsigned long long* dbgbuf = nullptr;
  DevBuf<unsigned long long> dbgdev;
  if (getenv("RPT_DEBUG_STAMPS")) {
    RPT_TRY(dbgdev.alloc(256));
    RPT_HIP(hipMemsetAsync(dbgdev.p, 0, 256 * 8, st));
    dbgbuf = dbgdev.p;
  }
  unsigned long long* tie_count = counters.p;
  unsigned int* big_count = reinterpret_cast<unsigned int*>(counters.p + 1);

  auto upload = [&](const std::vector<Seg>& v, DevBuf<Seg>& d) -> int32_t {
    RPT_TRY(d.ensure(v.size()));
    RPT_HIP(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(Seg), hipMemcpyHostToDevice, st));
    RPT_HIP(hipStreamSynchronize(st));  // v may be a temporary
    return RPT_OK;
  };

  // pending split nodes per level, each remembering which ping-pong buffer holds its points
  struct PNode {
    Seg seg;
    int buf;
  };
  std::vector<std::vector<PNode>> pending((size_t)Lused + kRmax + 1);
  int32_t* bufs[2] = {bufA.p, bufB.p};
  if (!splits[0].empty()) pending[0].push_back(PNode{splits[0][0], 0});
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

  for (int level = 0; level < Lused; ++level) {
    for (int b = 0; b < 2; ++b) {
      std::vector<Seg> small, big;
      for (const PNode& pn : pending[(size_t)level])
        if (pn.buf == b) (pn.seg.n > kSmallCap ? big : small).push_back(pn.seg);
      if (small.empty() && big.empty()) continue;
      ProfScope ps(ctx, RPT_PROF_SPLIT);
      int32_t* cur = bufs[b];
      int32_t* nxt = bufs[1 - b];

      if (!small.empty()) {  // whole subtrees in LDS, kRmax levels per launch
        RPT_TRY(upload(small, dsegs));
        hipLaunchKernelGGL(subtree_kernel<TK>, dim3((unsigned)small.size(), T), dim3(kSubThreads), 0, st,
                           cur, nxt, F, N, P, L, level, f->min_leaf, dsegs.p, f->thr.p, f->mglo.p,
                           f->mghi.p, f->nodes, tie_count, dbgbuf);
        if (level + kRmax < Lused) {
          std::vector<Seg> rest;
          for (const Seg& sgm : small) descend(sgm, level, 0, kRmax, rest);
          for (const Seg& sgm : rest) pending[(size_t)level + kRmax].push_back(PNode{sgm, 1 - b});
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
        RPT_HIP(hipStreamSynchronize(st));
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
          RPT_HIP(hipStreamSynchronize(st));
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
  if (dbgbuf) {
    unsigned long long hs[256];
    RPT_HIP(hipStreamSynchronize(st));
    RPT_HIP(hipMemcpy(hs, dbgbuf, sizeof(hs), hipMemcpyDeviceToHost));
    for (int i = 1; i < 256 && hs[i]; ++i) fprintf(stderr, "stamp %d: +%llu\n", i, hs[i] - hs[i - 1]);
  }
  unsigned long long ties = 0;
  RPT_HIP(hipMemcpyAsync(&ties, tie_count, 8, hipMemcpyDeviceToHost, st));
  RPT_HIP(hipStreamSynchronize(st));
  f->tie_nodes = (int64_t)ties;
  return RPT_OK;
}

}  // namespace

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
  RPT_HIP(hipStreamSynchronize(st));
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
