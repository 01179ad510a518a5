// project.hip — the projection batch  P[c][i] = R[c] `inner` x_i.
//
// Replaces the N scalar `inner` calls of partitionAtMedian (Internal.hs:504) and the
// per-node projection of `candidates` (RPTree.hs:303-304).  Because rp-tree keeps one
// hyperplane per tree LEVEL (Internal.hs:171-175), the projections of every point on a set
// of (tree, level) hyperplanes are one tall-skinny contraction  P[C x N] = R[C x d] * X^T
// that does not depend on the trees' current permutations.
//
// Kernels (all gfx950, wave64):
//   proj_exact<T,CB>   VALU, lane = point.  Reproduces innerSD/innerSS bit for bit: terms are
//                      added from the LAST index to the first, `acc = x*r + acc`, separate
//                      multiply and add (Internal.hs:364,382).  Dense-ified zeros of the
//                      hyperplane contribute an exact +-0 and never change the sum.
//   proj_mfma<TIn,TC>  v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 tiles: hyperplane
//                      fragments live in registers (A operand), the point tile is staged
//                      through LDS with coalesced whole-row HBM reads (B operand), output
//                      written tree-major.  k-ordered fma chain -> not bit-identical to the
//                      reference, within 1e-5*|x||r|.
//   proj_csr<T,CB>     CSR rows x dense-ified hyperplanes (innerSS semantics,
//                      Internal.hs:351-366), exact order as above.
//
// HBM traffic per launch (algorithmic): N*d*sizeof(x) + d*CB*8 + N*CB*sizeof(p).
#include <hip/hip_bf16.h>

#include <cstdlib>
#include <vector>

#include "common.h"
#include "codes.h"

namespace rpt {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// 16-byte register pieces (SSA-friendly ext vectors; struct pieces were demoted to scratch)
template <class T>
struct Vec16;
template <>
struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <>
struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); };

// ---------------------------------------------------------------------------------------
// transpose a block of hyperplanes R[C][d] -> Rt[blk][d][CB] (k-major, zero padded)
// ---------------------------------------------------------------------------------------
template <class T>
__global__ void transpose_R(const double* __restrict__ R, int C, int d, int CB,
                            T* __restrict__ Rt) {
  int64_t total = (int64_t)((C + CB - 1) / CB) * d * CB;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % CB);
    int k = (int)((i / CB) % d);
    int blk = (int)(i / ((int64_t)CB * d));
    int col = blk * CB + c;
    Rt[i] = col < C ? (T)R[(int64_t)col * d + k] : (T)0;
  }
}

__device__ inline double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ inline double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ inline float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ inline float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ inline double fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------------------------------
// exact-order dense kernel.  Block = 256 threads = 4 waves; a wave owns 64 consecutive
// rows (lane = row).  X is staged in k-chunks of KC through a wave-private LDS tile
// [64][KC+1] (odd dword-pair stride: ds_read_b64 by 32 lanes hits 32 distinct bank pairs).
// Chunks are visited from the LAST to the first and k descends inside a chunk, so each
// accumulator sees exactly the reference's right-nested order.
// Hyperplane values are wave-uniform: they are read through the scalar cache (s_load).
// ---------------------------------------------------------------------------------------
template <class T, int CB, int KC>
__global__ __launch_bounds__(256) void proj_exact(const T* __restrict__ X, int64_t n, int d,
                                                  const T* __restrict__ Rt /*[d][CB]*/,
                                                  T* __restrict__ P, int64_t ldp, int ncol) {
  constexpr int LDW = KC + 1;
  __shared__ T tile[4][kWave * LDW];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * kWave;
  T* my = tile[wave];

  T acc[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) acc[c] = (T)0;

  const int nchunk = (d + KC - 1) / KC;
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    const int k0 = ch * KC;
    const int klen = min(KC, d - k0);
    // ---- stage [64 rows][klen] : consecutive lanes walk along a row (coalesced) ----
    __syncthreads();
    for (int e = lane; e < kWave * KC; e += kWave) {
      int r = e / KC, k = e % KC;
      int64_t row = row0 + r;
      T v = (T)0;
      if (k < klen && row < n) v = X[row * d + k0 + k];
      my[r * LDW + k] = v;
    }
    __syncthreads();
    // ---- accumulate, k descending ----
    const T* rrow = Rt + (int64_t)k0 * CB;
    for (int k = klen - 1; k >= 0; --k) {
      const T x = my[lane * LDW + k];
      const T* rk = rrow + (int64_t)k * CB;  // wave-uniform address -> scalar loads
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[c] = add_rn(mul_rn(rk[c], x), acc[c]);
    }
  }
  const int64_t row = row0 + lane;
  if (row < n) {
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (c < ncol) P[(int64_t)c * ldp + row] = acc[c];
  }
}

// ---------------------------------------------------------------------------------------
// Exact-order kernel, LDS-broadcast form (rows of exactly D elements).  (Reading the 32
// hyperplane values of a k through wave-uniform scalar loads, as proj_exact does, is bounded by
// the scalar cache: four serialized s_load_dwordx16 per k — measured 1.1 ms per launch at C2.)
// Here a wave owns 128 rows (two per lane) and walks k-chunks of KC from the last to the first; the chunk of X ([128][KC]) and the chunk of hyperplanes ([KC][32]) are
// staged in a wave-private LDS slab (next chunk prefetched into registers meanwhile); per k the
// 32 hyperplane values are read with eight uniform-address ds_read_b128 (LDS broadcast) and
// each feeds two rows: acc = r*x + acc, separate multiply and add, k descending — the
// reference's innerSD order (Internal.hs:382), bit for bit.
// ---------------------------------------------------------------------------------------
template <class T, int D, int KC>
// (f64: 256 registers, two workgroups = two waves per SIMD, so one wave's chunk commits — which
// wait for HBM — run under the other's arithmetic; the f32 instantiation needs more registers)
__global__ __launch_bounds__(256, sizeof(T) == 8 ? 2 : 1) void proj_exact_lds(const T* __restrict__ X, int64_t n,
                                                         const T* __restrict__ Rt /*[D][32]*/,
                                                         T* __restrict__ P, int64_t ldp, int ncol,
                                                         int64_t ntiles,
                                                         uint16_t* __restrict__ Cd /* codes of the
                                                            same columns (codes.h) or null */,
                                                         int64_t ldc,
                                                         const unsigned long long* __restrict__ cmm,
                                                         unsigned int cmask /* columns with codes */) {
  constexpr int CB = 32;
  constexpr int ROWS = 128;
  CodeGeo<T> cg{(T)1, (T)0};
  if (Cd) cg = code_geo<T>(cmm[0], cmm[1]);
  constexpr int PIECE = 16 / (int)sizeof(T);
  constexpr int PPR = KC / PIECE;                  // pieces per row per chunk
  constexpr int NPX = ROWS * PPR / 64;             // X pieces per lane per chunk
  constexpr int NPR = KC * CB / PIECE / 64;        // hyperplane pieces per lane per chunk
  constexpr int NCH = D / KC;
  constexpr int LDW = KC + PIECE;                  // X row stride (keeps 16-B alignment)
  static_assert(NPR >= 1, "chunk too small");
  __shared__ __attribute__((aligned(16))) T xt[4][ROWS * LDW];
  __shared__ __attribute__((aligned(16))) T rt[4][KC * CB];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  T* myx = xt[wave];
  T* myr = rt[wave];
  typedef typename Vec16<T>::type Raw;
  const int64_t last_row = n - 1;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * 4;

  Raw sx[NPX], sr[NPR];
  auto issue = [&](int64_t t, int ch) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = i * 64 + lane;
      int64_t row = t * ROWS + p / PPR;
      row = row < last_row ? row : last_row;
      sx[i] = *reinterpret_cast<const Raw*>(X + row * D + ch * KC + (p % PPR) * PIECE);
    }
#pragma unroll
    for (int i = 0; i < NPR; ++i)
      sr[i] = *reinterpret_cast<const Raw*>(Rt + (int64_t)ch * KC * CB + (i * 64 + lane) * PIECE);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = i * 64 + lane;
      *reinterpret_cast<Raw*>(myx + (p / PPR) * LDW + (p % PPR) * PIECE) = sx[i];
    }
#pragma unroll
    for (int i = 0; i < NPR; ++i) *reinterpret_cast<Raw*>(myr + (i * 64 + lane) * PIECE) = sr[i];
  };

  T acc0[CB], acc1[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    acc0[c] = (T)0;
    acc1[c] = (T)0;
  }
  int64_t t = wave_global;
  int ch = NCH - 1;
  if (t < ntiles) issue(t, ch);
  while (t < ntiles) {
    __builtin_amdgcn_wave_barrier();
    commit();
    int64_t tn = t;
    int chn = ch - 1;
    if (chn < 0) {
      chn = NCH - 1;
      tn = t + wave_stride;
    }
    if (tn < ntiles) issue(tn, chn);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // One wave per SIMD: nothing hides a dependent instruction's latency.  Left to the compiler
    // every v_mul was followed at once by the v_add that consumes it and every LDS read by its
    // first use: the VALU sat idle half the time.  Here the chunk is one flat sequence of 16-byte
    // hyperplane pieces (k descending, columns ascending); the pieces travel through a ring of
    // eight registers, read kAhead pieces before their use, and the products of two pieces (four
    // columns x two rows) are all issued before the first add.
    constexpr int NV = CB / PIECE;           // pieces per hyperplane row (k)
    constexpr int NPC = KC * NV;             // pieces per chunk
    constexpr int kRing = 8, kAhead = 6;
    static_assert(NPC % 2 == 0 && kAhead < kRing && kAhead % 2 == 0, "pieces are consumed in pairs");
    Raw ring[kRing];
    T xk[2][2];
    auto piece_ptr = [&](int i) {  // same address in every lane (LDS broadcast)
      return reinterpret_cast<const Raw*>(myr + (KC - 1 - i / NV) * CB) + (i % NV);
    };
    xk[(KC - 1) & 1][0] = myx[lane * LDW + KC - 1];
    xk[(KC - 1) & 1][1] = myx[(lane + 64) * LDW + KC - 1];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) ring[i % kRing] = *piece_ptr(i);
#pragma unroll
    for (int i = 0; i < NPC; i += 2) {
      const int k = KC - 1 - i / NV, kb = k & 1;
      if (i % NV == 0 && k > 0) {  // the next row's two x values, a whole row ahead
        xk[kb ^ 1][0] = myx[lane * LDW + k - 1];
        xk[kb ^ 1][1] = myx[(lane + 64) * LDW + k - 1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (i + kAhead + u < NPC) ring[(i + kAhead + u) % kRing] = *piece_ptr(i + kAhead + u);
      const T x0 = xk[kb][0], x1 = xk[kb][1];
      constexpr int NPROD = 2 * PIECE;       // columns of the pair
      T p0[NPROD], p1[NPROD];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NPROD; ++j) {
        const T r = ring[(i + j / PIECE) % kRing][j % PIECE];
        p0[j] = mul_rn(r, x0);
        p1[j] = mul_rn(r, x1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int cbase = (i % NV) * PIECE;
#pragma unroll
      for (int j = 0; j < NPROD; ++j) {
        acc0[cbase + j] = add_rn(p0[j], acc0[cbase + j]);
        acc1[cbase + j] = add_rn(p1[j], acc1[cbase + j]);
      }
    }
    if (ch == 0) {
      const int64_t row0 = t * ROWS + lane, row1 = row0 + 64;
#pragma unroll
      for (int c = 0; c < CB; ++c)
        if (c < ncol) {
          if (row0 < n) P[(int64_t)c * ldp + row0] = acc0[c];
          if (row1 < n) P[(int64_t)c * ldp + row1] = acc1[c];
          if (Cd && ((cmask >> c) & 1u)) {
            if (row0 < n) Cd[(int64_t)c * ldc + row0] = code_of(acc0[c], cg);
            if (row1 < n) Cd[(int64_t)c * ldc + row1] = code_of(acc1[c], cg);
          }
        }
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        acc0[c] = (T)0;
        acc1[c] = (T)0;
      }
    }
    t = tn;
    ch = chn;
  }
}

// ---------------------------------------------------------------------------------------
// MFMA kernel.  D[m = hyperplane][n = point] = sum_k A[m][k] * B[k][n],
//   A[m][k] = R[c0+m][k]   (lane l: m = l&15, k = 4s + (l>>4)) — registers, loaded once per
//             k-chunk (hoisted out of the tile loop when d <= KCH),
//   B[k][n] = X[row0+n][k] (lane l: n = l&15, k = 4s + (l>>4)) — ds_read from the LDS tile.
// A wave walks 16-row tiles; the tile is staged with whole-row coalesced loads (16 B/lane)
// into a wave-private LDS tile with a 16-byte row pad (conflict-free ds_read_b64 for the
// 16 rows x 4 k pattern).  Output tree-major: for a fixed register the 16 lanes of a
// quarter-wave hold 16 consecutive points of one hyperplane (128-B segments).
// CBT = number of 16-column MFMA blocks per pass (2 -> 32 hyperplanes).
// ---------------------------------------------------------------------------------------
template <class TC>
struct Mfma;
template <>
struct Mfma<double> {
  typedef double4_t acc_t;
  __device__ static acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // v_mfma_f64_16x16x4_f64 C/D map: col = lane&15, row = (lane>>4) + 4*reg
  __device__ static int row_of(int lane, int reg) { return (lane >> 4) + 4 * reg; }
  // point-major tiles (proj_mfma_fast / _wide): M index m carries point 4 * (m & 3) + (m >> 2) of
  // the tile, so that a lane's four accumulators (rows q, q+4, q+8, q+12) are CONSECUTIVE points
  __device__ static int point_of_m(int m) { return 4 * (m & 3) + (m >> 2); }
};
template <>
struct Mfma<float> {
  typedef float4_t acc_t;
  __device__ static acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // v_mfma_f32_16x16x4_f32 C/D map: col = lane&15, row = 4*(lane>>4) + reg
  __device__ static int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
  __device__ static int point_of_m(int m) { return m; }  // rows 4q .. 4q+3 are consecutive already
};

template <class TIn>
__device__ inline float to_f32(TIn v);
template <>
__device__ inline float to_f32<float>(float v) { return v; }
template <>
__device__ inline float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

// Codes of a wave's finished tile (codes.h), after its projections have been stored.  The main
// loops of the MFMA kernels run at the register limit: the geometry of each column is fetched
// from LDS four values at a time, behind scheduling barriers, so that the epilogue's temporaries
// never overlap the loop's (the compiler would otherwise hoist all 4*CBT LDS reads).
template <class TC>
struct CodeCtl {   // kept in LDS: its registers would otherwise live through the main loop
  uint16_t* base;  // codes of column c0, point 0
  int64_t ld;
  TC s, b;         // the geometry (codes.h)
};
// Point-major tiles (proj_mfma_fast / _wide): D[m = point][n = hyperplane], i.e. the X tile is
// the MFMA's first operand and the hyperplane fragments the second.  A lane then holds FOUR
// CONSECUTIVE POINTS (row0 + 4 q + 0..3, q = lane >> 4) of ONE hyperplane (h * 16 + (lane & 15))
// per column tile h: its projections are one 32-byte (f64) run and its codes one 8-byte run —
// CBT code stores and 2 CBT (f64) projection stores per tile instead of 4 CBT + 4 CBT.  (The
// hyperplane-major form stored one point of four hyperplanes per lane: 2-byte code stores, one
// instruction per value, cost the f64 kernel 12 % whatever the arithmetic in front of them.)
template <class TC>
struct Vec4;
template <>
struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4), aligned(16))); };
template <>
struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4), aligned(16))); };
typedef unsigned short code4_t __attribute__((ext_vector_type(4), aligned(8)));

// bit h of a lane's mask: column h * 16 + (lane & 15) of the pass gets codes (its level is streamed)
template <int CBT>
__device__ inline unsigned int code_lane_mask(int lane, int c0, int ncol, int L, int Lc) {
  unsigned int m = 0;
#pragma unroll
  for (int h = 0; h < CBT; ++h) {
    const int col = h * 16 + (lane & 15);
    if (col < ncol && (c0 + col) % L < Lc) m |= 1u << h;
  }
  return m;
}

template <class TC, int CBT>
__device__ __forceinline__ void tile_load(typename Mfma<TC>::acc_t (&acc)[CBT], const TC* __restrict__ P,
                                          int64_t ldp, int c0, int ncol, int64_t row0, int64_t n,
                                          int lane) {
  const int64_t r4 = row0 + 4 * (lane >> 4);
#pragma unroll
  for (int h = 0; h < CBT; ++h) {
    const int col = h * 16 + (lane & 15);
    if (col < ncol) {
      const TC* p = P + (int64_t)(c0 + col) * ldp + r4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r4 + r < n) acc[h][r] = p[r];
    }
  }
}

// value of lane l ^ 1 (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ double lane_swap1(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), 0xb1, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0xb1, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_swap1(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xb1, 0xf, 0xf, true));
}

template <class TC, int CBT, bool CODES>
__device__ __forceinline__ void tile_store(const typename Mfma<TC>::acc_t (&acc)[CBT], TC* __restrict__ P,
                                           int64_t ldp, int c0, int ncol, int64_t row0, int64_t n,
                                           int lane, const CodeCtl<TC>* ctl, unsigned int cmask) {
  const int64_t r4 = row0 + 4 * (lane >> 4);
  if constexpr (sizeof(TC) == 8) {
    // f64: a lane holds 32 bytes (four points) of one hyperplane row, the row's 128 bytes of
    // this tile sit in four lanes 16 apart — stored as they are, every instruction writes half
    // of each line (16 of every 32 bytes).  Lanes l and l ^ 1 (neighbouring hyperplanes) swap
    // one 16-byte half instead, so that the first instruction writes the EVEN row of the pair
    // completely (eight lanes x 16 contiguous bytes) and the second the odd row.
    const bool whole = row0 + 15 < n && (ldp & 1) == 0 && ncol == CBT * 16;  // wave-uniform
    if (whole) {
      typedef double d2 __attribute__((ext_vector_type(2), aligned(16)));
      const int odd = lane & 1;
#pragma unroll
      for (int h = 0; h < CBT; ++h) {
        // even lane: keeps (a0, a1), gives (a2, a3); odd lane: keeps (b2, b3), gives (b0, b1)
        const double g0 = odd ? acc[h][0] : acc[h][2], g1 = odd ? acc[h][1] : acc[h][3];
        const double x0 = lane_swap1(g0), x1 = lane_swap1(g1);
        const int col = h * 16 + (lane & 15);
        TC* pe = P + (int64_t)(c0 + (col & ~1)) * ldp + r4 + 2 * odd;   // the pair's even row
        TC* po = pe + ldp;                                               // ... and odd row
        d2 ve, vo;
        if (odd) {
          ve = d2{x0, x1};                  // the even row's points 4q + 2, 4q + 3
          vo = d2{acc[h][2], acc[h][3]};    // own points 4q + 2, 4q + 3
        } else {
          ve = d2{acc[h][0], acc[h][1]};    // own points 4q, 4q + 1
          vo = d2{x0, x1};                  // the odd row's points 4q, 4q + 1
        }
        *reinterpret_cast<d2*>(pe) = ve;
        *reinterpret_cast<d2*>(po) = vo;
      }
    }
    if (whole) goto codes;
  }
  if (r4 >= n) return;
  {
  const bool vec = r4 + 3 < n && (ldp & 3) == 0;  // whole, 16-byte aligned runs
#pragma unroll
  for (int h = 0; h < CBT; ++h) {
    const int col = h * 16 + (lane & 15);
    if (col < ncol) {
      TC* p = P + (int64_t)(c0 + col) * ldp + r4;
      if (vec) {
        typename Vec4<TC>::type v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[h][r];
        *reinterpret_cast<typename Vec4<TC>::type*>(p) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r4 + r < n) p[r] = acc[h][r];
      }
    }
  }
  }
codes:
  if constexpr (CODES) {
    if (cmask == 0) return;
    const CodeCtl<TC> cc = *ctl;
    const CodeGeo<TC> g{cc.s, cc.b};
    const bool cvec = r4 + 3 < n && (cc.ld & 3) == 0;
#pragma unroll
    for (int h = 0; h < CBT; ++h)
      if ((cmask >> h) & 1u) {
        uint16_t* q = cc.base + (int64_t)(h * 16 + (lane & 15)) * cc.ld + r4;
        if (cvec) {
          code4_t v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = code_of(acc[h][r], g);
          *reinterpret_cast<code4_t*>(q) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r4 + r < n) q[r] = code_of(acc[h][r], g);
        }
      }
  }
}

template <class TIn, class TC, int CBT, int KCH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void proj_mfma(const TIn* __restrict__ X, int64_t n,
                                                        int d, const double* __restrict__ R,
                                                        int c0, int ncol, TC* __restrict__ P,
                                                        int64_t ldp, int64_t ntiles) {
  constexpr int STEPS = KCH / 4;
  constexpr int LDW = KCH + 16 / (int)sizeof(TC);  // row stride in elements (16-B pad)
  __shared__ __attribute__((aligned(16))) TC tile[WAVES][16 * LDW];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  TC* my = tile[wave];
  typedef typename Mfma<TC>::acc_t acc_t;

  const int nchunk = (d + KCH - 1) / KCH;
  TC a[CBT][STEPS];
  auto load_a = [&](int k0) {
#pragma unroll
    for (int h = 0; h < CBT; ++h) {
      const int col = h * 16 + m;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int k = k0 + 4 * s + q;
        a[h][s] = (col < ncol && k < d) ? (TC)R[(int64_t)(c0 + col) * d + k] : (TC)0;
      }
    }
  };
  if (nchunk == 1) load_a(0);

  const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * WAVES;
  for (int64_t t = wave_global; t < ntiles; t += wave_stride) {
    const int64_t row0 = t * 16;
    acc_t acc[CBT];
#pragma unroll
    for (int h = 0; h < CBT; ++h) acc[h] = acc_t{0, 0, 0, 0};

    for (int ch = 0; ch < nchunk; ++ch) {
      const int k0 = ch * KCH;
      if (nchunk > 1) load_a(k0);
      // ---- stage 16 rows x KCH: one instruction covers 64 lanes x VEC contiguous elems ----
      constexpr int VEC = 16 / (int)sizeof(TIn);  // elements per 16-B load
      const bool vec_ok = (d % VEC) == 0;
      __builtin_amdgcn_wave_barrier();
      if (vec_ok) {
        for (int e = lane; e < 16 * (KCH / VEC); e += 64) {
          const int r = e / (KCH / VEC), kv = (e % (KCH / VEC)) * VEC;
          const int64_t row = row0 + r;
          TC v[VEC];
          if (row < n && k0 + kv < d) {  // d % VEC == 0 -> the whole vector is in range
            struct alignas(16) Raw { TIn v[VEC]; };
            const Raw g = *reinterpret_cast<const Raw*>(X + row * d + k0 + kv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              if constexpr (sizeof(TIn) == sizeof(TC)) v[j] = (TC)g.v[j];
              else v[j] = (TC)to_f32<TIn>(g.v[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = (TC)0;
          }
#pragma unroll
          for (int j = 0; j < VEC; ++j) my[r * LDW + kv + j] = v[j];
        }
      } else {
        for (int e = lane; e < 16 * KCH; e += 64) {
          const int r = e / KCH, k = e % KCH;
          const int64_t row = row0 + r;
          TC v = (TC)0;
          if (row < n && k0 + k < d) {
            if constexpr (sizeof(TIn) == sizeof(TC)) v = (TC)X[row * d + k0 + k];
            else v = (TC)to_f32<TIn>(X[row * d + k0 + k]);
          }
          my[r * LDW + k] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- MFMA over the chunk ----
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const TC b = my[m * LDW + 4 * s + q];
#pragma unroll
        for (int h = 0; h < CBT; ++h) acc[h] = Mfma<TC>::run(a[h][s], b, acc[h]);
      }
    }
    // ---- store: P[c0 + hyperplane][row0 + point] ----
    const int64_t row = row0 + m;  // D col = lane&15 = point
    if (row < n) {
#pragma unroll
      for (int h = 0; h < CBT; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = h * 16 + Mfma<TC>::row_of(lane, r);
          if (col < ncol) P[(int64_t)(c0 + col) * ldp + row] = acc[h][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fast MFMA path for rows of exactly D elements (D = 128: BASELINE configs C2/C4).
// No per-element guards anywhere in the loop:
//   * hyperplane fragments come from a pre-padded buffer Apad[CBT][D/4][64] (one coalesced
//     load per fragment, zero where col >= ncol),
//   * a 16-row tile is ONE contiguous 16*D*sizeof(TIn) byte run of row-major X: every lane
//     loads 16 B pieces, piece p of the tile -> row p / (D*sizeof/16); rows past n are
//     clamped to the last row (their results are never stored),
//   * software pipeline (async-STAGE split): the next tile's pieces are loaded into
//     registers while the current tile is multiplied out of LDS; waves never synchronise
//     with each other (wave-private LDS tile).
// ---------------------------------------------------------------------------------------
template <class TC>
__global__ void pad_A_kernel(const double* __restrict__ R, int C, int d, int D, int nblk, int cbt,
                             int col0, int k0, TC* __restrict__ Apad /*[nblk][cbt][D/4][64]*/) {
  const int steps = D / 4;
  const int64_t total = (int64_t)nblk * cbt * steps * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int s = (int)((i >> 6) % steps);
    const int h = (int)((i / (64 * (int64_t)steps)) % cbt);
    const int blk = (int)(i / (64 * (int64_t)steps * cbt));
    const int col = col0 + blk * 16 * cbt + h * 16 + (lane & 15);
    const int k = k0 + 4 * s + (lane >> 4);
    Apad[i] = (col < C && k < d) ? (TC)R[(int64_t)col * d + k] : (TC)0;
  }
}

template <class TIn, class TC, int D, int CBT, bool CODES = false>
__device__ __forceinline__ void proj_fast_body(const TIn* __restrict__ X, int64_t n,
                                               const TC* __restrict__ Apad, int c0,
                                               int ncol, TC* __restrict__ P,
                                               int64_t ldp, int64_t ntiles,
                                               int64_t ldx /* row stride of X */,
                                               int accumulate /* P += instead of = */,
                                               int kvalid /* elements of the K chunk that
                                                             exist (the rest reads as 0) */,
                                               uint16_t* __restrict__ Cd /* codes.h; null
                                                  unless this pass completes the sums */,
                                               int64_t ldc,
                                               const unsigned long long* __restrict__ cmm,
                                               int cL, int cLc /* codes for the columns
                                                  whose level (column % cL) is < cLc */) {
  constexpr int STEPS = D / 4;
  constexpr int PIECE = 16 / (int)sizeof(TIn);            // elements per 16-B piece
  constexpr int PIECES_PER_ROW = D / PIECE;
  __shared__ CodeCtl<TC> cctl;
  unsigned int cmask = 0;
  if constexpr (CODES) {
    if (threadIdx.x == 0) {
      const CodeGeo<TC> g = code_geo<TC>(cmm[0], cmm[1]);
      cctl = CodeCtl<TC>{Cd + (int64_t)c0 * ldc, ldc, g.s, g.b};
    }
    cmask = code_lane_mask<CBT>(threadIdx.x & 63, c0, ncol, cL, cLc);
    __syncthreads();
  }
  constexpr int NP = 16 * PIECES_PER_ROW / 64;            // pieces per lane per tile
  constexpr int LDW = D + 16 / (int)sizeof(TC);           // LDS row stride (elements of TC)
  static_assert((16 * PIECES_PER_ROW) % 64 == 0, "tile must be a multiple of 64 pieces");
  __shared__ __attribute__((aligned(16))) TC tile[4][16 * LDW];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  TC* my = tile[wave];
  typedef typename Mfma<TC>::acc_t acc_t;
  struct alignas(16) Raw { TIn v[PIECE]; };

  TC a[CBT][STEPS];
#pragma unroll
  for (int h = 0; h < CBT; ++h)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) a[h][s] = Apad[(h * STEPS + s) * 64 + lane];

  const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * 4;
  const int64_t last_row = n - 1;

  Raw stage[NP];
  auto issue = [&](int64_t t) {
    const int64_t row0 = t * 16;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = i * 64 + lane;
      int64_t row = row0 + p / PIECES_PER_ROW;
      row = row < last_row ? row : last_row;
      if ((p % PIECES_PER_ROW) * PIECE < kvalid)
        stage[i] = *reinterpret_cast<const Raw*>(X + row * ldx + (p % PIECES_PER_ROW) * PIECE);
      else
        stage[i] = Raw{};  // past the end of a short row: kvalid is a multiple of PIECE
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = i * 64 + lane;
      TC* dst = my + (p / PIECES_PER_ROW) * LDW + (p % PIECES_PER_ROW) * PIECE;
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        if constexpr (sizeof(TIn) == sizeof(TC)) dst[j] = (TC)stage[i].v[j];
        else dst[j] = (TC)to_f32<TIn>(stage[i].v[j]);
      }
    }
  };

  int64_t t = wave_global;
  if (t < ntiles) issue(t);
  while (t < ntiles) {
    commit();                                   // waits for the staged pieces, writes LDS
    const int64_t tn = t + wave_stride;
    if (tn < ntiles) issue(tn);                 // in flight during the MFMA phase
    acc_t acc[CBT];
#pragma unroll
    for (int h = 0; h < CBT; ++h) acc[h] = acc_t{0, 0, 0, 0};
    // a later K chunk of rows longer than D: continue the sum
    if (accumulate) tile_load<TC, CBT>(acc, P, ldp, c0, ncol, t * 16, n, lane);
    const int mp = Mfma<TC>::point_of_m(m);  // the tile row this lane feeds (point-major tiles)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const TC b = my[mp * LDW + 4 * s + q];
#pragma unroll
      for (int h = 0; h < CBT; ++h) acc[h] = Mfma<TC>::run(b, a[h][s], acc[h]);
    }
    tile_store<TC, CBT, CODES>(acc, P, ldp, c0, ncol, t * 16, n, lane, &cctl, cmask);
    t = tn;
  }
}

template <class TIn, class TC, int D, int CBT, bool CODES = false>
__global__ __launch_bounds__(256, 2) void proj_mfma_fast(const TIn* __restrict__ X, int64_t n,
                                                         const TC* __restrict__ Apad, int c0, int ncol,
                                                         TC* __restrict__ P, int64_t ldp, int64_t ntiles,
                                                         int64_t ldx, int accumulate, int kvalid,
                                                         uint16_t* __restrict__ Cd, int64_t ldc,
                                                         const unsigned long long* __restrict__ cmm,
                                                         int cL, int cLc) {
  proj_fast_body<TIn, TC, D, CBT, CODES>(X, n, Apad, c0, ncol, P, ldp, ntiles, ldx, accumulate, kvalid, Cd,
                                         ldc, cmm, cL, cLc);
}

// Few rows, many hyperplanes (a query batch against every (tree, level) of a forest): ALL column
// groups of 32 in ONE launch, blockIdx.y = group (fragments Apad[group][2][D/4][64]); the rows are
// read once per group — nothing next to the launches this saves (C2: ten 96/128-column launches and
// fragment copies of 10 000 rows took 0.15 ms of a 1.5 ms query batch).  Same MFMA sequence per
// (row, hyperplane) as the other kernels: same bits.
template <class TIn, class TC, int D>
__global__ __launch_bounds__(256, 2) void proj_mfma_fast_groups(const TIn* __restrict__ X, int64_t n,
                                                                const TC* __restrict__ Apad, int C,
                                                                TC* __restrict__ P, int64_t ldp,
                                                                int64_t ntiles, int64_t ldx, int kvalid) {
  const int c0 = (int)blockIdx.y * 32;
  const int ncol = C - c0 < 32 ? C - c0 : 32;
  proj_fast_body<TIn, TC, D, 2, false>(X, n, Apad + (size_t)blockIdx.y * 2 * (D / 4) * 64, c0, ncol, P, ldp,
                                       ntiles, ldx, 0, kvalid, nullptr, 0, nullptr, 1, 0);
}

// ---------------------------------------------------------------------------------------
// Wide MFMA path (rows of exactly D elements): 16*CBT hyperplanes per pass over X, so the
// point set is read ceil(C / (16*CBT)) times instead of ceil(C / 32) times.  The hyperplane
// fragments no longer fit in registers (CBT * D/4 values per lane): they live in LDS, in
// fragment order (one conflict-free ds_read per MFMA), shared by the 8 waves of the one
// workgroup a CU holds.  Each wave stages a 1/KS K-slice of a 16-row tile at a time (16 rows x
// D/KS elements, wave-private LDS region) through two register stages: a slice is requested
// two MFMA phases before it is needed.
//   LDS = CBT*D/4*64*sizeof(TC) + 8 * 16*(D/KS + pad)*sizeof(TC)
//       f64: CBT 4, KS 2: 64 + 66 KB;  CBT 6, KS 4: 96 + 34 KB
// f64: one v_mfma_f64_16x16x4 is 64 cycles and a tile costs CBT*32 of them per wave: at
// CBT = 6 the matrix pipe needs 0.31 ms per pass and HBM (1.79 GB) about as long — the kernel
// sits where the two rooflines cross, which is the point of reading X less often.
// ---------------------------------------------------------------------------------------
template <class TC, int D, int CBT, int KS, int WPB>
constexpr size_t wide_smem_bytes() {
  return ((size_t)CBT * (D / 4) * 64 + (size_t)WPB * 16 * (D / KS + 16 / sizeof(TC))) * sizeof(TC) +
         sizeof(CodeCtl<TC>);  // + code destination and geometry
}

template <class TIn, class TC, int D, int CBT, int KS, int WPB, bool CODES = false>
__global__ __launch_bounds__(WPB * 64, 1) void proj_mfma_wide(const TIn* __restrict__ X, int64_t n,
                                                         const TC* __restrict__ Apad, int c0,
                                                         int ncol, TC* __restrict__ P,
                                                         int64_t ldp, int64_t ntiles,
                                                         int64_t ldx /* row stride of X */,
                                                         int accumulate /* P += instead of = */,
                                                         int kvalid /* elements of the K chunk that
                                                                       exist (the rest reads as 0) */,
                                                         uint16_t* __restrict__ Cd /* codes.h; null
                                                            unless this pass completes the sums */,
                                                         int64_t ldc,
                                                         const unsigned long long* __restrict__ cmm,
                                                         int cL, int cLc /* codes for the columns
                                                            whose level (column % cL) is < cLc */) {
  constexpr int STEPS = D / 4, SSTEPS = STEPS / KS, KW = D / KS;
  constexpr int PIECE = 16 / (int)sizeof(TIn);  // elements per 16-B piece
  constexpr int PPR = KW / PIECE;               // pieces per row slice
  constexpr int NP = 16 * PPR / 64;             // pieces per lane per slice
  constexpr int LDW = KW + 16 / (int)sizeof(TC);
  static_assert((16 * PPR) % 64 == 0 && NP >= 1, "slice must be a multiple of 64 pieces");
  static_assert(KS % 2 == 0, "two register stages alternate per slice");
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_smem[];
  TC* As = reinterpret_cast<TC*>(wide_smem);                   // [CBT][STEPS][64]
  TC* tiles = As + (size_t)CBT * STEPS * 64;                   // [WPB][16 * LDW]
  CodeCtl<TC>* cctl = reinterpret_cast<CodeCtl<TC>*>(tiles + (size_t)WPB * 16 * LDW);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  TC* my = tiles + (size_t)wave * 16 * LDW;
  typedef typename Mfma<TC>::acc_t acc_t;
  struct alignas(16) Raw { TIn v[PIECE]; };

  for (int i = threadIdx.x; i < CBT * STEPS * 64; i += WPB * 64) As[i] = Apad[i];
  unsigned int cmask = 0;
  if constexpr (CODES) {
    if (threadIdx.x == 0) {
      const CodeGeo<TC> g = code_geo<TC>(cmm[0], cmm[1]);
      *cctl = CodeCtl<TC>{Cd + (int64_t)c0 * ldc, ldc, g.s, g.b};
    }
    cmask = code_lane_mask<CBT>(lane, c0, ncol, cL, cLc);
  }
  __syncthreads();

  const int64_t wave_global = (int64_t)blockIdx.x * WPB + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * WPB;
  const int64_t last_row = n - 1;

  Raw stg0[NP], stg1[NP];
  auto issue = [&](Raw (&stage)[NP], int64_t t, int slice) {
    const int64_t row0 = t * 16;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = i * 64 + lane;
      int64_t row = row0 + p / PPR;
      row = row < last_row ? row : last_row;
      if (slice * KW + (p % PPR) * PIECE < kvalid)
        stage[i] = *reinterpret_cast<const Raw*>(X + row * ldx + slice * KW + (p % PPR) * PIECE);
      else
        stage[i] = Raw{};  // past the end of a short row: kvalid is a multiple of PIECE
    }
  };
  auto commit = [&](const Raw (&stage)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = i * 64 + lane;
      TC* dst = my + (p / PPR) * LDW + (p % PPR) * PIECE;
#pragma unroll
      for (int j = 0; j < PIECE; ++j) {
        if constexpr (sizeof(TIn) == sizeof(TC)) dst[j] = (TC)stage[i].v[j];
        else dst[j] = (TC)to_f32<TIn>(stage[i].v[j]);
      }
    }
  };

  int64_t t = wave_global;
  if (t < ntiles) {
    issue(stg0, t, 0);
    issue(stg1, t, 1);
  }
  while (t < ntiles) {
    const int64_t tn = t + wave_stride;
    acc_t acc[CBT];
#pragma unroll
    for (int h = 0; h < CBT; ++h) acc[h] = acc_t{0, 0, 0, 0};
    // a later K chunk of rows longer than D: continue the sum
    if (accumulate) tile_load<TC, CBT>(acc, P, ldp, c0, ncol, t * 16, n, lane);
    const int mp = Mfma<TC>::point_of_m(m);  // the tile row this lane feeds (point-major tiles)
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      // slice sl -> LDS, then request the slice two ahead into the stage just freed
      if (sl % 2 == 0) commit(stg0);
      else commit(stg1);
      if (sl + 2 < KS) {
        if (sl % 2 == 0) issue(stg0, t, sl + 2);
        else issue(stg1, t, sl + 2);
      } else if (tn < ntiles) {
        if (sl % 2 == 0) issue(stg0, tn, sl + 2 - KS);
        else issue(stg1, tn, sl + 2 - KS);
      }
      if (sl * KW < kvalid) {  // a slice beyond a short row holds zeros only
#pragma unroll
        for (int s = 0; s < SSTEPS; ++s) {
          const TC b = my[mp * LDW + 4 * s + q];
#pragma unroll
          for (int h = 0; h < CBT; ++h)
            acc[h] = Mfma<TC>::run(b, As[(h * STEPS + sl * SSTEPS + s) * 64 + lane], acc[h]);
        }
      }
    }
    tile_store<TC, CBT, CODES>(acc, P, ldp, c0, ncol, t * 16, n, lane, cctl, cmask);
    t = tn;
  }
}

// ---------------------------------------------------------------------------------------
// CSR rows x dense-ified hyperplanes.  16 lanes per row (one lane per hyperplane of the
// block, CB = 16), 4 rows per wave; every lane walks its row's nonzeros from the last to
// the first: acc = val*r[col] + acc  (innerSS, Internal.hs:353-366: only matching indices
// contribute; a zero of the dense-ified hyperplane adds an exact zero).
// Rt[d][16] k-major: the 16 lanes of a row read 128 contiguous bytes (L2 resident).
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// bf16 data on the bf16 matrix pipe: proj_bf16x3
//   The f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate: 10 M x 768
//   bf16 points against 256 hyperplanes are 3.9 TFLOP = 25 ms of matrix pipe at its peak, five
//   times the HBM time of the same pass.  A bf16 row times a hyperplane rounded to bf16 would
//   miss the 1e-5 tolerance (8 significant bits), so the hyperplane is SPLIT: r = r_hi + r_mid +
//   r_lo with three bf16 terms (24 significant bits, what the f32 kernels keep of it), each
//   product x * r_part is exact in f32, and three v_mfma_f32_16x16x32_bf16 per tile replace
//   eight f32 MFMAs — 3/16 of the f32 pipe time, f32 accumulation as before.
//   D[m = hyperplane][n = point].  Workgroup = 8 waves, tile = 128 (or 64) hyperplanes x 256
//   points x K: every wave owns 32 points (2 n-tiles) x all column tiles, accumulators in
//   registers over the whole K, P written once.  B = rows of X, loaded straight from HBM as
//   fragments (lane l: row l&15, 16 bytes at k = 8*(l>>4)), four k-steps ahead in a register
//   ring.  A = the split hyperplanes in fragment order (split_A_bf16x3), streamed from L2
//   through two LDS chunk buffers of 2 k-steps (48 KB each), next chunk staged in registers
//   while the current one is multiplied; one workgroup barrier per chunk.  With d <= 64 * 2 the
//   two buffers hold all of A and nothing is re-staged.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kB3KC = 2;     // k-steps (32 k each) per A chunk with three hyperplane terms (four with two terms)

// hyperplanes [c0, c0 + ncol) as three bf16 terms in fragment order:
// out[chunk][ks][part][mt][lane] (16 bytes = 8 bf16): R[c0 + 16 mt + (lane & 15)][k .. k + 7],
// k = 32 (2 chunk + ks) + 8 (lane >> 4); zero outside the columns / past d.
__global__ void split_A_bf16x3(const double* __restrict__ R, int d, int c0, int ncol, int cbt,
                               int nch, int kc /* k-steps per chunk */, int np /* terms kept: 2 or 3 */,
                               uint4* __restrict__ out) {
  const int per_chunk = kc * np * cbt * 64;
  const int total = nch * kc * cbt * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63;
    int r = i >> 6;
    const int mt = r % cbt;
    r /= cbt;
    const int ks = r % kc, ch = r / kc;
    const int col = mt * 16 + (lane & 15);
    const int k0 = (ch * kc + ks) * 32 + 8 * (lane >> 4);
    unsigned int w[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int j = 0; j < 8; ++j) {
      double v = (col < ncol && k0 + j < d) ? R[(int64_t)(c0 + col) * d + k0 + j] : 0.0;
      for (int p3 = 0; p3 < 3; ++p3) {
        const __hip_bfloat16 b = __float2bfloat16((float)v);
        v -= (double)__bfloat162float(b);
        const unsigned int bits = (unsigned int)__builtin_bit_cast(unsigned short, b);
        w[p3][j >> 1] |= bits << ((j & 1) * 16);
      }
    }
    for (int p3 = 0; p3 < np; ++p3)
      out[(size_t)ch * per_chunk + ((ks * np + p3) * cbt + mt) * 64 + lane] =
          make_uint4(w[p3][0], w[p3][1], w[p3][2], w[p3][3]);
  }
}

template <int CBT, int NT /* 16-point tiles per wave, even */, bool RESIDENT /* both chunks stay in LDS */,
          int WAVES /* per workgroup */, class TP = float /* type of P */,
          int NTERM = 1 /* bf16 terms of a row: X[term][n][d]; 2 = dense-ified SVector rows (launch_csr_dense_mfma) */,
          int NP = 3 /* bf16 terms of a hyperplane */, int KC = kB3KC /* k-steps per A chunk: 2 or 4 */,
          bool CODES = false /* codes.h next to P for the columns whose level is streamed */,
          int RD = 4 /* k-steps the row requests run ahead (2 or 4) */>
__global__ __launch_bounds__(WAVES * 64) void proj_bf16x3(
    const __hip_bfloat16* __restrict__ X, int64_t n, int d, const uint4* __restrict__ Aimg,
    int nch /* even */, int c0, int ncol, TP* __restrict__ P, int64_t ldp, int64_t ntiles,
    uint16_t* __restrict__ Cd, int64_t ldc, const unsigned long long* __restrict__ cmm, int cL, int cLc) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds_a[];  // [2][CH16]
  constexpr int CH16 = KC * NP * CBT * 64;    // uint4 per chunk
  constexpr int kB3NT = NT, kB3Pts = WAVES * NT * 16, NTHR = WAVES * 64;  // points per workgroup tile
  constexpr int ST = CH16 / NTHR;              // uint4 a thread stages per chunk
  static_assert(KC == 2 || KC == 4, "the B ring is addressed by (chunk parity, k-step)");
  static_assert(NT % 2 == 0 && CH16 % NTHR == 0, "point tiles are stored in pairs; whole staging rounds");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nks = nch * KC;                   // k-steps per tile, a multiple of 4
  constexpr bool resident = RESIDENT;
  const int kg = 8 * (lane >> 4);             // the lane's k offset inside a k-step
  CodeGeo<float> cgeo{1.f, 0.f};
  unsigned int cmask = 0;  // bit mt: hyperplane mt * 16 + (lane & 15) of the pass gets codes
  if constexpr (CODES) {
    cgeo = code_geo<float>(cmm[0], cmm[1]);
    cmask = code_lane_mask<CBT>(lane, c0, ncol, cL, cLc);
  }
  // prologue: chunk 0 (and chunk 1 when resident) into LDS
  for (int s = 0; s < ST; ++s) lds_a[s * NTHR + tid] = Aimg[s * NTHR + tid];
  if constexpr (resident)
    for (int s = 0; s < ST; ++s) lds_a[CH16 + s * NTHR + tid] = Aimg[CH16 + s * NTHR + tid];
  __syncthreads();

  // B fragment of k-step ks of tile t for n-tile nt: zero past d; rows past n clamped
  auto load_b = [&](int64_t t, int ks, int nt, int term) -> uint4 {
    const int k = ks * 32 + kg;
    if (t >= ntiles || k >= d) return make_uint4(0, 0, 0, 0);
    int64_t row = t * kB3Pts + wave * (kB3NT * 16) + nt * 16 + (lane & 15);
    row = row < n ? row : n - 1;
    return *reinterpret_cast<const uint4*>(X + ((int64_t)term * n + row) * (int64_t)d + k);
  };

  int64_t tile = blockIdx.x;
  uint4 bf[RD][NTERM][kB3NT];  // ring: k-step s lives in bf[s & (RD - 1)]
#pragma unroll
  for (int u = 0; u < RD; ++u)
#pragma unroll
    for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
      for (int nt = 0; nt < kB3NT; ++nt) bf[u][tm][nt] = load_b(tile, u, nt, tm);

  for (; tile < ntiles; tile += gridDim.x) {
    f32x4 acc[CBT][kB3NT];
#pragma unroll
    for (int mt = 0; mt < CBT; ++mt)
#pragma unroll
      for (int nt = 0; nt < kB3NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cp = 0; cp < nch; cp += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = cp + h;
        uint4 stage[ST];
        if constexpr (!resident) {  // the chunk after this one (of the next tile at the end)
          const int cn = c + 1 == nch ? 0 : c + 1;
#pragma unroll
          for (int s = 0; s < ST; ++s) stage[s] = Aimg[(size_t)cn * CH16 + s * NTHR + tid];
        }
        const uint4* ab = lds_a + h * CH16;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
          const int u = (h * KC + ks) & (RD - 1);
          bf16x8 b[NTERM][kB3NT];
#pragma unroll
          for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
            for (int nt = 0; nt < kB3NT; ++nt) b[tm][nt] = __builtin_bit_cast(bf16x8, bf[u][tm][nt]);
          // refill the ring slot four k-steps ahead (the next tile's first k-steps at the end)
          {
            const int sn = c * KC + ks + RD;
            const int64_t tn = sn >= nks ? tile + gridDim.x : tile;
            const int kn = sn >= nks ? sn - nks : sn;
#pragma unroll
            for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
              for (int nt = 0; nt < kB3NT; ++nt) bf[u][tm][nt] = load_b(tn, kn, nt, tm);
          }
#pragma unroll 1  // (unrolled, the fragment reads of a k-step are hoisted together: no faster, or spills)
          for (int p3 = 0; p3 < NP; ++p3)
#pragma unroll
            for (int mt = 0; mt < CBT; ++mt) {
              const bf16x8 a = __builtin_bit_cast(bf16x8, ab[((ks * NP + p3) * CBT + mt) * 64 + lane]);
              // (two row terms: one fragment read feeds 2 NT MFMAs — the LDS reads per MFMA halve)
#pragma unroll
              for (int tm = 0; tm < NTERM; ++tm)
#pragma unroll
                for (int nt = 0; nt < kB3NT; ++nt)
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tm][nt], a, acc[mt][nt], 0, 0, 0);
            }
        }
        if constexpr (!resident) {
#pragma unroll
          for (int s = 0; s < ST; ++s) lds_a[(h ^ 1) * CH16 + s * NTHR + tid] = stage[s];
          __syncthreads();
        }
      }
    }
    // The rows are the MFMA's FIRST operand: D[row = 4 (lane >> 4) + r][col = lane & 15] = (point, hyperplane), a
    // lane holds FOUR CONSECUTIVE POINTS of one hyperplane per (mt, nt): 16-byte stores, and the codes of a
    // streamed level (codes.h) as 8-byte stores.  Two neighbouring point tiles then trade lane halves inside every
    // row of sixteen lanes (DPP row_ror:8): lanes 0..7 of a row keep their own tile-0 values and lanes 8..15 take
    // the tile-1 values of THE SAME eight hyperplanes, so that eight lanes hold 32 consecutive points of one
    // hyperplane and a store writes whole 128-byte lines of P (half lines — straight from the accumulators, or
    // hyperplane-major word stores —: WRITE_SIZE 6.6 GB for 5.1 GB of P, and 4 % of the pass).
    // ONE running pointer: the addresses of the straightforward loop are hoisted out of the tile loop (60
    // registers; with two row terms they spilled, and a scratch reload waits for every row request in flight).
    {
      const int hi8 = (lane >> 3) & 1;
      const bool vec = (ldp & 3) == 0;  // 16-byte aligned runs
      const bool cvec = CODES && (ldc & 3) == 0;
      auto ror8 = [](float v) -> float {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
      };
#pragma unroll
      for (int np2 = 0; np2 < kB3NT; np2 += 2) {
        // this lane's four points: tile np2 + hi8, rows 4 q .. 4 q + 3
        const int64_t pt = tile * kB3Pts + wave * (kB3NT * 16) + (np2 + hi8) * 16 + 4 * (lane >> 4);
        int col = lane & 7;  // ... of the hyperplanes col (first store) and col + 8 (second) of every column tile
        TP* dst = P + (int64_t)(c0 + col) * ldp + pt;
        uint16_t* q = CODES ? Cd + (int64_t)(c0 + col) * ldc + pt : nullptr;
#pragma unroll
        for (int mt = 0; mt < CBT; ++mt) {
          float lo[4], hi[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t0 = ror8(acc[mt][np2 + 1][r]), t1 = ror8(acc[mt][np2][r]);
            lo[r] = hi8 ? t0 : acc[mt][np2][r];
            hi[r] = hi8 ? acc[mt][np2 + 1][r] : t1;
          }
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float(&v4)[4] = hf ? hi : lo;
            if (col + 8 * hf < ncol) {
              if (vec && pt + 3 < n) {
                typename Vec4<TP>::type v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (TP)v4[r];
                *reinterpret_cast<typename Vec4<TP>::type*>(dst) = v;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (pt + r < n) dst[r] = (TP)v4[r];
              }
            }
            if constexpr (CODES) {
              // (cmask bit mt is about hyperplane mt * 16 + (lane & 15): the first store's for lanes 0..7 of a
              // row, the second's for lanes 8..15 — the bit of lane ^ 8 comes through the same rotation)
              const unsigned int mine = (cmask >> mt) & 1u;
              const unsigned int other = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mine, 0x128, 0xf, 0xf, true);
              const unsigned int want = hf == hi8 ? mine : other;
              if (want) {
                if (cvec && pt + 3 < n) {
                  code4_t c4;
#pragma unroll
                  for (int r = 0; r < 4; ++r) c4[r] = code_of(v4[r], cgeo);
                  *reinterpret_cast<code4_t*>(q) = c4;
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (pt + r < n) q[r] = code_of(v4[r], cgeo);
                }
              }
              q += 8 * ldc;
            }
            dst += 8 * ldp;
          }
          col += 16;
        }
      }
    }
  }
}

template <class T, int CB>
__global__ __launch_bounds__(256) void proj_csr(const int64_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ col,
                                                const T* __restrict__ val, int64_t n,
                                                const T* __restrict__ Rt /*[d][CB]*/,
                                                T* __restrict__ P, int64_t ldp, int ncol) {
  static_assert(CB == 16, "16 lanes per row");
  const int lane = threadIdx.x & 63;
  const int c = lane & 15;
  const int64_t row = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 +
                      (lane >> 4);
  if (row >= n) return;
  const int64_t a = rowptr[row], b = rowptr[row + 1];
  T acc = (T)0;
  for (int64_t j = b - 1; j >= a; --j) {
    const T x = val[j];
    const T r = Rt[(int64_t)col[j] * CB + c];
    acc = add_rn(mul_rn(r, x), acc);
  }
  if (c < ncol) P[(int64_t)c * ldp + row] = acc;
}

// ---------------------------------------------------------------------------------------
// CSR fast path: the 16-hyperplane tile Rt[d][16] lives in LDS (d*16*sizeof(T) <= 128 KB, e.g.
// 100 KB for the 784-dim C3 rows), one workgroup of 16 waves per CU.  A wave works on 16 rows:
// FOUR lanes per row, each lane owning four of the 16 hyperplanes, so one nonzero costs a lane
// two 16-byte LDS reads and four multiply-adds (16 lanes x one hyperplane each needed an LDS
// read and three broadcasts per single multiply-add: the kernel was bound by instruction issue
// at 2.3 TB/s).  The row's nonzeros are fetched sixteen at a time (four per lane), and broadcast
// one by one in REVERSE order inside the quad (DPP quad_perm, a VALU move): every lane accumulates
// `acc = val*r[col] + acc` from the last nonzero to the first — the reference's innerSS order
// (Internal.hs:353-366).  Padding entries (val 0) add an exact zero.
// ---------------------------------------------------------------------------------------
template <int U>
__device__ inline int quad_bcast(int x) {  // lane U of every quad to the whole quad
  return __builtin_amdgcn_mov_dpp(x, U | (U << 2) | (U << 4) | (U << 6), 0xf, 0xf, true);
}
template <int U>
__device__ inline float quad_bcast(float x) {
  return __int_as_float(quad_bcast<U>(__float_as_int(x)));
}
template <int U>
__device__ inline double quad_bcast(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = quad_bcast<U>((int)(b & 0xffffffffLL)), hi = quad_bcast<U>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// LDS row stride of the hyperplane tile in elements: 16 + 2 — with 16 the sixteen quads of a
// wave, each reading the 16-element row of a different column index, all start at bank 0 or 32
constexpr int kCsrLd = 18;

// one nonzero of the batch: element E of quad lane U, broadcast to the quad
template <class T, int U, int E>
__device__ inline void csr_term(const int (&mycol)[4], const T (&myval)[4], const T* rl, int q,
                                T (&acc)[4]) {
  const int cu = quad_bcast<U>(mycol[E]);
  const T vu = quad_bcast<U>(myval[E]);
  const T* r = rl + cu * kCsrLd + 4 * q;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = add_rn(mul_rn(vu, r[k]), acc[k]);
}
// the 16 nonzeros of a batch from the LAST (lane 3, element 3) to the first (lane 0, element 0)
template <class T, int I>
__device__ inline void csr_batch(const int (&mycol)[4], const T (&myval)[4], const T* rl, int q,
                                 T (&acc)[4]) {
  csr_term<T, I / 4, I % 4>(mycol, myval, rl, q, acc);
  if constexpr (I > 0) csr_batch<T, I - 1>(mycol, myval, rl, q, acc);
}

template <class T>
__global__ __launch_bounds__(1024) void proj_csr_lds(const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const T* __restrict__ val, int64_t n, int d,
                                                     const T* __restrict__ Rt /*[d][16]*/,
                                                     T* __restrict__ P, int64_t ldp, int ncol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* rl = reinterpret_cast<T*>(smem_raw);
  for (int i = threadIdx.x; i < d * 16; i += blockDim.x) rl[(i >> 4) * kCsrLd + (i & 15)] = Rt[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int q = lane & 3;  // hyperplanes 4q .. 4q+3
  const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wave_stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t ngroups = (n + 15) / 16;
  for (int64_t g = wave_global; g < ngroups; g += wave_stride) {
    const int64_t row = g * 16 + (lane >> 2);
    const bool rv = row < n;
    const int64_t a = rv ? rowptr[row] : 0, b = rv ? rowptr[row + 1] : 0;
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    int64_t j0 = b;
    while (__any(j0 > a)) {
      // a batch of 16 nonzeros [j0 - 16, j0): lane q of the quad loads four consecutive ones,
      // so the quad reads one 64-byte run of columns and one 128-byte run of values
      int mycol[4];
      T myval[4];
      const int64_t i4 = j0 - 16 + 4 * q;
      if (i4 >= a && j0 > a) {  // the lane's four nonzeros all exist: two/three wide loads
        struct __attribute__((packed, aligned(4))) C4 { int v[4]; };
        struct __attribute__((packed, aligned(4))) V4 { T v[4]; };
        const C4 c4 = *reinterpret_cast<const C4*>(col + i4);
        const V4 v4 = *reinterpret_cast<const V4*>(val + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mycol[e] = c4.v[e];
          myval[e] = v4.v[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t idx = i4 + e;
          const bool ok = idx >= a && j0 > a;
          mycol[e] = ok ? col[idx] : 0;
          myval[e] = ok ? val[idx] : (T)0;
        }
      }
      csr_batch<T, 15>(mycol, myval, rl, q, acc);
      j0 -= 16;
    }
    if (rv) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * q + k < ncol) P[(int64_t)(4 * q + k) * ldp + row] = acc[k];
    }
  }
}

// ---------------------------------------------------------------------------------------
// CSR, 32 hyperplanes per pass: the same scheme with EIGHT hyperplanes per lane (four lanes per
// row), so the CSR arrays are read once per 32 hyperplanes instead of once per 16.
//
// The 16-column kernel is not bound by HBM but by the LDS: every (nonzero, hyperplane) pair
// fetches its own hyperplane component (rows of a wave hit unrelated columns: no reuse), and
// with padded tile rows the four rows of a ds_read_b128 lane group land on random 64-byte
// quarters of the 256-byte bank row — two of four collide on average (~2.1 LDS cycles per
// group instead of 1).  Here the tile row of a column is exactly one bank row (32 f64 = 256
// bytes, no padding) cut in four 64-byte PIECES, a quad reads one piece per instruction (lane q
// its 16 bytes), and the four quads of a lane group read four DIFFERENT pieces: quad r starts at
// piece rho(r) and walks (rho + j) mod 4 over its four reads.  Whatever columns the rows hit,
// the group covers each bank once: conflict-free by construction.  A lane's accumulators are
// therefore rotated by its rho; they are rotated back once per row when P is written.
// (f32: rows are 128 bytes = two pieces; two of a group's four quads share a piece and collide
// when their columns have the same parity — 1.5 cycles per group on average.)
//
// The tile Rt[k][32] of a column range [k_lo, k_hi) lives in LDS.  When all d rows fit (f32 up
// to d ~ 1100, f64 up to d ~ 570) one launch does everything; otherwise the columns are cut in
// two halves and a pass is TWO launches: the first walks every row's nonzeros with column >=
// k_mid (CSR columns ascend, so that is the tail [split, end) of the row, found once per
// dataset), from the last to the first, and leaves the partial sums in P; the second starts
// from those sums and walks the head [begin, split).  Together they perform exactly the
// reference's right-nested sum (Internal.hs:353-366), in the same order, so P is bit-identical;
// each launch reads half of the CSR arrays.
// ---------------------------------------------------------------------------------------
template <class T>
struct Csr32 {
  static constexpr int kPer = 16 / (int)sizeof(T);      // hyperplanes per 16-byte read: 2 / 4
  static constexpr int kReads = 8 / kPer;               // reads per lane and nonzero: 4 / 2
  static constexpr int kRowBytes = 32 * (int)sizeof(T);  // one tile row: 256 / 128
  struct __attribute__((aligned(16))) Vec {
    T v[kPer];
  };
};

// start piece of a quad: distinct over the four quads of every ds_read_b128 lane group
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same in the upper half: quads {0,3,5,6}
// and {1,2,4,7})
__device__ inline int csr_rho(int lane) { return ((lane >> 2) & 7) >> 1; }

// the four nonzeros of quad lane U, from the last (element 3) to the first: the LDS reads of a
// nonzero are issued two nonzeros ahead of its arithmetic, so a wave waits for the LDS once per
// group, not once per read
template <class T, int U, bool FUSED>
__device__ inline void csr_group32(const int (&mycolb)[4], const T (&myval)[4],
                                   const unsigned char* const (&rp)[Csr32<T>::kReads],
                                   T (&acc)[Csr32<T>::kReads][Csr32<T>::kPer]) {
  using G = Csr32<T>;
  using Vec = typename G::Vec;
  Vec ra[G::kReads], rb[G::kReads];
#define RPT_CSR_ISSUE(E, buf)                                                    \
  {                                                                              \
    const int cb = quad_bcast<U>(mycolb[E]); /* (column - k_lo) * kRowBytes */   \
    _Pragma("unroll") for (int j = 0; j < G::kReads; ++j)                        \
        buf[j] = *reinterpret_cast<const Vec*>(rp[j] + cb);                      \
  }
#define RPT_CSR_MATH(E, buf)                                                     \
  {                                                                              \
    const T vu = quad_bcast<U>(myval[E]);                                        \
    _Pragma("unroll") for (int j = 0; j < G::kReads; ++j)                        \
        _Pragma("unroll") for (int e = 0; e < G::kPer; ++e)                      \
            acc[j][e] = FUSED ? fma_rn(vu, buf[j].v[e], acc[j][e])               \
                              : add_rn(mul_rn(vu, buf[j].v[e]), acc[j][e]);      \
  }
  RPT_CSR_ISSUE(3, ra)
  RPT_CSR_ISSUE(2, rb)
  __builtin_amdgcn_sched_barrier(0);
  RPT_CSR_MATH(3, ra)
  __builtin_amdgcn_sched_barrier(0);
  RPT_CSR_ISSUE(1, ra)
  __builtin_amdgcn_sched_barrier(0);
  RPT_CSR_MATH(2, rb)
  __builtin_amdgcn_sched_barrier(0);
  RPT_CSR_ISSUE(0, rb)
  __builtin_amdgcn_sched_barrier(0);
  RPT_CSR_MATH(1, ra)
  RPT_CSR_MATH(0, rb)
#undef RPT_CSR_ISSUE
#undef RPT_CSR_MATH
}

// split[row] = index of the row's first nonzero with column >= k_mid (rowptr[row + 1] if none)
__global__ void csr_split_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                 int64_t n, int k_mid, int64_t* __restrict__ split) {
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n;
       row += (int64_t)gridDim.x * blockDim.x) {
    int64_t a = rowptr[row], b = rowptr[row + 1];
    while (a < b) {
      const int64_t m = (a + b) >> 1;
      if (col[m] >= k_mid) b = m;
      else a = m + 1;
    }
    split[row] = a;
  }
}

// part: 0 = the whole row, 1 = the tail [split, end) from zero, 2 = the head [begin, split)
// continuing the sums the tail launch left in P.
//
// The wave's loop is software-pipelined over (row group, batch of 16 nonzeros): the column /
// value loads of the NEXT batch — of the next row group when the current rows are finished, whose
// row range was itself fetched a whole group earlier, and in part 2 that group's partial sums —
// are issued before the arithmetic of the current batch.  Without this a wave paid a dependent
// HBM latency for the row range, one per batch and (part 2) one for the sums: at C3 the kernel
// ran at 2.8 TB/s with VALU and LDS half idle.
constexpr int kCsr32Threads = 768;  // 3 waves per SIMD: 168 VGPRs for the two batches in flight

template <class T, bool FUSED>
__global__ __launch_bounds__(kCsr32Threads) void proj_csr_lds32(
    const int64_t* __restrict__ rowptr, const int64_t* __restrict__ split,
    const int32_t* __restrict__ col, const T* __restrict__ val, int64_t n, int k_lo, int k_hi,
    const T* __restrict__ Rt /*[d][32]*/, T* __restrict__ P, int64_t ldp, int ncol, int part) {
  using G = Csr32<T>;
  constexpr int NR = G::kReads, NP = G::kPer;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  {  // the tile is a straight copy of rows [k_lo, k_hi) of Rt
    const typename G::Vec* src = reinterpret_cast<const typename G::Vec*>(Rt + (int64_t)k_lo * 32);
    typename G::Vec* dst = reinterpret_cast<typename G::Vec*>(smem_raw);
    const int nvec = (k_hi - k_lo) * 32 / NP;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();  // the only barrier: waves leave independently below
  const int lane = threadIdx.x & 63;
  const int q = lane & 3;
  const int rho = csr_rho(lane) & (NR - 1);
  const unsigned char* rp[NR];  // read j inside tile row 0: piece (rho + j) mod NR, slot q
#pragma unroll
  for (int j = 0; j < NR; ++j) rp[j] = smem_raw + (((rho + j) & (NR - 1)) * 64 + 16 * q);
  const int64_t wave_stride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t ngroups = (n + 15) / 16;
  int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= ngroups) return;

  struct __attribute__((packed, aligned(4))) C4 { int v[4]; };
  struct __attribute__((packed, aligned(4))) V4 { T v[4]; };
  // the lane's row range in group gg ([0, 0) past the end)
  auto range = [&](int64_t gg, int64_t& ra, int64_t& rb) {
    const int64_t r = gg * 16 + (lane >> 2);
    ra = 0;
    rb = 0;
    if (gg < ngroups && r < n) {
      ra = part == 1 ? split[r] : rowptr[r];
      rb = part == 2 ? split[r] : rowptr[r + 1];
    }
  };
  // the lane's four nonzeros [jj - 16 + 4q, +4) of the batch ending at jj, clipped to [ra, jj)
  auto fetch = [&](int64_t ra, int64_t jj, C4& c, V4& v) {
    const int64_t i4 = jj - 16 + 4 * q;
    if (i4 >= ra && jj > ra) {  // all four exist: wide loads
      c = *reinterpret_cast<const C4*>(col + i4);
      v = *reinterpret_cast<const V4*>(val + i4);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t idx = i4 + e;
        const bool ok = idx >= ra && jj > ra;
        c.v[e] = ok ? col[idx] : k_lo;  // padding: value 0 times tile row 0
        v.v[e] = ok ? val[idx] : (T)0;
      }
    }
  };
  // hyperplane of accumulator (j, e) in this lane: piece (rho + j) mod NR, slot q
  auto column = [&](int j, int e) { return 4 * NP * ((rho + j) & (NR - 1)) + NP * q + e; };
  // partial sums of the lane's row in group gg (part 2), in accumulator order.  A lane touches 16
  // bytes of a 128-byte line per access — the group's other quads fill the line within the same
  // few instructions, L2 merges them.
  auto sums = [&](int64_t gg, T (&dst)[NR][NP]) {
    const int64_t r = gg * 16 + (lane >> 2);
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        const int c = column(j, e);
        dst[j][e] = (gg < ngroups && r < n && c < ncol) ? P[(int64_t)c * ldp + r] : (T)0;
      }
  };

  int64_t a, b, an, bn;
  range(g, a, b);
  int64_t j0 = b;
  C4 cc;
  V4 vc;
  fetch(a, j0, cc, vc);
  range(g + wave_stride, an, bn);
  T acc[NR][NP], nat[NR][NP];
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[j][e] = nat[j][e] = (T)0;
  if (part == 2) sums(g, acc);
  for (;;) {
    // ---- what comes after this batch (wave-uniform) ----
    const bool more = __any(j0 - 16 > a);
    const int64_t gn = g + wave_stride;
    const bool last = !more && gn >= ngroups;
    const int64_t na = more ? a : an, nj0 = more ? j0 - 16 : bn;
    C4 cn;
    V4 vn;
    if (!last) fetch(na, nj0, cn, vn);
    if (!more && !last && part == 2) sums(gn, nat);
    // ---- this batch ----
    int mycolb[4];
    T myval[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mycolb[e] = (cc.v[e] - k_lo) * G::kRowBytes;
      myval[e] = vc.v[e];
    }
    // quad lane U holds batch positions 4U .. 4U+3; a row with rem = j0 - a nonzeros left fills
    // the positions from 16 - rem up, so group U is empty in every row of the wave unless some
    // row has rem >= 13 - 4U (row lengths vary: the wave runs to its longest row)
    const int64_t rem = j0 - a;
    if (__any(rem >= 1)) {
      csr_group32<T, 3, FUSED>(mycolb, myval, rp, acc);
      if (__any(rem >= 5)) {
        csr_group32<T, 2, FUSED>(mycolb, myval, rp, acc);
        if (__any(rem >= 9)) {
          csr_group32<T, 1, FUSED>(mycolb, myval, rp, acc);
          if (__any(rem >= 13)) csr_group32<T, 0, FUSED>(mycolb, myval, rp, acc);
        }
      }
    }
    if (!more) {  // the rows of this group are complete
      const int64_t row = g * 16 + (lane >> 2);
      if (row < n) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
          for (int e = 0; e < NP; ++e) {
            const int c = column(j, e);
            if (c < ncol) P[(int64_t)c * ldp + row] = acc[j][e];
          }
      }
      if (last) break;
      g = gn;
      a = an;
      b = bn;
      range(g + wave_stride, an, bn);  // used a whole group from now
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < NP; ++e) acc[j][e] = nat[j][e];  // zeros unless part 2
    }
    j0 = nj0;
    cc = cn;
    vc = vn;
  }
}

// one wide pass over a 128-element K chunk of X (columns [k0, k0 + D) of every row) for the
// CBT*16 hyperplanes padded into fragment order at Ab
template <class TIn, class TC, int D, int CBT, int KS>
int32_t launch_wide(rpt_ctx* ctx, const rpt_dataset* ds, int k0, int kvalid, int accumulate, int c0,
                    int ncol, const TC* Ab, TC* P, int64_t ntiles, int64_t blocks,
                    const CodeOut* co /* null: no codes from this pass */) {
  constexpr int WPB = 8;
  constexpr size_t smem = wide_smem_bytes<TC, D, CBT, KS, WPB>();
  static DeviceOnce attr_once;  // per device: one process may drive several (rpt_comm_init)
  RPT_TRY(attr_once.run(ctx->device, [&]() -> int32_t {
    RPT_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&proj_mfma_wide<TIn, TC, D, CBT, KS, WPB, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    RPT_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&proj_mfma_wide<TIn, TC, D, CBT, KS, WPB, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    return RPT_OK;
  }));
  if (co)
    hipLaunchKernelGGL((proj_mfma_wide<TIn, TC, D, CBT, KS, WPB, true>), dim3((unsigned)blocks),
                       dim3(WPB * 64), smem, ctx->stream, (const TIn*)ds->X + k0, ds->n, Ab, c0,
                       ncol, P, ds->n, ntiles, (int64_t)ds->d, accumulate, kvalid, co->codes, co->ld,
                       co->mm, co->L, co->Lc);
  else
    hipLaunchKernelGGL((proj_mfma_wide<TIn, TC, D, CBT, KS, WPB, false>), dim3((unsigned)blocks),
                       dim3(WPB * 64), smem, ctx->stream, (const TIn*)ds->X + k0, ds->n, Ab, c0,
                       ncol, P, ds->n, ntiles, (int64_t)ds->d, accumulate, kvalid,
                       (uint16_t*)nullptr, (int64_t)0, (const unsigned long long*)nullptr, 1, 0);
  return RPT_OK;
}

template <class TIn, class TC>
int32_t launch_mfma(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                    TC* P, const CodeOut* co, bool* codes_written) {
  const int64_t n = ds->n;
  const int64_t ntiles = (n + 15) / 16;
  constexpr int kPiece = 16 / (int)sizeof(TIn);
  if (ds->d % kPiece == 0) {  // rows are 16-byte aligned: the pipelined kernels
    constexpr int D = 128;
    // Rows are projected one 128-element K chunk at a time: every chunk is a pass of the same
    // kernels over columns [k0, k0 + 128) of X (row stride d), the chunks after the first
    // continue the sums in P; a last chunk shorter than 128 reads as zero past the row's end
    // (and the wide kernel skips its empty K slices).
    const int nkc = (ds->d + D - 1) / D;
    if (nkc == 1 && C > 32 && !co && n <= 65536 && !ctx->opt.proj_narrow) {  // see proj_mfma_fast_groups
      const int ngrp = (C + 31) / 32;
      const size_t frag1 = (size_t)(D / 4) * 64;
      DevBuf<TC> Ag;
      RPT_TRY(Ag.alloc((size_t)ngrp * 2 * frag1));
      hipLaunchKernelGGL(pad_A_kernel<TC>, dim3(64), dim3(256), 0, ctx->stream, R_dev, C, ds->d, D, ngrp, 2,
                         0, 0, Ag.p);
      int64_t bx = (ntiles + 3) / 4;
      const int64_t capx = ((int64_t)ctx->n_cu * 8 + ngrp - 1) / ngrp;
      if (bx > capx) bx = capx;
      if (bx < 1) bx = 1;
      ProfScope pn(ctx, RPT_PROF_PROJECT);
      hipLaunchKernelGGL((proj_mfma_fast_groups<TIn, TC, D>), dim3((unsigned)bx, (unsigned)ngrp), dim3(256), 0,
                         ctx->stream, (const TIn*)ds->X, n, Ag.p, C, P, n, ntiles, (int64_t)ds->d, ds->d);
      RPT_HIP(hipGetLastError());
      return RPT_OK;  // Ag returns to the stream-ordered allocator
    }
    // Column passes.  Full passes take 96 hyperplanes per read of X (6 column tiles: the point
    // where the f64 matrix pipe and HBM take about as long).  What is left goes to the smallest
    // shape that holds it (the matrix pipe pays for padded columns too): <= 32 columns the
    // register-resident 32-column kernel, <= 64 four tiles, else six — except that a tail of at
    // most 32 columns after a full pass is folded into that pass (8 tiles, 128 columns: cheaper
    // than one more read of X).
    struct Pass {
      int c0, ncol, tiles;  // tiles == 0: the 32-column kernel
    };
    std::vector<Pass> passes;
    constexpr int KS8 = sizeof(TC) == 8 ? 8 : 4;  // 8 tiles of 8-byte fragments are 128 KB of LDS
    {
      const bool wide_ok = !ctx->opt.proj_narrow;
      int c0 = 0;
      while (c0 < C) {
        const int left = C - c0;
        int take, tiles;
        if (!wide_ok || left <= 32) {
          take = left < 32 ? left : 32;
          tiles = 0;
        } else if (left <= 64) {
          take = left;
          tiles = 4;
        } else if (left <= 96) {
          take = left;
          tiles = 6;
        } else if (left <= 128) {
          take = left;
          tiles = 8;
        } else if (sizeof(TC) == 4 && left <= 160) {
          // f32 fragments are half the size: up to ten column tiles fit the LDS, and what is left of
          // a 10 M-point shard's 136 columns after a 96-column pass (40) is a pass bound by its read of X
          // (round 4; C4 shard build 14.7 / 14.9 -> 14.4 / 13.8 ms on one box)
          take = left;
          tiles = left <= 144 ? 9 : 10;
        } else {
          take = 96;
          tiles = 6;
        }
        passes.push_back(Pass{c0, take, tiles});
        c0 += take;
      }
    }
    const size_t frag = (size_t)(D / 4) * 64;  // fragment words of one 16-column tile
    size_t total_tiles = 0;
    for (const Pass& ps : passes) total_tiles += ps.tiles ? ps.tiles : 2;
    DevBuf<TC> Afrag;
    RPT_TRY(Afrag.alloc(total_tiles * nkc * frag));
    // fragment-order copies of the hyperplanes, one per (column pass, K chunk)
    {
      size_t off = 0;
      for (int kc = 0; kc < nkc; ++kc)
        for (const Pass& ps : passes) {
          const int tl = ps.tiles ? ps.tiles : 2;
          hipLaunchKernelGGL(pad_A_kernel<TC>, dim3(16), dim3(256), 0, ctx->stream, R_dev, C, ds->d,
                             D, 1, tl, ps.c0, kc * D, Afrag.p + off);
          off += (size_t)tl * frag;
        }
    }
    int64_t wblocks = (ntiles + 7) / 8;
    if (wblocks > ctx->n_cu) wblocks = ctx->n_cu;
    int64_t blocks = (ntiles + 3) / 4;
    const int64_t cap = (int64_t)ctx->n_cu * 2;
    if (blocks > cap) blocks = cap;
    size_t off = 0;
    for (int kc = 0; kc < nkc; ++kc) {
      const int k0 = kc * D, accumulate = kc > 0;
      const int kvalid = ds->d - k0 < D ? ds->d - k0 : D;
      const CodeOut* cop = (co && kc == nkc - 1) ? co : nullptr;  // codes of the COMPLETE sums
      for (const Pass& ps : passes) {
        const TC* Ab = Afrag.p + off;
        off += (size_t)(ps.tiles ? ps.tiles : 2) * frag;
        if (ps.tiles) {
          // resolved into class 0 as well
          ProfScope pw(ctx, RPT_PROF_PROJECT_WIDE);
          switch (ps.tiles) {
            case 10:
            case 9:
              if constexpr (sizeof(TC) == 4) {
                if (ps.tiles == 10)
                  RPT_TRY((launch_wide<TIn, TC, D, 10, 4>(ctx, ds, k0, kvalid, accumulate, ps.c0, ps.ncol, Ab, P,
                                                          ntiles, wblocks, cop)));
                else
                  RPT_TRY((launch_wide<TIn, TC, D, 9, 4>(ctx, ds, k0, kvalid, accumulate, ps.c0, ps.ncol, Ab, P,
                                                         ntiles, wblocks, cop)));
              }
              break;
            case 8:
              RPT_TRY((launch_wide<TIn, TC, D, 8, KS8>(ctx, ds, k0, kvalid, accumulate, ps.c0,
                                                       ps.ncol, Ab, P, ntiles, wblocks, cop)));
              break;
            case 6:
              RPT_TRY((launch_wide<TIn, TC, D, 6, 4>(ctx, ds, k0, kvalid, accumulate, ps.c0,
                                                     ps.ncol, Ab, P, ntiles, wblocks, cop)));
              break;
            default:
              RPT_TRY((launch_wide<TIn, TC, D, 4, 2>(ctx, ds, k0, kvalid, accumulate, ps.c0,
                                                     ps.ncol, Ab, P, ntiles, wblocks, cop)));
          }
        } else {
          ProfScope pn(ctx, RPT_PROF_PROJECT);
#define RPT_FAST(CBT_, CODES_)                                                                  \
  hipLaunchKernelGGL((proj_mfma_fast<TIn, TC, D, CBT_, CODES_>), dim3((unsigned)blocks), dim3(256), \
                     0, ctx->stream, (const TIn*)ds->X + k0, n, Ab, ps.c0, ps.ncol, P, n, ntiles,  \
                     (int64_t)ds->d, accumulate, kvalid, cop ? cop->codes : (uint16_t*)nullptr,    \
                     cop ? cop->ld : (int64_t)0, cop ? cop->mm : (const unsigned long long*)nullptr, \
                     cop ? cop->L : 1, cop ? cop->Lc : 0)
          if (ps.ncol > 16) {
            if (cop) RPT_FAST(2, true);
            else RPT_FAST(2, false);
          } else {
            if (cop) RPT_FAST(1, true);
            else RPT_FAST(1, false);
          }
#undef RPT_FAST
        }
      }
    }
    RPT_HIP(hipGetLastError());
    if (co && codes_written) *codes_written = true;
    return RPT_OK;  // the fragment buffer returns to the allocator, which recycles it only after
                    // the stream has been synchronised
  }
  constexpr int WAVES = 4;
  int64_t blocks = (ntiles + WAVES - 1) / WAVES;
  const int64_t cap = (int64_t)ctx->n_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int ncol = C - c0 < 32 ? C - c0 : 32;
    ProfScope ps(ctx, RPT_PROF_PROJECT);
    if (ncol > 16)
      hipLaunchKernelGGL((proj_mfma<TIn, TC, 2, 128, WAVES>), dim3((unsigned)blocks),
                         dim3(WAVES * 64), 0, ctx->stream, (const TIn*)ds->X, n, ds->d, R_dev,
                         c0, ncol, P, n, ntiles);
    else
      hipLaunchKernelGGL((proj_mfma<TIn, TC, 1, 128, WAVES>), dim3((unsigned)blocks),
                         dim3(WAVES * 64), 0, ctx->stream, (const TIn*)ds->X, n, ds->d, R_dev,
                         c0, ncol, P, n, ntiles);
  }
  RPT_HIP(hipGetLastError());
  return RPT_OK;
}

// bf16 rows of 16-byte granularity on the bf16 matrix pipe (see proj_bf16x3)
template <int CBT, int NT, int WAVES, class TP, int NTERM, int NP = 3, int KC = kB3KC, int RD = 4>
int32_t launch_bf16x3_pass(rpt_ctx* ctx, const __hip_bfloat16* X, int64_t n, int d, const uint4* Aimg, int nch,
                           int c0, int ncol, TP* P, const CodeOut* co = nullptr) {
  const int64_t ntiles = (n + WAVES * NT * 16 - 1) / (WAVES * NT * 16);
  int64_t blocks = ntiles < ctx->n_cu ? ntiles : ctx->n_cu;
  if (blocks < 1) blocks = 1;
  constexpr size_t smem = (size_t)2 * KC * NP * CBT * 64 * 16;
  static_assert(smem <= 160 * 1024, "two A chunks in LDS");
  constexpr bool kCanCode = std::is_same<TP, float>::value && NTERM == 1;  // (codes from the f32 sums)
#define RPT_B3_KERNEL(RES, CODES) proj_bf16x3<CBT, NT, RES, WAVES, TP, NTERM, NP, KC, CODES, RD>
  static DeviceOnce attr_once;
  RPT_TRY(attr_once.run(ctx->device, [&]() -> int32_t {
    RPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&RPT_B3_KERNEL(true, false)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    RPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&RPT_B3_KERNEL(false, false)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if constexpr (kCanCode) {
      RPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&RPT_B3_KERNEL(true, true)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      RPT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&RPT_B3_KERNEL(false, true)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    return RPT_OK;
  }));
  const dim3 grid((unsigned)blocks), block(WAVES * 64);
  if constexpr (kCanCode) {
    if (co) {
      if (nch == 2)
        hipLaunchKernelGGL((RPT_B3_KERNEL(true, true)), grid, block, smem, ctx->stream, X, n, d, Aimg, nch, c0, ncol,
                           P, n, ntiles, co->codes, co->ld, co->mm, co->L, co->Lc);
      else
        hipLaunchKernelGGL((RPT_B3_KERNEL(false, true)), grid, block, smem, ctx->stream, X, n, d, Aimg, nch, c0, ncol,
                           P, n, ntiles, co->codes, co->ld, co->mm, co->L, co->Lc);
      return RPT_OK;
    }
  }
  if (nch == 2)
    hipLaunchKernelGGL((RPT_B3_KERNEL(true, false)), grid, block, smem, ctx->stream, X, n, d, Aimg, nch, c0, ncol, P,
                       n, ntiles, (uint16_t*)nullptr, (int64_t)0, (const unsigned long long*)nullptr, 1, 0);
  else
    hipLaunchKernelGGL((RPT_B3_KERNEL(false, false)), grid, block, smem, ctx->stream, X, n, d, Aimg, nch, c0, ncol, P,
                       n, ntiles, (uint16_t*)nullptr, (int64_t)0, (const unsigned long long*)nullptr, 1, 0);
#undef RPT_B3_KERNEL
  return RPT_OK;
}

// P[C][n] = the rows X[NTERM][n][d] (NTERM bf16 terms each) against the hyperplanes R_dev[C][d] split into
// bf16 terms.  bf16 DATA (NTERM = 1) takes TWO hyperplane terms: r = r_hi + r_mid + e with |e_i| <= 2^-17 |r_i|
// (two roundings to 8 significant bits), every product x_i * r_part exact in f32, so
// |P - x.r| <= 2^-17 sum |x_i||r_i| <= 7.6e-6 |x||r| by Cauchy-Schwarz, inside north_star's 1e-5 with the f32
// accumulation (measured: 4.6e-7 |x||r| at d = 768; three terms: 1.0e-7) — a third less matrix-pipe and LDS work
// than three terms (6.7 -> 5.0 ms per 128 hyperplanes over 10 M x 768).  Option proj_bf16_terms = 3 keeps the
// third term.  Dense-ified SVector rows (NTERM = 2) already spend their 2^-17 on the ROW split and keep three.
template <class TP, int NTERM>
int32_t launch_bf16x3_rows(rpt_ctx* ctx, const __hip_bfloat16* X, int64_t n, int d, const double* R_dev,
                           int32_t C, TP* P, const CodeOut* co = nullptr) {
  // (rows shorter than 64 elements keep three terms: their few products do not average the split's error down —
  // 7.1e-6 |x||r| was measured on rows of 8 — and cost next to nothing)
  const bool two = NTERM == 1 && ctx->opt.proj_bf16_terms != 3 && d >= 64;
  const int np = two ? 2 : 3;
  // two terms leave room for chunks of four k-steps (128 KB of LDS for 128 hyperplanes): half the barriers
  const int kc = (two && d > 128) ? 4 : kB3KC;
  int nch = (d + 32 * kc - 1) / (32 * kc);
  nch += nch & 1;  // even: chunk c always lives in LDS buffer c & 1
  struct Pass {
    int c0, ncol, cbt;
  };
  std::vector<Pass> passes;
  for (int c0 = 0; c0 < C;) {
    const int left = C - c0;
    const int take = left < 128 ? left : 128;
    passes.push_back(Pass{c0, take, take <= 64 ? 4 : 8});
    c0 += take;
  }
  auto image16 = [&](const Pass& ps) { return (size_t)nch * kc * np * ps.cbt * 64; };
  size_t total16 = 0;
  for (const Pass& ps : passes) total16 += image16(ps);
  DevBuf<uint4> Aimg;
  RPT_TRY(Aimg.alloc(total16));
  size_t off = 0;
  for (const Pass& ps : passes) {
    hipLaunchKernelGGL(split_A_bf16x3, dim3(64), dim3(256), 0, ctx->stream, R_dev, d, ps.c0,
                       ps.ncol, ps.cbt, nch, kc, np, Aimg.p + off);
    off += image16(ps);
  }
  off = 0;
  for (const Pass& ps : passes) {
    ProfScope pw(ctx, RPT_PROF_PROJECT_WIDE);
    const uint4* img = Aimg.p + off;
    // rows of up to 128 elements (A resident in LDS, no staging): four waves of 64 points each
    // — half the LDS fragment reads per MFMA, accumulators in AGPRs — are 14 % faster than
    // eight waves of 32 points (4 M x 128 x 416: 2.98 -> 2.56 ms); with the chunked A stream of
    // longer rows the single wave per SIMD hides less and loses (2 M x 768 x 256: 3.09 -> 3.27 ms),
    // and so do 64 points per wave at two waves per SIMD (spills), two waves per point group with 64
    // hyperplanes each, twelve waves, and two workgroups of four waves
    if constexpr (NTERM == 1) {
      if (two) {
        if (ps.cbt == 8 && nch == 2 && kc == 2)
          RPT_TRY((launch_bf16x3_pass<8, 4, 4, TP, 1, 2, 2>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
        else if (ps.cbt == 8 && ctx->opt.proj_bf16_terms != 8)
          // sixteen waves (four per SIMD at 117 registers), rows requested TWO k-steps ahead: as many row
          // requests in flight per CU and as far ahead in time as eight waves with four (more in flight is
          // slower), but 512 points per workgroup tile — half the hyperplane image traffic and half
          // the barriers per point: 5.05 -> 4.75 ms per 128 hyperplanes over 10 M x 768 (option value 8: the
          // eight-wave shape)
          RPT_TRY((launch_bf16x3_pass<8, 2, 16, TP, 1, 2, 4, 2>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
        else if (ps.cbt == 8)
          RPT_TRY((launch_bf16x3_pass<8, 2, 8, TP, 1, 2, 4>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
        else if (kc == 2)
          RPT_TRY((launch_bf16x3_pass<4, 2, 8, TP, 1, 2, 2>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
        else
          RPT_TRY((launch_bf16x3_pass<4, 2, 8, TP, 1, 2, 4>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
        off += image16(ps);
        continue;
      }
    }
    if (ps.cbt == 8 && nch == 2)
      RPT_TRY((launch_bf16x3_pass<8, 4, 4, TP, NTERM>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
    else if (ps.cbt == 8)
      RPT_TRY((launch_bf16x3_pass<8, 2, 8, TP, NTERM>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
    else
      RPT_TRY((launch_bf16x3_pass<4, 2, 8, TP, NTERM>(ctx, X, n, d, img, nch, ps.c0, ps.ncol, P, co)));
    off += image16(ps);
  }
  RPT_HIP(hipGetLastError());
  return RPT_OK;  // the image returns to the stream-ordered allocator
}

int32_t launch_bf16x3(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C, float* P,
                      const CodeOut* co, bool* codes_written) {
  if (!ctx->opt.proj_bf16_codes) co = nullptr;
  RPT_TRY((launch_bf16x3_rows<float, 1>(ctx, (const __hip_bfloat16*)ds->X, ds->n, ds->d, R_dev, C, P, co)));
  if (co && codes_written) *codes_written = true;
  return RPT_OK;
}

// ---- SVector rows on the matrix pipe (round 4; RPT_PROJ_MFMA = the tolerance mode on CSR data) ----
// The segmented CSR kernel is instruction-issue bound (one multiply-add per (nonzero, hyperplane):
// 10.6 ms per C3 forest with FMAs).  Dense-ified, the same contraction is proj_bf16x3's: the rows as
// TWO bf16 terms x = x_hi + x_lo (|x - x_hi - x_lo| <= 2^-17 |x|), each multiplied with the hyperplanes'
// three bf16 terms (24 bits) on v_mfma_f32_16x16x32_bf16 in ONE pass (proj_bf16x3<..., NTERM = 2>: both
// row terms ride the B ring, an A fragment feeds four MFMAs), f32 accumulation:
// |P - r.x| <= 2^-17 sum |x_i||r_i| + the f32 accumulation <= 1e-5 |x||r|
// (north_star's tolerance; tests/test_gpu_parity.py::test_project_csr_dense_mfma).  The dense terms
// ([2][n][d] bf16: 3.1 GB at C3 against 1.8 GB of CSR arrays) are built once per dataset.
template <class T>
__global__ __launch_bounds__(256) void csr_densify_bf16x2_kernel(const int64_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 const T* __restrict__ val, int64_t n, int d,
                                                                 uint16_t* __restrict__ out /* [2][n][d], zeroed */) {
  // one wave per row: the row's nonzeros, two bf16 terms each
  const int lane = threadIdx.x & 63;
  const int64_t wave_g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = wave_g; row < n; row += nwaves) {
    const int64_t a = rowptr[row], b = rowptr[row + 1];
    for (int64_t j = a + lane; j < b; j += 64) {
      const int c = col[j];
      if (c < 0 || c >= d) continue;  // (unchecked SVector invariants, Internal.hs:106-131: ignored like innerSS would)
      const double v = (double)val[j];
      const __hip_bfloat16 hi = __float2bfloat16((float)v);
      const double rest = v - (double)__bfloat162float(hi);
      const __hip_bfloat16 lo = __float2bfloat16((float)rest);
      out[row * (int64_t)d + c] = __builtin_bit_cast(unsigned short, hi);
      out[((int64_t)n + row) * (int64_t)d + c] = __builtin_bit_cast(unsigned short, lo);
    }
  }
}

// optional memory: anything that fails leaves the dataset on the segmented kernel
template <class T>
void ensure_csr_dense(rpt_ctx* ctx, const rpt_dataset* ds) {
  if (ds->csr_dense_state != 0) return;
  ds->csr_dense_state = -1;
  if (ds->d % 8 != 0 || ds->n <= 0) return;
  const size_t bytes = (size_t)2 * (size_t)ds->n * (size_t)ds->d * 2;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  if (bytes + ((size_t)8 << 30) > free_b) return;  // leave room for the build itself
  void* p = nullptr;
  if (dev_alloc(&p, bytes) != hipSuccess || !p) {
    (void)hipGetLastError();
    return;
  }
  if (hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) {
    dev_free(p);
    return;
  }
  hipLaunchKernelGGL(csr_densify_bf16x2_kernel<T>, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                     ds->rowptr, ds->col, (const T*)ds->val, ds->n, ds->d, (uint16_t*)p);
  if (hipGetLastError() != hipSuccess) {
    dev_free(p);
    return;
  }
  ds->csr_dense = (uint16_t*)p;
  ds->csr_dense_state = 1;
}

template <class T>
int32_t launch_csr_dense_mfma(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C, T* P) {
  return launch_bf16x3_rows<T, 2>(ctx, reinterpret_cast<const __hip_bfloat16*>(ds->csr_dense), ds->n, ds->d,
                                  R_dev, C, P);
}

template <class T>
int32_t launch_exact_dense(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                           T* P, const CodeOut* co, bool* codes_written) {
  const int64_t n = ds->n;
  const int d = ds->d;
  constexpr int CB = 32;
  const int nblk = (C + CB - 1) / CB;
  DevBuf<T> Rt;
  RPT_TRY(Rt.alloc((size_t)nblk * d * CB));
  hipLaunchKernelGGL(transpose_R<T>, dim3(256), dim3(256), 0, ctx->stream, R_dev, C, d, CB, Rt.p);
  const int64_t blocks = (n + 255) / 256;
  for (int b = 0; b < nblk; ++b) {
    const int c0 = b * CB;
    const int ncol = C - c0 < CB ? C - c0 : CB;
    ProfScope ps(ctx, RPT_PROF_PROJECT);
    unsigned int cmask = 0;  // columns of this block whose level is streamed (codes.h)
    if (co)
      for (int c = 0; c < ncol; ++c)
        if ((c0 + c) % co->L < co->Lc) cmask |= 1u << c;
    if (d == 128) {
      const int64_t nt128 = (n + 127) / 128;
      int64_t lb = (nt128 + 3) / 4;
      if (lb > (int64_t)ctx->n_cu * 2) lb = (int64_t)ctx->n_cu * 2;
      hipLaunchKernelGGL((proj_exact_lds<T, 128, 8>), dim3((unsigned)lb), dim3(256), 0, ctx->stream,
                         (const T*)ds->X, n, Rt.p + (size_t)b * d * CB, P + (int64_t)c0 * n, n, ncol,
                         nt128, co ? co->codes + (int64_t)c0 * co->ld : (uint16_t*)nullptr,
                         co ? co->ld : (int64_t)0, co ? co->mm : (const unsigned long long*)nullptr,
                         cmask);
    } else {
      hipLaunchKernelGGL((proj_exact<T, CB, 32>), dim3((unsigned)blocks), dim3(256), 0,
                         ctx->stream, (const T*)ds->X, n, d, Rt.p + (size_t)b * d * CB,
                         P + (int64_t)c0 * n, n, ncol);
    }
  }
  RPT_HIP(hipGetLastError());
  if (co && codes_written && d == 128) *codes_written = true;
  return RPT_OK;  // Rt returns to the stream-ordered allocator
}

// split point of every CSR row at column k_mid, computed once per dataset
static int32_t ensure_csr_split(rpt_ctx* ctx, const rpt_dataset* ds, int k_mid) {
  if (ds->csr_split && ds->csr_split_k == k_mid) return RPT_OK;
  if (ds->csr_split) dev_free(ds->csr_split);
  ds->csr_split = nullptr;
  void* p = nullptr;
  if (dev_alloc(&p, (size_t)(ds->n > 0 ? ds->n : 1) * 8) != hipSuccess)
    return fail(RPT_E_NOMEM, "CSR split points");
  int64_t blocks = (ds->n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(csr_split_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ds->rowptr,
                     ds->col, ds->n, k_mid, (int64_t*)p);
  if (hipGetLastError() != hipSuccess) {
    dev_free(p);
    return fail(RPT_E_HIP, "csr_split_kernel launch");
  }
  ds->csr_split = (int64_t*)p;
  ds->csr_split_k = k_mid;
  return RPT_OK;
}

template <class T>
int32_t launch_csr(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C, T* P,
                   bool fused) {
  const int64_t n = ds->n;
  const int d = ds->d;
  // ---- 32 hyperplanes per pass over the CSR arrays (whole rows, or two column halves) ----
  const size_t tile32 = (size_t)d * Csr32<T>::kRowBytes;
  const int k_mid = (d + 1) / 2;
  const size_t half32 = (size_t)k_mid * Csr32<T>::kRowBytes;
  const size_t kLdsMax = 144 * 1024;
  if (C > 16 && half32 <= kLdsMax && !ctx->opt.proj_narrow) {
    const bool whole = tile32 <= kLdsMax;
    constexpr int CB = 32;
    const int nblk = (C + CB - 1) / CB;
    DevBuf<T> Rt;
    RPT_TRY(Rt.alloc((size_t)nblk * d * CB));
    hipLaunchKernelGGL(transpose_R<T>, dim3(256), dim3(256), 0, ctx->stream, R_dev, C, d, CB, Rt.p);
    if (!whole) RPT_TRY(ensure_csr_split(ctx, ds, k_mid));
    const size_t smem = whole ? tile32 : half32;
    auto kern = fused ? proj_csr_lds32<T, true> : proj_csr_lds32<T, false>;
    static DeviceOnce attr_once[2];
    RPT_TRY(attr_once[fused].run(ctx->device, [&]() -> int32_t {
      RPT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kLdsMax));
      return RPT_OK;
    }));
    for (int b = 0; b < nblk; ++b) {
      const int c0 = b * CB;
      const int ncol = C - c0 < CB ? C - c0 : CB;
      const T* rt = Rt.p + (size_t)b * d * CB;
      T* pb = P + (int64_t)c0 * n;
      if (whole) {
        ProfScope ps(ctx, RPT_PROF_PROJECT);
        hipLaunchKernelGGL(kern, dim3((unsigned)ctx->n_cu), dim3(kCsr32Threads), smem, ctx->stream,
                           ds->rowptr, (const int64_t*)nullptr, ds->col, (const T*)ds->val, n, 0, d,
                           rt, pb, n, ncol, 0);
      } else {
        {
          ProfScope ps(ctx, RPT_PROF_PROJECT);
          hipLaunchKernelGGL(kern, dim3((unsigned)ctx->n_cu), dim3(kCsr32Threads), smem,
                             ctx->stream, ds->rowptr, (const int64_t*)ds->csr_split, ds->col,
                             (const T*)ds->val, n, k_mid, d, rt, pb, n, ncol, 1);
        }
        {
          ProfScope ps(ctx, RPT_PROF_PROJECT);
          hipLaunchKernelGGL(kern, dim3((unsigned)ctx->n_cu), dim3(kCsr32Threads), smem,
                             ctx->stream, ds->rowptr, (const int64_t*)ds->csr_split, ds->col,
                             (const T*)ds->val, n, 0, k_mid, rt, pb, n, ncol, 2);
        }
      }
    }
    RPT_HIP(hipGetLastError());
    return RPT_OK;  // Rt returns to the stream-ordered allocator
  }
  constexpr int CB = 16;
  const int nblk = (C + CB - 1) / CB;
  DevBuf<T> Rt;
  RPT_TRY(Rt.alloc((size_t)nblk * d * CB));
  hipLaunchKernelGGL(transpose_R<T>, dim3(256), dim3(256), 0, ctx->stream, R_dev, C, d, CB, Rt.p);
  const size_t tile = (size_t)d * kCsrLd * sizeof(T);  // padded rows, see kCsrLd
  const bool lds_path = tile <= 144 * 1024;  // else the hyperplane tile is read through L2
  if (lds_path && tile > 64 * 1024)
    RPT_HIP(hipFuncSetAttribute((const void*)proj_csr_lds<T>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile));
  const int64_t blocks = (n + 15) / 16;  // 256 threads = 4 waves x 4 rows
  for (int b = 0; b < nblk; ++b) {
    const int c0 = b * CB;
    const int ncol = C - c0 < CB ? C - c0 : CB;
    ProfScope ps(ctx, RPT_PROF_PROJECT);
    if (lds_path)
      hipLaunchKernelGGL(proj_csr_lds<T>, dim3((unsigned)ctx->n_cu), dim3(1024), tile, ctx->stream,
                         ds->rowptr, ds->col, (const T*)ds->val, n, d, Rt.p + (size_t)b * d * CB,
                         P + (int64_t)c0 * n, n, ncol);
    else
      hipLaunchKernelGGL((proj_csr<T, CB>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                         ds->rowptr, ds->col, (const T*)ds->val, n, Rt.p + (size_t)b * d * CB,
                         P + (int64_t)c0 * n, n, ncol);
  }
  RPT_HIP(hipGetLastError());
  return RPT_OK;  // Rt returns to the stream-ordered allocator
}

}  // namespace

// whether project_columns(.., co) would write codes for this dataset in this mode (the callers
// skip the sample pass otherwise): the pipelined dense kernels only
bool project_writes_codes(const rpt_ctx* ctx, const rpt_dataset* ds, int32_t mode) {
  if (ds->csr) return false;
  if (mode == RPT_PROJ_AUTO) mode = ds->dtype == RPT_F64 ? RPT_PROJ_EXACT : RPT_PROJ_MFMA;
  if (mode == RPT_PROJ_EXACT) return ds->dtype != RPT_BF16 && ds->d == 128;
  const int piece = 16 / (int)dtype_size(ds->dtype);
  if (ds->d % piece != 0) return false;
  // bf16 rows on the bf16 pipe: proj_bf16x3 HAS a code epilogue (option proj_bf16_codes), but its 8-byte code runs
  // are half lines and the epilogue runs with the matrix pipe idle: +3.0 ms of projection per C5 shard build
  // against the 2.4 ms of the coalesced pass over the stored keys (pcode_kernel) — the pass stays the default
  if (ds->dtype == RPT_BF16 && !ctx->opt.proj_bf16_codes)
    return !(ds->d % 8 == 0 && (reinterpret_cast<uintptr_t>(ds->X) & 15) == 0 && !ctx->opt.proj_bf16_f32);
  return true;
}

int32_t project_columns(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                        int32_t mode, void* P_dev, const CodeOut* co, bool* codes_written) {
  if (codes_written) *codes_written = false;
  RPT_ARG(mode == RPT_PROJ_AUTO || mode == RPT_PROJ_EXACT || mode == RPT_PROJ_MFMA,
          "unknown projection mode");
  if (ds->n == 0) return RPT_OK;
  if (mode == RPT_PROJ_AUTO) mode = ds->dtype == RPT_F64 ? RPT_PROJ_EXACT : RPT_PROJ_MFMA;
  if (ds->csr) {
    // CSR x dense is not MFMA shaped: both modes use the segmented kernel, RPT_PROJ_MFMA (the
    // tolerance mode) with one fused multiply-add per term instead of the reference's two roundings
    const bool fused = mode == RPT_PROJ_MFMA;
    // ... and, when the rows fit dense-ified (d % 8 == 0, memory), the tolerance mode runs on the
    // matrix pipe (launch_csr_dense_mfma); few rows (query batches) stay on the segmented kernel
    if (fused && !ctx->opt.proj_csr_nodense && ds->n >= 65536) {
      if (ds->dtype == RPT_F64) ensure_csr_dense<double>(ctx, ds);
      else ensure_csr_dense<float>(ctx, ds);
      if (ds->csr_dense_state == 1) {
        if (ds->dtype == RPT_F64) return launch_csr_dense_mfma<double>(ctx, ds, R_dev, C, (double*)P_dev);
        return launch_csr_dense_mfma<float>(ctx, ds, R_dev, C, (float*)P_dev);
      }
    }
    if (ds->dtype == RPT_F64) return launch_csr<double>(ctx, ds, R_dev, C, (double*)P_dev, fused);
    return launch_csr<float>(ctx, ds, R_dev, C, (float*)P_dev, fused);
  }
  if (mode == RPT_PROJ_EXACT) {
    if (ds->dtype == RPT_F64)
      return launch_exact_dense<double>(ctx, ds, R_dev, C, (double*)P_dev, co, codes_written);
    if (ds->dtype == RPT_F32)
      return launch_exact_dense<float>(ctx, ds, R_dev, C, (float*)P_dev, co, codes_written);
    return fail(RPT_E_UNSUPPORTED, "exact-order projection is defined for f64/f32 data");
  }
  if (ds->dtype == RPT_F64)
    return launch_mfma<double, double>(ctx, ds, R_dev, C, (double*)P_dev, co, codes_written);
  if (ds->dtype == RPT_F32)
    return launch_mfma<float, float>(ctx, ds, R_dev, C, (float*)P_dev, co, codes_written);
  // bf16: three bf16 MFMAs per tile against the split hyperplanes when the rows allow 16-byte
  // fragment loads, else the f32-MFMA kernels on converted inputs (option proj_bf16_f32: force them)
  if (ds->d % 8 == 0 && (reinterpret_cast<uintptr_t>(ds->X) & 15) == 0 && !ctx->opt.proj_bf16_f32)
    return launch_bf16x3(ctx, ds, R_dev, C, (float*)P_dev, co, codes_written);
  return launch_mfma<__hip_bfloat16, float>(ctx, ds, R_dev, C, (float*)P_dev, co, codes_written);
}

}  // namespace rpt
