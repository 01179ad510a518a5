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

#include "common.h"

namespace rpt {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

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
};
template <>
struct Mfma<float> {
  typedef float4_t acc_t;
  __device__ static acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // v_mfma_f32_16x16x4_f32 C/D map: col = lane&15, row = 4*(lane>>4) + reg
  __device__ static int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <class TIn>
__device__ inline float to_f32(TIn v);
template <>
__device__ inline float to_f32<float>(float v) { return v; }
template <>
__device__ inline float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

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
// CSR rows x dense-ified hyperplanes.  16 lanes per row (one lane per hyperplane of the
// block, CB = 16), 4 rows per wave; every lane walks its row's nonzeros from the last to
// the first: acc = val*r[col] + acc  (innerSS, Internal.hs:353-366: only matching indices
// contribute; a zero of the dense-ified hyperplane adds an exact zero).
// Rt[d][16] k-major: the 16 lanes of a row read 128 contiguous bytes (L2 resident).
// ---------------------------------------------------------------------------------------
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

template <class TIn, class TC>
int32_t launch_mfma(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                    TC* P) {
  const int64_t n = ds->n;
  const int64_t ntiles = (n + 15) / 16;
  constexpr int WAVES = 4;
  int64_t blocks = (ntiles + WAVES - 1) / WAVES;
  const int64_t cap = (int64_t)ctx->n_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int ncol = C - c0 < 32 ? C - c0 : 32;
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

template <class T>
int32_t launch_exact_dense(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                           T* P) {
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
    hipLaunchKernelGGL((proj_exact<T, CB, 32>), dim3((unsigned)blocks), dim3(256), 0,
                       ctx->stream, (const T*)ds->X, n, d, Rt.p + (size_t)b * d * CB,
                       P + (int64_t)c0 * n, n, ncol);
  }
  RPT_HIP(hipGetLastError());
  RPT_HIP(hipStreamSynchronize(ctx->stream));  // Rt is freed on return
  return RPT_OK;
}

template <class T>
int32_t launch_csr(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C, T* P) {
  const int64_t n = ds->n;
  const int d = ds->d;
  constexpr int CB = 16;
  const int nblk = (C + CB - 1) / CB;
  DevBuf<T> Rt;
  RPT_TRY(Rt.alloc((size_t)nblk * d * CB));
  hipLaunchKernelGGL(transpose_R<T>, dim3(256), dim3(256), 0, ctx->stream, R_dev, C, d, CB, Rt.p);
  const int64_t blocks = (n + 15) / 16;  // 256 threads = 4 waves x 4 rows
  for (int b = 0; b < nblk; ++b) {
    const int c0 = b * CB;
    const int ncol = C - c0 < CB ? C - c0 : CB;
    hipLaunchKernelGGL((proj_csr<T, CB>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       ds->rowptr, ds->col, (const T*)ds->val, n, Rt.p + (size_t)b * d * CB,
                       P + (int64_t)c0 * n, n, ncol);
  }
  RPT_HIP(hipGetLastError());
  RPT_HIP(hipStreamSynchronize(ctx->stream));
  return RPT_OK;
}

}  // namespace

int32_t project_columns(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                        int32_t mode, void* P_dev) {
  RPT_ARG(mode == RPT_PROJ_AUTO || mode == RPT_PROJ_EXACT || mode == RPT_PROJ_MFMA,
          "unknown projection mode");
  if (ds->n == 0) return RPT_OK;
  if (mode == RPT_PROJ_AUTO) mode = ds->dtype == RPT_F64 ? RPT_PROJ_EXACT : RPT_PROJ_MFMA;
  if (ds->csr) {
    // CSR x dense is not MFMA shaped; both modes use the segmented kernel
    if (ds->dtype == RPT_F64) return launch_csr<double>(ctx, ds, R_dev, C, (double*)P_dev);
    return launch_csr<float>(ctx, ds, R_dev, C, (float*)P_dev);
  }
  if (mode == RPT_PROJ_EXACT) {
    if (ds->dtype == RPT_F64) return launch_exact_dense<double>(ctx, ds, R_dev, C, (double*)P_dev);
    if (ds->dtype == RPT_F32) return launch_exact_dense<float>(ctx, ds, R_dev, C, (float*)P_dev);
    return fail(RPT_E_UNSUPPORTED, "exact-order projection is defined for f64/f32 data");
  }
  if (ds->dtype == RPT_F64) return launch_mfma<double, double>(ctx, ds, R_dev, C, (double*)P_dev);
  if (ds->dtype == RPT_F32) return launch_mfma<float, float>(ctx, ds, R_dev, C, (float*)P_dev);
  return launch_mfma<__hip_bfloat16, float>(ctx, ds, R_dev, C, (float*)P_dev);
}

}  // namespace rpt
