// Cost of sorting 1024 u32 keys held 16 per lane by ONE wave (index = lane * 16 + r: the four
// closest exchange distances are register-to-register, the other six cross lanes), the core of
// the sort-based wave subtree kernel (split.hip).  Prints ms for `waves` independent sorts x `reps`
// sorts per wave and checks the result of the last one on the host.
//   hipcc -O3 --offload-arch=gfx950 wsort_bench.hip -o wsort_bench && ./wsort_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int E = 16;

#ifndef VARIANT
#define VARIANT 1
#endif
template <int D>
__device__ __forceinline__ unsigned int lane_xor(unsigned int v) {
#if VARIANT == 0
  return (unsigned int)__shfl_xor((int)v, D);
#else
  if constexpr (D == 1) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
  else if constexpr (D == 2) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
  else if constexpr (D == 4) return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
  else if constexpr (D == 8) return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);
  else if constexpr (D == 16) return (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
  else return (unsigned int)__shfl_xor((int)v, 32);
#endif
}

// compare-exchange across lanes at lane distance D; upper = this lane keeps the larger key
template <int D>
__device__ __forceinline__ void cross_stage(unsigned int (&k)[E], bool upper) {
#if VARIANT == 1
  if constexpr (D == 32) {  // v_permlane32_swap: both halves see (low half's key, high half's key)
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(k[r], k[r], false, false);
      const unsigned int a = sw[0], b = sw[1];
      const unsigned int mn = a < b ? a : b, mx = a < b ? b : a;
      k[r] = upper ? mx : mn;
    }
    return;
  }
#endif
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const unsigned int o = lane_xor<D>(k[r]);
    const unsigned int mn = k[r] < o ? k[r] : o, mx = k[r] < o ? o : k[r];
    k[r] = upper ? mx : mn;
  }
}

template <int J>
__device__ __forceinline__ void reg_stage(unsigned int (&k)[E]) {
#pragma unroll
  for (int r = 0; r < E; ++r)
    if ((r & J) == 0) {
      const unsigned int a = k[r], b = k[r + J];
      k[r] = a < b ? a : b;
      k[r + J] = a < b ? b : a;
    }
}

// keys of lanes whose block sorts descending are kept complemented, so every exchange is an
// ascending one; phase KK (block size) decides the direction: bit KK of index = lane * 16 + r
template <int KK>
__device__ __forceinline__ void phase(unsigned int (&k)[E], int lane, unsigned int& flip) {
  if constexpr (KK < 16) {
    // direction depends on r only: descending blocks are handled by swapping the roles
#pragma unroll
    for (int r = 0; r < E; ++r)
      if (r & KK) k[r] = ~k[r];
    if constexpr (KK >= 16) reg_stage<8>(k);
    if constexpr (KK >= 8) reg_stage<(KK >= 8 ? 4 : 1)>(k);
    if constexpr (KK >= 4) reg_stage<(KK >= 4 ? 2 : 1)>(k);
    reg_stage<1>(k);
#pragma unroll
    for (int r = 0; r < E; ++r)
      if (r & KK) k[r] = ~k[r];
  } else {
    const unsigned int want = (KK < 1024 && ((lane * 16) & KK)) ? ~0u : 0u;
    const unsigned int x = want ^ flip;
    flip = want;
#pragma unroll
    for (int r = 0; r < E; ++r) k[r] ^= x;
    if constexpr (KK >= 1024) cross_stage<32>(k, (lane & 32) != 0);
    if constexpr (KK >= 512) cross_stage<16>(k, (lane & 16) != 0);
    if constexpr (KK >= 256) cross_stage<8>(k, (lane & 8) != 0);
    if constexpr (KK >= 128) cross_stage<4>(k, (lane & 4) != 0);
    if constexpr (KK >= 64) cross_stage<2>(k, (lane & 2) != 0);
    if constexpr (KK >= 32) cross_stage<1>(k, (lane & 1) != 0);
    reg_stage<8>(k);
    reg_stage<4>(k);
    reg_stage<2>(k);
    reg_stage<1>(k);
  }
}

__device__ __forceinline__ void sort1024(unsigned int (&k)[E], int lane) {
  unsigned int flip = 0;
  phase<2>(k, lane, flip);
  phase<4>(k, lane, flip);
  phase<8>(k, lane, flip);
  phase<16>(k, lane, flip);
  phase<32>(k, lane, flip);
  phase<64>(k, lane, flip);
  phase<128>(k, lane, flip);
  phase<256>(k, lane, flip);
  phase<512>(k, lane, flip);
  phase<1024>(k, lane, flip);  // want = 0: keys leave un-complemented
}

__global__ __launch_bounds__(256) void bench(const unsigned int* __restrict__ in,
                                             unsigned int* __restrict__ out, int reps) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned int k[E];
  const uint4* p = reinterpret_cast<const uint4*>(in + w * 1024 + lane * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = p[q];
    k[4 * q] = v.x;
    k[4 * q + 1] = v.y;
    k[4 * q + 2] = v.z;
    k[4 * q + 3] = v.w;
  }
  for (int it = 0; it < reps; ++it) {
    if (it) {
#pragma unroll
      for (int r = 0; r < E; ++r) k[r] = k[r] * 2654435761u + (unsigned int)(lane * 16 + r);
    }
    sort1024(k, lane);
  }
  uint4* o = reinterpret_cast<uint4*>(out + w * 1024 + lane * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = make_uint4(k[4 * q], k[4 * q + 1], k[4 * q + 2], k[4 * q + 3]);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 32768;
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  std::vector<unsigned int> h((size_t)waves * 1024);
  unsigned int s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = s;
  }
  unsigned int *din, *dout;
  hipMalloc(&din, h.size() * 4);
  hipMalloc(&dout, h.size() * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pass = 0; pass < 3; ++pass) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench, dim3(waves / 4), dim3(256), 0, 0, din, dout, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("waves %d, %d sorts of 1024 u32 per wave: %.3f ms\n", waves, reps, ms);
  }
  hipLaunchKernelGGL(bench, dim3(waves / 4), dim3(256), 0, 0, din, dout, 1);
  std::vector<unsigned int> o(h.size());
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int w = 0; w < waves; ++w) {
    std::vector<unsigned int> ref(h.begin() + (size_t)w * 1024, h.begin() + (size_t)(w + 1) * 1024);
    std::sort(ref.begin(), ref.end());
    for (int i = 0; i < 1024; ++i) bad += ref[i] != o[(size_t)w * 1024 + i];
  }
  printf("mismatches after one sort: %zu\n", bad);
  return bad != 0;
}
