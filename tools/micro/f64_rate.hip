// Issue-rate check of the f64 VALU instructions on gfx950: wave64 v_mul_f64 / v_add_f64 / v_fma_f64.
// 8 independent chains per lane, 4 waves per SIMD on every CU; prints cycles per wave-instruction
// per SIMD (4 = full rate: 16 lanes per clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double x, double y) {
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = x + threadIdx.x + i;
  const double one = 1.0, nz = -0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      double p;
      if (MODE == 0) {  // mul + add
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(a[i]), "v"(y));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(p), "v"(a[i]));
      } else if (MODE == 1) {  // the same values from two FMAs
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(a[i]), "v"(y), "v"(nz));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(p), "v"(one), "v"(a[i]));
      } else if (MODE == 2) {  // one fused
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(y));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[i]), "v"(y));
      } else if (MODE == 3) {  // mul only
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
      } else {  // add only
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* out, int blocks) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 1.0000001);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0000001);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks*4 waves / (CUs*4 SIMDs) waves, each iters*16 instructions
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const double waves_per_simd = (double)blocks * 4 / (pr.multiProcessorCount * 4);
  const double instr = waves_per_simd * iters * 16.0;
  printf("%-12s %8.3f ms  %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
  double* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  run<0>("mul+add", out, blocks);
  run<1>("fma+fma", out, blocks);
  run<2>("fma", out, blocks);
  run<3>("mul", out, blocks);
  run<4>("add", out, blocks);
  hipFree(out);
  return 0;
}
