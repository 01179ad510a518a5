// Measures the dense v_mfma_f64_16x16x4_f64 issue rate of the whole chip (the guide's peak table
// has no FP64 matrix row): every wave runs a chain-free stream of MFMAs on 8 accumulators.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// usage: mfma_f64_peak [waves per SIMD = 2] [iterations = 20000]
#include <cstdlib>
int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  const int blocks = p.multiProcessorCount * wps, iters = argc > 2 ? atoi(argv[2]) : 20000;
  double* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 8 * 2048.0;
  printf("CUs %d clock %d MHz: %.3f ms, %.2f TFLOP/s f64 MFMA (16x16x4), %.1f cycles per MFMA per SIMD at the reported clock\n",
         p.multiProcessorCount, p.clockRate / 1000, ms, flops / (ms * 1e-3) / 1e12,
         (ms * 1e-3) * (p.clockRate * 1e3) / ((double)iters * 8 * wps));
  printf("  (%d waves per SIMD, %d MFMAs per wave; a short run shows the rate before the clock settles)\n", wps, iters * 8);
  return 0;
}
