// How long does a wave take to issue the stores of one projection tile — 16*CBT rows of P, 128
// contiguous bytes in each — as a function of the distance between the rows?  (P[c][N]: rows are
// N * 8 bytes apart; a point-blocked layout would bring them within 32 KB.)
// Every wave of a full grid writes `tiles` tiles round-robin; per tile 2 x CBT store instructions
// of 64 lanes x 16 B (lane: row l & 15 of the column tile, bytes 32 (l >> 4) + 16 k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void stores(char* base, long long row_stride, long long tile_stride,
                                              int cbt, int tiles, long long wave_tiles_stride,
                                              unsigned long long* ticks, int pattern, int nsleep) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 8 + (threadIdx.x >> 6);
  unsigned long long acc = 0;
  float4 v = make_float4(lane, 1, 2, 3);
  for (int t = 0; t < tiles; ++t) {
    char* tb = base + (wave + (long long)t * wave_tiles_stride) * tile_stride;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int h = 0; h < cbt; ++h)
      for (int k = 0; k < 2; ++k) {
        char* p = tb + (long long)(h * 16 + (lane & 15)) * row_stride +
                  (pattern ? 16 * (lane >> 4) + 64 * k : 32 * (lane >> 4) + 16 * k);
        *reinterpret_cast<float4*>(p) = v;
      }
    acc += __builtin_readcyclecounter() - c0;
    // stand-in for the matrix phase: the stores drain meanwhile
    for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(100);
  }
  if (lane == 0) atomicAdd(ticks, acc);
}

int main() {
  const long long N = 1000000;
  const int cbt = 6, tiles = 30;
  const long long waves = 256 * 8;
  char* buf;
  const size_t bytes = (size_t)416 * N * 8;  // the C2 projection matrix: 3.3 GB
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  unsigned long long* ticks;
  hipMalloc(&ticks, 8);
  struct Case { const char* name; long long row_stride, tile_stride; } cases[] = {
      {"rows N*8 B apart (P[c][N])", N * 8, 128},
      {"rows 2 MB apart", 2 << 20, 128},
      {"rows 256 KB apart", 256 << 10, 128},
      {"rows 32 KB apart (4096-point blocks)", 32 << 10, 128},
      {"rows 128 B apart (tile-contiguous)", 128, 128 * 96},
  };
  for (int pattern = 0; pattern < 2; ++pattern)
  for (int nsleep = 0; nsleep <= 2; nsleep += 2)
  for (const Case& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(ticks, 0, 8);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(stores, dim3(256), dim3(512), 0, 0, buf, c.row_stride, c.tile_stride, cbt, tiles,
                         waves, ticks, pattern, nsleep);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
      if (rep) printf("%s sleep %d  %-40s %8.0f ticks per tile (12 store instructions), kernel %.3f ms = %.2f TB/s\n",
                      pattern ? "16 B pieces adjacent " : "16 B pieces interleaved", nsleep, c.name,
                      (double)h / (waves * tiles), ms, waves * tiles * 12288.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
