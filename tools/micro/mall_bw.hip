// Does a consumer kernel read a producer's output from the memory-side cache (Infinity Cache) instead of HBM?
// Producer writes chunk i (S bytes), consumer reads chunk i, for many distinct chunks of a 16 GB arena.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/mall_bw tools/micro/mall_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) producer(double2* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 v = make_double2((double)i, 1.0);
    if (NT) { __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y); }
    else dst[i] = v;
  }
}
template <bool NT>
__global__ void __launch_bounds__(256) consumer(const double2* __restrict__ src, size_t n, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 v;
    if (NT) { v.x = __builtin_nontemporal_load(&src[i].x); v.y = __builtin_nontemporal_load(&src[i].y); }
    else v = src[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345) *sink = acc;
}

template <bool NTW, bool NTR>
void run(char* arena, size_t arena_bytes, size_t S, double* sink) {
  const size_t chunks = arena_bytes / S, n = S / 16;
  const int grid = 256 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // interleaved: write chunk i, read chunk i
  float ms_w = 0, ms_r = 0, ms;
  for (size_t i = 0; i < chunks; ++i) {
    double2* p = (double2*)(arena + i * S);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(producer<NTW>, dim3(grid), dim3(256), 0, 0, p, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms_w += ms;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(consumer<NTR>, dim3(grid), dim3(256), 0, 0, p, n, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms_r += ms;
  }
  // cold read: read every chunk again, in the same order (the arena is far larger than any cache)
  float ms_c = 0;
  for (size_t i = 0; i < chunks; ++i) {
    double2* p = (double2*)(arena + i * S);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(consumer<NTR>, dim3(grid), dim3(256), 0, 0, p, n, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms_c += ms;
  }
  const double gb = (double)chunks * S / 1e9;
  printf("S=%6zu MB ntw=%d ntr=%d  write %.0f GB/s  read-after-write %.0f GB/s  cold read %.0f GB/s\n", S >> 20, (int)NTW, (int)NTR,
         gb / ms_w * 1e3, gb / ms_r * 1e3, gb / ms_c * 1e3);
  fflush(stdout);
}

int main() {
  const size_t arena_bytes = (size_t)16 << 30;
  char* arena; CK(hipMalloc(&arena, arena_bytes)); CK(hipMemset(arena, 0, arena_bytes));
  double* sink; CK(hipMalloc(&sink, 8));
  for (size_t mb : {16, 32, 64, 128, 256, 512, 2048}) {
    run<false, false>(arena, arena_bytes, mb << 20, sink);
    run<true, false>(arena, arena_bytes, mb << 20, sink);
    run<true, true>(arena, arena_bytes, mb << 20, sink);
  }
  return 0;
}
