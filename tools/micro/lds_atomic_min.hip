// LDS returning atomic-min throughput: 64-bit keys against 32-bit keys (what the NN scatter kernel pays per candidate)
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_atomic_min tools/micro/lds_atomic_min.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename K, int MODE>   // MODE 0: returning atomicMin, 1: non-returning, 2: plain read + compare only
__global__ void __launch_bounds__(256) k(unsigned* out, int iters) {
  __shared__ K key[4096 + 8];
  for (int i = threadIdx.x; i < 4096; i += 256) key[i] = (K)~(K)0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned q = (s >> 12) & 4095u;
    const K mine = ((K)(s | 1u) << (sizeof(K) * 8 - 32)) | (K)threadIdx.x;
    if (MODE == 0) { const K old = atomicMin(&key[q], mine); acc += (unsigned)(old < mine); }
    else if (MODE == 1) { atomicMin(&key[q], mine); }
    else { acc += (unsigned)(key[q] < mine); }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename K, int MODE>
void run(const char* name, unsigned* out) {
  const int iters = 2048, grid = 256 * 12;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s %.3f ms  %.1f G lane-ops/s\n", name, ms, (double)grid * 256 * iters / ms / 1e6); fflush(stdout);
}

int main() {
  unsigned* out; CK(hipMalloc(&out, 64));
  run<unsigned long long, 0>("u64 atomicMin returning", out);
  run<unsigned long long, 1>("u64 atomicMin", out);
  run<unsigned long long, 2>("u64 read + compare", out);
  run<unsigned, 0>("u32 atomicMin returning", out);
  run<unsigned, 1>("u32 atomicMin", out);
  run<unsigned, 2>("u32 read + compare", out);
  return 0;
}
