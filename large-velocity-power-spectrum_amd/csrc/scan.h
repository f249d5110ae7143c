// Exclusive prefix sum of 32-bit counts (n entries -> n+1 offsets): three small kernels.
// Shared by the NN cell list (nn.hip) and the brick deposit (deposit.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// exclusive scan of `count` into `start` (n+1 entries), three small kernels
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__global__ void __launch_bounds__(SCAN_BLOCK)
    scan_tile_sums(const unsigned* __restrict__ count, long long n, unsigned* __restrict__ tile_sum) {
  __shared__ unsigned red[SCAN_BLOCK / 64];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  unsigned s = 0;
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + (long long)k * SCAN_BLOCK + threadIdx.x;
    if (i < n) s += count[i];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int k = 0; k < SCAN_BLOCK / 64; ++k) t += red[k];
    tile_sum[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(1024) scan_tile_offsets(unsigned* __restrict__ tile_sum, long long ntiles) {
  // single workgroup: serial-over-chunks exclusive scan of the tile sums, in place
  __shared__ unsigned sh[1024];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long long base = 0; base < ntiles; base += 1024) {
    const long long i = base + threadIdx.x;
    const unsigned v = (i < ntiles) ? tile_sum[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      unsigned add = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0u;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    const unsigned incl = sh[threadIdx.x];
    const unsigned c = carry;
    if (i < ntiles) tile_sum[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
}

// (count and start may be the SAME array -- the NN cell list scans its table in place: no __restrict__ on them; a thread loads
//  its SCAN_ITEMS counts before it stores, and no thread stores into another thread's counts)
__global__ void __launch_bounds__(SCAN_BLOCK)
    scan_apply(const unsigned* count, long long n, const unsigned* __restrict__ tile_off, unsigned* start) {
  // each thread owns SCAN_ITEMS consecutive counts
  __shared__ unsigned sh[SCAN_BLOCK];
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned v[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? count[base + k] : 0u;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
    unsigned add = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned run = tile_off[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) start[base + k] = run;
    run += v[k];
  }
  if (base <= n - 1 && n - 1 < base + SCAN_ITEMS) start[n] = run;  // total
}


// enqueue the three scan kernels; tile_scratch needs scan_tiles(n)+1 entries
inline long long scan_tiles(long long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
inline void launch_exclusive_scan(hipStream_t stream, const unsigned* count, long long n,
                                  unsigned* tile_scratch, unsigned* start) {
  const long long nt = scan_tiles(n);
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)nt), dim3(SCAN_BLOCK), 0, stream, count, n, tile_scratch);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(1024), 0, stream, tile_scratch, nt);
  hipLaunchKernelGGL(scan_apply, dim3((unsigned)nt), dim3(SCAN_BLOCK), 0, stream, count, n, tile_scratch, start);
}

}  // namespace
