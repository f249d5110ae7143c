// Ceiling of the y pass's access pattern: read T contiguous lines, write T-element segments at row stride OK.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/transpose_bw tools/micro/transpose_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float2 cf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int T, int KC, bool NT, int GROUP>
__global__ void __launch_bounds__(1024) tr(const cf* __restrict__ in, cf* __restrict__ out, int A, int B, int NC,
                                           long long SA, long long SB, long long OB, long long OK) {
  __shared__ cf buf[T * KC + T * KC / 32 + 64];
  constexpr int PER = T * KC / 1024;
  const int tiles = A / T, chunks = NC / KC;
  unsigned bid = blockIdx.x;
  if (GROUP > 1) {
    const unsigned span = 8 * GROUP;
    if (bid / span < gridDim.x / span) { const unsigned base = (bid / span) * span, h = bid % span; bid = base + GROUP * (h % 8) + (h / 8); }
  }
  const int a0 = (bid % tiles) * T;
  const int c = (bid / tiles) % chunks;
  const int b = bid / (tiles * chunks);
  const cf* src = in + (long long)b * SB + (long long)a0 * SA + c * KC;
  cf v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = threadIdx.x + i * 1024, t = e / KC, k = e % KC;
    if (NT) { double raw = __builtin_nontemporal_load((const double*)&src[(long long)t * SA + k]); v[i] = *(cf*)&raw; }
    else v[i] = src[(long long)t * SA + k];
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = threadIdx.x + i * 1024, t = e / KC, k = e % KC;
    const int s = k * T + (t ^ ((k >> (T >= 16 ? 0 : (T == 8 ? 1 : 2))) & (T - 1)));
    buf[s] = v[i];
  }
  __syncthreads();
  cf* dst = out + (long long)b * OB + (long long)(c * KC) * OK + a0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = threadIdx.x + i * 1024, k = e / T, t = e % T;
    const int s = k * T + (t ^ ((k >> (T >= 16 ? 0 : (T == 8 ? 1 : 2))) & (T - 1)));
    const cf val = buf[s];
    if (NT) __builtin_nontemporal_store(*(const double*)&val, (double*)&dst[(long long)k * OK + t]);
    else dst[(long long)k * OK + t] = val;
  }
}

template <int T, int KC, bool NT, int GROUP>
void run(const char* name, const cf* in, cf* out, int A, int B, int NC) {
  const long long SA = (long long)B * NC, SB = NC, OK = A, OB = (long long)NC * A;
  const long long grid = (long long)(A / T) * (NC / KC) * B;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((tr<T, KC, NT, GROUP>), dim3((unsigned)grid), dim3(1024), 0, 0, in, out, A, B, NC, SA, SB, OB, OK);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((tr<T, KC, NT, GROUP>), dim3((unsigned)grid), dim3(1024), 0, 0, in, out, A, B, NC, SA, SB, OB, OK);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = 16.0 * A * B * NC;
  printf("%-28s T=%2d KC=%4d nt=%d grp=%d  %.3f ms  %.0f GB/s\n", name, T, KC, (int)NT, GROUP, ms, bytes / ms / 1e6); fflush(stdout);
}

int main(int argc, char** argv) {
  const int A = argc > 1 ? atoi(argv[1]) : 512, B = 1024, NC = 2048;
  const size_t n = (size_t)A * B * NC;
  cf *in, *out; CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
  CK(hipMemset(in, 1, n * 8)); CK(hipMemset(out, 0, n * 8));
  printf("A=%d (row stride %d B) B=%d NC=%d  %.1f GB each way\n", A, A * 8, B, NC, n * 8 / 1e9);
  run<4, 2048, false, 4>("seg 32 B", in, out, A, B, NC);
  run<8, 1024, false, 2>("seg 64 B", in, out, A, B, NC);
  run<8, 2048, false, 2>("seg 64 B whole lines", in, out, A, B, NC);
  run<8, 2048, false, 1>("seg 64 B whole, ungrouped", in, out, A, B, NC);
  run<8, 2048, true, 2>("seg 64 B whole lines nt", in, out, A, B, NC);
  run<16, 1024, false, 1>("seg 128 B", in, out, A, B, NC);
  run<16, 1024, true, 1>("seg 128 B nt", in, out, A, B, NC);
  run<16, 512, false, 1>("seg 128 B kc512", in, out, A, B, NC);
  run<32, 512, false, 1>("seg 256 B", in, out, A, B, NC);
  run<32, 512, true, 1>("seg 256 B nt", in, out, A, B, NC);
  run<64, 256, false, 1>("seg 512 B", in, out, A, B, NC);
  return 0;
}
