// What v_permlane16_swap / v_permlane32_swap do on gfx950 (a lane-row transpose primitive for the line FFT's second exchange).
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/permlane_swap tools/micro/permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  u2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r32.x; out[64 + l] = r32.y; out[128 + l] = r16.x; out[192 + l] = r16.y;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32.x (new vdst)", "swap32.y (new vsrc)", "swap16.x (new vdst)", "swap16.y (new vsrc)"};
  for (int r = 0; r < 4; ++r) { printf("%s: rows of 16 lanes start with", names[r]); for (int g = 0; g < 4; ++g) printf(" %u", h[64 * r + 16 * g]); printf("\n"); }
  return 0;
}
