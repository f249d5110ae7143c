// Un-fused forms of the binning stage, kept so that the reference's two-step API
// (_pair_power then _hist_sample, vpower/interp.py:1440-1482; pair_power / hist_sample,
// scripts/parallel_optimized.py:145-190) has a device implementation too.  The hot path
// never materialises these arrays: it bins straight out of the x pass (fft.hip).
#pragma clang fp contract(off)

#include "vps_internal.h"

namespace {

__global__ void __launch_bounds__(256)
    pair_k_kernel(const double* __restrict__ kaxes, int N, double* __restrict__ out) {
  const long long n3 = (long long)N * N * N;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  const int iz = (int)(i % N), iy = (int)((i / N) % N), ix = (int)(i / ((long long)N * N));
  const double kx = kaxes[ix], ky = kaxes[N + iy], kz = kaxes[2 * N + iz];
  double s = kx * kx;
  s = s + ky * ky;
  s = s + kz * kz;
  out[i] = sqrt(s);
}

// numpy.histogram with explicit edges: edges[i] <= k < edges[i+1], last bin right-closed
__global__ void __launch_bounds__(256)
    hist_pairs_kernel(const double* __restrict__ k, const double* __restrict__ w, long long n,
                      const double* __restrict__ edges, int nbins, double* __restrict__ psum,
                      unsigned long long* __restrict__ nsample) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* e = reinterpret_cast<double*>(smem_raw);
  double* hs = e + nbins + 1;
  unsigned* hc = reinterpret_cast<unsigned*>(hs + nbins);
  for (int i = threadIdx.x; i <= nbins; i += blockDim.x) e[i] = edges[i];
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    hs[i] = 0.0;
    hc[i] = 0u;
  }
  __syncthreads();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const double x = k[i];
    if (!(x >= e[0]) || !(x <= e[nbins])) continue;
    int lo = 0, hi = nbins;  // invariant: e[lo] <= x, and (hi == nbins or x < e[hi])
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (x >= e[mid]) lo = mid; else hi = mid;
    }
    atomicAdd(&hs[lo], w ? w[i] : 1.0);
    atomicAdd(&hc[lo], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
    if (hc[i]) {
      atomicAdd(&psum[i], hs[i]);
      atomicAdd(&nsample[i], (unsigned long long)hc[i]);
    }
  }
}

}  // namespace

extern "C" {

int vps_pair_k(vps_ctx* ctx, int N, const double* kx_host, const double* ky_host,
               const double* kz_host, double* out_dev) {
  VPS_ENTER(ctx);
  if (N < 1 || !kx_host || !ky_host || !kz_host || !out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_pair_k: bad arguments");
  const size_t one = (size_t)N * sizeof(double);
  const size_t need = 3 * one;
  if (need > ctx->axes_cap) {
    VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_axes) VPS_HIP_CHECK(ctx, hipFree(ctx->d_axes));
    ctx->d_axes = nullptr;
    ctx->axes_cap = 0;
    VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_axes, need));
    ctx->axes_cap = need;
  }
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_axes, kx_host, one, hipMemcpyHostToDevice, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_axes + N, ky_host, one, hipMemcpyHostToDevice, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_axes + 2 * N, kz_host, one, hipMemcpyHostToDevice, ctx->stream));
  const long long n3 = (long long)N * N * N;
  const long long blocks = (n3 + 255) / 256;
  if (blocks > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_pair_k: N too large");
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(pair_k_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->d_axes, N, out_dev);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

int vps_hist_pairs(vps_ctx* ctx, const double* k_dev, const double* w_dev, int64_t n,
                   const double* edges_host, int nbins, double* psum_dev,
                   unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  if (n < 0 || nbins < 1 || nbins > 8192 || !edges_host || !psum_dev || !nsample_dev)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_hist_pairs: bad arguments");
  for (int i = 0; i < nbins; ++i)
    if (!(edges_host[i] <= edges_host[i + 1]))
      return vps_fail(ctx, VPS_ERR_ARG, "vps_hist_pairs: edges must increase monotonically");
  if (n == 0) return VPS_OK;
  if (!k_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_hist_pairs: null k");
  double* d_edges = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&d_edges, sizeof(double) * (nbins + 1)));
  hipError_t e = hipMemcpy(d_edges, edges_host, sizeof(double) * (nbins + 1), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d_edges);
    VPS_HIP_CHECK(ctx, e);
  }
  long long blocks = (n + 255) / 256;
  const long long cap = (long long)ctx->num_cu * 8;
  if (blocks > cap) blocks = cap;
  const size_t lds = sizeof(double) * (2 * nbins + 1) + sizeof(unsigned) * nbins;
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(hist_pairs_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, k_dev, w_dev,
                       (long long)n, d_edges, nbins, psum_dev, nsample_dev);
  }
  e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_edges);
  VPS_HIP_CHECK(ctx, e);
  VPS_HIP_CHECK(ctx, e2);
  return VPS_OK;
}

}  // extern "C"
