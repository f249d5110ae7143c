// Particle preprocessing of the reference drivers, on the device:
//   shift to the origin   coords[:, a] -= min(coords[:, a])                (scripts/parallel_optimized.py:280-282,
//                                                                            vpower/interp.py:169-175)
//   remove bulk velocity  v[:, a] -= sum(mass * v[:, a]) / sum(mass)        (script:285-289, interp.py:178-182)
// The minimum and the subtraction are exact in the dtype of the coordinates (bit-identical to
// numpy); the mass-weighted mean is accumulated in float64 (numpy sums float32 pairwise in float32,
// so the bulk velocity agrees to float32 rounding, not bit for bit).
#pragma clang fp contract(off)

#include "vps_internal.h"

namespace {

struct PrepScratch {
  unsigned long long vmin[3];  // order-preserving images of the per-axis minima
  double wsum[4];              // sum m vx, sum m vy, sum m vz, sum m
};

__device__ __forceinline__ unsigned long long ordered_of(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
inline double unordered_of(unsigned long long u) {
  u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
  double d;
  memcpy(&d, &u, sizeof(d));
  return d;
}

__global__ void prep_init(PrepScratch* s) {
  if (threadIdx.x < 3) s->vmin[threadIdx.x] = ~0ull;
  if (threadIdx.x < 4) s->wsum[threadIdx.x] = 0.0;
}

template <typename F>
__global__ void __launch_bounds__(256)
    prep_reduce(const F* __restrict__ pos, const float* __restrict__ vel, const float* __restrict__ mass,
                long long np, int want_min, int want_bulk, PrepScratch* s) {
  double lo[3] = {INFINITY, INFINITY, INFINITY};
  double w[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < np;
       i += (long long)gridDim.x * blockDim.x) {
    if (want_min) {
#pragma unroll
      for (int a = 0; a < 3; ++a) lo[a] = fmin(lo[a], (double)pos[i * 3 + a]);
    }
    if (want_bulk) {
      const double m = (double)mass[i];
#pragma unroll
      for (int a = 0; a < 3; ++a) w[a] += m * (double)vel[i * 3 + a];
      w[3] += m;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = fmin(lo[a], __shfl_down(lo[a], off, 64));
#pragma unroll
    for (int a = 0; a < 4; ++a) w[a] += __shfl_down(w[a], off, 64);
  }
  // one set of atomics per WORKGROUP, not per wave: the seven words share a cache line, and returning atomics on one line
  // serialise at the memory side (the NN bounding-box pass spent most of its time there, nn.hip)
  __shared__ double red[4][7];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) red[threadIdx.x >> 6][a] = lo[a];
#pragma unroll
    for (int a = 0; a < 4; ++a) red[threadIdx.x >> 6][3 + a] = w[a];
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int a = threadIdx.x;
    if (a < 3) {
      const double m = fmin(fmin(red[0][a], red[1][a]), fmin(red[2][a], red[3][a]));
      if (want_min && m != INFINITY) atomicMin(&s->vmin[a], ordered_of(m));
    } else if (want_bulk) {
      atomicAdd(&s->wsum[a - 3], (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]));
    }
  }
}

template <typename F>
__global__ void __launch_bounds__(256)
    prep_apply(F* __restrict__ pos, float* __restrict__ vel, long long np, int do_shift, int do_bulk,
               F mx, F my, F mz, float bx, float by, float bz) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  if (do_shift) {
    pos[i * 3 + 0] -= mx;
    pos[i * 3 + 1] -= my;
    pos[i * 3 + 2] -= mz;
  }
  if (do_bulk) {
    vel[i * 3 + 0] -= bx;
    vel[i * 3 + 1] -= by;
    vel[i * 3 + 2] -= bz;
  }
}

template <typename F>
int prep_run(vps_ctx* ctx, F* pos, float* vel, const float* mass, int64_t np, int shift, int bulk,
             double* out_min, double* out_bulk) {
  PrepScratch* d = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&d, sizeof(PrepScratch)));
  PrepScratch h;
  long long blocks = (np + 255) / 256;
  if (blocks > (long long)ctx->num_cu * 8) blocks = (long long)ctx->num_cu * 8;
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(prep_init, dim3(1), dim3(64), 0, ctx->stream, d);
    hipLaunchKernelGGL(prep_reduce<F>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, pos, vel, mass,
                       (long long)np, shift, bulk, d);
  }
  hipError_t e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  VPS_HIP_CHECK(ctx, e);
  double mn[3] = {0, 0, 0}, bv[3] = {0, 0, 0};
  if (shift)
    for (int a = 0; a < 3; ++a) mn[a] = unordered_of(h.vmin[a]);
  if (bulk) {
    if (!(h.wsum[3] != 0.0)) return vps_fail(ctx, VPS_ERR_ARG, "vps_preprocess: total mass is zero");
    // the reference subtracts a float32 value from float32 velocities (numpy keeps the dtype)
    for (int a = 0; a < 3; ++a) bv[a] = (double)(float)(h.wsum[a] / h.wsum[3]);
  }
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(prep_apply<F>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, pos, vel,
                       (long long)np, shift, bulk, (F)mn[0], (F)mn[1], (F)mn[2], (float)bv[0], (float)bv[1],
                       (float)bv[2]);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  if (out_min) memcpy(out_min, mn, sizeof(mn));
  if (out_bulk) memcpy(out_bulk, bv, sizeof(bv));
  return VPS_OK;
}

// sum m, sum m v_c, sum m |v|^2 in float64: the totals behind check_conservation (interp.py:1269-1319)
__global__ void __launch_bounds__(256)
    totals_kernel(const float* __restrict__ v, long long vs, long long vc, const float* __restrict__ mass,
                  long long n, double* __restrict__ out) {
  double w[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double m = (double)mass[i];
    const double vx = (double)v[i * vs], vy = (double)v[i * vs + vc], vz = (double)v[i * vs + 2 * vc];
    w[0] += m;
    w[1] += m * vx;
    w[2] += m * vy;
    w[3] += m * vz;
    w[4] += m * ((vx * vx + vy * vy) + vz * vz);
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 5; ++a) w[a] += __shfl_down(w[a], off, 64);
  }
  __shared__ double red[4][5];     // (one set of atomics per workgroup: see prep_reduce)
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 5; ++a) red[threadIdx.x >> 6][a] = w[a];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int a = threadIdx.x;
    atomicAdd(&out[a], (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]));
  }
}

}  // namespace

extern "C" int vps_totals(vps_ctx* ctx, const float* v_dev, int64_t v_elem_stride, int64_t v_comp_stride,
                          const float* mass_dev, int64_t n, double out_host[5]) {
  VPS_ENTER(ctx);
  if (n < 0 || !out_host || (n > 0 && (!v_dev || !mass_dev))) return vps_fail(ctx, VPS_ERR_ARG, "vps_totals: bad arguments");
  for (int a = 0; a < 5; ++a) out_host[a] = 0.0;
  if (n == 0) return VPS_OK;
  double* d = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&d, 5 * sizeof(double)));
  hipError_t e = hipMemsetAsync(d, 0, 5 * sizeof(double), ctx->stream);
  if (e == hipSuccess) {
    long long blocks = (n + 255) / 256;
    if (blocks > (long long)ctx->num_cu * 8) blocks = (long long)ctx->num_cu * 8;
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(totals_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, v_dev, (long long)v_elem_stride,
                       (long long)v_comp_stride, mass_dev, (long long)n, d);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  VPS_HIP_CHECK(ctx, e);
  return VPS_OK;
}

extern "C" int vps_preprocess(vps_ctx* ctx, void* pos_dev, int pos_is_f64, float* vel_dev, const float* mass_dev,
                              int64_t np, int shift_to_origin, int remove_bulk_velocity, double* min_out_host,
                              double* bulk_out_host) {
  VPS_ENTER(ctx);
  if (np < 1) return vps_fail(ctx, VPS_ERR_ARG, "vps_preprocess: need at least one particle");
  if ((np + 255) / 256 > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_preprocess: np too large for one launch");
  if (!pos_dev || (remove_bulk_velocity && (!vel_dev || !mass_dev)))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_preprocess: null buffer");
  if (pos_is_f64)
    return prep_run<double>(ctx, reinterpret_cast<double*>(pos_dev), vel_dev, mass_dev, np, shift_to_origin,
                            remove_bulk_velocity, min_out_host, bulk_out_host);
  return prep_run<float>(ctx, reinterpret_cast<float*>(pos_dev), vel_dev, mass_dev, np, shift_to_origin,
                         remove_bulk_velocity, min_out_host, bulk_out_host);
}
