// Stage A1 (nearest-grid-point deposition) and A3 (field algebra).
//
// Replaces deposit_to_grid (vpower/interp.py:996-1015) and the elementwise steps
// of ann_interp_to_field / BoxField.{momentum,kinetic_energy}_power
// (vpower/interp.py:272-273, 523-525, 546).
//
// Cell indices must be BIT EXACT with numpy's  int((pos // Lcell) % N)  (SURVEY.md
// Q13): numpy evaluates floor_divide and remainder with its divmod algorithm in the
// dtype of `pos`, so that algorithm is restated here and this translation unit is
// compiled without floating-point contraction.
#pragma clang fp contract(off)

#include "vps_internal.h"

namespace {

template <typename F>
__device__ __forceinline__ F dev_fmod(F a, F b);
template <>
__device__ __forceinline__ float dev_fmod<float>(float a, float b) { return fmodf(a, b); }
template <>
__device__ __forceinline__ double dev_fmod<double>(double a, double b) { return fmod(a, b); }

// numpy npy_divmod: the quotient part (numpy/_core/src/npymath/npy_math_internal.h.src)
template <typename F>
__device__ __forceinline__ F np_floor_divide(F a, F b) {
  F mod = dev_fmod<F>(a, b);
  F div = (a - mod) / b;
  if (mod != F(0)) {
    if ((b < F(0)) != (mod < F(0))) {
      mod += b;
      div -= F(1);
    }
  }
  F fd;
  if (div != F(0)) {
    fd = floor(div);
    if (div - fd > F(0.5)) fd += F(1);
  } else {
    fd = copysign(F(0), a / b);
  }
  return fd;
}

// numpy npy_divmod: the modulus part
template <typename F>
__device__ __forceinline__ F np_remainder(F a, F b) {
  F mod = dev_fmod<F>(a, b);
  if (mod != F(0)) {
    if ((b < F(0)) != (mod < F(0))) mod += b;
  } else {
    mod = copysign(F(0), b);
  }
  return mod;
}

template <typename F>
__device__ __forceinline__ int cell_of(F x, F lcell, F nsize) {
  // int((x // Lcell) % N): float -> int cast truncates
  return (int)np_remainder<F>(np_floor_divide<F>(x, lcell), nsize);
}

template <typename F>
__global__ void __launch_bounds__(256) cell_index_kernel(const F* __restrict__ pos, long long np,
                                                         F lcell, F nsize, int* __restrict__ cell) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) cell[i * 3 + a] = cell_of<F>(pos[i * 3 + a], lcell, nsize);
}

// One thread per particle, C float atomics into the channel-major slab grid.
// Sparse regime of the BASELINE configs (<= 0.08 particles per cell): collisions are
// rare, the cost is the scattered read-modify-write traffic itself.
template <typename F, int C>
__global__ void __launch_bounds__(256)
    deposit_ngp_kernel(const F* __restrict__ pos, const float* __restrict__ payload, long long np,
                       F lcell, F nsize, int N, int x0, int nx, float* __restrict__ grid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int cx = cell_of<F>(pos[i * 3 + 0], lcell, nsize) - x0;
  if (cx < 0 || cx >= nx) return;
  const int cy = cell_of<F>(pos[i * 3 + 1], lcell, nsize);
  const int cz = cell_of<F>(pos[i * 3 + 2], lcell, nsize);
  if ((unsigned)cy >= (unsigned)N || (unsigned)cz >= (unsigned)N) return;  // NaN / inf positions
  const long long cell = ((long long)cx * N + cy) * N + cz;
  const long long plane = (long long)nx * N * N;
  float val[C];
  if constexpr (C == 4) {
    const float4 p4 = *reinterpret_cast<const float4*>(payload + i * 4);
    val[0] = p4.x; val[1] = p4.y; val[2] = p4.z; val[3] = p4.w;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) val[c] = payload[i * C + c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) atomicAdd(grid + c * plane + cell, val[c]);
}

// [rho vx, rho vy, rho vz, rho] per particle (interp.py:199-213)
__global__ void __launch_bounds__(256)
    rhov_kernel(const float* __restrict__ vel, const float* __restrict__ rho, long long np,
                float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const float r = rho[i];
  *reinterpret_cast<float4*>(out + i * 4) =
      make_float4(vel[i * 3 + 0] * r, vel[i * 3 + 1] * r, vel[i * 3 + 2] * r, r);
}

__global__ void __launch_bounds__(256)
    field_algebra_kernel(float* __restrict__ ch, long long ncell, int quantity, int flags, float vol) {
  const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= ncell) return;
  float4 a = *reinterpret_cast<float4*>(ch + i0);
  float4 b = *reinterpret_cast<float4*>(ch + ncell + i0);
  float4 c = *reinterpret_cast<float4*>(ch + 2 * ncell + i0);
  const float4 r = *reinterpret_cast<float4*>(ch + 3 * ncell + i0);
  float4 mm = r;
  float* pa = &a.x; float* pb = &b.x; float* pc = &c.x; float* pm = &mm.x;
  const float* pr = &r.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float vx, vy, vz, m;
    if (flags & VPS_FLAG_INPUT_IS_VM) {
      vx = pa[j]; vy = pb[j]; vz = pc[j]; m = pr[j];
    } else {
      const float rho = pr[j];
      // v = (rho v)/rho; empty cells give 0 (the NaN->0 rule of interp.py:329-331)
      vx = rho != 0.f ? pa[j] / rho : 0.f;
      vy = rho != 0.f ? pb[j] / rho : 0.f;
      vz = rho != 0.f ? pc[j] / rho : 0.f;
      m = rho * vol;
    }
    if (quantity == VPS_VELOCITY || quantity == VPS_VM) {
      pa[j] = vx; pb[j] = vy; pc[j] = vz;
      pm[j] = m;
    } else if (quantity == VPS_MOMENTUM) {
      pa[j] = vx * m;
      pb[j] = ((flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG) ? vx : vy) * m;
      pc[j] = ((flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG) ? vx : vz) * m;
    } else {
      pa[j] = m * ((vx * vx + vy * vy) + vz * vz);
    }
  }
  *reinterpret_cast<float4*>(ch + i0) = a;
  if (quantity != VPS_ENERGY) {
    *reinterpret_cast<float4*>(ch + ncell + i0) = b;
    *reinterpret_cast<float4*>(ch + 2 * ncell + i0) = c;
  }
  if (quantity == VPS_VM) *reinterpret_cast<float4*>(ch + 3 * ncell + i0) = mm;
}

template <typename F>
int deposit_dispatch(vps_ctx* ctx, const void* pos, const float* payload, int64_t np, int C, int N,
                     double Lbox, int x0, int nx, float* grid) {
  // Lcell = Lbox/float(N) in double, then cast to the dtype of pos: what numpy's weak
  // Python-float scalar does in `pos // Lcell`
  const F lcell = (F)(Lbox / (double)N);
  const F nsz = (F)N;
  const unsigned blocks = (unsigned)((np + 255) / 256);
  vps_launch_timer tm(ctx, VPS_K_DEPOSIT);
  const F* p = reinterpret_cast<const F*>(pos);
  switch (C) {
    case 1: hipLaunchKernelGGL((deposit_ngp_kernel<F, 1>), dim3(blocks), dim3(256), 0, ctx->stream, p, payload, np, lcell, nsz, N, x0, nx, grid); break;
    case 3: hipLaunchKernelGGL((deposit_ngp_kernel<F, 3>), dim3(blocks), dim3(256), 0, ctx->stream, p, payload, np, lcell, nsz, N, x0, nx, grid); break;
    case 4: hipLaunchKernelGGL((deposit_ngp_kernel<F, 4>), dim3(blocks), dim3(256), 0, ctx->stream, p, payload, np, lcell, nsz, N, x0, nx, grid); break;
    default: return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "deposit: C=%d channels (supported: 1,3,4)", C);
  }
  return VPS_OK;
}

}  // namespace

extern "C" {

int vps_cell_index(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, int64_t np, int N, double Lbox,
                   int32_t* cell_dev) {
  if (!ctx) return VPS_ERR_ARG;
  if (np < 0 || N < 1 || !(Lbox > 0)) return vps_fail(ctx, VPS_ERR_ARG, "vps_cell_index: bad np/N/Lbox");
  if (np == 0) return VPS_OK;
  if (!pos_dev || !cell_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_cell_index: null buffer");
  const unsigned blocks = (unsigned)((np + 255) / 256);
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    if (pos_is_f64)
      hipLaunchKernelGGL(cell_index_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const double*>(pos_dev), (long long)np,
                         Lbox / (double)N, (double)N, cell_dev);
    else
      hipLaunchKernelGGL(cell_index_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const float*>(pos_dev), (long long)np,
                         (float)(Lbox / (double)N), (float)N, cell_dev);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

int vps_deposit_ngp(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                    int64_t np, int C, int N, double Lbox, int x0, int nx, float* grid_dev) {
  if (!ctx) return VPS_ERR_ARG;
  if (np < 0 || N < 1 || !(Lbox > 0)) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_ngp: bad np/N/Lbox");
  if (x0 < 0 || nx < 1 || x0 + nx > N) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_ngp: slab [%d,%d) outside [0,%d)", x0, x0 + nx, N);
  if (np == 0) return VPS_OK;
  if (!pos_dev || !payload_dev || !grid_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_ngp: null buffer");
  if ((np + 255) / 256 > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_deposit_ngp: np too large for one launch");
  int rc = pos_is_f64 ? deposit_dispatch<double>(ctx, pos_dev, payload_dev, np, C, N, Lbox, x0, nx, grid_dev)
                      : deposit_dispatch<float>(ctx, pos_dev, payload_dev, np, C, N, Lbox, x0, nx, grid_dev);
  if (rc) return rc;
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

int vps_density_velocity_vector(vps_ctx* ctx, const float* vel_dev, const float* rho_dev, int64_t np,
                                float* out_dev) {
  if (!ctx) return VPS_ERR_ARG;
  if (np < 0) return vps_fail(ctx, VPS_ERR_ARG, "vps_density_velocity_vector: np < 0");
  if (np == 0) return VPS_OK;
  if (!vel_dev || !rho_dev || !out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_density_velocity_vector: null buffer");
  if ((np + 255) / 256 > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "np too large for one launch");
  {
    vps_launch_timer tm(ctx, VPS_K_MISC);
    hipLaunchKernelGGL(rhov_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, vel_dev,
                       rho_dev, (long long)np, out_dev);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

int vps_field_algebra(vps_ctx* ctx, int quantity, int flags, double Lcell, float* chans_dev,
                      int64_t ncell) {
  if (!ctx) return VPS_ERR_ARG;
  if (quantity < 0 || quantity > 3) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: quantity %d", quantity);
  if (ncell < 0 || (ncell & 3)) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: ncell must be a multiple of 4");
  if (ncell == 0) return VPS_OK;
  if (!chans_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: null buffer");
  const long long nthreads = ncell / 4;
  const unsigned blocks = (unsigned)((nthreads + 255) / 256);
  {
    vps_launch_timer tm(ctx, VPS_K_ALGEBRA);
    hipLaunchKernelGGL(field_algebra_kernel, dim3(blocks), dim3(256), 0, ctx->stream, chans_dev,
                       (long long)ncell, quantity, flags, (float)(Lcell * Lcell * Lcell));
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

}  // extern "C"
