// Stage A2: exact nearest-neighbour resampling of particle payloads onto a lattice.
//
// Replaces ann_interpolate (vpower/interp.py:1018-1049: pyann.nn2 k=1, eps=0) and the
// per-cell Annoy query loop of scripts/parallel_optimized.py:337-358, as an EXACT
// 1-NN search: squared distance ((qx-px)^2 + (qy-py)^2) + (qz-pz)^2 in float64 (no
// contraction), lowest original particle index on exact ties -- the rule the oracle
// states, so indices are bit exact.
//
// Method: counting-sort the particles into an M^3 cell list spanning their bounding
// box (about two particles per cell), then one thread per lattice point searches
// Chebyshev rings of cells around its own cell until the best squared distance is
// provably smaller than the distance to every unsearched cell.  Lattice points are
// z-fastest, so a wave's 64 queries walk the same few cell runs.
#pragma clang fp contract(off)

#include <cmath>

#include "vps_internal.h"
#include "scan.h"

namespace {

struct NnHeader {
  unsigned long long bmin[3];  // order-preserving images of the bounding box
  unsigned long long bmax[3];
};

__device__ __forceinline__ unsigned long long f64_to_ordered(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __host__ inline double ordered_to_f64(unsigned long long u) {
  u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
  double d;
  memcpy(&d, &u, sizeof(d));
  return d;
}

__global__ void nn_init_header(NnHeader* h) {
  if (threadIdx.x < 3) {
    h->bmin[threadIdx.x] = ~0ull;
    h->bmax[threadIdx.x] = 0ull;
  }
}

template <typename F>
__global__ void __launch_bounds__(256) nn_bbox_kernel(const F* __restrict__ pos, long long np, NnHeader* h) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < np;
       i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = (double)pos[i * 3 + a];
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (lo[a] <= hi[a]) {
        atomicMin(&h->bmin[a], f64_to_ordered(lo[a]));
        atomicMax(&h->bmax[a], f64_to_ordered(hi[a]));
      }
    }
  }
}

struct NnGrid {
  double lo[3];
  double inv_w[3];
  double w[3];
  int M;
};

__device__ __forceinline__ int cell_coord(double v, double lo, double inv_w, int M) {
  const double f = floor((v - lo) * inv_w);
  int c = (f >= (double)M) ? M - 1 : ((f < 0.0) ? 0 : (int)f);
  return c;
}

template <typename F>
__global__ void __launch_bounds__(256)
    nn_count_kernel(const F* __restrict__ pos, long long np, NnGrid g, unsigned* __restrict__ count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int cx = cell_coord((double)pos[i * 3 + 0], g.lo[0], g.inv_w[0], g.M);
  const int cy = cell_coord((double)pos[i * 3 + 1], g.lo[1], g.inv_w[1], g.M);
  const int cz = cell_coord((double)pos[i * 3 + 2], g.lo[2], g.inv_w[2], g.M);
  atomicAdd(&count[((long long)cx * g.M + cy) * g.M + cz], 1u);
}

template <typename F>
__global__ void __launch_bounds__(256)
    nn_fill_kernel(const F* __restrict__ pos, long long np, NnGrid g, const unsigned* __restrict__ start,
                   unsigned* __restrict__ fill, F* __restrict__ spos, int* __restrict__ sidx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const F px = pos[i * 3 + 0], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
  const int cx = cell_coord((double)px, g.lo[0], g.inv_w[0], g.M);
  const int cy = cell_coord((double)py, g.lo[1], g.inv_w[1], g.M);
  const int cz = cell_coord((double)pz, g.lo[2], g.inv_w[2], g.M);
  const long long c = ((long long)cx * g.M + cy) * g.M + cz;
  const unsigned slot = start[c] + atomicAdd(&fill[c], 1u);
  spos[(long long)slot * 3 + 0] = px;
  spos[(long long)slot * 3 + 1] = py;
  spos[(long long)slot * 3 + 2] = pz;
  sidx[slot] = (int)i;
}

template <typename F, int C>
__global__ void __launch_bounds__(256)
    nn_query_kernel(const F* __restrict__ spos, const int* __restrict__ sidx,
                    const unsigned* __restrict__ start, NnGrid g, const double* __restrict__ qx,
                    const double* __restrict__ qy, const double* __restrict__ qz, int x0, int nx,
                    int nqy, int nqz, const float* __restrict__ payload, float* __restrict__ out,
                    int* __restrict__ nn_idx) {
  const long long nq = (long long)nx * nqy * nqz;
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int iz = (int)(q % nqz);
  const int iy = (int)((q / nqz) % nqy);
  const int ix = (int)(q / ((long long)nqz * nqy));
  const double Q[3] = {qx[x0 + ix], qy[iy], qz[iz]};
  const int M = g.M;
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = cell_coord(Q[a], g.lo[a], g.inv_w[a], M);

  double best = INFINITY;
  int best_i = 0x7fffffff;
  for (int r = 0; r < M; ++r) {
    const int xlo = max(c[0] - r, 0), xhi = min(c[0] + r, M - 1);
    const int ylo = max(c[1] - r, 0), yhi = min(c[1] + r, M - 1);
    for (int cx = xlo; cx <= xhi; ++cx) {
      const bool xedge = (cx == c[0] - r) || (cx == c[0] + r);
      for (int cy = ylo; cy <= yhi; ++cy) {
        const bool edge = xedge || (cy == c[1] - r) || (cy == c[1] + r);
        const long long rowbase = ((long long)cx * M + cy) * M;
        // shell cells of this (cx,cy) column: the whole z run on an x/y face, else the two caps
        const int nruns = (edge || r == 0) ? 1 : 2;
        for (int run = 0; run < nruns; ++run) {
          int z0, z1;
          if (edge || r == 0) {
            z0 = max(c[2] - r, 0);
            z1 = min(c[2] + r, M - 1);
          } else {
            z0 = z1 = (run == 0) ? c[2] - r : c[2] + r;
            if (z0 < 0 || z0 >= M) continue;
          }
          const unsigned s = start[rowbase + z0], e = start[rowbase + z1 + 1];
          for (unsigned j = s; j < e; ++j) {
            const double dx = Q[0] - (double)spos[(long long)j * 3 + 0];
            const double dy = Q[1] - (double)spos[(long long)j * 3 + 1];
            const double dz = Q[2] - (double)spos[(long long)j * 3 + 2];
            double d2 = dx * dx;
            d2 = d2 + dy * dy;
            d2 = d2 + dz * dz;
            const int oi = sidx[j];
            if (d2 < best || (d2 == best && oi < best_i)) {
              best = d2;
              best_i = oi;
            }
          }
        }
      }
    }
    // lower bound on the distance to anything outside the searched (2r+1)^3 block
    double bound = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (c[a] + r + 1 <= M - 1) bound = fmin(bound, (g.lo[a] + (double)(c[a] + r + 1) * g.w[a]) - Q[a]);
      if (c[a] - r - 1 >= 0) bound = fmin(bound, Q[a] - (g.lo[a] + (double)(c[a] - r) * g.w[a]));
    }
    if (bound == INFINITY) break;  // whole grid searched
    // slack: a particle may sit one rounding error outside its cell's nominal extent
    bound -= 1e-6 * fmax(g.w[0], fmax(g.w[1], g.w[2]));
    if (bound > 0.0 && best < bound * bound) break;
  }
  if (nn_idx) nn_idx[q] = best_i;
  if (out) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) out[(long long)ch * nq + q] = payload[(long long)best_i * C + ch];
  }
}

int nn_grid_side(int64_t np) {
  double m = std::cbrt((double)np / 2.0);
  int M = (int)m;
  if (M < 1) M = 1;
  if (M > 1024) M = 1024;
  return M;
}

struct NnLayout {
  size_t header, count, fill, start, tiles, spos, sidx, total;
  long long ncell, ntiles;
  int M;
};

NnLayout nn_layout(int64_t np, int is_f64) {
  NnLayout l;
  l.M = nn_grid_side(np);
  l.ncell = (long long)l.M * l.M * l.M;
  l.ntiles = (l.ncell + SCAN_TILE - 1) / SCAN_TILE;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  l.header = off; off = align(off + sizeof(NnHeader));
  l.count = off;  off = align(off + sizeof(unsigned) * l.ncell);
  l.fill = off;   off = align(off + sizeof(unsigned) * l.ncell);
  l.start = off;  off = align(off + sizeof(unsigned) * (l.ncell + 1));
  l.tiles = off;  off = align(off + sizeof(unsigned) * (l.ntiles + 1));
  l.spos = off;   off = align(off + (size_t)np * 3 * (is_f64 ? 8 : 4));
  l.sidx = off;   off = align(off + (size_t)np * 4);
  l.total = off;
  return l;
}

template <typename F>
int nn_run(vps_ctx* ctx, const F* pos, const float* payload, int64_t np, int C, int x0, int nx,
           int nqy, int nqz, const double* dqx, const double* dqy, const double* dqz, float* out,
           int* nn_idx, char* work) {
  const NnLayout l = nn_layout(np, sizeof(F) == 8);
  NnHeader* hdr = reinterpret_cast<NnHeader*>(work + l.header);
  unsigned* count = reinterpret_cast<unsigned*>(work + l.count);
  unsigned* fill = reinterpret_cast<unsigned*>(work + l.fill);
  unsigned* start = reinterpret_cast<unsigned*>(work + l.start);
  unsigned* tiles = reinterpret_cast<unsigned*>(work + l.tiles);
  F* spos = reinterpret_cast<F*>(work + l.spos);
  int* sidx = reinterpret_cast<int*>(work + l.sidx);
  const unsigned pblocks = (unsigned)((np + 255) / 256);

  NnHeader h;
  {
    vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
    hipLaunchKernelGGL(nn_init_header, dim3(1), dim3(64), 0, ctx->stream, hdr);
    const unsigned rb = pblocks < 2048u ? pblocks : 2048u;
    hipLaunchKernelGGL(nn_bbox_kernel<F>, dim3(rb), dim3(256), 0, ctx->stream, pos, (long long)np, hdr);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  NnGrid g;
  g.M = l.M;
  for (int a = 0; a < 3; ++a) {
    const double lo = ordered_to_f64(h.bmin[a]), hi = ordered_to_f64(h.bmax[a]);
    if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi))
      return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: particle positions are not finite");
    double ext = hi - lo;
    if (!(ext > 0)) ext = 1.0;
    g.lo[a] = lo;
    g.w[a] = ext / (double)l.M;
    g.inv_w[a] = (double)l.M / ext;
  }
  VPS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, sizeof(unsigned) * l.ncell, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemsetAsync(fill, 0, sizeof(unsigned) * l.ncell, ctx->stream));
  {
    vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
    hipLaunchKernelGGL(nn_count_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, count);
    launch_exclusive_scan(ctx->stream, count, l.ncell, tiles, start);
    hipLaunchKernelGGL(nn_fill_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, start, fill, spos, sidx);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  const long long nq = (long long)nx * nqy * nqz;
  const long long qblocks = (nq + 255) / 256;
  if (qblocks > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_nn_resample: too many queries for one launch");
  {
    vps_launch_timer tm(ctx, VPS_K_NN_QUERY);
#define VPS_NNQ(CC)                                                                                  \
  hipLaunchKernelGGL((nn_query_kernel<F, CC>), dim3((unsigned)qblocks), dim3(256), 0, ctx->stream,   \
                     spos, sidx, start, g, dqx, dqy, dqz, x0, nx, nqy, nqz, payload, out, nn_idx)
    switch (C) {
      case 1: VPS_NNQ(1); break;
      case 3: VPS_NNQ(3); break;
      case 4: VPS_NNQ(4); break;
      default: return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "nn_resample: C=%d channels (supported: 1,3,4)", C);
    }
#undef VPS_NNQ
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

}  // namespace

extern "C" {

size_t vps_nn_workspace_bytes(int64_t np, int pos_is_f64) {
  if (np < 1) return 256;
  return nn_layout(np, pos_is_f64).total;
}

int vps_nn_resample(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                    int64_t np, int C, const double* qx_host, int nqx, const double* qy_host, int nqy,
                    const double* qz_host, int nqz, int x0, int nx, float* out_dev,
                    int32_t* nn_idx_dev, void* work_dev) {
  if (!ctx) return VPS_ERR_ARG;
  if (np < 1) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: need at least one particle");
  if (np > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_nn_resample: np exceeds int32 indices");
  if (nqx < 1 || nqy < 1 || nqz < 1 || !qx_host || !qy_host || !qz_host)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: bad lattice");
  if (x0 < 0 || nx < 1 || x0 + nx > nqx) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: slab [%d,%d) outside [0,%d)", x0, x0 + nx, nqx);
  if (!pos_dev || !work_dev || (!out_dev && !nn_idx_dev)) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: null buffer");
  if (out_dev && !payload_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: null payload");
  const size_t need = (size_t)(nqx + nqy + nqz) * sizeof(double);
  if (need > ctx->axes_cap) {
    VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_axes) VPS_HIP_CHECK(ctx, hipFree(ctx->d_axes));
    ctx->d_axes = nullptr;
    ctx->axes_cap = 0;
    VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_axes, need));
    ctx->axes_cap = need;
  }
  double* dqx = ctx->d_axes;
  double* dqy = dqx + nqx;
  double* dqz = dqy + nqy;
  // pageable-host copies: the runtime stages them before returning
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(dqx, qx_host, sizeof(double) * nqx, hipMemcpyHostToDevice, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(dqy, qy_host, sizeof(double) * nqy, hipMemcpyHostToDevice, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemcpyAsync(dqz, qz_host, sizeof(double) * nqz, hipMemcpyHostToDevice, ctx->stream));
  char* work = reinterpret_cast<char*>(work_dev);
  if (pos_is_f64)
    return nn_run<double>(ctx, reinterpret_cast<const double*>(pos_dev), payload_dev, np, C, x0, nx, nqy,
                          nqz, dqx, dqy, dqz, out_dev, nn_idx_dev, work);
  return nn_run<float>(ctx, reinterpret_cast<const float*>(pos_dev), payload_dev, np, C, x0, nx, nqy, nqz,
                       dqx, dqy, dqz, out_dev, nn_idx_dev, work);
}

}  // extern "C"
