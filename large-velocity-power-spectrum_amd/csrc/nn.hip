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
#include <cstdlib>

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

// sorted record of a particle: float32 position (exact for float32 input, rounded for
// float64 input) + original index, one 16-byte load per candidate
template <typename F>
__global__ void __launch_bounds__(256)
    nn_fill_kernel(const F* __restrict__ pos, long long np, NnGrid g, const unsigned* __restrict__ start,
                   unsigned* __restrict__ fill, float4* __restrict__ srec) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const F px = pos[i * 3 + 0], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
  const int cx = cell_coord((double)px, g.lo[0], g.inv_w[0], g.M);
  const int cy = cell_coord((double)py, g.lo[1], g.inv_w[1], g.M);
  const int cz = cell_coord((double)pz, g.lo[2], g.inv_w[2], g.M);
  const long long c = ((long long)cx * g.M + cy) * g.M + cz;
  const unsigned slot = start[c] + atomicAdd(&fill[c], 1u);
  srec[slot] = make_float4((float)px, (float)py, (float)pz, __int_as_float((int)i));
}

// One thread per lattice point.  The cell columns that rings 0 and 1 of every point of a
// workgroup's tile can touch are staged in LDS once (bucket offsets + records); points that
// need ring 2 or more (rare at about one particle per cell) continue from global memory.
// Candidates are screened in float32 against a bound that provably keeps every particle
// whose exact float64 distance could tie or beat the best (err bounds the float32 distance
// error); survivors are re-evaluated exactly in float64 from the original coordinates, with
// the lowest-index tie rule.
constexpr int NN_MAXCELL = 1024;   // staged bucket-offset entries (4 KiB)
constexpr int NN_MAXREC = 1024;    // staged records (16 KiB)

struct NnBest {
  double best;
  float screen;
  int idx;
};

template <typename F>
__device__ __forceinline__ void nn_consider(const F* __restrict__ pos, const float4 rec, const double (&Q)[3],
                                            const float (&Qf)[3], float err, NnBest& b) {
  const float fx = Qf[0] - rec.x, fy = Qf[1] - rec.y, fz = Qf[2] - rec.z;
  const float d2f = (fx * fx + fy * fy) + fz * fz;
  if (d2f > b.screen) return;
  const int oi = __float_as_int(rec.w);
  double px, py, pz;
  if constexpr (sizeof(F) == 4) {
    px = (double)rec.x; py = (double)rec.y; pz = (double)rec.z;   // exact
  } else {
    px = pos[(long long)oi * 3 + 0]; py = pos[(long long)oi * 3 + 1]; pz = pos[(long long)oi * 3 + 2];
  }
  const double dx = Q[0] - px, dy = Q[1] - py, dz = Q[2] - pz;
  double d2 = dx * dx;
  d2 = d2 + dy * dy;
  d2 = d2 + dz * dz;
  if (d2 < b.best || (d2 == b.best && oi < b.idx)) {
    b.best = d2;
    b.idx = oi;
    // anything with true distance <= sqrt(best) has float32 distance <= sqrt(best)+err
    const float rb = sqrtf((float)d2) * 1.000001f + err;
    b.screen = rb * rb * 1.000001f;
  }
}

// true when everything outside the searched (2r+1)^3 block is provably farther than best
__device__ __forceinline__ bool nn_done(const NnGrid& g, const int (&c)[3], const double (&Q)[3], int r,
                                        double best, bool& exhausted) {
  const int M = g.M;
  double bound = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (c[a] + r + 1 <= M - 1) bound = fmin(bound, (g.lo[a] + (double)(c[a] + r + 1) * g.w[a]) - Q[a]);
    if (c[a] - r - 1 >= 0) bound = fmin(bound, Q[a] - (g.lo[a] + (double)(c[a] - r) * g.w[a]));
  }
  exhausted = (bound == INFINITY);  // whole grid searched
  if (exhausted) return true;
  // slack: a particle may sit one rounding error outside its cell's nominal extent
  bound -= 1e-6 * fmax(g.w[0], fmax(g.w[1], g.w[2]));
  return bound > 0.0 && best < bound * bound;
}

// Workgroup = 4 waves = a 4 x 4 x 16 tile of lattice points; wave w owns the 4 x 4 x 4
// sub-block at z offset 4w.  All lanes of a wave scan the SAME staged candidates (the union
// of the cells that rings 0 and 1 of the sub-block can touch): uniform loop bounds, LDS
// broadcast reads, no divergence outside the rare float64 re-check.
constexpr int NN_BX = 4, NN_BY = 4, NN_BZ = 16;

__device__ __forceinline__ int wave_min(int v) {
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

template <typename F, int C>
__global__ void __launch_bounds__(256)
    nn_query_kernel(const F* __restrict__ pos, const float4* __restrict__ srec,
                    const unsigned* __restrict__ start, NnGrid g, float err,
                    const double* __restrict__ qx, const double* __restrict__ qy,
                    const double* __restrict__ qz, int x0, int nx, int nqy, int nqz,
                    const float* __restrict__ payload, float* __restrict__ out,
                    int* __restrict__ nn_idx) {
  __shared__ unsigned lstart[NN_MAXCELL];
  __shared__ float4 lrec[NN_MAXREC];
  __shared__ int range[6];        // min/max of cx, cy, cz over the tile
  __shared__ unsigned colbase[65];
  __shared__ int staged_flag;
  __shared__ int lbest[NN_BX * NN_BY * NN_BZ];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntz = (nqz + NN_BZ - 1) / NN_BZ, nty = (nqy + NN_BY - 1) / NN_BY;
  const long long tile = blockIdx.x;
  const int tz0 = (int)(tile % ntz) * NN_BZ;
  const int ty0 = (int)((tile / ntz) % nty) * NN_BY;
  const int tx0 = (int)(tile / ((long long)ntz * nty)) * NN_BX;
  const int li = lane >> 4, lj = (lane >> 2) & 3, lk = lane & 3;
  const int ix = tx0 + li, iy = ty0 + lj, iz = tz0 + wv * 4 + lk;
  const bool valid = ix < nx && iy < nqy && iz < nqz;
  const int M = g.M;
  const double Q[3] = {valid ? qx[x0 + ix] : qx[x0], valid ? qy[iy] : qy[0], valid ? qz[iz] : qz[0]};
  const float Qf[3] = {(float)Q[0], (float)Q[1], (float)Q[2]};
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = cell_coord(Q[a], g.lo[a], g.inv_w[a], M);

  if (threadIdx.x < 6) range[threadIdx.x] = (threadIdx.x & 1) ? -1 : M;
  __syncthreads();
  // per-wave ranges by shuffles, tile ranges by one LDS atomic per wave and bound
  const int wx0 = wave_min(valid ? c[0] : M), wx1 = wave_max(valid ? c[0] : -1);
  const int wy0 = wave_min(valid ? c[1] : M), wy1 = wave_max(valid ? c[1] : -1);
  const int wz0 = wave_min(valid ? c[2] : M), wz1 = wave_max(valid ? c[2] : -1);
  if (lane == 0 && wx1 >= 0) {
    atomicMin(&range[0], wx0); atomicMax(&range[1], wx1);
    atomicMin(&range[2], wy0); atomicMax(&range[3], wy1);
    atomicMin(&range[4], wz0); atomicMax(&range[5], wz1);
  }
  __syncthreads();
  // cell columns that rings 0 and 1 of the whole tile can touch
  const int cx0 = max(range[0] - 1, 0), cx1 = min(range[1] + 1, M - 1);
  const int cy0 = max(range[2] - 1, 0), cy1 = min(range[3] + 1, M - 1);
  const int cz0 = max(range[4] - 1, 0), cz1 = min(range[5] + 1, M - 1);
  const int ncy = cy1 - cy0 + 1, ncz = cz1 - cz0 + 1;
  const int ncol = (cx1 - cx0 + 1) * ncy;
  const bool fits = range[1] >= 0 && ncol <= 64 && ncol * (ncz + 1) <= NN_MAXCELL;
  if (fits && (int)threadIdx.x < ncol) {
    const long long row = ((long long)(cx0 + threadIdx.x / ncy) * M + (cy0 + threadIdx.x % ncy)) * M;
    colbase[threadIdx.x + 1] = start[row + cz1 + 1] - start[row + cz0];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    colbase[0] = 0;
    if (fits)
      for (int k = 1; k <= ncol; ++k) {
        tot += colbase[k];
        colbase[k] = tot;
      }
    staged_flag = (fits && tot <= NN_MAXREC) ? 1 : 0;
  }
  __syncthreads();
  const bool staged = staged_flag != 0;
  if (staged) {
    for (int col = wv; col < ncol; col += 4) {
      const long long row = ((long long)(cx0 + col / ncy) * M + (cy0 + col % ncy)) * M;
      const unsigned g0 = start[row + cz0];
      const unsigned base = colbase[col];
      for (int k = lane; k <= ncz; k += 64) lstart[col * (ncz + 1) + k] = start[row + cz0 + k] - g0 + base;
      const unsigned n = colbase[col + 1] - base;
      for (unsigned j = lane; j < n; j += 64) lrec[base + j] = srec[g0 + j];
    }
  }
  __syncthreads();

  NnBest b{INFINITY, INFINITY, 0x7fffffff};
  int r0 = 0;
  bool finished = !valid;
  if (staged && wx1 >= 0) {
    // this wave's share of the staged region: wave-uniform bounds
    // (readfirstlane: the values are wave-uniform by construction; telling the compiler keeps
    // the loop control on the scalar unit)
    const int sx0 = __builtin_amdgcn_readfirstlane(max(wx0 - 1, 0)), sx1 = __builtin_amdgcn_readfirstlane(min(wx1 + 1, M - 1));
    const int sy0 = __builtin_amdgcn_readfirstlane(max(wy0 - 1, 0)), sy1 = __builtin_amdgcn_readfirstlane(min(wy1 + 1, M - 1));
    const int sz0 = __builtin_amdgcn_readfirstlane(max(wz0 - 1, 0)), sz1 = __builtin_amdgcn_readfirstlane(min(wz1 + 1, M - 1));
    // pass 1, float32 only and branch-free: the TWO smallest float32 squared distances and where the
    // smallest one sits in the staged records
    float b1 = INFINITY, b2 = INFINITY;
    int j1 = -1;
    for (int cx = sx0; cx <= sx1; ++cx)
      for (int cy = sy0; cy <= sy1; ++cy) {
        const int colrow = ((cx - cx0) * ncy + (cy - cy0)) * (ncz + 1) - cz0;
        const unsigned s = __builtin_amdgcn_readfirstlane(lstart[colrow + sz0]);
        const unsigned e = __builtin_amdgcn_readfirstlane(lstart[colrow + sz1 + 1]);
        for (unsigned j = s; j < e; ++j) {
          const float4 rec = lrec[j];
          const float fx = Qf[0] - rec.x, fy = Qf[1] - rec.y, fz = Qf[2] - rec.z;
          const float d2f = (fx * fx + fy * fy) + fz * fz;
          b2 = __builtin_amdgcn_fmed3f(b1, b2, d2f);   // b1 <= b2: the middle one is the new runner-up
          j1 = d2f < b1 ? (int)j : j1;
          b1 = d2f < b1 ? d2f : b1;                    // (fminf's NaN rules cost a dozen instructions)
        }
      }
    // The exact nearest particle p* satisfies d(p*) <= d(argmin32) <= sqrt(b1)+err, hence
    // d32(p*) <= sqrt(b1) + 2 err =: screen.  When the runner-up lies beyond the screen the float32 winner is
    // the exact one and a single float64 evaluation (for the termination test) finishes the lane; only if some
    // lane of the wave has a contender inside its screen -- near ties, rare -- the wave makes a second sweep
    // that re-evaluates those contenders in float64 (nn_consider screens against b.screen).
    {
      const float rb = sqrtf(b1) * 1.000001f + 2.f * err;
      b.screen = rb * rb * 1.000001f;
    }
    const bool contested = valid && !(b2 > b.screen);
    if (__any(contested)) {
      for (int cx = sx0; cx <= sx1; ++cx)
        for (int cy = sy0; cy <= sy1; ++cy) {
          const int colrow = ((cx - cx0) * ncy + (cy - cy0)) * (ncz + 1) - cz0;
          const unsigned s = __builtin_amdgcn_readfirstlane(lstart[colrow + sz0]);
          const unsigned e = __builtin_amdgcn_readfirstlane(lstart[colrow + sz1 + 1]);
          for (unsigned j = s; j < e; ++j) nn_consider<F>(pos, lrec[j], Q, Qf, err, b);
        }
    } else if (j1 >= 0) {
      nn_consider<F>(pos, lrec[j1], Q, Qf, err, b);
    }
    if (valid) {
      bool exhausted;
      finished = nn_done(g, c, Q, 1, b.best, exhausted);
    }
    r0 = 2;
  }
  // general search from global memory: rings r0, r0+1, ... (rare after a staged scan)
  for (int r = r0; r < M && !finished; ++r) {
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
          for (unsigned j = s; j < e; ++j) nn_consider<F>(pos, srec[j], Q, Qf, err, b);
        }
      }
    }
    bool exhausted;
    finished = nn_done(g, c, Q, r, b.best, exhausted);
  }
  // results leave through LDS so that 16 consecutive z (64 bytes per channel) are stored together
  lbest[(li * NN_BY + lj) * NN_BZ + wv * 4 + lk] = b.idx;
  __syncthreads();
  {
    const int zz = threadIdx.x & 15, j = (threadIdx.x >> 4) & 3, i = threadIdx.x >> 6;
    const int ox = tx0 + i, oy = ty0 + j, oz = tz0 + zz;
    if (ox < nx && oy < nqy && oz < nqz) {
      const int bi = lbest[(i * NN_BY + j) * NN_BZ + zz];
      const long long nq = (long long)nx * nqy * nqz;
      const long long q = ((long long)ox * nqy + oy) * nqz + oz;
      if (nn_idx) nn_idx[q] = bi;
      if (out) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out[(long long)ch * nq + q] = payload[(long long)bi * C + ch];
      }
    }
  }
}

int nn_grid_side(int64_t np) {
  // cells per axis from the particles per cell: 1.5 measured best (sweep 0.7 .. 3) -- the staged
  // union of cells per 4x4x4 query block grows with smaller cells, the ring-2 fallbacks with larger
  const double ppc = 1.5;
  double m = std::cbrt((double)np / ppc);
  int M = (int)m;
  if (M < 1) M = 1;
  if (M > 1024) M = 1024;
  return M;
}

struct NnLayout {
  size_t header, count, fill, start, tiles, srec, total;
  long long ncell, ntiles;
  int M;
};

NnLayout nn_layout(int64_t np, int /*is_f64*/) {
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
  l.srec = off;   off = align(off + (size_t)np * sizeof(float4));
  l.total = off;
  return l;
}

template <typename F>
int nn_run(vps_ctx* ctx, const F* pos, const float* payload, int64_t np, int C, int x0, int nx,
           int nqy, int nqz, const double* dqx, const double* dqy, const double* dqz, double qmax,
           float* out, int* nn_idx, char* work) {
  const NnLayout l = nn_layout(np, sizeof(F) == 8);
  NnHeader* hdr = reinterpret_cast<NnHeader*>(work + l.header);
  unsigned* count = reinterpret_cast<unsigned*>(work + l.count);
  unsigned* fill = reinterpret_cast<unsigned*>(work + l.fill);
  unsigned* start = reinterpret_cast<unsigned*>(work + l.start);
  unsigned* tiles = reinterpret_cast<unsigned*>(work + l.tiles);
  float4* srec = reinterpret_cast<float4*>(work + l.srec);
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
  double pmax = 0.0;
  for (int a = 0; a < 3; ++a) {
    const double lo = ordered_to_f64(h.bmin[a]), hi = ordered_to_f64(h.bmax[a]);
    if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi))
      return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: particle positions are not finite");
    double ext = hi - lo;
    if (!(ext > 0)) ext = 1.0;
    g.lo[a] = lo;
    g.w[a] = ext / (double)l.M;
    g.inv_w[a] = (double)l.M / ext;
    pmax = std::fmax(pmax, std::fmax(std::fabs(lo), std::fabs(hi)));
  }
  // bound on |float32 distance - exact distance|: each coordinate is rounded to float32 once
  // (relative 2^-24) on both sides of the subtraction, three components
  const float err = (float)((qmax + pmax) * 2.5e-7);
  VPS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, sizeof(unsigned) * l.ncell, ctx->stream));
  VPS_HIP_CHECK(ctx, hipMemsetAsync(fill, 0, sizeof(unsigned) * l.ncell, ctx->stream));
  {
    vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
    hipLaunchKernelGGL(nn_count_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, count);
    launch_exclusive_scan(ctx->stream, count, l.ncell, tiles, start);
    hipLaunchKernelGGL(nn_fill_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, start, fill, srec);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  const long long qblocks = (long long)((nx + NN_BX - 1) / NN_BX) * ((nqy + NN_BY - 1) / NN_BY) * ((nqz + NN_BZ - 1) / NN_BZ);
  if (qblocks > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_nn_resample: too many queries for one launch");
  {
    vps_launch_timer tm(ctx, VPS_K_NN_QUERY);
#define VPS_NNQ(CC)                                                                                  \
  hipLaunchKernelGGL((nn_query_kernel<F, CC>), dim3((unsigned)qblocks), dim3(256), 0, ctx->stream, \
                     pos, srec, start, g, err, dqx, dqy, dqz, x0, nx, nqy, nqz, payload, out, nn_idx)
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
  VPS_ENTER(ctx);
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
  double qmax = 0.0;
  for (int i = 0; i < nqx; ++i) qmax = std::fmax(qmax, std::fabs(qx_host[i]));
  for (int i = 0; i < nqy; ++i) qmax = std::fmax(qmax, std::fabs(qy_host[i]));
  for (int i = 0; i < nqz; ++i) qmax = std::fmax(qmax, std::fabs(qz_host[i]));
  if (!std::isfinite(qmax)) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample: lattice axes are not finite");
  if (pos_is_f64)
    return nn_run<double>(ctx, reinterpret_cast<const double*>(pos_dev), payload_dev, np, C, x0, nx, nqy,
                          nqz, dqx, dqy, dqz, qmax, out_dev, nn_idx_dev, work);
  return nn_run<float>(ctx, reinterpret_cast<const float*>(pos_dev), payload_dev, np, C, x0, nx, nqy, nqz,
                       dqx, dqy, dqz, qmax, out_dev, nn_idx_dev, work);
}

}  // extern "C"
