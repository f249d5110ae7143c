// Stage A2: exact nearest-neighbour resampling of particle payloads onto a lattice.
//
// Replaces ann_interpolate (vpower/interp.py:1018-1049: pyann.nn2 k=1, eps=0) and the
// per-cell Annoy query loop of scripts/parallel_optimized.py:337-358, as an EXACT
// 1-NN search: squared distance ((qx-px)^2 + (qy-py)^2) + (qz-pz)^2 in float64 (no
// contraction), lowest original particle index on exact ties -- the rule the oracle
// states, so indices are bit exact.
//
// Method: sort the particles into an M^3 cell list spanning their bounding box (about 1.5
// particles per cell; two-level LDS bucket sort, nb_* kernels), then
//   * uniformly spaced lattices (both reference lattices): nn_column_kernel where the lattice is
//     about as fine as the particles are dense -- a lane keeps the minima of a column of 32 lattice
//     points in registers and a wave walks the nearby particles --, else nn_scatter_kernel -- a
//     16^3 tile of lattice points keeps its running minima in LDS and every nearby particle lowers
//     the minima inside its own R-box; the few points these cannot settle (voids, near-ties)
//     are finished by nn_fallback_kernel;
//   * any other lattice axes: nn_query_kernel -- one thread per lattice point, a wave-wide
//     staged union of candidate cells, then Chebyshev rings of cells from global memory
//     until the best squared distance is provably smaller than the distance to every
//     unsearched cell (nn_ring_search, also the fallback's search).
#pragma clang fp contract(off)

#include <cmath>
#include <cstdlib>

#include <type_traits>

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
  // A thread takes FOUR consecutive particles per trip = 12 consecutive coordinates = three 16-byte loads (float; six for double):
  // whole cache lines per wave instruction.  (One particle per thread -- three 4-byte loads 12 bytes apart, every line touched by
  // three instructions -- left the pass latency bound: 0.8 ms for 0.6 GB; four strided particles per trip 0.58 ms.)
  typedef typename std::conditional<sizeof(F) == 4, float4, double2>::type V;
  constexpr int PER = sizeof(V) / sizeof(F), NV = 12 / PER;
  // (a coordinate array that does not start on a 16-byte boundary -- a slice handed in through the C ABI -- goes through the
  //  scalar tail below, whole)
  const bool aligned = (reinterpret_cast<unsigned long long>(pos) & 15ull) == 0;
  const long long ngroups = aligned ? np / 4 : 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const V* pv = reinterpret_cast<const V*>(pos);
  for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += stride) {
    V v[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u] = pv[gi * NV + u];
    const F* f = reinterpret_cast<const F*>(v);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      lo[j % 3] = fmin(lo[j % 3], (double)f[j]);
      hi[j % 3] = fmax(hi[j % 3], (double)f[j]);
    }
  }
  for (long long i = 4 * ngroups + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += stride) {   // the last np % 4
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], (double)pos[i * 3 + a]);
      hi[a] = fmax(hi[a], (double)pos[i * 3 + a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  // one set of atomics per WORKGROUP: returning atomics on one word peak near 90 per microsecond, and the six words share a
  // cache line -- one set per wave (8192 waves) was most of the pass: 0.6 ms for 0.6 GB
  __shared__ double red[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[threadIdx.x >> 6][a] = lo[a];
      red[threadIdx.x >> 6][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    double l0 = red[0][a], h0 = red[0][3 + a];
    for (int w = 1; w < 4; ++w) {
      l0 = fmin(l0, red[w][a]);
      h0 = fmax(h0, red[w][3 + a]);
    }
    if (l0 <= h0) {
      atomicMin(&h->bmin[a], f64_to_ordered(l0));
      atomicMax(&h->bmax[a], f64_to_ordered(h0));
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

// ------------------------------------------------------------------------------------------------
// Cell list by a two-level LDS bucket sort (the scheme of deposit.hip, with a cell's linear index as the key): the
// counting sort above pays one random global atomic per particle twice (count, fill: 2.2 + 4.5 ms at 5e7 particles) and
// a random 16-byte store.  Here
//   level 1: chunks of NB_CHUNK particles; per-chunk LDS histogram over groups of 2^gshift consecutive cells ->
//            table[group][chunk] -> exclusive scan -> each chunk ranks its particles with LDS atomics, stages
//            {record, key} in LDS in group order and streams every (chunk, group) run out contiguously;
//   level 2: one workgroup per group: LDS histogram over the group's cells, LDS scan (-> start[]), second sweep places
//            the records at their final slots (all inside the group's few hundred KB: the L2 merges them).
// Records inside one cell come out in no particular order (as with the atomic fill).
// ------------------------------------------------------------------------------------------------
constexpr int NB_THREADS = 1024;
constexpr int NB_ITEMS = 4;
constexpr int NB_CHUNK = NB_THREADS * NB_ITEMS;
constexpr int NB_MAXG_SHIFT = 15;         // cells per group <= 32768 (128 KB of LDS counters in level 2)
constexpr int NB_MAXGROUPS = 2048;

struct NbGeom {
  int gshift, ngroups;
  long long nchunks, ncell;
};

template <typename F>
__device__ __forceinline__ unsigned nb_cell_of(const F* __restrict__ pos, long long i, const NnGrid& g) {
  const int cx = cell_coord((double)pos[i * 3 + 0], g.lo[0], g.inv_w[0], g.M);
  const int cy = cell_coord((double)pos[i * 3 + 1], g.lo[1], g.inv_w[1], g.M);
  const int cz = cell_coord((double)pos[i * 3 + 2], g.lo[2], g.inv_w[2], g.M);
  return (unsigned)((cx * g.M + cy) * g.M + cz);     // M <= 1024: fits 30 bits
}

template <typename F>
__global__ void __launch_bounds__(NB_THREADS)
    nb_hist_kernel(const F* __restrict__ pos, long long np, NnGrid g, NbGeom s, unsigned* __restrict__ keys,
                   unsigned* __restrict__ table) {
  extern __shared__ unsigned nb_lds[];
  for (int i = threadIdx.x; i < s.ngroups; i += NB_THREADS) nb_lds[i] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * NB_CHUNK;
#pragma unroll
  for (int k = 0; k < NB_ITEMS; ++k) {
    const long long i = base + (long long)k * NB_THREADS + threadIdx.x;
    if (i < np) {
      const unsigned key = nb_cell_of<F>(pos, i, g);
      keys[i] = key;
      atomicAdd(&nb_lds[key >> s.gshift], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < s.ngroups; i += NB_THREADS) table[(long long)i * s.nchunks + blockIdx.x] = nb_lds[i];
}

// exclusive scan of the LDS array a[0..n), n <= 4 * NB_THREADS, in place; returns the total (all threads call it)
__device__ __forceinline__ unsigned nb_block_scan(unsigned* a, int n, unsigned* scratch) {
  const int per = (n + NB_THREADS - 1) / NB_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned v[4], mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = tid * per + k;
    v[k] = (k < per && idx < n) ? a[idx] : 0u;
    mine += v[k];
  }
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  unsigned before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NB_THREADS / 64; ++w) {
    const unsigned t = scratch[w];
    if (w < wave) before += t;
    total += t;
  }
  unsigned run = before + inc - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = tid * per + k;
    if (k < per && idx < n) {
      a[idx] = run;
      run += v[k];
    }
  }
  __syncthreads();
  return total;
}

// level-1 scatter, LDS-staged; consecutive chunks own adjacent runs of every group and are dealt to the SAME XCD
// (blockIdx % 8, speed only) so that its L2 can merge the partly written lines at run boundaries
template <typename F>
__global__ void __launch_bounds__(NB_THREADS)
    nb_scatter_kernel(const F* __restrict__ pos, long long np, const unsigned* __restrict__ keys, NbGeom s,
                      const unsigned* __restrict__ table_start, float4* __restrict__ rec1, unsigned* __restrict__ key1) {
  extern __shared__ __attribute__((aligned(16))) unsigned nb_lds[];
  float4* stage = reinterpret_cast<float4*>(nb_lds);                      // [NB_CHUNK]
  unsigned* skey = nb_lds + NB_CHUNK * 4;                                  // [NB_CHUNK]
  unsigned* gdest = skey + NB_CHUNK;                                       // [NB_CHUNK] global slot of staged record p
  unsigned* gbase = gdest + NB_CHUNK;                                      // [ngroups] first global slot of this chunk's run
  unsigned* lstart = gbase + s.ngroups;                                    // [ngroups] counts, then local exclusive starts
  unsigned* scratch = lstart + s.ngroups;                                  // [NB_THREADS / 64]
  const long long per_xcd = (s.nchunks + 7) / 8;
  const long long chunk = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (chunk >= s.nchunks) return;
  for (int i = threadIdx.x; i < s.ngroups; i += NB_THREADS) {
    gbase[i] = table_start[(long long)i * s.nchunks + chunk];
    lstart[i] = 0;
  }
  __syncthreads();
  const long long base = chunk * NB_CHUNK;
  unsigned key[NB_ITEMS], rk[NB_ITEMS];
  float4 rec[NB_ITEMS];
  bool ok[NB_ITEMS];
#pragma unroll
  for (int k = 0; k < NB_ITEMS; ++k) {
    const long long i = base + (long long)k * NB_THREADS + threadIdx.x;
    ok[k] = i < np;
    if (ok[k]) {
      key[k] = keys[i];
      rec[k] = make_float4((float)pos[i * 3 + 0], (float)pos[i * 3 + 1], (float)pos[i * 3 + 2], __int_as_float((int)i));
    }
  }
#pragma unroll
  for (int k = 0; k < NB_ITEMS; ++k)
    if (ok[k]) rk[k] = atomicAdd(&lstart[key[k] >> s.gshift], 1u);
  __syncthreads();
  const unsigned total = nb_block_scan(lstart, s.ngroups, scratch);
#pragma unroll
  for (int k = 0; k < NB_ITEMS; ++k)
    if (ok[k]) {
      const unsigned grp = key[k] >> s.gshift;
      const unsigned p_ = lstart[grp] + rk[k];
      gdest[p_] = gbase[grp] + rk[k];
      stage[p_] = rec[k];
      skey[p_] = key[k];
    }
  __syncthreads();
  for (unsigned t = threadIdx.x; t < total; t += NB_THREADS) {
    const unsigned d = gdest[t];
    rec1[d] = stage[t];
    key1[d] = skey[t];
  }
}

__global__ void __launch_bounds__(NB_THREADS)
    nb_fine_kernel(const float4* __restrict__ rec1, const unsigned* __restrict__ key1, NbGeom s,
                   const unsigned* __restrict__ table_start, unsigned* __restrict__ start, float4* __restrict__ srec) {
  extern __shared__ unsigned nb_lds[];          // [G] counts -> cursors, then scan scratch
  const int G = 1 << s.gshift;
  unsigned* cur = nb_lds;
  unsigned* scratch = nb_lds + G;
  const int grp = blockIdx.x;
  const unsigned gs = table_start[(long long)grp * s.nchunks];
  const unsigned ge = table_start[(long long)(grp + 1) * s.nchunks];   // [ngroups * nchunks] = total
  for (int i = threadIdx.x; i < G; i += NB_THREADS) cur[i] = 0;
  __syncthreads();
  constexpr int U = 4;   // loads of U strides are issued together: the sweeps are latency bound otherwise
  for (unsigned j0 = gs + threadIdx.x; j0 < ge; j0 += U * NB_THREADS) {
    unsigned key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + u * NB_THREADS;
      key[u] = j < ge ? key1[j] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (key[u] != 0xffffffffu) atomicAdd(&cur[key[u] & (G - 1)], 1u);
  }
  __syncthreads();
  // exclusive scan of the G counters, NB_THREADS * 4 at a time (G <= 32768 = 8 rounds)
  unsigned carry = 0;
  for (int c0 = 0; c0 < G; c0 += 4 * NB_THREADS) {
    const int n = min(4 * NB_THREADS, G - c0);
    const unsigned tot = nb_block_scan(cur + c0, n, scratch);
    for (int f = threadIdx.x; f < n; f += NB_THREADS) {
      const unsigned at = gs + carry + cur[c0 + f];
      cur[c0 + f] = at;
      const long long cell = (long long)grp * G + c0 + f;
      if (cell < s.ncell) start[cell] = at;
    }
    carry += tot;
    __syncthreads();
  }
  if (grp == s.ngroups - 1 && threadIdx.x == 0) start[s.ncell] = ge;
  __syncthreads();
  for (unsigned j0 = gs + threadIdx.x; j0 < ge; j0 += U * NB_THREADS) {
    unsigned key[U];
    float4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + u * NB_THREADS;
      key[u] = 0xffffffffu;
      if (j < ge) {
        key[u] = key1[j];
        r[u] = rec1[j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (key[u] != 0xffffffffu) srec[atomicAdd(&cur[key[u] & (G - 1)], 1u)] = r[u];
  }
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

__device__ __forceinline__ float4 srec_load(const float4* __restrict__ srec, unsigned i) { return srec[i]; }

// What a lattice point receives from its nearest particle's payload (C = 4: [rho v, rho]):
//   NN_RAW       the payload itself, channel-major
//   NN_VM        the BoxField form v = rho v / rho, mass = rho * Lcell^3 (interp.py:272-273): 4 channels
//   NN_VELOCITY  v alone: 3 channels          NN_MOMENTUM  p = v * mass = rho v * Lcell^3 (interp.py:523-525): 3 channels
//   NN_MOMBUG    the reference's momentum slip (py = pz = vx * mass): 3 channels
//   NN_ENERGY    E = mass |v|^2 = Lcell^3 |rho v|^2 / rho (interp.py:546): 1 channel
// -- the quantity a spectrum needs, formed where the winner is known, so that neither a fourth channel nor a weighted z pass
// moves bytes for it (C3: momentum -- 12.9 instead of 17.2 GB written, 4.3 GB less read per component in the z pass).
enum { NN_RAW = 0, NN_VM = 1, NN_VELOCITY = 2, NN_MOMENTUM = 3, NN_MOMBUG = 4, NN_ENERGY = 5 };
struct NnEmit {
  float vol;   // Lcell^3
  int form;
};
__host__ __device__ inline int nn_form_channels(int form, int C) {
  return form == NN_RAW ? C : form == NN_VM ? 4 : form == NN_ENERGY ? 1 : 3;
}
// the 1..4 output values of one lattice point from its payload
__device__ __forceinline__ float4 nn_form_apply(float4 v, NnEmit em) {
  const float inv = v.w != 0.f ? 1.f / v.w : 0.f;
  switch (em.form) {
    case NN_VM: return make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * em.vol);
    case NN_VELOCITY: return make_float4(v.x * inv, v.y * inv, v.z * inv, 0.f);
    case NN_MOMENTUM: return make_float4((v.x * inv) * (v.w * em.vol), (v.y * inv) * (v.w * em.vol), (v.z * inv) * (v.w * em.vol), 0.f);
    case NN_MOMBUG: { const float px = (v.x * inv) * (v.w * em.vol); return make_float4(px, px, px, 0.f); }
    case NN_ENERGY: {
      const float vx = v.x * inv, vy = v.y * inv, vz = v.z * inv;
      return make_float4((v.w * em.vol) * ((vx * vx + vy * vy) + vz * vz), 0.f, 0.f, 0.f);
    }
    default: return v;
  }
}
template <int C>
__device__ __forceinline__ void nn_emit(const float* __restrict__ payload, long long bi, long long q, long long nq,
                                        float* __restrict__ out, NnEmit em) {
  if constexpr (C == 4) {
    const float4 v = nn_form_apply(*reinterpret_cast<const float4*>(payload + bi * 4), em);
    const int nch = nn_form_channels(em.form, 4);
    out[q] = v.x;
    if (nch > 1) {
      out[nq + q] = v.y;
      out[2 * nq + q] = v.z;
    }
    if (nch > 3) out[3 * nq + q] = v.w;
  } else {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) out[(long long)ch * nq + q] = payload[bi * C + ch];
  }
}

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


// Exact search of one lattice point from global memory: Chebyshev rings r0, r0+1, ... of cells around the point's
// own cell c, until the best squared distance is provably smaller than the distance to every unsearched cell.
template <typename F>
__device__ __forceinline__ void nn_ring_search(const F* __restrict__ pos, const float4* __restrict__ srec,
                                               const unsigned* __restrict__ start, const NnGrid& g, const int (&c)[3],
                                               const double (&Q)[3], const float (&Qf)[3], float err, int r0, NnBest& b) {
  const int M = g.M;
  bool finished = false;
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
                    int* __restrict__ nn_idx, NnEmit em) {
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
  if (!finished) nn_ring_search<F>(pos, srec, start, g, c, Q, Qf, err, r0, b);
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
      if (out) nn_emit<C>(payload, bi, q, nq, out, em);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Particle-centric search on a UNIFORM lattice (both reference lattices are: interp.py:1063 is a
// linspace, scripts/parallel_optimized.py:343-346 is i * LCELL): the default path.
//
// The query-centric kernel above evaluates, for each of the ~21 lattice points per particle of C3,
// every candidate of a wave-wide union of cells (~200 distance evaluations per point where one point
// needs ~40).  Here the roles are swapped.  A workgroup owns a 16^3 tile of lattice points whose
// running minima live in LDS as 64-bit keys (float32 squared distance bits << 32 | particle index);
// every particle within a radius R of the tile visits only the lattice points inside its own R-box
// (about (2R/h)^3 of them) and lowers their keys with one LDS atomic-min.  Per (point, particle) pair
// the cost is one add, one LDS read of the current minimum and a compare; the atomic runs only when the
// pair can still win.  R follows the local density (R = kappa * n_local^(-1/3), kappa = 1.15: about 12 box visits, 6
// of them inside the sphere, per lattice point), so dense and sparse tiles do comparable work per lattice point.
//
// Exactness (the oracle's rule: smallest float64 squared distance, lowest index on ties):
//   * distances are evaluated in float32 on coordinates taken RELATIVE TO THE TILE CENTRE (subtracted in
//     float64 first), so the error bound `err` of a float32 distance is ~2^-24 of the tile size, not of
//     the box size;
//   * a particle that could be the exact nearest one or tie with it has a float32 distance inside the
//     screen (sqrt(b1) + 2 err)^2 of the smallest one b1 (as in the kernel above).  The atomic-min returns the
//     previous key: of the final smallest and second-smallest keys, whichever arrives later sees the other,
//     so comparing (old, mine) catches every point whose runner-up lies inside the winner's screen -- such a
//     point is marked `contested`;
//   * a point is `resolved` when its winner's distance is provably <= R: every particle within R has
//     visited it (the R-box of a particle covers its R-ball).
// Unresolved or contested points (voids, near-ties, duplicates: 2e-3 of the points at C3 with kappa = 1.15) are
// appended to a list and finished by nn_fallback_kernel with the exact float64 ring search.
//
// (Measured alternative, not kept: a point-centric "gather" form inside this kernel for sparse tiles -- region particles
// staged in LDS, every wave collecting the particles inside the R-box of a 4x4x4 block of points through the staged cell
// columns and evaluating them for its 64 points in registers, two candidates per packed instruction, no atomics: 35 ms at
// C3 against 33.5 ms for the scatter -- 17 ms in the ~45 evaluations per point, 8.6 ms collecting the candidates, 9.4 ms
// in what both forms share (region set-up, 17 GB of output) -- and its 28 KB of LDS cost the scatter form a workgroup
// per CU.)
// ------------------------------------------------------------------------------------------------
// A tile's search radius is shrunk (x 0.7) until the cell columns of its region fit the staging table.  When the particles are
// much denser than the lattice (cells narrower than about half a lattice step) the tile's own footprint does not fit at any
// radius: after this many steps (0.7^40 = 6e-7) the tile gives up and leaves all its points to the exact fallback -- the loop
// used to have no such exit and would never have ended there.
constexpr int NN_MAX_SHRINK = 40;
constexpr int NT_T = 16;                  // lattice tile edge
constexpr int NT_PTS = NT_T * NT_T * NT_T;
constexpr int NT_MAXCOL = 1024;           // cell columns of a tile's search region staged in LDS
constexpr int NT_THREADS = 256;
constexpr int NT_SEG = 1024;              // region particles sorted by box size at a time

struct NnScatterParams {
  const float4* srec;
  const unsigned* start;
  NnGrid g;
  const double* qx;   // device lattice axes (the slab's x rows start at qx[x0])
  const double* qy;
  const double* qz;
  int x0, nx, nqy, nqz;
  double a0[3], h[3];   // uniform model of the axes, q_a[i] ~ a0[a] + i h[a] (|deviation| <= 0.2 h, host-checked)
  float slack[3];       // measured max |q_a[i] - (a0 + i h)| / |h|, plus 2e-3 for the float32 index arithmetic
  float kappa;
  int ablate;           // timing experiments only: skip the atomics (results are garbage)
  const float* payload;
  float* out;
  NnEmit em;             // what a lattice point receives (C = 4): NN_RAW payload, NN_VM, NN_MOMENTUM, ... (nn_emit)
  int* nn_idx;
  unsigned* list;        // unresolved lattice points (offsets inside the slab), capacity = all points
  unsigned* list_count;
  const float* tile_r;   // column kernel: search radius per tile (nn_tile_radius_kernel), or NULL: computed in the kernel
#ifdef VPS_NN_STAMPS
  unsigned long long* stamps;   // timing-only builds: cycles per phase, summed over waves
#endif
};

// exclusive scan of a[0..1023] in LDS by 256 threads (4 entries each); returns the total
__device__ __forceinline__ unsigned block_scan_1024(unsigned* a, unsigned* wsum, int tid) {
  unsigned v0 = a[4 * tid], v1 = a[4 * tid + 1], v2 = a[4 * tid + 2], v3 = a[4 * tid + 3];
  const unsigned mine = v0 + v1 + v2 + v3;
  unsigned incl = mine;
  const int lane = tid & 63;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  unsigned run = base + incl - mine;
  a[4 * tid] = run; run += v0;
  a[4 * tid + 1] = run; run += v1;
  a[4 * tid + 2] = run; run += v2;
  a[4 * tid + 3] = run;
  __syncthreads();
  return total;
}

template <typename F, int C>
__global__ void __launch_bounds__(NT_THREADS) nn_scatter_kernel(const F* __restrict__ pos, const NnScatterParams p) {
  __shared__ unsigned long long key[NT_PTS + 8];
  __shared__ unsigned char contested[NT_PTS];
  __shared__ unsigned colbase[NT_MAXCOL + 4];   // exclusive prefix of the column run lengths
  __shared__ unsigned colg0[NT_MAXCOL];         // first record of each column's z run
  __shared__ float qf[3][NT_T];                 // lattice coordinates relative to the tile centre
  __shared__ unsigned wsum[4];
  __shared__ unsigned s_count;
  __shared__ unsigned stash[NT_SEG];          // packed box (and size class) of the region particles being sorted
  __shared__ unsigned short order[NT_SEG];    // their order, most rows first
  __shared__ unsigned cls[65];

  const int tid = threadIdx.x;
  const int ntz = (p.nqz + NT_T - 1) / NT_T, nty = (p.nqy + NT_T - 1) / NT_T;
  const long long tile = blockIdx.x;
  const int t0[3] = {(int)(tile / ((long long)ntz * nty)) * NT_T, (int)((tile / ntz) % nty) * NT_T, (int)(tile % ntz) * NT_T};
  const int nq[3] = {p.nx, p.nqy, p.nqz};
  const double* ax[3] = {p.qx + p.x0, p.qy, p.qz};
  int nt[3];
  double lo[3], hi[3], org[3];
  const NnGrid& g = p.g;
  const int M = g.M;
  const unsigned* __restrict__ start = p.start;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    nt[a] = min(NT_T, nq[a] - t0[a]);
    const double qa = ax[a][t0[a]], qb = ax[a][t0[a] + nt[a] - 1];
    lo[a] = fmin(qa, qb);
    hi[a] = fmax(qa, qb);
    org[a] = 0.5 * (qa + qb);
  }
  if (tid < 3 * NT_T) {
    const int a = tid / NT_T, i = tid % NT_T;
    qf[a][i] = (float)(ax[a][t0[a] + min(i, nt[a] - 1)] - org[a]);
  }
  if (tid == 0) s_count = 0;
  __syncthreads();

  // ---- local density -> search radius ----
  int c0[3], c1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    c0[a] = cell_coord(lo[a], g.lo[a], g.inv_w[a], M);
    c1[a] = cell_coord(hi[a], g.lo[a], g.inv_w[a], M);
  }
  {
    const int ny0 = c1[1] - c0[1] + 1, ncol0 = (c1[0] - c0[0] + 1) * ny0;
    unsigned cnt = 0;
    for (int col = tid; col < ncol0; col += NT_THREADS) {
      const long long row = ((long long)(c0[0] + col / ny0) * M + (c0[1] + col % ny0)) * M;
      cnt += start[row + c1[2] + 1] - start[row + c0[2]];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((tid & 63) == 0 && cnt) atomicAdd(&s_count, cnt);
  }
  __syncthreads();
  const double hmax = fmax(fabs(p.h[0]), fmax(fabs(p.h[1]), fabs(p.h[2])));
  const double rmax = (double)NT_T * hmax;
  double R;
  {
    const double vol = ((c1[0] - c0[0] + 1) * g.w[0]) * ((c1[1] - c0[1] + 1) * g.w[1]) * ((c1[2] - c0[2] + 1) * g.w[2]);
    const unsigned cnt = s_count;
    R = cnt ? (double)p.kappa * cbrt(vol / (double)cnt) : rmax;
    R = fmin(R, rmax);
  }
  // region of cells whose particles can lie within R of the tile; shrink R until its columns fit the staging
  int r0[3], r1[3], ncy, ncol;
  for (int shrink = 0;; ++shrink) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r0[a] = cell_coord(lo[a] - R, g.lo[a], g.inv_w[a], M);
      r1[a] = cell_coord(hi[a] + R, g.lo[a], g.inv_w[a], M);
    }
    ncy = r1[1] - r0[1] + 1;
    ncol = (r1[0] - r0[0] + 1) * ncy;
    if (ncol <= NT_MAXCOL) break;
    if (shrink >= NN_MAX_SHRINK) {   // the tile ALONE covers more cell columns than can be staged (particles far denser than the
      R = 0.0;                       // lattice): no region, every point of the tile goes to the exact fallback
      ncol = 0;
      break;
    }
    R *= 0.7;
  }
  const float Rf = (float)R;
  // float32 error bound of a distance between tile-relative coordinates (see nn_run for the constant)
  double half = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) half = fmax(half, 0.5 * (hi[a] - lo[a]));
  const float err = (float)((2.0 * half + R + hmax) * 2.5e-7);
  // keys start at R^2: a pair farther than that never enters
  const float c_init = Rf * Rf;
  // cheap upper bound of the screen (sqrt(c)(1+1e-6) + 2 err)^2 (1+1e-6) of a current minimum c <= R^2
  const float alpha = 1.0f + 4e-6f;
  const float beta = 4.01f * err * (Rf * 1.01f) + 4.01f * err * err;
  {
    const unsigned long long k0 = ((unsigned long long)__float_as_uint(c_init) << 32) | 0xffffffffull;
    for (int i = tid; i < NT_PTS; i += NT_THREADS) {
      key[i] = k0;
      contested[i] = 0;
    }
    for (int col = tid; col < NT_MAXCOL; col += NT_THREADS) {
      unsigned n = 0;
      if (col < ncol) {
        const long long row = ((long long)(r0[0] + col / ncy) * M + (r0[1] + col % ncy)) * M;
        const unsigned g0 = start[row + r0[2]];
        colg0[col] = g0;
        n = start[row + r1[2] + 1] - g0;
      }
      colbase[col] = n;
    }
  }
  __syncthreads();
  const unsigned total = block_scan_1024(colbase, wsum, tid);
  if (tid == 0) colbase[NT_MAXCOL] = total;
  __syncthreads();

  // ---- scatter: one particle per thread, particles dealt out in order of their box size ----
  // A particle's work is its (ix, iy) rows of up to 8 z-points.  Boxes differ a lot (those of the halo particles are
  // clipped by the tile), and a wave runs as long as its biggest box: so the region's particles are first counting-sorted
  // in LDS by their number of rows (largest first, NT_SEG particles at a time), and every wave then gets 64 particles of
  // about the same size; the rows of a box are walked in ONE loop (ix, iy advanced together), so different box shapes
  // with the same row count cost the same.
  float inv_h[3], a0f[3];   // index of a tile-relative coordinate v on axis a: (v - a0f) * inv_h
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    inv_h[a] = (float)(1.0 / p.h[a]);
    a0f[a] = (float)(p.a0[a] + (double)((a == 0 ? p.x0 : 0) + t0[a]) * p.h[a] - org[a]);
  }
  unsigned* key32 = reinterpret_cast<unsigned*>(key);
  // tile-relative position and record of flattened region particle j
  auto fetch_particle = [&](unsigned j, float (&pr)[3], int& oi) {
    int a_ = 0, b_ = ncol;   // column of item j: largest col with colbase[col] <= j
    while (b_ - a_ > 1) {
      const int m = (a_ + b_) >> 1;
      if (colbase[m] <= j) a_ = m; else b_ = m;
    }
    const float4 rec = srec_load(p.srec, colg0[a_] + (j - colbase[a_]));
    oi = __float_as_int(rec.w);
    double pd[3];
    if constexpr (sizeof(F) == 4) {
      pd[0] = (double)rec.x; pd[1] = (double)rec.y; pd[2] = (double)rec.z;
    } else {
      pd[0] = pos[(long long)oi * 3 + 0]; pd[1] = pos[(long long)oi * 3 + 1]; pd[2] = pos[(long long)oi * 3 + 2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) pr[a] = (float)(pd[a] - org[a]);
  };
  for (unsigned seg0 = 0; seg0 < total; seg0 += NT_SEG) {
    const unsigned nseg = min((unsigned)NT_SEG, total - seg0);
    if (tid < 64) cls[tid] = 0;
    __syncthreads();
    // pass 1: the box of every particle of the segment -> stash, histogram of the row-count classes
    for (unsigned jl = tid; jl < nseg; jl += NT_THREADS) {
      float pr[3];
      int oi;
      fetch_particle(seg0 + jl, pr, oi);
      // lattice indices (tile-local) whose coordinate can be within R of the particle: every integer of
      // [(p - R - a0)/h - slack, (p + R - a0)/h + slack], slack = measured deviation of the axis from uniform
      // (in units of h) + float32 rounding of this arithmetic
      unsigned pk = 0;
      bool any = true;
      int n_[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float u0 = (pr[a] - Rf - a0f[a]) * inv_h[a], u1 = (pr[a] + Rf - a0f[a]) * inv_h[a];
        const float ulo = fminf(u0, u1), uhi = fmaxf(u0, u1);
        const int lo_ = max(0, (int)ceilf(ulo - p.slack[a])), hi_ = min(nt[a] - 1, (int)floorf(uhi + p.slack[a]));
        any = any && (lo_ <= hi_);
        n_[a] = hi_ - lo_ + 1;
        pk |= ((unsigned)(lo_ & 15) | ((unsigned)(hi_ & 15) << 4)) << (8 * a);
      }
      if (any) {
        const int c_ = 63 - min(63, n_[0] * n_[1] - 1);            // class 0 = 64 or more rows
        pk |= 0x80000000u | ((unsigned)c_ << 24);
        atomicAdd(&cls[c_], 1u);
      }
      stash[jl] = pk;
    }
    __syncthreads();
    if (tid < 64) {   // exclusive scan of the 64 class counts by the first wave
      const unsigned v = cls[tid];
      unsigned incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        if (tid >= off) incl += o;
      }
      cls[tid] = incl - v;
      if (tid == 63) cls[64] = incl;
    }
    __syncthreads();
    const unsigned nvalid = cls[64];
    for (unsigned jl = tid; jl < nseg; jl += NT_THREADS) {
      const unsigned pk = stash[jl];
      if (pk & 0x80000000u) order[atomicAdd(&cls[(pk >> 24) & 63u], 1u)] = (unsigned short)jl;
    }
    __syncthreads();
    // pass 2: the scatter proper
    for (unsigned k_ = tid; k_ < nvalid; k_ += NT_THREADS) {
      const unsigned jl = order[k_];
      const unsigned pk = stash[jl];
      float pr[3];
      int oi;
      fetch_particle(seg0 + jl, pr, oi);
      const int i0x = (int)(pk & 15u), i1x = (int)((pk >> 4) & 15u), i0y = (int)((pk >> 8) & 15u), i1y = (int)((pk >> 12) & 15u);
      const int i0z = (int)((pk >> 16) & 15u), i1z = (int)((pk >> 20) & 15u);
      const int nrows = (i1x - i0x + 1) * (i1y - i0y + 1);
      const unsigned long long mine_lo = (unsigned long long)(unsigned)oi;
      // z in chunks of 8 (one chunk when 2R < 8 h, the usual case): the chunk's fz^2 stay in registers, the 8 current
      // minima of a row are read together, so the LDS latency is paid once per row, not once per pair
      for (int zb = i0z; zb <= i1z; zb += 8) {
        float fz2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float fz = qf[2][min(zb + k, NT_T - 1)] - pr[2];
          fz2[k] = (zb + k <= i1z) ? fz * fz : INFINITY;     // beyond the box: never a candidate
        }
        int ix = i0x, iy = i0y;
        float fx = qf[0][ix] - pr[0];
        float fx2 = fx * fx;
        for (int t_ = 0; t_ < nrows; ++t_) {
          const float fy = qf[1][iy] - pr[1];
          const float t2 = fx2 + fy * fy;
          const int q0 = (ix * NT_T + iy) * NT_T + zb;
          // The 8 current minima of the row are screened into a bit mask: a pair goes on only if it can still win or tie.
          // The survivors (few per lane and row) are then worked off lowest bit first, so a wave issues as many atomic
          // instructions per row as its busiest lane has survivors, not eight.  The atomic-min returns the previous key:
          // of the final smallest and second-smallest keys whichever arrives later sees the other one, so comparing
          // (old, mine) catches every point whose runner-up lies inside the winner's screen -- `contested`.
          unsigned m = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float cur = __uint_as_float(key32[2 * (q0 + k) + 1]);   // (the array is padded by 8 keys)
            const float d2 = t2 + fz2[k];
            m |= (d2 <= cur * alpha + beta && d2 <= c_init) ? (1u << k) : 0u;
          }
          if (p.ablate) m = 0;
          while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            const float fz = qf[2][min(zb + k, NT_T - 1)] - pr[2];
            const float d2 = t2 + fz * fz;
            const int q = q0 + k;
            const unsigned long long mine = ((unsigned long long)__float_as_uint(d2) << 32) | mine_lo;
            const unsigned long long old = atomicMin(&key[q], mine);
            const unsigned long long klo = old < mine ? old : mine, khi = old < mine ? mine : old;
            const float dlo = __uint_as_float((unsigned)(klo >> 32)), dhi = __uint_as_float((unsigned)(khi >> 32));
            // (a key that still carries the start value 0xffffffff is nobody's candidate)
            if ((unsigned)khi != 0xffffffffu && (unsigned)klo != 0xffffffffu && (unsigned)klo != (unsigned)khi &&
                dhi <= dlo * alpha + beta) {
              const float rb = sqrtf(dlo) * 1.000001f + 2.f * err;
              if (dhi <= rb * rb * 1.000001f) contested[q] = 1;
            }
          }
          if (++iy > i1y) {            // next row of the box
            iy = i0y;
            ix = min(ix + 1, NT_T - 1);
            fx = qf[0][ix] - pr[0];
            fx2 = fx * fx;
          }
        }
      }
    }
    __syncthreads();   // stash / order are rewritten by the next segment
  }

  // ---- epilogue: 16 consecutive z per row leave together ----
  const long long nqs = (long long)p.nx * p.nqy * p.nqz;
  for (int i = tid; i < NT_PTS; i += NT_THREADS) {
    const int iz = i % NT_T, iy = (i / NT_T) % NT_T, ix = i / (NT_T * NT_T);
    if (ix >= nt[0] || iy >= nt[1] || iz >= nt[2]) continue;
    const unsigned long long k = key[i];
    const unsigned bi = p.ablate ? 0u : (unsigned)k;     // (timing experiments: any valid index)
    const float d = __uint_as_float((unsigned)(k >> 32));
    const long long q = ((long long)(t0[0] + ix) * p.nqy + (t0[1] + iy)) * p.nqz + (t0[2] + iz);
    // settled: the winner's whole screen lies inside the sphere of radius R (every particle that could beat or tie it
    // has therefore visited), and no runner-up was seen inside that screen (so the float32 winner is the exact one;
    // ties and near-ties land in the fallback)
    const float rb = sqrtf(d) * 1.000001f + 2.f * err;
    const float screen = rb * rb * 1.000001f;
    const bool ok = (bi != 0xffffffffu && screen <= c_init && !contested[i]) || p.ablate;
    if (ok) {
      if (p.nn_idx) p.nn_idx[q] = (int)bi;
      if (p.out) nn_emit<C>(p.payload, bi, q, nqs, p.out, p.em);
    } else {
      p.list[atomicAdd(p.list_count, 1u)] = (unsigned)q;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Column-register search (the default where the lattice is about as fine as the particles are dense: a few lattice
// points per particle, both BASELINE NN configs).  The scatter kernel above spends two thirds of its time around LDS
// atomics issued from partly filled waves.  Here nothing is shared: a LANE owns a column of NC_TZ = 32 lattice points
// (fixed ix, iy; all z of the tile) and keeps their running minima -- smallest and second-smallest float32 squared
// distance and the winner's index -- in 96 registers; a WAVE owns an 8 x 8 patch of columns and walks, one particle at
// a time, the list of particles whose R-disc meets its patch.  A particle's coordinates are wave-uniform (v_readlane
// of a register that holds 64 list entries at once), every lane evaluates it against its own column, and only for
// the 8 z-points of the particle's z-window [z0, z0 + 8), which is where its R-ball can reach (R <= (3.5 - slack) h_z
// makes 8 points enough).  The window start decides which registers are touched, so the wave's list is counting-sorted
// by z0 first and each of the 25 classes runs its own straight-line code with static register indices: per (particle,
// z) one subtract, one fma, and on the KEY (distance bits | staged slot, see NC_KEYBITS) a bit-field insert, an unsigned
// median-of-three (new runner-up) and an unsigned minimum -- no LDS traffic, no atomics, no divergence, results independent
// of any order.  About 90 pair evaluations per lattice point instead of the scatter kernel's 12, at a tenth of the cost each.
// Exactness is argued as for the scatter kernel: every particle within R of a point has been evaluated for it (the
// staged region covers the tile + R, a wave's list every disc that meets its patch, a window every z within R); the
// point is settled when the winner's screen (sqrt(b1) + 2 err)^2 lies inside R^2 and the runner-up outside that
// screen; all other points (voids, near ties, duplicates) go to the list of nn_fallback_kernel.
// The epilogue forms v = rho v / rho, m = rho Lcell^3 (or gathers the payload) and writes each lane's 32 consecutive
// z as whole 128-byte lines.
// ------------------------------------------------------------------------------------------------
constexpr int NC_TX = 16, NC_TY = 16, NC_TZ = 32;
constexpr int NC_THREADS = 256;
constexpr int NC_WIN = 8;                       // z-points a particle is evaluated for
constexpr int NC_NCLS = NC_TZ - NC_WIN + 1;     // window starts 0 .. 24
constexpr int NC_SEG = 1216;                    // region particles staged in LDS at a time (slot NC_SEG: the dummy record)
constexpr int NC_MAXCOL = 1024;
constexpr int NC_SMAX = NC_SEG / 4 - 2;         // steps of a wave's padded quadrant lists (two more are read ahead)

// search radius of a tile from the particles in the cells c0..c1 that hold its lattice points
__device__ __forceinline__ double nc_radius(const NnGrid& g, const int (&c0)[3], const int (&c1)[3], unsigned cnt, float kappa,
                                            float slack_z, double hz) {
  const double rcap = (0.5 * (NC_WIN - 1) - (double)slack_z) * fabs(hz) * 0.999;
  const double vol = ((c1[0] - c0[0] + 1) * g.w[0]) * ((c1[1] - c0[1] + 1) * g.w[1]) * ((c1[2] - c0[2] + 1) * g.w[2]);
  const double R = cnt ? (double)kappa * cbrt(vol / (double)cnt) : rcap;
  return fmin(R, rcap);
}

// one thread per tile of the column kernel: its search radius (what the kernel's own prologue would compute, moved out of
// its critical path -- the tile's cells, their particle count, kappa * n^(-1/3), the window cap)
__global__ void __launch_bounds__(256) nn_tile_radius_kernel(const NnScatterParams p, long long ntiles, float* __restrict__ tile_r) {
  const long long tile = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= ntiles) return;
  constexpr int TD[3] = {NC_TX, NC_TY, NC_TZ};
  const int ntz = (p.nqz + NC_TZ - 1) / NC_TZ, nty = (p.nqy + NC_TY - 1) / NC_TY;
  const int t0[3] = {(int)(tile / ((long long)ntz * nty)) * NC_TX, (int)((tile / ntz) % nty) * NC_TY, (int)(tile % ntz) * NC_TZ};
  const int nq[3] = {p.nx, p.nqy, p.nqz};
  const double* ax[3] = {p.qx + p.x0, p.qy, p.qz};
  const NnGrid& g = p.g;
  int c0[3], c1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int nt = min(TD[a], nq[a] - t0[a]);
    const double qa = ax[a][t0[a]], qb = ax[a][t0[a] + nt - 1];
    c0[a] = cell_coord(fmin(qa, qb), g.lo[a], g.inv_w[a], g.M);
    c1[a] = cell_coord(fmax(qa, qb), g.lo[a], g.inv_w[a], g.M);
  }
  unsigned cnt = 0;
  for (int cx = c0[0]; cx <= c1[0]; ++cx)
    for (int cy = c0[1]; cy <= c1[1]; ++cy) {
      const long long row = ((long long)cx * g.M + cy) * g.M;
      cnt += p.start[row + c1[2] + 1] - p.start[row + c0[2]];
    }
  tile_r[tile] = (float)nc_radius(g, c0, c1, cnt, p.kappa, p.slack[2], p.h[2]);
}

__device__ __forceinline__ float nc_readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Running minima as KEYS: the float32 squared distance with its low NC_KEYBITS mantissa bits replaced by the particle's slot in
// the staged list -- (d2 & ~mask) | slot.  Squared distances are non-negative, so their bit patterns order like unsigned
// integers, and one v_min_u32 keeps winner and distance together (no compare + two selects), one v_med3_u32 the runner-up.
// The price is that the distance is known to 2^-12 only; the settle test of the epilogue works with the bounds that leaves
// (near ties inside 2.5e-4 go to the exact fallback: 4e5 more points at C3, against a sixth of the loop's instructions).
// Slot NC_RESOLVED marks a key whose winner has already been looked up (earlier segment; bi[] holds its index).
constexpr int NC_KEYBITS = 11;
constexpr unsigned NC_SLOTMASK = (1u << NC_KEYBITS) - 1u;
constexpr unsigned NC_RESOLVED = NC_SLOTMASK;
static_assert(NC_SEG < (int)NC_RESOLVED, "staged slots must fit the key's low bits");

__device__ __forceinline__ unsigned nc_bfi(unsigned mask, unsigned a, unsigned b) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));   // (mask & a) | (~mask & b); hipcc emits and + or
  return r;
}
__device__ __forceinline__ unsigned nc_umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));     // (no builtin for the unsigned median)
  return r;
}

// Window class Z0 of a wave's lists: steps [beg, end).  The wave's 8 x 8 patch is walked as four 4 x 4 QUADRANTS of 16 lanes,
// each with its own list of the particles whose R-disc meets it (a disc covers 32 h^2 of the 207 h^2 footprint of the whole
// patch + R, of a quadrant's 108 h^2: half the pair evaluations).  The window start still decides which registers are touched,
// so the four lists advance class by class in lockstep: entry (step, quadrant) of `ordq` is a staged slot, or the dummy slot
// whose record is NaN -- its keys lose every unsigned comparison -- where a quadrant has fewer particles of this class than the
// longest of the four.  A lane reads its quadrant's slot and record from LDS one step ahead (the four addresses of a wave
// broadcast inside their 16 lanes) where the whole-patch form read lanes of a register.
typedef float nc_f4 __attribute__((ext_vector_type(4)));
// The read pipeline of a lane through its quadrant's padded list: record and slot of the current step, slot of the next one.
// It runs straight through the class boundaries (the steps of class c + 1 follow those of class c), so a class costs no
// start-up reads of its own.
struct NcPipe {
  nc_f4 r_cur;
  unsigned s_cur, s_nxt;
  unsigned o_lds;       // LDS byte address of the current step's entry of this lane's quadrant
  unsigned p_lds;       // LDS byte address of the staged records
};

// one step: the record of step i + 1 and the slot of step i + 2 are requested, step i (rec, slot) is worked through, then the
// wait.  PIPE: requests and wait are inline assembly -- left to itself the compiler folds the pipeline back into "read the
// slot, wait, read the record, wait" in front of every step (volatile reads become flat loads with a wait each; offsets made
// opaque get the reads serialised on a shared register).  The wait is tied to the step's last key, so it cannot be scheduled
// in front of the arithmetic.  Between request and wait the destination registers r_n / s_n2 hold nothing yet: the compiler
// must not copy or spill them there.  It has no reason to in the common path -- the build checks exactly that on the device
// assembly of every build (vpower/_asmcheck.py) and refuses to link otherwise -- and the rare paths (further segments of a clump, next to spilling staging code) take
// the form without assembly (PIPE = false).
template <int Z0, bool PIPE>
__device__ __forceinline__ void nc_step(unsigned (&k1)[NC_TZ], unsigned (&k2)[NC_TZ], const float (&qz)[NC_WIN], float qxl, float qyl,
                                        const nc_f4 rec, const unsigned slot, const unsigned s_nxt, unsigned& o_lds, unsigned p_lds,
                                        nc_f4& r_n, unsigned& s_n2) {
  typedef __attribute__((address_space(3))) const nc_f4 lds_f4;
  typedef __attribute__((address_space(3))) const unsigned short lds_u16;
  if constexpr (PIPE) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_u16 %1, %3 offset:16"
                 : "=&v"(r_n), "=&v"(s_n2)
                 : "v"(p_lds + s_nxt * 16u), "v"(o_lds));   // (beyond a list's end: the padding -- read, not used)
  } else {
    r_n = *(lds_f4*)(uintptr_t)(p_lds + s_nxt * 16u);
    s_n2 = *(lds_u16*)(uintptr_t)(o_lds + 16u);
  }
  o_lds += 8;
  const float fx = qxl - rec.x, fy = qyl - rec.y;
  const float t2 = fmaf(fy, fy, fx * fx);
  // (two z per packed instruction -- v_pk_add_f32 / v_pk_fma_f32 -- measured: 14.9 against 14.7 ms, they issue at half rate)
#pragma unroll
  for (int k = 0; k < NC_WIN; ++k) {
    const float fz = qz[k] - rec.z;
    const float d2 = fmaf(fz, fz, t2);
    const unsigned key = nc_bfi(NC_SLOTMASK, slot, __float_as_uint(d2));   // (d2 & ~mask) | slot
    k2[Z0 + k] = nc_umed3(k1[Z0 + k], k2[Z0 + k], key);                  // k1 <= k2: the middle one is the new runner-up
    k1[Z0 + k] = min(k1[Z0 + k], key);
  }
  if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r_n), "+v"(s_n2), "+v"(k1[Z0 + NC_WIN - 1]));
}

template <int Z0, bool PIPE>
__device__ __forceinline__ void nc_run_class(unsigned (&k1)[NC_TZ], unsigned (&k2)[NC_TZ], NcPipe& pp, unsigned beg, unsigned end,
                                             float qzv, float qxl, float qyl) {
  if (beg >= end) return;   // (wave-uniform)
  float qz[NC_WIN];
#pragma unroll
  for (int k = 0; k < NC_WIN; ++k) qz[k] = nc_readlane(qzv, Z0 + k);   // (lane z of qzv holds the tile's z coordinate z)
  unsigned i = beg;
  if constexpr (PIPE) {
    for (; i + 2 <= end; i += 2) {   // two steps per trip: the records alternate between two register sets without copies
      nc_f4 r_b, r_a;
      unsigned s_2, s_3;
      nc_step<Z0, PIPE>(k1, k2, qz, qxl, qyl, pp.r_cur, pp.s_cur, pp.s_nxt, pp.o_lds, pp.p_lds, r_b, s_2);
      nc_step<Z0, PIPE>(k1, k2, qz, qxl, qyl, r_b, pp.s_nxt, s_2, pp.o_lds, pp.p_lds, r_a, s_3);
      pp.r_cur = r_a;
      pp.s_cur = s_2;
      pp.s_nxt = s_3;
    }
  }
  for (; i < end; ++i) {
    nc_f4 r_b;
    unsigned s_2;
    nc_step<Z0, PIPE>(k1, k2, qz, qxl, qyl, pp.r_cur, pp.s_cur, pp.s_nxt, pp.o_lds, pp.p_lds, r_b, s_2);
    pp.r_cur = r_b;
    pp.s_cur = pp.s_nxt;
    pp.s_nxt = s_2;
  }
}

// cendv: lane c holds the END step of class c (common to the four quadrants)
template <int Z0, bool PIPE>
__device__ __forceinline__ void nc_run_all(unsigned (&k1)[NC_TZ], unsigned (&k2)[NC_TZ], NcPipe& pp, unsigned cendv, float qzv,
                                           float qxl, float qyl) {
  const unsigned beg = Z0 ? (unsigned)__builtin_amdgcn_readlane((int)cendv, Z0 ? Z0 - 1 : 0) : 0u;
  const unsigned end = (unsigned)__builtin_amdgcn_readlane((int)cendv, Z0);
  nc_run_class<Z0, PIPE>(k1, k2, pp, beg, end, qzv, qxl, qyl);
  if constexpr (Z0 + 1 < NC_NCLS) nc_run_all<Z0 + 1, PIPE>(k1, k2, pp, cendv, qzv, qxl, qyl);
}

template <typename F, int C>
__global__ void __launch_bounds__(NC_THREADS, 4) nn_column_kernel(const F* __restrict__ pos, const NnScatterParams p) {
  // one LDS block [ P | order | colbase | colg0 ]; the epilogue re-uses its head as the winners' image
  constexpr int NC_PBYTES = (NC_SEG + 1) * 16;           // staged records + the dummy
  constexpr int NC_SMEM = NC_PBYTES + 4 * NC_SEG * 2 + (NC_MAXCOL + 4) * 4 + NC_MAXCOL * 4;
  constexpr int NC_IMG = 64 * 33;                        // ints per wave: [column][z], one pad word per column
  static_assert(4 * NC_IMG * 4 <= NC_SMEM, "winner image must fit the staging block (all of it is dead by the epilogue)");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NC_SMEM];
  float4* P = reinterpret_cast<float4*>(smem);                                         // staged region particles: tile-relative position, index
  unsigned short (*order)[NC_SEG] = reinterpret_cast<unsigned short (*)[NC_SEG]>(smem + NC_PBYTES);   // per wave: [step][quadrant] staged slots, steps grouped by z-window start
  unsigned* colbase = reinterpret_cast<unsigned*>(smem + NC_PBYTES + 4 * NC_SEG * 2);   // [NC_MAXCOL + 4] exclusive prefix of the column run lengths
  unsigned* colg0 = colbase + NC_MAXCOL + 4;                                            // [NC_MAXCOL] first record of each column's z run
  __shared__ float qf[3][NC_TZ];                        // lattice coordinates relative to the tile centre
  __shared__ unsigned cls[4][32];                       // per wave: END step of every window class (common to the four quadrants)
  __shared__ unsigned qcnt[4][4][32];                   // per wave and quadrant: class counts, then write cursors
  __shared__ unsigned wsum[4];
  __shared__ unsigned s_count, s_n;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef VPS_NN_STAMPS
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define NC_STAMP(i)                                                                         \
  do {                                                                                      \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                           \
    if (lane == 0) atomicAdd(&p.stamps[i], now_ - st_prev);                                 \
    st_prev = __builtin_amdgcn_s_memtime();                                                 \
  } while (0)
#else
#define NC_STAMP(i) do {} while (0)
#endif
  constexpr int TD[3] = {NC_TX, NC_TY, NC_TZ};
  const int ntz = (p.nqz + NC_TZ - 1) / NC_TZ, nty = (p.nqy + NC_TY - 1) / NC_TY;
  // (every XCD walking ONE contiguous range of tiles, so that its L2 serves the halo particles neighbouring tiles share:
  // measured 16.5 against 16.2 ms -- not kept)
  const long long tile = blockIdx.x;
  const int t0[3] = {(int)(tile / ((long long)ntz * nty)) * NC_TX, (int)((tile / ntz) % nty) * NC_TY, (int)(tile % ntz) * NC_TZ};
  const int nq[3] = {p.nx, p.nqy, p.nqz};
  const double* ax[3] = {p.qx + p.x0, p.qy, p.qz};
  int nt[3];
  double lo[3], hi[3], org[3];
  const NnGrid& g = p.g;
  const int M = g.M;
  const unsigned* __restrict__ start = p.start;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    nt[a] = min(TD[a], nq[a] - t0[a]);
    const double qa = ax[a][t0[a]], qb = ax[a][t0[a] + nt[a] - 1];
    lo[a] = fmin(qa, qb);
    hi[a] = fmax(qa, qb);
    org[a] = 0.5 * (qa + qb);
  }
  if (tid < 3 * NC_TZ) {
    const int a = tid / NC_TZ, i = tid % NC_TZ;
    qf[a][i] = (float)(ax[a][t0[a] + min(i, nt[a] - 1)] - org[a]);
  }
  if (tid == 0) {
    s_count = 0;
    const float nanf_ = __uint_as_float(0x7fc00000u);
    P[NC_SEG] = make_float4(nanf_, nanf_, nanf_, __int_as_float(-1));   // the dummy record: every key made from it is a NaN pattern
  }
  __syncthreads();

  // ---- local density -> search radius (as in nn_scatter_kernel), capped by what a window of NC_WIN z-points covers ----
  const double hmax = fmax(fabs(p.h[0]), fmax(fabs(p.h[1]), fabs(p.h[2])));
  double R;
  if (p.tile_r) {
    R = (double)p.tile_r[tile];     // nn_tile_radius_kernel did this once for all tiles (one dependent round trip less here)
  } else {
    int c0[3], c1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      c0[a] = cell_coord(lo[a], g.lo[a], g.inv_w[a], M);
      c1[a] = cell_coord(hi[a], g.lo[a], g.inv_w[a], M);
    }
    const int ny0 = c1[1] - c0[1] + 1, ncol0 = (c1[0] - c0[0] + 1) * ny0;
    unsigned cnt = 0;
    for (int col = tid; col < ncol0; col += NC_THREADS) {
      const long long row = ((long long)(c0[0] + col / ny0) * M + (c0[1] + col % ny0)) * M;
      cnt += start[row + c1[2] + 1] - start[row + c0[2]];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0 && cnt) atomicAdd(&s_count, cnt);
    __syncthreads();
    R = nc_radius(g, c0, c1, s_count, p.kappa, p.slack[2], p.h[2]);
  }
  int r0[3], r1[3], ncy, ncol;
  for (int shrink = 0;; ++shrink) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r0[a] = cell_coord(lo[a] - R, g.lo[a], g.inv_w[a], M);
      r1[a] = cell_coord(hi[a] + R, g.lo[a], g.inv_w[a], M);
    }
    ncy = r1[1] - r0[1] + 1;
    ncol = (r1[0] - r0[0] + 1) * ncy;
    if (ncol <= NC_MAXCOL) break;
    if (shrink >= NN_MAX_SHRINK) {   // the tile ALONE covers more cell columns than can be staged (particles far denser than the
      R = 0.0;                       // lattice): no region, every point of the tile goes to the exact fallback
      ncol = 0;
      break;
    }
    R *= 0.7;
  }
  const float Rf = (float)R;
  float halff[3];
  double half = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    half = fmax(half, 0.5 * (hi[a] - lo[a]));
    halff[a] = (float)(0.5 * (hi[a] - lo[a]));
  }
  // float32 error bound of a distance between tile-relative coordinates (nn_run: 2.5e-7 per unit of coordinate size)
  const float err = (float)((2.0 * half + R + hmax) * 2.5e-7);
  const float c_init = Rf * Rf;
  const float reach = Rf * 1.00001f + 4.f * err;    // a particle farther than this from a box cannot be within R of its points
  for (int col = tid; col < NC_MAXCOL; col += NC_THREADS) {
    unsigned n = 0;
    if (col < ncol) {
      const long long row = ((long long)(r0[0] + col / ncy) * M + (r0[1] + col % ncy)) * M;
      const unsigned g0 = start[row + r0[2]];
      colg0[col] = g0;
      n = start[row + r1[2] + 1] - g0;
    }
    colbase[col] = n;
  }
  __syncthreads();
  const unsigned total = block_scan_1024(colbase, wsum, tid);

  // this lane's column and this wave's patch
  const int px0 = (wv & 1) * 8, py0 = (wv >> 1) * 8;
  const int ix = px0 + (lane >> 3), iy = py0 + (lane & 7);
  const float qxl = qf[0][min(ix, NC_TX - 1)], qyl = qf[1][min(iy, NC_TY - 1)];
  // the patch's four quadrants: x halves a = 0, 1 (columns px0 + 4 a .. + 3), y halves b likewise; centre and half extent
  // (tile-relative).  A lane's quadrant is 2 a + b of its own column.
  float qcx[2], qhx[2], qcy[2], qhy[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float xa = qf[0][px0 + 4 * a], xb = qf[0][px0 + 4 * a + 3], ya = qf[1][py0 + 4 * a], yb = qf[1][py0 + 4 * a + 3];
    qcx[a] = 0.5f * (xa + xb); qhx[a] = 0.5f * fabsf(xb - xa);
    qcy[a] = 0.5f * (ya + yb); qhy[a] = 0.5f * fabsf(yb - ya);
  }
  const int quad = ((lane >> 5) & 1) * 2 + ((lane >> 2) & 1);
  const float qzv = qf[2][lane & 31];   // the tile's z coordinates, one per lane (read back lane by lane: v_readlane)
  const bool patch_live = px0 < nt[0] && py0 < nt[1];     // (wave-uniform) the patch holds lattice points at all
  const float inv_hz = (float)(1.0 / p.h[2]);
  const float a0z = (float)(p.a0[2] + (double)t0[2] * p.h[2] - org[2]);   // model coordinate of the tile's z index 0
  // ---- staging: the next particles of the region that can lie within R of the tile -> P (tile-relative, compacted) ----
  // Every thread takes up to NC_CAND candidates j = cursor + k * 256 + tid; all their loads are in flight together.  Slots are
  // handed out in the order of j (per-wave ballots + a prefix over the 6 x 4 wave counts), so that when P overflows the cut is
  // clean: everything from the first candidate without a slot on is staged by the next segment.
  constexpr int NC_CAND = 6;
  __shared__ unsigned wcnt[NC_CAND][4];
  __shared__ unsigned s_next;
  auto stage = [&](unsigned& cursor) -> unsigned {
    __syncthreads();   // the previous segment's P / order are consumed
    if (tid == 0) s_next = 0xffffffffu;
    float4 rec[NC_CAND];
    bool in[NC_CAND];
    {
      // column of candidate j: the largest col with colbase[col] <= j -- a branch-free binary search with a fixed number of
      // steps, all NC_CAND searches of a thread advancing together (one LDS latency per step, not per step and candidate)
      unsigned jj[NC_CAND];
      int col[NC_CAND];
#pragma unroll
      for (int k = 0; k < NC_CAND; ++k) {
        jj[k] = cursor + (unsigned)(k * NC_THREADS + tid);
        in[k] = jj[k] < total;
        col[k] = 0;
      }
      for (int step = NC_MAXCOL / 2; step >= 1; step >>= 1) {
        if (step >= 2 * ncol) continue;      // (uniform) beyond the table
#pragma unroll
        for (int k = 0; k < NC_CAND; ++k) {
          const int m = col[k] + step;
          const unsigned cb = colbase[min(m, NC_MAXCOL)];
          col[k] = (m < ncol && cb <= jj[k]) ? m : col[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NC_CAND; ++k)
        if (in[k]) rec[k] = srec_load(p.srec, colg0[col[k]] + (jj[k] - colbase[col[k]]));
    }
#pragma unroll
    for (int k = 0; k < NC_CAND; ++k) {
      if (in[k]) {
        double pd[3];
        if constexpr (sizeof(F) == 4) {
          pd[0] = (double)rec[k].x; pd[1] = (double)rec[k].y; pd[2] = (double)rec[k].z;
        } else {
          const long long oi = __float_as_int(rec[k].w);
          pd[0] = pos[oi * 3 + 0]; pd[1] = pos[oi * 3 + 1]; pd[2] = pos[oi * 3 + 2];
        }
        rec[k].x = (float)(pd[0] - org[0]);
        rec[k].y = (float)(pd[1] - org[1]);
        rec[k].z = (float)(pd[2] - org[2]);
        in[k] = fabsf(rec[k].x) <= halff[0] + reach && fabsf(rec[k].y) <= halff[1] + reach && fabsf(rec[k].z) <= halff[2] + reach;
      }
      const unsigned long long bal = __ballot(in[k]);
      if (lane == 0) wcnt[k][wv] = (unsigned)__popcll(bal);
    }
    __syncthreads();
    unsigned run = 0, n_all = 0;
#pragma unroll
    for (int k = 0; k < NC_CAND; ++k) {
      unsigned base = run;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned c_ = wcnt[k][w];
        if (w < wv) base += c_;
        run += c_;
      }
      const unsigned long long bal = __ballot(in[k]);
      const unsigned slot = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      if (in[k]) {
        if (slot < (unsigned)NC_SEG) P[slot] = rec[k];
        else atomicMin(&s_next, cursor + (unsigned)(k * NC_THREADS + tid));
      }
      n_all = run;
    }
    __syncthreads();
    const unsigned nx_ = s_next;
    cursor = nx_ != 0xffffffffu ? nx_ : min(cursor + (unsigned)(NC_CAND * NC_THREADS), total);
    return min(n_all, (unsigned)NC_SEG);
  };
  unsigned cursor = 0;
  NC_STAMP(0);
  unsigned staged = total ? stage(cursor) : 0u;   // the first segment is staged before the 96 accumulator registers exist
  NC_STAMP(1);
  unsigned k1[NC_TZ], k2[NC_TZ];     // smallest and second-smallest key per lattice point of this lane's column
  int bi[NC_TZ];                     // winner's particle index, looked up at the end of the segment that set it
#pragma unroll
  for (int z = 0; z < NC_TZ; ++z) {
    k1[z] = 0x7f800000u | NC_RESOLVED;     // +infinity, nobody's
    k2[z] = 0x7f800000u | NC_RESOLVED;
    bi[z] = -1;
  }
  auto process = [&](const unsigned n, auto pipe_tag) __attribute__((always_inline)) {
    constexpr bool PIPE = decltype(pipe_tag)::value;
    if (!patch_live || n == 0) return;   // (wave-uniform; stage()'s barriers are reached by every wave all the same)
    // ---- this wave's lists: per quadrant the particles whose R-disc meets it, counting-sorted by z-window start; the four
    // lists share the step axis -- class c takes max over the quadrants of its counts, shorter lists are padded ----
    auto wave_sync = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // z-window start of a particle, or -1: none of this wave's; `mask`: the quadrants (bit 2 a + b) its disc meets
    auto classify = [&](const float4 q, unsigned& mask) -> int {
      mask = 0;
      float dx2[2], dy2[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float dx = fmaxf(fabsf(q.x - qcx[a]) - qhx[a], 0.f), dy = fmaxf(fabsf(q.y - qcy[a]) - qhy[a], 0.f);
        dx2[a] = dx * dx;
        dy2[a] = dy * dy;
      }
      const float r2 = reach * reach;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (dx2[a] + dy2[b] <= r2) mask |= 1u << (2 * a + b);
      if (!mask) return -1;
      const float u0 = (q.z - Rf - a0z) * inv_hz, u1 = (q.z + Rf - a0z) * inv_hz;
      const int zl = (int)ceilf(fminf(u0, u1) - p.slack[2]), zh = (int)floorf(fmaxf(u0, u1) + p.slack[2]);
      if (zh < 0 || zl > nt[2] - 1) return -1;
      return min(max(zl, 0), NC_NCLS - 1);
    };
    unsigned* cnt = &qcnt[wv][0][0];
    unsigned lo_ = 0;
    while (lo_ < n) {   // (one trip -- unless a clump makes the lists longer than their LDS room: then the staged range is halved)
      unsigned hi_ = n, steps;
      for (;;) {
        cnt[lane] = 0;
        cnt[lane + 64] = 0;
        wave_sync();
        for (unsigned s_ = lo_ + lane; s_ < hi_; s_ += 64) {
          unsigned mask;
          const int c_ = classify(P[s_], mask);
          if (c_ >= 0) {
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_)
              if (mask & (1u << q_)) atomicAdd(&cnt[q_ * 32 + c_], 1u);
          }
        }
        wave_sync();
        {   // class c (lane c): steps = the longest of the four counts; exclusive prefix -> first step; cursors per quadrant
          const unsigned c0_ = cnt[lane & 31], c1_ = cnt[32 + (lane & 31)], c2_ = cnt[64 + (lane & 31)], c3_ = cnt[96 + (lane & 31)];
          const unsigned v = lane < NC_NCLS ? max(max(c0_, c1_), max(c2_, c3_)) : 0u;
          unsigned incl = v;
          for (int off = 1; off < 32; off <<= 1) {
            const unsigned o_ = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o_;
          }
          steps = (unsigned)__builtin_amdgcn_readlane((int)incl, 31);
          wave_sync();
          if (lane < 32) {
            cls[wv][lane] = incl;             // END step of class c
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) cnt[q_ * 32 + lane] = incl - v;
          }
        }
        if (steps <= (unsigned)NC_SMAX) break;     // (uniform)
        hi_ = lo_ + (hi_ - lo_) / 2;                // steps <= particles of the range: ends at the latest at NC_SMAX of them
      }
      {   // padding: every entry of the used steps (+ the two read ahead) starts as the dummy slot
        unsigned* o32 = reinterpret_cast<unsigned*>(order[wv]);
        const unsigned words = (steps + 2) * 2;
        for (unsigned w_ = lane; w_ < words; w_ += 64) o32[w_] = (unsigned)NC_SEG | ((unsigned)NC_SEG << 16);
      }
      wave_sync();
      for (unsigned s_ = lo_ + lane; s_ < hi_; s_ += 64) {
        unsigned mask;
        const int c_ = classify(P[s_], mask);
        if (c_ >= 0) {
#pragma unroll
          for (int q_ = 0; q_ < 4; ++q_)
            if (mask & (1u << q_)) order[wv][atomicAdd(&cnt[q_ * 32 + c_], 1u) * 4 + q_] = (unsigned short)s_;
        }
      }
      wave_sync();
      NC_STAMP(2);
      // (s_setprio 3 outside the class loops / 0 inside -- so that the latency-bound phases never queue behind another wave's
      // VALU stream -- measured: 18.5 against 18.4 ms, no effect)
      {
        NcPipe pp;
        const unsigned short* o = order[wv] + quad;
        pp.p_lds = (unsigned)(size_t)P;
        pp.o_lds = (unsigned)(size_t)o;
        pp.s_cur = o[0];
        pp.s_nxt = o[4];
        pp.r_cur = *reinterpret_cast<const nc_f4*>(&P[pp.s_cur]);
        nc_run_all<0, PIPE>(k1, k2, pp, cls[wv][lane & 31], qzv, qxl, qyl);
      }
      lo_ = hi_;
    }
    // winners set in this segment: their particle index, while the staged list still holds it
#pragma unroll
    for (int z = 0; z < NC_TZ; ++z) {
      const unsigned slot = k1[z] & NC_SLOTMASK;
      if (slot != NC_RESOLVED) {
        bi[z] = __float_as_int(P[slot].w);
        k1[z] |= NC_RESOLVED;
      }
    }
    NC_STAMP(3);
  };
  process(staged, std::true_type());
  // regions of more than NC_SEG particles (clumps): further segments.  Rare -- and marked so, because staging next to the
  // 96 live accumulators spills, which must not leak into the common path
  while (__builtin_expect(cursor < total, 0)) {
    staged = stage(cursor);
    process(staged, std::false_type());
  }

  // ---- epilogue ----
  // (1) settle test per point, the open ones appended to the fallback's list with ONE atomic per wave; (2) the winners go
  // through an LDS image [column][z] so that 8 lanes share a column's 32 z: every store instruction then writes whole 128-byte
  // lines, and a lane has four payload gathers per column in flight, two columns at a time.
  __syncthreads();   // every wave is done with P / order: the block becomes the winners' image
  NC_STAMP(5);
  int* img = reinterpret_cast<int*>(smem) + wv * NC_IMG;
  const long long nqs = (long long)p.nx * p.nqy * p.nqz;
  {
    const bool col_live = ix < nt[0] && iy < nt[1];
    const long long q0 = ((long long)(t0[0] + ix) * p.nqy + (t0[1] + iy)) * p.nqz + t0[2];
    unsigned open = 0;
#pragma unroll
    for (int z = 0; z < NC_TZ; ++z) {
      // settled: the winner's whole screen lies inside the sphere of radius R (every particle that could beat or tie it was
      // evaluated) and the runner-up lies outside that screen (the float32 winner is the exact one)
      // (the keys hold the distances to 2^-12: b1 from above, b2 from below)
      const float b1hi = __uint_as_float(k1[z] | NC_SLOTMASK), b2lo = __uint_as_float(k2[z] & ~NC_SLOTMASK);
      const float rb = sqrtf(b1hi) * 1.000001f + 2.f * err;
      const float screen = rb * rb * 1.000001f;
      const bool ok = bi[z] >= 0 && screen <= c_init && b2lo > screen;
      if (!ok && z < nt[2] && col_live) open |= 1u << z;
      img[lane * 33 + z] = max(bi[z], 0);
    }
    const unsigned mine = (unsigned)__popc(open);
    unsigned incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const unsigned wave_total = __builtin_amdgcn_readlane(incl, 63);
    if (wave_total) {   // (wave-uniform)
      unsigned base = 0;
      if (lane == 63) base = atomicAdd(p.list_count, wave_total);
      base = __builtin_amdgcn_readlane(base, 63) + incl - mine;
      while (open) {
        const int z = __builtin_ctz(open);
        open &= open - 1;
        p.list[base++] = (unsigned)(q0 + z);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  NC_STAMP(6);
  const bool wide = nt[2] == NC_TZ && (p.nqz & 3) == 0 && (nqs & 3) == 0 &&
                    (!p.out || (reinterpret_cast<size_t>(p.out) & 15) == 0) &&
                    (!p.nn_idx || (reinterpret_cast<size_t>(p.nn_idx) & 15) == 0);
  const int zc = lane & 7;
  // round r handles the columns (lane >> 3) + 8 (2 r + u), u = 0, 1.  The gathers of round r + 1 are requested BEFORE the
  // stores of round r are issued: a wave's memory operations retire in order, so gathers queued behind stores would wait
  // for the stores' acknowledgement as well.
  auto col_of = [&](int r, int u) { return (lane >> 3) + 8 * (2 * r + u); };
  auto col_live = [&](int col) { return px0 + (col >> 3) < nt[0] && py0 + (col & 7) < nt[1]; };
  auto col_q = [&](int col) {
    return ((long long)(t0[0] + px0 + (col >> 3)) * p.nqy + (t0[1] + py0 + (col & 7))) * p.nqz + t0[2] + 4 * zc;
  };
  if (wide && C == 4 && p.out) {
    if constexpr (C == 4) {
      int w[2][2][4];
      float4 v[2][2][4];
      auto request = [&](int r, int slot) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = col_of(r, u);
#pragma unroll
          for (int k = 0; k < 4; ++k) w[slot][u][k] = img[col * 33 + 4 * zc + k];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[slot][u][k] = *reinterpret_cast<const float4*>(p.payload + (long long)w[slot][u][k] * 4);
        }
      };
      auto emit = [&](int r, int slot) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = col_of(r, u);
          if (!col_live(col)) continue;
          const long long q = col_q(col);
          if (p.nn_idx) *reinterpret_cast<int4*>(p.nn_idx + q) = make_int4(w[slot][u][0], w[slot][u][1], w[slot][u][2], w[slot][u][3]);
          float4 (&x)[4] = v[slot][u];
          if (p.em.form != NN_RAW) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = nn_form_apply(x[k], p.em);
          }
          const int nch = nn_form_channels(p.em.form, 4);      // (uniform)
          float* o = p.out + q;
          *reinterpret_cast<float4*>(o) = make_float4(x[0].x, x[1].x, x[2].x, x[3].x);
          if (nch > 1) {
            *reinterpret_cast<float4*>(o + nqs) = make_float4(x[0].y, x[1].y, x[2].y, x[3].y);
            *reinterpret_cast<float4*>(o + 2 * nqs) = make_float4(x[0].z, x[1].z, x[2].z, x[3].z);
          }
          if (nch > 3) *reinterpret_cast<float4*>(o + 3 * nqs) = make_float4(x[0].w, x[1].w, x[2].w, x[3].w);
        }
      };
      request(0, 0);
      request(1, 1);
      emit(0, 0);
      request(2, 0);
      emit(1, 1);
      request(3, 1);
      emit(2, 0);
      emit(3, 1);
    }
  } else {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = col_of(r, u);
        if (!col_live(col)) continue;
        const long long q = col_q(col);
        int w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = img[col * 33 + 4 * zc + k];
        if (wide) {
          if (p.nn_idx) *reinterpret_cast<int4*>(p.nn_idx + q) = make_int4(w[0], w[1], w[2], w[3]);
          if (p.out) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
              *reinterpret_cast<float4*>(p.out + (long long)ch * nqs + q) =
                  make_float4(p.payload[(long long)w[0] * C + ch], p.payload[(long long)w[1] * C + ch],
                              p.payload[(long long)w[2] * C + ch], p.payload[(long long)w[3] * C + ch]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * zc + k < nt[2]) {
              if (p.nn_idx) p.nn_idx[q + k] = w[k];
              if (p.out) nn_emit<C>(p.payload, w[k], q + k, nqs, p.out, p.em);
            }
        }
      }
    }
  }
  NC_STAMP(4);
}

// the lattice points the scatter pass could not settle: exact float64 ring search, one point per thread
template <typename F, int C>
__global__ void __launch_bounds__(256)
    nn_fallback_kernel(const F* __restrict__ pos, const float4* __restrict__ srec, const unsigned* __restrict__ start,
                       NnGrid g, float err, const double* __restrict__ qx, const double* __restrict__ qy,
                       const double* __restrict__ qz, int x0, int nx, int nqy, int nqz,
                       const unsigned* __restrict__ list, const unsigned* __restrict__ list_count,
                       const float* __restrict__ payload, float* __restrict__ out, int* __restrict__ nn_idx, NnEmit em) {
  const unsigned n = *list_count;
  const long long nqs = (long long)nx * nqy * nqz;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned q = list[i];
    const int iz = (int)(q % (unsigned)nqz), iy = (int)((q / (unsigned)nqz) % (unsigned)nqy), ix = (int)(q / ((unsigned)nqz * (unsigned)nqy));
    const double Q[3] = {qx[x0 + ix], qy[iy], qz[iz]};
    const float Qf[3] = {(float)Q[0], (float)Q[1], (float)Q[2]};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = cell_coord(Q[a], g.lo[a], g.inv_w[a], g.M);
    NnBest b{INFINITY, INFINITY, 0x7fffffff};
    nn_ring_search<F>(pos, srec, start, g, c, Q, Qf, err, 0, b);
    if (nn_idx) nn_idx[q] = b.idx;
    if (out) nn_emit<C>(payload, b.idx, q, nqs, out, em);
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
  size_t header, count, fill, start, tiles, srec, list, list_count, keys, key1, rec1, table, total;
  long long ncell, ntiles;
  int M;
  bool sorted;      // cell list by the two-level LDS bucket sort (else: counting sort with global atomics)
  NbGeom nb;
};

NnLayout nn_layout(int64_t np, int /*is_f64*/, int64_t nq_slab) {
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
  l.list_count = off; off = align(off + sizeof(unsigned));
  l.list = off;   off = align(off + (size_t)(nq_slab > 0 ? nq_slab : 0) * sizeof(unsigned));   // unresolved points of the scatter pass
  // two-level sort: groups of 2^gshift cells, as few as keep a group's counters inside LDS
  NbGeom& nb = l.nb;
  nb.ncell = l.ncell;
  nb.nchunks = (np + NB_CHUNK - 1) / NB_CHUNK;
  nb.gshift = 0;
  while (nb.gshift < NB_MAXG_SHIFT && ((l.ncell + (1LL << nb.gshift) - 1) >> nb.gshift) > 512) ++nb.gshift;
  nb.ngroups = (int)((l.ncell + (1LL << nb.gshift) - 1) >> nb.gshift);
  l.sorted = np >= 4 * NB_CHUNK && nb.ngroups <= NB_MAXGROUPS && vps_option("nn_build_atomic", 0) == 0;
  l.keys = l.key1 = l.rec1 = l.table = off;
  if (l.sorted) {
    const long long nt = (long long)nb.ngroups * nb.nchunks;
    l.keys = off;  off = align(off + (size_t)np * sizeof(unsigned));
    l.key1 = off;  off = align(off + (size_t)np * sizeof(unsigned));
    l.rec1 = off;  off = align(off + (size_t)np * sizeof(float4));
    l.table = off; off = align(off + (size_t)(nt + 1) * sizeof(unsigned));
    if ((size_t)(scan_tiles(nt) + 1) > (size_t)(l.ntiles + 1))     // the scan's tile scratch is shared with the cell scan's
      l.sorted = false;
  }
  l.total = off;
  return l;
}

// q[i] ~ a0 + i h with |deviation| <= 0.2 |h| (what the scatter kernel's index ranges allow for)
bool nn_axis_uniform(const double* q, int n, double* a0, double* h, double* dev) {
  *a0 = q[0];
  *h = n > 1 ? (q[n - 1] - q[0]) / (double)(n - 1) : 1.0;
  *dev = 0.0;
  if (!(*h != 0.0) || !std::isfinite(*h)) return false;
  for (int i = 0; i < n; ++i) {
    const double d = std::fabs(q[i] - (*a0 + (double)i * *h)) / std::fabs(*h);
    if (!(d <= 0.2)) return false;
    *dev = std::fmax(*dev, d);
  }
  return true;
}

struct NnAxesModel {
  bool uniform;
  double a0[3], h[3], dev[3];
};

template <typename F>
int nn_run(vps_ctx* ctx, const F* pos, const float* payload, int64_t np, int C, int x0, int nx,
           int nqy, int nqz, const double* dqx, const double* dqy, const double* dqz, double qmax,
           const NnAxesModel& model, float* out, int* nn_idx, char* work, NnEmit em) {
  const long long nq_slab = (long long)nx * nqy * nqz;
  const NnLayout l = nn_layout(np, sizeof(F) == 8, nq_slab);
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
    const unsigned rb = pblocks < 1024u ? pblocks : 1024u;
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
  if (l.sorted) {
    const NbGeom& nb = l.nb;
    unsigned* keys = reinterpret_cast<unsigned*>(work + l.keys);
    unsigned* key1 = reinterpret_cast<unsigned*>(work + l.key1);
    float4* rec1 = reinterpret_cast<float4*>(work + l.rec1);
    unsigned* table = reinterpret_cast<unsigned*>(work + l.table);
    const size_t lds_h = sizeof(unsigned) * nb.ngroups;
    const size_t lds_s = sizeof(unsigned) * ((size_t)NB_CHUNK * 6 + 2 * nb.ngroups + NB_THREADS / 64);
    const size_t lds_f = sizeof(unsigned) * ((1u << nb.gshift) + NB_THREADS / 64);
    auto ks = nb_scatter_kernel<F>;
    VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    if (lds_f > 64 * 1024)
      VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(nb_fine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
    hipLaunchKernelGGL(nb_hist_kernel<F>, dim3((unsigned)nb.nchunks), dim3(NB_THREADS), lds_h, ctx->stream, pos, (long long)np, g, nb,
                       keys, table);
    launch_exclusive_scan(ctx->stream, table, (long long)nb.ngroups * nb.nchunks, tiles, table);   // in place
    const unsigned sgrid = (unsigned)(((nb.nchunks + 7) / 8) * 8);
    hipLaunchKernelGGL(ks, dim3(sgrid), dim3(NB_THREADS), lds_s, ctx->stream, pos, (long long)np, keys, nb, table, rec1, key1);
    hipLaunchKernelGGL(nb_fine_kernel, dim3((unsigned)nb.ngroups), dim3(NB_THREADS), lds_f, ctx->stream, rec1, key1, nb, table, start,
                       srec);
  } else {
    VPS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, sizeof(unsigned) * l.ncell, ctx->stream));
    VPS_HIP_CHECK(ctx, hipMemsetAsync(fill, 0, sizeof(unsigned) * l.ncell, ctx->stream));
    vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
    hipLaunchKernelGGL(nn_count_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, count);
    launch_exclusive_scan(ctx->stream, count, l.ncell, tiles, start);
    hipLaunchKernelGGL(nn_fill_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, g, start, fill, srec);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  // uniform lattice (both reference lattices): column-register search (lattice about as fine as the particles are dense) or
  // particle-centric scatter, + exact fallback for the few open points
  if (model.uniform && nq_slab < 0xffffffffLL && vps_option("nn_query_centric", 0) == 0) {
    NnScatterParams sp{};
    sp.srec = srec;
    sp.start = start;
    sp.g = g;
    sp.qx = dqx; sp.qy = dqy; sp.qz = dqz;
    sp.x0 = x0; sp.nx = nx; sp.nqy = nqy; sp.nqz = nqz;
    for (int a = 0; a < 3; ++a) { sp.a0[a] = model.a0[a]; sp.h[a] = model.h[a]; sp.slack[a] = (float)(model.dev[a] + 2e-3); }
    sp.kappa = (float)vps_option("nn_kappa", 1.15);
    if (!(sp.kappa > 0.2f && sp.kappa < 8.f)) sp.kappa = 1.15f;
#ifdef VPS_TIMING_VARIANTS
    sp.ablate = vps_option("nn_ablate", 0) != 0 ? 1 : 0;   // timing-only build: garbage results
#else
    sp.ablate = 0;
#endif
    sp.payload = payload;
    sp.out = out;
    sp.em = em;
    sp.nn_idx = nn_idx;
    sp.list = reinterpret_cast<unsigned*>(work + l.list);
    sp.list_count = reinterpret_cast<unsigned*>(work + l.list_count);
#ifdef VPS_NN_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc(&d_stamps, 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), ctx->stream);
    sp.stamps = d_stamps;
#endif
    // Which search: a z-window of NC_WIN lattice points must hold the R-ball of a particle at the MEAN density (denser tiles
    // shrink R, sparser ones are capped and leave more points to the fallback); option nn_column: 1 force, 0 never
    const double slack_z = model.dev[2] + 2e-3;
    const double rcap = (0.5 * (NC_WIN - 1) - slack_z) * std::fabs(model.h[2]) * 0.999;
    const double spacing = std::cbrt((g.w[0] * l.M) * (g.w[1] * l.M) * (g.w[2] * l.M) / (double)np);
    const double colopt = vps_option("nn_column", -1);
    const bool column = colopt >= 0 ? colopt != 0 : (double)sp.kappa * spacing <= rcap;
    const long long tiles = column ? (long long)((nx + NC_TX - 1) / NC_TX) * ((nqy + NC_TY - 1) / NC_TY) * ((nqz + NC_TZ - 1) / NC_TZ)
                                   : (long long)((nx + NT_T - 1) / NT_T) * ((nqy + NT_T - 1) / NT_T) * ((nqz + NT_T - 1) / NT_T);
    if (tiles > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_nn_resample: too many queries for one launch");
    VPS_HIP_CHECK(ctx, hipMemsetAsync(sp.list_count, 0, sizeof(unsigned), ctx->stream));
    sp.tile_r = nullptr;
    if (column && tiles <= l.ncell) {   // (the cell counters are dead once the list is built: room for one float per tile)
      float* tr = reinterpret_cast<float*>(count);
      vps_launch_timer tm(ctx, VPS_K_NN_BUILD);
      hipLaunchKernelGGL(nn_tile_radius_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, ctx->stream, sp, tiles, tr);
      sp.tile_r = tr;
    }
    {
      // (each launch has its own event bracket: a bracket around both would also count the host's gap between them)
#define VPS_NNS(CC)                                                                                            \
  do {                                                                                                         \
    {                                                                                                          \
      vps_launch_timer tm(ctx, VPS_K_NN_QUERY);                                                                \
      if (column)                                                                                              \
        hipLaunchKernelGGL((nn_column_kernel<F, CC>), dim3((unsigned)tiles), dim3(NC_THREADS), 0, ctx->stream, pos, sp); \
      else                                                                                                     \
        hipLaunchKernelGGL((nn_scatter_kernel<F, CC>), dim3((unsigned)tiles), dim3(NT_THREADS), 0, ctx->stream, pos, sp); \
    }                                                                                                          \
    vps_launch_timer tm2(ctx, VPS_K_NN_QUERY);                                                                 \
    hipLaunchKernelGGL((nn_fallback_kernel<F, CC>), dim3((unsigned)(ctx->num_cu * 4)), dim3(256), 0, ctx->stream, \
                       pos, srec, start, g, err, dqx, dqy, dqz, x0, nx, nqy, nqz, sp.list, sp.list_count, payload, \
                       out, nn_idx, em);                                                                       \
  } while (0)
      switch (C) {
        case 1: VPS_NNS(1); break;
        case 3: VPS_NNS(3); break;
        case 4: VPS_NNS(4); break;
        default: return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "nn_resample: C=%d channels (supported: 1,3,4)", C);
      }
#undef VPS_NNS
    }
    VPS_HIP_CHECK(ctx, hipGetLastError());
#ifdef VPS_NN_STAMPS
    {
      unsigned long long h_st[8];
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipMemcpy(h_st, d_stamps, sizeof(h_st), hipMemcpyDeviceToHost);
      double tot = 0;
      for (int i = 0; i < 7; ++i) tot += (double)h_st[i];
      fprintf(stderr, "[vps] nn column stamps: setup %.1f%% stage %.1f%% lists %.1f%% classes %.1f%% barrier %.1f%% settle %.1f%% output %.1f%% (sum %.3g cycles)\n",
              100 * h_st[0] / tot, 100 * h_st[1] / tot, 100 * h_st[2] / tot, 100 * h_st[3] / tot, 100 * h_st[5] / tot, 100 * h_st[6] / tot, 100 * h_st[4] / tot, tot);
    }
#endif
    if (ctx->timing || vps_option("nn_stats", 0) != 0) {   // diagnostics: how many points the scatter pass left open
      unsigned open_pts = 0;
      VPS_HIP_CHECK(ctx, hipMemcpyAsync(&open_pts, sp.list_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
      VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      ctx->nn_open_points = open_pts;
      if (vps_option("nn_stats", 0) != 0) fprintf(stderr, "[vps] nn: %u of %lld lattice points left to the exact fallback\n", open_pts, nq_slab);
    }
    return VPS_OK;
  }
  const long long qblocks = (long long)((nx + NN_BX - 1) / NN_BX) * ((nqy + NN_BY - 1) / NN_BY) * ((nqz + NN_BZ - 1) / NN_BZ);
  if (qblocks > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_nn_resample: too many queries for one launch");
  {
    vps_launch_timer tm(ctx, VPS_K_NN_QUERY);
#define VPS_NNQ(CC)                                                                                  \
  hipLaunchKernelGGL((nn_query_kernel<F, CC>), dim3((unsigned)qblocks), dim3(256), 0, ctx->stream, \
                     pos, srec, start, g, err, dqx, dqy, dqz, x0, nx, nqy, nqz, payload, out, nn_idx, em)
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

size_t vps_nn_workspace_bytes(int64_t np, int pos_is_f64, int64_t nq_slab) {
  if (np < 1) return 256;
  return nn_layout(np, pos_is_f64, nq_slab).total;
}

static int nn_resample_impl(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                            int64_t np, int C, const double* qx_host, int nqx, const double* qy_host, int nqy,
                            const double* qz_host, int nqz, int x0, int nx, float* out_dev,
                            int32_t* nn_idx_dev, void* work_dev, NnEmit em);

int vps_nn_resample(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                    int64_t np, int C, const double* qx_host, int nqx, const double* qy_host, int nqy,
                    const double* qz_host, int nqz, int x0, int nx, float* out_dev,
                    int32_t* nn_idx_dev, void* work_dev) {
  VPS_ENTER(ctx);
  return nn_resample_impl(ctx, pos_dev, pos_is_f64, payload_dev, np, C, qx_host, nqx, qy_host, nqy, qz_host, nqz, x0, nx,
                          out_dev, nn_idx_dev, work_dev, NnEmit{0.f, NN_RAW});
}

int vps_nn_resample_quantity(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* rhov_dev,
                             int64_t np, const double* qx_host, int nqx, const double* qy_host, int nqy,
                             const double* qz_host, int nqz, int x0, int nx, double Lcell, int quantity, int flags,
                             float* out_dev, int32_t* nn_idx_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!(Lcell > 0) || !out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample_quantity: bad Lcell / null output");
  int form;
  switch (quantity) {
    case VPS_VM: form = NN_VM; break;
    case VPS_VELOCITY: form = NN_VELOCITY; break;
    case VPS_MOMENTUM: form = (flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG) ? NN_MOMBUG : NN_MOMENTUM; break;
    case VPS_ENERGY: form = NN_ENERGY; break;
    default: return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample_quantity: quantity %d", quantity);
  }
  return nn_resample_impl(ctx, pos_dev, pos_is_f64, rhov_dev, np, 4, qx_host, nqx, qy_host, nqy, qz_host, nqz, x0, nx,
                          out_dev, nn_idx_dev, work_dev, NnEmit{(float)(Lcell * Lcell * Lcell), form});
}

int vps_nn_resample_field(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* rhov_dev,
                          int64_t np, const double* qx_host, int nqx, const double* qy_host, int nqy,
                          const double* qz_host, int nqz, int x0, int nx, double Lcell, float* out_dev,
                          int32_t* nn_idx_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!(Lcell > 0) || !out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_nn_resample_field: bad Lcell / null output");
  return nn_resample_impl(ctx, pos_dev, pos_is_f64, rhov_dev, np, 4, qx_host, nqx, qy_host, nqy, qz_host, nqz, x0, nx,
                          out_dev, nn_idx_dev, work_dev, NnEmit{(float)(Lcell * Lcell * Lcell), NN_VM});
}

static int nn_resample_impl(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                            int64_t np, int C, const double* qx_host, int nqx, const double* qy_host, int nqy,
                            const double* qz_host, int nqz, int x0, int nx, float* out_dev,
                            int32_t* nn_idx_dev, void* work_dev, NnEmit em) {
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
  NnAxesModel model;
  model.uniform = nn_axis_uniform(qx_host, nqx, &model.a0[0], &model.h[0], &model.dev[0]) &&
                  nn_axis_uniform(qy_host, nqy, &model.a0[1], &model.h[1], &model.dev[1]) &&
                  nn_axis_uniform(qz_host, nqz, &model.a0[2], &model.h[2], &model.dev[2]);
  {   // an axis with a single point has no spacing of its own: give it its neighbours' (it only scales the search radius cap)
    const int nn[3] = {nqx, nqy, nqz};
    double hm = 0.0;
    for (int a = 0; a < 3; ++a)
      if (nn[a] > 1) hm = std::fmax(hm, std::fabs(model.h[a]));
    for (int a = 0; a < 3; ++a)
      if (nn[a] == 1) model.h[a] = hm > 0.0 ? hm : 1.0;
  }
  if (pos_is_f64)
    return nn_run<double>(ctx, reinterpret_cast<const double*>(pos_dev), payload_dev, np, C, x0, nx, nqy,
                          nqz, dqx, dqy, dqz, qmax, model, out_dev, nn_idx_dev, work, em);
  return nn_run<float>(ctx, reinterpret_cast<const float*>(pos_dev), payload_dev, np, C, x0, nx, nqy, nqz,
                       dqx, dqy, dqz, qmax, model, out_dev, nn_idx_dev, work, em);
}

}  // extern "C"
