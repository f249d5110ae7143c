// Stage A1 (nearest-grid-point deposition) and A3 (field algebra).
//
// Replaces deposit_to_grid (vpower/interp.py:996-1015), density_velocity_vector
// (:199-213) and the elementwise steps of ann_interp_to_field /
// BoxField.{momentum,kinetic_energy}_power (:272-273, 523-525, 546).
//
// Deposition is a bucket scheme built for HBM3E + 160 KiB LDS:
//   1. sort    : the particles' records {cell-in-bucket, payload[C]} are grouped by BUCKET (a brick of
//                bx*by*bz cells whose C float channels fit an LDS tile, or a z-pass pencil on the
//                fused path) by a two-level LDS bucket sort -- chunk histograms, one scan, LDS-staged
//                runs, one workgroup per coarse group (see "Two-level bucket sort" below).  No global
//                atomic per particle, no random 20-byte writes.  A one-atomic-per-particle ranking
//                (brick_rank_kernel / brick_scatter_kernel) remains for bucket counts or key ranges
//                the sort does not cover.
//   2. buckets : one workgroup per bucket adds its records into an LDS tile and then either streams
//                the WHOLE tile out with full-width coalesced stores (rows of bz cells), applying the
//                field algebra (v = rho v / rho, momentum, kinetic energy) on the way out
//                (brick_accumulate_kernel), or feeds it straight into the z-pass FFT
//                (pencil_fft_z_kernel, fft.hip).
// Every cell of the slab is therefore written exactly once by plain stores: there is no
// grid memset, no global float atomic and no separate algebra pass.
//
// Cell indices must be BIT EXACT with numpy's  int((pos // Lcell) % N)  (SURVEY.md
// Q13): numpy evaluates floor_divide and remainder with its divmod algorithm in the
// dtype of `pos`, so that algorithm is restated here and this translation unit is
// compiled without floating-point contraction.
#pragma clang fp contract(off)

#include <type_traits>

#include "vps_internal.h"
#include "scan.h"

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

// fmod(q, n) for an integer-valued q and a positive integer n, both below 2^mantissa: one division.
// Exact: a non-integer q/n is at least 1/n away from every integer while its rounding error is below
// (q/n) 2^-mantissa < 1/n, so trunc() cannot cross one; trunc(q/n) n <= |q| and the remainder are exact.
// (fmod proper is a long routine, in double a very long one.)
template <typename F>
__device__ __forceinline__ F fmod_integral(F q, F n) {
  constexpr F lim = sizeof(F) == 4 ? F(16777216.0) : F(9007199254740992.0);
  if (fabs(q) < lim && n < lim) {
    const F r = q - trunc(q / n) * n;
    return copysign(r, q);          // fmod's zero carries the sign of the dividend
  }
  return dev_fmod<F>(q, n);
}

template <typename F>
__device__ __forceinline__ int cell_of(F x, F lcell, F nsize) {
  // int((x // Lcell) % N): float -> int cast truncates.
  // Fast path (every in-range input): numpy's floor_divide returns the exact floor of the real quotient
  // x / Lcell whenever that is far below 2^mantissa ((x - fmod) / Lcell is then an exact integer), and
  // the correctly rounded q = x / Lcell has the same floor unless q itself is an integer -- where the
  // sign of the exact residual fma(-q, Lcell, x) says whether the real quotient lies just below it.
  // The remainder of the integer fd by the integer N is one more division (exact: fd / N is at least
  // 1/N from any integer it does not equal, its rounding error below that).
  constexpr F lim = sizeof(F) == 4 ? F(4194304.0) : F(2251799813685248.0);   // 2^22, 2^51
  const F q = x / lcell;
  if (fabs(q) < lim && nsize < lim && lcell > F(0) && nsize >= F(1)) {
    F fd = floor(q);
    if (fd == q && fma(-q, lcell, x) < F(0)) fd -= F(1);
    return (int)(fd - floor(fd / nsize) * nsize);
  }
  // anything else (huge, inf, NaN): numpy's npy_divmod steps verbatim
  const F fdn = np_floor_divide<F>(x, lcell);
  F mod = fmod_integral<F>(fdn, nsize);
  if (mod != F(0)) {
    if ((nsize < F(0)) != (mod < F(0))) mod += nsize;
  } else {
    mod = copysign(F(0), nsize);
  }
  return (int)mod;
}

template <typename F>
__global__ void __launch_bounds__(256) cell_index_kernel(const F* __restrict__ pos, long long np,
                                                         F lcell, F nsize, int* __restrict__ cell) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) cell[i * 3 + a] = cell_of<F>(pos[i * 3 + a], lcell, nsize);
}

// ---- brick geometry ------------------------------------------------------------
struct Bricks {
  int N, x0, nx;        // grid and slab
  int bx, by, bz;       // brick extent in cells
  int nbx, nby, nbz;    // bricks per axis of the slab
  int cells;            // bx*by*bz
  int pow2, sy, sz;     // by, bz powers of two: log2(by), log2(bz)
};

// brick id and cell-in-brick of a particle, or false when it is outside the slab
template <typename F>
__device__ __forceinline__ bool locate(const F* __restrict__ pos, long long i, F lcell, F nsize,
                                       const Bricks& b, unsigned& brick, unsigned& loc) {
  const int cx = cell_of<F>(pos[i * 3 + 0], lcell, nsize) - b.x0;
  if (cx < 0 || cx >= b.nx) return false;
  const int cy = cell_of<F>(pos[i * 3 + 1], lcell, nsize);
  const int cz = cell_of<F>(pos[i * 3 + 2], lcell, nsize);
  if ((unsigned)cy >= (unsigned)b.N || (unsigned)cz >= (unsigned)b.N) return false;  // NaN / inf
  const int ix = cx / b.bx, iy = cy / b.by, iz = cz / b.bz;
  brick = (unsigned)((ix * b.nby + iy) * b.nbz + iz);
  loc = (unsigned)(((cx - ix * b.bx) * b.by + (cy - iy * b.by)) * b.bz + (cz - iz * b.bz));
  return true;
}

// Pass 1: one returning atomic per particle gives both the brick histogram and the
// particle's rank inside its brick.  key = brick * cells + cell-in-brick (0xffff.. = outside).
template <typename F>
__global__ void __launch_bounds__(256)
    brick_rank_kernel(const F* __restrict__ pos, long long np, F lcell, F nsize, Bricks b,
                      unsigned* __restrict__ count, unsigned long long* __restrict__ keys,
                      unsigned* __restrict__ ranks) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  unsigned brick, loc;
  if (locate<F>(pos, i, lcell, nsize, b, brick, loc)) {
    ranks[i] = atomicAdd(&count[brick], 1u);
    keys[i] = (unsigned long long)brick * (unsigned)b.cells + loc;
  } else {
    keys[i] = ~0ull;
  }
}

// Pass 2 (after the scan): record = {loc, payload[C]} as (C+1) 32-bit words goes to slot
// start[brick] + rank.  RHOV: payload is built from velocity and density on the fly
// ([rho vx, rho vy, rho vz, rho], interp.py:199-213).
template <int C, bool RHOV>
__global__ void __launch_bounds__(256)
    brick_scatter_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ ranks,
                         const float* __restrict__ payload, const float* __restrict__ rho, long long np,
                         unsigned cells, const unsigned* __restrict__ start,
                         unsigned* __restrict__ records) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const unsigned long long key = keys[i];
  if (key == ~0ull) return;
  const unsigned brick = (unsigned)(key / cells), loc = (unsigned)(key % cells);
  float val[C];
  if constexpr (RHOV) {
    static_assert(C == 4, "rho*v payload has four channels");
    const float r = rho[i];
    val[0] = payload[i * 3 + 0] * r;
    val[1] = payload[i * 3 + 1] * r;
    val[2] = payload[i * 3 + 2] * r;
    val[3] = r;
  } else if constexpr (C == 4) {
    const float4 p4 = *reinterpret_cast<const float4*>(payload + i * 4);
    val[0] = p4.x; val[1] = p4.y; val[2] = p4.z; val[3] = p4.w;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) val[c] = payload[i * C + c];
  }
  unsigned* rec = records + (size_t)(start[brick] + ranks[i]) * (C + 1);
  rec[0] = loc;
#pragma unroll
  for (int c = 0; c < C; ++c) rec[1 + c] = __float_as_uint(val[c]);
}

// ------------------------------------------------------------------------------
// Two-level bucket sort: the same records / start[] as rank -> scan -> scatter above, but
// without a global atomic per particle (memory-side atomics cap that pass at ~2.4e10
// particles/s) and without 20-byte random writes.
//   level 1: chunks of SORT_CHUNK particles; per-chunk LDS histogram over coarse groups of
//            2^gshift consecutive buckets -> table[group][chunk] -> exclusive scan -> each chunk
//            ranks its particles in LDS, stages its records in LDS in group order and streams
//            {key, payload} into its own contiguous run of every group
//   level 2: one workgroup per group: LDS histogram over the group's buckets, LDS scan
//            (-> start[]), second sweep places {loc, payload} at its final slot
// Slots inside one bucket come out in no particular order (as with the atomic ranks).
// ------------------------------------------------------------------------------
#ifndef VPS_SORT_THREADS
#define VPS_SORT_THREADS 1024
#endif
#ifndef VPS_SORT_ITEMS
#define VPS_SORT_ITEMS 2
#endif
constexpr int SORT_THREADS = VPS_SORT_THREADS;   // level 1: chunk = SORT_THREADS * SORT_ITEMS particles
constexpr int SORT_ITEMS = VPS_SORT_ITEMS;
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;
constexpr int FINE_THREADS = 1024;  // level 2: one big workgroup per group
constexpr unsigned SORT_INVALID = 0xffffffffu;
// words per level-1 record {key, payload[C]}.  (Padding the 5-word record of C = 4 to an aligned 32-byte sector was
// measured: the level-1 scatter gains 10 %, level 2 loses 50 % to the extra bytes.)
__host__ __device__ constexpr int sort_rec1_words(int C) { return C + 1; }

struct SortGeom {
  int gshift, ngroups;     // buckets per group = 1 << gshift
  int cshift;              // log2(cells) when cells is a power of two, else -1
  unsigned cells;
  long long nbuckets, nchunks;
};

// Keys.  The full key of a particle is bucket * cells + cell-in-bucket: K = unsigned while that fits 32 bits, unsigned long
// long beyond (C4 on one GPU: 2^19 pencils x 2^14 cells).  It only lives in the keys[] array between the level-1 histogram
// and the level-1 scatter; the level-1 RECORD carries the key relative to its group's first bucket (< 2^gshift * cells),
// which is all level 2 -- one workgroup per group -- needs, and always 32 bits.
template <typename K>
__device__ __forceinline__ constexpr K sort_invalid() { return (K)~(K)0; }

template <typename K>
__device__ __forceinline__ unsigned sort_bucket_of(K key, const SortGeom& g) {
  return (unsigned)(g.cshift >= 0 ? (key >> g.cshift) : (key / g.cells));
}

template <typename F, typename K>
__global__ void __launch_bounds__(SORT_THREADS)
    sort_hist_kernel(const F* __restrict__ pos, long long np, F lcell, F nsize, Bricks b, SortGeom g,
                     K* __restrict__ keys, unsigned* __restrict__ table) {
  extern __shared__ unsigned sort_lds[];
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS) sort_lds[i] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * SORT_CHUNK;
#pragma unroll 4
  for (int k = 0; k < SORT_ITEMS; ++k) {
    const long long i = base + (long long)k * SORT_THREADS + threadIdx.x;
    if (i < np) {
      unsigned brick, loc;
      K key = sort_invalid<K>();
      if (locate<F>(pos, i, lcell, nsize, b, brick, loc)) {
        key = (K)brick * g.cells + loc;
        atomicAdd(&sort_lds[brick >> g.gshift], 1u);
      }
      keys[i] = key;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS)
    table[(long long)i * g.nchunks + blockIdx.x] = sort_lds[i];
}

// ---- slab compaction ---------------------------------------------------------------------------------------------------
// A rank that holds a REPLICATED particle set but deposits one x-slab of it (scripts/parallel_optimized.py:272-276 loads the
// whole snapshot on every rank) first filters: one pass over the positions keeps the particles whose cell lies in the slab and
// writes their {key, [rho v, rho]} to compact arrays -- a workgroup reserves its output range with ONE global atomic.  The
// two-level sort then runs on the compact arrays only: its tables, chunks and records are sized for the slab, and the seven
// eighths of the particles that belong to other ranks are read once instead of twice.
// 256 threads, four consecutive particles per thread: their 12 coordinates are three 16-byte loads of one contiguous 12 KB
// block per workgroup; [rho v, rho] of the ones inside is requested BEFORE the workgroup's output range is reserved, so that
// those loads and the atomic's round trip overlap.
constexpr int COMPACT_THREADS = 256;
constexpr int COMPACT_ITEMS = 4;
constexpr int COMPACT_BLOCK = 2048;            // output slots reserved per atomic (>= the particles of one trip)
constexpr int COMPACT_MAXGRID = 2048;          // workgroups of the compaction launch (each may leave one block partly unused)
template <typename F, typename K>
__global__ void __launch_bounds__(COMPACT_THREADS)
    slab_compact_kernel(const F* __restrict__ pos, const float* __restrict__ vel, const float* __restrict__ rho, long long np,
                        F lcell, F nsize, Bricks b, SortGeom g, K* __restrict__ ckeys, float4* __restrict__ cpay,
                        unsigned long long* __restrict__ counter, long long cap) {
  __shared__ unsigned wcount[COMPACT_THREADS / 64];
  __shared__ unsigned long long blk_next;        // the block reserved for the part of a trip that does not fit the current one
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Output slots come in BLOCKS of COMPACT_BLOCK reserved with one global atomic each (a returning atomic on one word peaks
  // near 90 per microsecond: one per trip -- 10^6 of them at 10^9 particles -- would take longer than reading the particles).
  // A trip whose records do not fit the rest of the current block spills into the next; what a workgroup leaves unused at the
  // end is filled with invalid keys, which the sort skips.
  unsigned long long blk_base = 0;               // (uniform over the workgroup, kept in registers)
  unsigned blk_used = COMPACT_BLOCK;             // nothing reserved yet
  constexpr long long PER = (long long)COMPACT_THREADS * COMPACT_ITEMS;
  const long long nchunk = (np + PER - 1) / PER;
  for (long long c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const long long i0 = c * PER + (long long)threadIdx.x * COMPACT_ITEMS;      // this thread's first particle
    F q[3 * COMPACT_ITEMS];
    if (i0 + COMPACT_ITEMS <= np) {
      if constexpr (sizeof(F) == 4) {
        const float4* src = reinterpret_cast<const float4*>(pos + i0 * 3);       // (i0 * 3 floats = a multiple of 48 bytes)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 v4 = src[k];
          q[4 * k] = v4.x; q[4 * k + 1] = v4.y; q[4 * k + 2] = v4.z; q[4 * k + 3] = v4.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3 * COMPACT_ITEMS; ++k) q[k] = pos[i0 * 3 + k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3 * COMPACT_ITEMS; ++k) q[k] = (i0 * 3 + k < np * 3) ? pos[i0 * 3 + k] : F(0);
    }
    K key[COMPACT_ITEMS];
    bool in[COMPACT_ITEMS];
    float4 pay[COMPACT_ITEMS];
    unsigned mine = 0;
#pragma unroll
    for (int k = 0; k < COMPACT_ITEMS; ++k) {
      const long long i = i0 + k;
      const int cx = cell_of<F>(q[3 * k], lcell, nsize) - b.x0;
      in[k] = i < np && cx >= 0 && cx < b.nx;
      key[k] = 0;
      if (in[k]) {
        const int cy = cell_of<F>(q[3 * k + 1], lcell, nsize), cz = cell_of<F>(q[3 * k + 2], lcell, nsize);
        in[k] = (unsigned)cy < (unsigned)b.N && (unsigned)cz < (unsigned)b.N;      // NaN / inf: nowhere (as `locate`)
        if (in[k]) {
          const int ix = cx / b.bx, iy = cy / b.by, iz = cz / b.bz;
          const unsigned brick = (unsigned)((ix * b.nby + iy) * b.nbz + iz);
          const unsigned loc = (unsigned)(((cx - ix * b.bx) * b.by + (cy - iy * b.by)) * b.bz + (cz - iz * b.bz));
          key[k] = (K)brick * g.cells + loc;
          const float r = rho[i];
          pay[k] = make_float4(vel[i * 3 + 0] * r, vel[i * 3 + 1] * r, vel[i * 3 + 2] * r, r);
          ++mine;
        }
      }
    }
    // slots: thread-major inside the workgroup (a thread's particles are consecutive), one atomic per workgroup and trip
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wcount[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_THREADS / 64; ++w) {
      const unsigned t = wcount[w];
      if (w < wave) before += t;
      total += t;
    }
    const bool spill = blk_used + total > (unsigned)COMPACT_BLOCK;        // (uniform; total <= PER <= COMPACT_BLOCK)
    if (spill && threadIdx.x == 0) blk_next = atomicAdd(counter, (unsigned long long)COMPACT_BLOCK);
    __syncthreads();
    const unsigned long long nb = spill ? blk_next : blk_base;
    unsigned p_ = before + incl - mine;            // this thread's first record inside the trip
#pragma unroll
    for (int k = 0; k < COMPACT_ITEMS; ++k) {
      if (in[k]) {
        const unsigned at = blk_used + p_;
        const unsigned long long slot = at < (unsigned)COMPACT_BLOCK ? blk_base + at : nb + (at - COMPACT_BLOCK);
        if ((long long)slot < cap) {
          ckeys[slot] = key[k];
          cpay[slot] = pay[k];
        }
        ++p_;
      }
    }
    if (spill) {
      blk_used = blk_used + total - COMPACT_BLOCK;
      blk_base = nb;
    } else {
      blk_used += total;
    }
    __syncthreads();   // wcount / blk_next are rewritten by the next trip
  }
  // the unused tail of the last block: invalid keys
  if (blk_used < (unsigned)COMPACT_BLOCK)
    for (unsigned at = blk_used + threadIdx.x; at < (unsigned)COMPACT_BLOCK; at += COMPACT_THREADS)
      if ((long long)(blk_base + at) < cap) ckeys[blk_base + at] = sort_invalid<K>();
}

// level-1 histogram of keys that are already there (the compacted slab)
template <typename K>
__global__ void __launch_bounds__(SORT_THREADS)
    sort_hist_keys_kernel(const K* __restrict__ keys, long long n, SortGeom g, unsigned* __restrict__ table) {
  extern __shared__ unsigned sort_lds[];
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS) sort_lds[i] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * SORT_CHUNK;
#pragma unroll 4
  for (int k = 0; k < SORT_ITEMS; ++k) {
    const long long i = base + (long long)k * SORT_THREADS + threadIdx.x;
    if (i < n) {
      const K key = keys[i];
      if (key != sort_invalid<K>()) atomicAdd(&sort_lds[sort_bucket_of<K>(key, g) >> g.gshift], 1u);   // (block tails of the compaction)
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS)
    table[(long long)i * g.nchunks + blockIdx.x] = sort_lds[i];
}

template <int C, bool RHOV>
__device__ __forceinline__ void load_payload(const float* __restrict__ payload, const float* __restrict__ rho,
                                             long long i, float val[C]) {
  if constexpr (RHOV) {
    static_assert(C == 4, "rho*v payload has four channels");
    const float r = rho[i];
    val[0] = payload[i * 3 + 0] * r;
    val[1] = payload[i * 3 + 1] * r;
    val[2] = payload[i * 3 + 2] * r;
    val[3] = r;
  } else if constexpr (C == 4) {
    const float4 p4 = *reinterpret_cast<const float4*>(payload + i * 4);
    val[0] = p4.x; val[1] = p4.y; val[2] = p4.z; val[3] = p4.w;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) val[c] = payload[i * C + c];
  }
}

template <int C, bool RHOV, typename K>
__global__ void __launch_bounds__(SORT_THREADS)
    sort_scatter_kernel(const K* __restrict__ keys, const float* __restrict__ payload,
                        const float* __restrict__ rho, long long np, SortGeom g,
                        const unsigned* __restrict__ table_start, unsigned* __restrict__ rec1) {
  extern __shared__ unsigned sort_lds[];
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS)
    sort_lds[i] = table_start[(long long)i * g.nchunks + blockIdx.x];
  __syncthreads();
  const long long base = (long long)blockIdx.x * SORT_CHUNK;
#pragma unroll 4
  for (int k = 0; k < SORT_ITEMS; ++k) {
    const long long i = base + (long long)k * SORT_THREADS + threadIdx.x;
    if (i >= np) continue;
    const K key = keys[i];
    if (key == sort_invalid<K>()) continue;
    float val[C];
    load_payload<C, RHOV>(payload, rho, i, val);
    const unsigned grp = sort_bucket_of<K>(key, g) >> g.gshift;
    const unsigned slot = atomicAdd(&sort_lds[grp], 1u);
    constexpr int W = sort_rec1_words(C);
    unsigned w[W];
    w[0] = (unsigned)(key - (K)((unsigned long long)grp << g.gshift) * g.cells);   // relative to the group's first bucket
#pragma unroll
    for (int c = 0; c < C; ++c) w[1 + c] = __float_as_uint(val[c]);
#pragma unroll
    for (int c = C + 1; c < W; ++c) w[c] = 0;
    unsigned* rec = rec1 + (size_t)slot * W;
    if constexpr (W == 2) {
      *reinterpret_cast<uint2*>(rec) = make_uint2(w[0], w[1]);
    } else if constexpr (W == 4) {
      *reinterpret_cast<uint4*>(rec) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
      for (int c = 0; c < W; ++c) rec[c] = w[c];
    }
  }
}

// Exclusive scan of the LDS array a[0..n), n <= 4 * NT, in place; returns the total.
// `scratch` holds NT/64 words.  All NT threads of the workgroup must call it.
template <int NT>
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned* a, int n, unsigned* scratch) {
  const int per = (n + NT - 1) / NT;
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
  for (int w = 0; w < NT / 64; ++w) {
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

// Level-1 scatter, LDS-staged: the chunk's records are first placed in LDS in group order, then
// streamed out word by word, so each (chunk, group) run leaves the CU as contiguous stores instead
// of 64 scattered dwords per instruction.  Consecutive chunks own adjacent runs of every group:
// they are dealt to the SAME XCD (blockIdx % 8, speed only) so that its L2 can merge the partly
// written lines at run boundaries.
template <int C, bool RHOV, typename K>
__global__ void __launch_bounds__(SORT_THREADS)
    sort_scatter_staged_kernel(const K* __restrict__ keys, const float* __restrict__ payload,
                               const float* __restrict__ rho, long long np, SortGeom g,
                               const unsigned* __restrict__ table_start, unsigned* __restrict__ rec1) {
  constexpr int W = sort_rec1_words(C);
  extern __shared__ unsigned sort_lds[];
  unsigned* gbase = sort_lds;                        // [ngroups] first global slot of this chunk's run
  unsigned* lstart = gbase + g.ngroups;              // [ngroups] counts, then local exclusive starts
  unsigned* scratch = lstart + g.ngroups;            // [SORT_THREADS / 64]
  unsigned* gdest = scratch + SORT_THREADS / 64;     // [SORT_CHUNK] global slot of staged record p
  unsigned* stage = gdest + SORT_CHUNK;              // [SORT_CHUNK * W]
  const long long per_xcd = (g.nchunks + 7) / 8;
  const long long chunk = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (chunk >= g.nchunks) return;
  for (int i = threadIdx.x; i < g.ngroups; i += SORT_THREADS) {
    gbase[i] = table_start[(long long)i * g.nchunks + chunk];
    lstart[i] = 0;
  }
  __syncthreads();
  const long long base = chunk * SORT_CHUNK;
  K key[SORT_ITEMS];
  unsigned grp[SORT_ITEMS], rk[SORT_ITEMS];
  float val[SORT_ITEMS][C];
#pragma unroll
  for (int k = 0; k < SORT_ITEMS; ++k) {
    const long long i = base + (long long)k * SORT_THREADS + threadIdx.x;
    key[k] = (i < np) ? keys[i] : sort_invalid<K>();
    if (key[k] != sort_invalid<K>()) load_payload<C, RHOV>(payload, rho, i, val[k]);
  }
#pragma unroll
  for (int k = 0; k < SORT_ITEMS; ++k) {
    if (key[k] != sort_invalid<K>()) {
      grp[k] = sort_bucket_of<K>(key[k], g) >> g.gshift;
      rk[k] = atomicAdd(&lstart[grp[k]], 1u);
    }
  }
  __syncthreads();
  const unsigned total = block_exclusive_scan<SORT_THREADS>(lstart, g.ngroups, scratch);
#pragma unroll
  for (int k = 0; k < SORT_ITEMS; ++k) {
    if (key[k] != sort_invalid<K>()) {
      const unsigned p = lstart[grp[k]] + rk[k];
      gdest[p] = gbase[grp[k]] + rk[k];
      stage[p * W] = (unsigned)(key[k] - (K)((unsigned long long)grp[k] << g.gshift) * g.cells);   // relative to the group's first bucket
#pragma unroll
      for (int c = 0; c < C; ++c) stage[p * W + 1 + c] = __float_as_uint(val[k][c]);
    }
  }
  __syncthreads();
  for (unsigned t = threadIdx.x; t < total * W; t += SORT_THREADS) {
    const unsigned rec = t / W, wd = t - rec * W;
    rec1[(size_t)gdest[rec] * W + wd] = stage[t];
  }
}

template <int C>
__global__ void __launch_bounds__(FINE_THREADS)
    sort_fine_kernel(const unsigned* __restrict__ rec1, SortGeom g, const unsigned* __restrict__ table_start,
                     unsigned* __restrict__ start, unsigned* __restrict__ records) {
  constexpr int W = sort_rec1_words(C);
  extern __shared__ unsigned sort_lds[];          // [G] counts -> cursors, then scan scratch
  const int G = 1 << g.gshift;
  unsigned* cur = sort_lds;
  unsigned* scratch = sort_lds + G;
  const int grp = blockIdx.x;
  const unsigned gs = table_start[(long long)grp * g.nchunks];
  const unsigned ge = table_start[(long long)(grp + 1) * g.nchunks];   // [ngroups*nchunks] = total
  for (int i = threadIdx.x; i < G; i += FINE_THREADS) cur[i] = 0;
  __syncthreads();
  constexpr int U = 4;   // loads of U strides are issued together: the sweeps are latency bound otherwise
  // The second sweep reads the level-1 records with streaming loads: they are dead after it, and what
  // should stay in the caches are the final records it writes (the accumulation kernel reads them next:
  // pencil kernel -10 %).  The first sweep keeps plain loads so that the second finds the lines.
  // Streaming hints on the level-1 scatter itself cost 35 %.
  for (unsigned j0 = gs + threadIdx.x; j0 < ge; j0 += U * FINE_THREADS) {
    unsigned key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + u * FINE_THREADS;
      key[u] = j < ge ? rec1[(size_t)j * W] : SORT_INVALID;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (key[u] != SORT_INVALID) atomicAdd(&cur[sort_bucket_of<unsigned>(key[u], g) & (G - 1)], 1u);   // (keys relative to the group)
  }
  __syncthreads();
  block_exclusive_scan<FINE_THREADS>(cur, G, scratch);
  for (int f = threadIdx.x; f < G; f += FINE_THREADS) {
    const unsigned at = gs + cur[f];
    cur[f] = at;
    const long long bucket = (long long)grp * G + f;
    if (bucket < g.nbuckets) start[bucket] = at;
  }
  if (grp == g.ngroups - 1 && threadIdx.x == 0) start[g.nbuckets] = ge;
  __syncthreads();
  for (unsigned j0 = gs + threadIdx.x; j0 < ge; j0 += U * FINE_THREADS) {
    unsigned r[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + u * FINE_THREADS;
      r[u][0] = SORT_INVALID;
      if (j < ge) {
        const unsigned* src = rec1 + (size_t)j * W;
#pragma unroll
        for (int c = 0; c < W; ++c) r[u][c] = __builtin_nontemporal_load(&src[c]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u][0] == SORT_INVALID) continue;
      const unsigned bucket = sort_bucket_of<unsigned>(r[u][0], g);
      const size_t slot = atomicAdd(&cur[bucket & (G - 1)], 1u);
      unsigned* dst = records + slot * W;
      dst[0] = r[u][0] - bucket * g.cells;
#pragma unroll
      for (int c = 1; c < W; ++c) dst[c] = r[u][c];
    }
  }
}

// Epilogue of the brick kernel: what is written for a cell from its C accumulated channels.
//   EPI_RAW      : the C channels as they are (deposit_to_grid)
//   EPI_ALGEBRA  : channels are [rho vx, rho vy, rho vz, rho] -> fields of `quantity`
enum { EPI_RAW = 0, EPI_ALGEBRA = 1 };

__device__ __forceinline__ void algebra_cell(float a, float b, float c, float rho, int quantity, int flags,
                                             float vol, float out[4]) {
  // v = (rho v)/rho; empty cells give 0 (the NaN->0 rule of interp.py:329-331)
  float vx, vy, vz, m;
  if (flags & VPS_FLAG_INPUT_IS_VM) {
    vx = a; vy = b; vz = c; m = rho;
  } else {
    // one hardware reciprocal per cell (v_rcp_f32, 1 ulp; the result is compared at 2e-5 with a
    // float64 reference) instead of three IEEE divisions (about a dozen instructions each)
    const float inv = rho != 0.f ? __builtin_amdgcn_rcpf(rho) : 0.f;
    vx = a * inv;
    vy = b * inv;
    vz = c * inv;
    m = rho * vol;
  }
  if (quantity == VPS_MOMENTUM) {
    out[0] = vx * m;
    out[1] = ((flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG) ? vx : vy) * m;
    out[2] = ((flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG) ? vx : vz) * m;
    out[3] = m;
  } else if (quantity == VPS_ENERGY) {
    out[0] = m * ((vx * vx + vy * vy) + vz * vz);
    out[1] = out[2] = 0.f;
    out[3] = m;
  } else {  // VPS_VELOCITY, VPS_VM
    out[0] = vx; out[1] = vy; out[2] = vz; out[3] = m;
  }
}

// QUANT is the (compile-time) quantity of the algebra epilogue; NOUT its channel count
template <int C, int EPI, int QUANT>
__global__ void __launch_bounds__(256)
    brick_accumulate_kernel(const unsigned* __restrict__ records, const unsigned* __restrict__ start,
                            Bricks b, long long nbricks, int flags, float vol,
                            float* __restrict__ grid) {
  constexpr int quantity = QUANT;
  constexpr int NOUT = (EPI == EPI_RAW) ? C : (QUANT == VPS_ENERGY ? 1 : (QUANT == VPS_VM ? 4 : 3));
  // Persistent workgroups walk the bricks: the streaming stores of one brick stay in flight
  // while the next bucket is being accumulated, and the tile is re-zeroed as it is read.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);  // [C][cells]
  const int cells = b.cells;
  const bool vec4 = ((b.bz & 3) == 0) && ((b.N & 3) == 0);
  if ((cells & 3) == 0) {
    for (int i = threadIdx.x; i < C * cells / 4; i += blockDim.x)
      reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int i = threadIdx.x; i < C * cells; i += blockDim.x) tile[i] = 0.f;
  }
  const long long plane = (long long)b.nx * b.N * b.N;
  long long brick = blockIdx.x;
  unsigned s = 0, e = 0;
  if (brick < nbricks) {
    s = start[brick];
    e = start[brick + 1];
  }
  __syncthreads();
  for (; brick < nbricks; brick += gridDim.x) {
    for (unsigned j = s + threadIdx.x; j < e; j += blockDim.x) {
      const unsigned* rec = records + (size_t)j * (C + 1);
      const unsigned loc = rec[0];
#pragma unroll
      for (int c = 0; c < C; ++c) atomicAdd(&tile[c * cells + loc], __uint_as_float(rec[1 + c]));   // (vps_lds_add: slower here)
    }
    // bucket bounds of the next brick: in flight during the stream-out below
    const long long nb = brick + gridDim.x;
    if (nb < nbricks) {
      s = start[nb];
      e = start[nb + 1];
    }
    __syncthreads();
    // stream the tile out: rows of bz cells are contiguous in the grid
    const int iz = (int)(brick % b.nbz), iy = (int)((brick / b.nbz) % b.nby);
    const int ix = (int)(brick / ((long long)b.nbz * b.nby));
    const int gx0 = ix * b.bx, gy0 = iy * b.by, gz0 = iz * b.bz;
    if (vec4) {
      const int q4 = cells / 4;
      for (int i = threadIdx.x; i < q4; i += blockDim.x) {
        const int loc = i * 4;
        int lz, ly, lx;
        if (b.pow2) {
          lz = loc & (b.bz - 1);
          ly = (loc >> b.sz) & (b.by - 1);
          lx = loc >> (b.sz + b.sy);
        } else {
          lz = loc % b.bz; ly = (loc / b.bz) % b.by; lx = loc / (b.bz * b.by);
        }
        const int gx = gx0 + lx, gy = gy0 + ly, gz = gz0 + lz;
        const bool inside = gx < b.nx && gy < b.N && gz < b.N;   // partial bricks at the slab edge
        const long long cell = ((long long)gx * b.N + gy) * b.N + gz;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (EPI == EPI_RAW) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            float4* t = reinterpret_cast<float4*>(tile + c * cells + loc);
            const float4 val = *t;
            *t = zero4;
            if (inside) *reinterpret_cast<float4*>(grid + c * plane + cell) = val;
          }
        } else {
          float4* t0 = reinterpret_cast<float4*>(tile + loc);
          float4* t1 = reinterpret_cast<float4*>(tile + cells + loc);
          float4* t2 = reinterpret_cast<float4*>(tile + 2 * cells + loc);
          float4* t3 = reinterpret_cast<float4*>(tile + 3 * cells + loc);
          const float4 c0 = *t0, c1 = *t1, c2 = *t2, c3 = *t3;
          *t0 = zero4; *t1 = zero4; *t2 = zero4; *t3 = zero4;
          if (inside) {
            float rx[4], ry[4], rz[4], rw[4];
            algebra_cell(c0.x, c1.x, c2.x, c3.x, quantity, flags, vol, rx);
            algebra_cell(c0.y, c1.y, c2.y, c3.y, quantity, flags, vol, ry);
            algebra_cell(c0.z, c1.z, c2.z, c3.z, quantity, flags, vol, rz);
            algebra_cell(c0.w, c1.w, c2.w, c3.w, quantity, flags, vol, rw);
#pragma unroll
            for (int c = 0; c < NOUT; ++c)
              {   // streaming store: the grid is written once; the cache should keep the records (-33 %)
                typedef float vf4 __attribute__((ext_vector_type(4)));
                vf4 q; q.x = rx[c]; q.y = ry[c]; q.z = rz[c]; q.w = rw[c];
                __builtin_nontemporal_store(q, reinterpret_cast<vf4*>(grid + c * plane + cell));
              }
          }
        }
      }
    } else {
      for (int loc = threadIdx.x; loc < cells; loc += blockDim.x) {
        const int lz = loc % b.bz, ly = (loc / b.bz) % b.by, lx = loc / (b.bz * b.by);
        const int gx = gx0 + lx, gy = gy0 + ly, gz = gz0 + lz;
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          v[c] = tile[c * cells + loc];
          tile[c * cells + loc] = 0.f;
        }
        if (gx >= b.nx || gy >= b.N || gz >= b.N) continue;
        const long long cell = ((long long)gx * b.N + gy) * b.N + gz;
        if constexpr (EPI == EPI_RAW) {
#pragma unroll
          for (int c = 0; c < C; ++c) grid[c * plane + cell] = v[c];
        } else {
          float r[4];
          algebra_cell(v[0], v[1], v[2], v[3], quantity, flags, vol, r);
#pragma unroll
          for (int c = 0; c < NOUT; ++c) grid[c * plane + cell] = r[c];
        }
      }
    }
    __syncthreads();   // tile is all zero again
  }
}

// In-place algebra on an existing 4-channel grid (used after the NN resample and by
// BoxField.spctrm on user-supplied fields).
// `out` == nullptr: in place; else the channels of the result go to out[c][ncell] and ch is only read
// (a second quantity of the same field: no copy of the four input channels is needed).
__global__ void __launch_bounds__(256)
    field_algebra_kernel(float* ch, long long ncell, int quantity, int flags, float vol, float* out) {
  const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= ncell) return;
  float* dst = out ? out : ch;
  float4 a = *reinterpret_cast<float4*>(ch + i0);
  float4 b = *reinterpret_cast<float4*>(ch + ncell + i0);
  float4 c = *reinterpret_cast<float4*>(ch + 2 * ncell + i0);
  float4 m = *reinterpret_cast<float4*>(ch + 3 * ncell + i0);
  float r[4];
  algebra_cell(a.x, b.x, c.x, m.x, quantity, flags, vol, r);
  a.x = r[0]; b.x = r[1]; c.x = r[2]; m.x = r[3];
  algebra_cell(a.y, b.y, c.y, m.y, quantity, flags, vol, r);
  a.y = r[0]; b.y = r[1]; c.y = r[2]; m.y = r[3];
  algebra_cell(a.z, b.z, c.z, m.z, quantity, flags, vol, r);
  a.z = r[0]; b.z = r[1]; c.z = r[2]; m.z = r[3];
  algebra_cell(a.w, b.w, c.w, m.w, quantity, flags, vol, r);
  a.w = r[0]; b.w = r[1]; c.w = r[2]; m.w = r[3];
  *reinterpret_cast<float4*>(dst + i0) = a;
  if (quantity != VPS_ENERGY) {
    *reinterpret_cast<float4*>(dst + ncell + i0) = b;
    *reinterpret_cast<float4*>(dst + 2 * ncell + i0) = c;
  }
  if (quantity == VPS_VM) *reinterpret_cast<float4*>(dst + 3 * ncell + i0) = m;
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

// Higher-order mass assignment (cloud-in-cell: 2 cells per axis, triangular-shaped-cloud: 3): every particle becomes
// S = 8 or 27 weighted sub-particles sitting at the centres of the cells it touches (periodic), which the NGP
// deposit then adds up -- no new deposit kernel, and with replicated particles no halo exchange between slabs.
// Not in the reference (it offers NGP interp.py:996, NN :1018 and Voxelize :280); SURVEY.md section 8(f-4).
template <typename F>
__global__ void __launch_bounds__(256)
    assign_expand_kernel(const F* __restrict__ pos, const float* __restrict__ payload, long long np, int C, int N,
                         double inv_lcell, double lcell, int order, float* __restrict__ pos_out,
                         float* __restrict__ payload_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  int c0[3];
  float w[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double s = (double)pos[i * 3 + a] * inv_lcell;
    if (order == 2) {            // CIC: cells floor(s - 1/2) and the next one, weights 1 - f, f
      const double f0 = floor(s - 0.5);
      const float f = (float)(s - 0.5 - f0);
      c0[a] = (int)f0;
      w[a][0] = 1.f - f;
      w[a][1] = f;
      w[a][2] = 0.f;
    } else {                     // TSC: the cell holding the particle and its two neighbours
      const double ic = floor(s);
      const float d = (float)(s - (ic + 0.5));
      c0[a] = (int)ic - 1;
      w[a][0] = 0.5f * (0.5f - d) * (0.5f - d);
      w[a][1] = 0.75f - d * d;
      w[a][2] = 0.5f * (0.5f + d) * (0.5f + d);
    }
  }
  const int S1 = order, S = S1 * S1 * S1;
  float pay[4];
  for (int c = 0; c < C; ++c) pay[c] = payload[i * C + c];
  for (int j = 0; j < S; ++j) {
    const int jx = j / (S1 * S1), jy = (j / S1) % S1, jz = j % S1;
    const float wt = w[0][jx] * w[1][jy] * w[2][jz];
    const int cx = ((c0[0] + jx) % N + N) % N, cy = ((c0[1] + jy) % N + N) % N, cz = ((c0[2] + jz) % N + N) % N;
    float* po = pos_out + (i * S + j) * 3;
    po[0] = (float)(((double)cx + 0.5) * lcell);
    po[1] = (float)(((double)cy + 0.5) * lcell);
    po[2] = (float)(((double)cz + 0.5) * lcell);
    for (int c = 0; c < C; ++c) payload_out[(i * S + j) * C + c] = pay[c] * wt;
  }
}

// LDS tile of one brick: 32 KiB -> four bricks resident per CU, enough workgroups in
// flight to hide the bucket-read -> LDS-add -> stream-out dependency chain of each one.  From 1024^3 on twice that
// (8 x 8 x 64 cells, two bricks per CU): a brick's stores then cover 64 rows of 256 bytes per field instead of 32, and the
// sort has half the buckets -- measured (sort + accumulate & write, ms): 1024^3 / 5e7 particles 2.34 + 3.08 against
// 2.39 + 3.53; 2048^3 / 1e8 4.6 + 16.9 against 5.6 + 20.4 (103 GB written at 6.1 instead of 5.1 TB/s); 512^3 / 1e7 0.37 + 0.36
// against 0.38 + 0.34; 128 KiB (one brick per CU) 4.8 + 20.4; rows of 512 / 1024 bytes (bz = 128 / 256) 19.0 / 19.4 at 32 KiB.
#ifndef VPS_BRICK_LDS_BYTES
#define VPS_BRICK_LDS_BYTES (32 * 1024)
#endif
#ifndef VPS_BRICK_BZ
#define VPS_BRICK_BZ 64
#endif

int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

Bricks make_bricks(int N, int x0, int nx, int C) {
  Bricks b;
  b.N = N; b.x0 = x0; b.nx = nx;
  const int max_cells = ((N >= 1024 ? 2 : 1) * (VPS_BRICK_LDS_BYTES)) / (4 * C);
  b.bz = N < VPS_BRICK_BZ ? N : VPS_BRICK_BZ;          // up to 256-byte rows
  if (b.bz > max_cells) b.bz = max_cells;
  int rest = max_cells / b.bz;
  b.by = pow2_floor(rest < 8 ? (rest < 1 ? 1 : rest) : 8);
  if (b.by > N) b.by = N;
  rest = max_cells / (b.bz * b.by);
  b.bx = pow2_floor(rest < 1 ? 1 : rest);
  if (b.bx > nx) b.bx = nx;
  b.nbx = (nx + b.bx - 1) / b.bx;
  b.nby = (N + b.by - 1) / b.by;
  b.nbz = (N + b.bz - 1) / b.bz;
  b.cells = b.bx * b.by * b.bz;
  b.pow2 = ((b.by & (b.by - 1)) == 0) && ((b.bz & (b.bz - 1)) == 0);
  b.sy = b.sz = 0;
  while ((1 << b.sy) < b.by) ++b.sy;
  while ((1 << b.sz) < b.bz) ++b.sz;
  return b;
}

// pencil buckets of the fused deposit -> z-pass path: (x, TP y-lines, all z)
Bricks make_pencils(int N, int x0, int nx, int TP) {
  Bricks b;
  b.N = N; b.x0 = x0; b.nx = nx;
  b.bx = 1; b.by = TP; b.bz = N;
  b.nbx = nx; b.nby = N / TP; b.nbz = 1;
  b.cells = TP * N;
  b.pow2 = 1;
  b.sy = b.sz = 0;
  while ((1 << b.sy) < b.by) ++b.sy;
  while ((1 << b.sz) < b.bz) ++b.sz;
  return b;
}

struct DepLayout {
  size_t count, start, tiles, keys, ranks, records, table, table_start, table_tiles, rec1, total;
  long long nbricks;
  long long cap;    // records the workspace has room for: np, or the caller's bound on the particles inside the slab
  long long cap_in; // entries of the compacted input arrays (cap + the block tails of slab_compact_kernel)
  bool recompute;   // cap < np: the slab's particles are compacted first (slab_compact_kernel); keys[] / cpay hold cap entries
  size_t cpay, counter;
  bool two_level;
  bool wide_keys;   // bucket * cells + cell does not fit 32 bits: 64-bit keys[] (the level-1 records stay 32-bit, see SortGeom)
  SortGeom geom;
};

// groups per launch the two-level sort aims for (level-2 workgroups); option sort_groups overrides (tuning)
int sort_target_groups() {
  int v = (int)vps_option("sort_groups", 512);
  if (v < 1) v = 1;
  if (v > 4096) v = 4096;
  return v;
}

bool sort_staged() { return vps_option("sort_staged", 1) != 0; }

// option sort_atomic forces the one-atomic-per-particle ranking (kept for bucket counts / key ranges the
// two-level sort does not cover, and as a cross-check in the tests)
bool sort_force_atomic() { return vps_option("sort_atomic", 0) != 0; }

size_t sort_staged_lds(const SortGeom& g, int C) {
  return sizeof(unsigned) * (2 * (size_t)g.ngroups + SORT_THREADS / 64 + (size_t)SORT_CHUNK * (1 + sort_rec1_words(C)));
}

// np_cap >= 0: the caller's bound on the number of particles inside the slab (vps_count_in_slab): the record arrays are sized
// for it and the key array disappears -- for ranks that hold a replicated particle set but deposit one slab of it
DepLayout dep_layout(int64_t np, int C, const Bricks& b, int64_t np_cap = -1) {
  DepLayout l;
  l.nbricks = (long long)b.nbx * b.nby * b.nbz;
  SortGeom& g = l.geom;
  g.cells = (unsigned)b.cells;
  g.cshift = -1;
  if ((b.cells & (b.cells - 1)) == 0) {
    g.cshift = 0;
    while ((1 << g.cshift) < b.cells) ++g.cshift;
  }
  g.nbuckets = l.nbricks;
  g.gshift = 3;
  while (((l.nbricks + (1ll << g.gshift) - 1) >> g.gshift) > sort_target_groups()) ++g.gshift;
  // more buckets than target x 4096 (the bricks of a 2048^3 grid: 4.2e6): rather more level-1 groups than the
  // one-atomic-per-particle ranking -- measured at 2048^3 / 1e8 particles: 5.4 ms with 1024 groups against 8.4 ms
  // (6.2 with 2048 groups, 8.6 with 4096: the group tables grow with them)
  if (g.gshift > 12 && ((l.nbricks + 4095) >> 12) <= 2048) g.gshift = 12;
  g.ngroups = (int)((l.nbricks + (1ll << g.gshift) - 1) >> g.gshift);
  g.nchunks = (np + SORT_CHUNK - 1) / SORT_CHUNK;     // (slab-sized workspaces: re-set below)
  l.wide_keys = (unsigned long long)l.nbricks * (unsigned long long)b.cells >= 0xffffffffull;
  l.two_level = !sort_force_atomic() && g.gshift <= 12 && l.nbricks < 0x7fffffffLL &&
                ((unsigned long long)b.cells << g.gshift) < 0xffffffffull;
  l.recompute = np_cap >= 0 && np_cap < np && l.two_level && sort_staged() && sort_staged_lds(g, C) <= 160 * 1024 && C == 4;
  l.cap = l.recompute ? np_cap : np;
  // compacted arrays: the slab's particles + one partly used block per workgroup of the compaction launch
  {
    const long long trips = (np + (long long)COMPACT_THREADS * COMPACT_ITEMS - 1) / ((long long)COMPACT_THREADS * COMPACT_ITEMS);
    const long long slack = std::min<long long>(trips, COMPACT_MAXGRID) * COMPACT_BLOCK;
    if (l.recompute && l.cap + slack >= np / 2) {     // nothing to gain: sort all particles in place, as without a bound
      l.recompute = false;
      l.cap = np;
    }
    l.cap_in = l.recompute ? l.cap + slack : l.cap;
  }
  if (l.recompute) g.nchunks = (l.cap_in + SORT_CHUNK - 1) / SORT_CHUNK;     // the sort sees the compacted particles only
  const long long ntable = (long long)g.ngroups * g.nchunks;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  l.count = off;   off = align(off + sizeof(unsigned) * l.nbricks);
  l.start = off;   off = align(off + sizeof(unsigned) * (l.nbricks + 1));
  l.tiles = off;   off = align(off + sizeof(unsigned) * (scan_tiles(l.nbricks) + 1));
  l.keys = off;    off = align(off + (size_t)l.cap_in * sizeof(unsigned long long));
  l.cpay = l.counter = off;
  if (l.recompute) {
    l.cpay = off;    off = align(off + (size_t)l.cap_in * sizeof(float4));
    l.counter = off; off = align(off + sizeof(unsigned long long));
  }
  l.ranks = off;   off = align(off + (size_t)l.cap_in * sizeof(unsigned));
  l.records = off; off = align(off + (size_t)l.cap_in * (C + 1) * sizeof(unsigned));
  l.table = l.table_start = l.table_tiles = l.rec1 = off;
  if (l.two_level) {
    l.table = off;        off = align(off + sizeof(unsigned) * (ntable + 1));
    l.table_start = off;  off = align(off + sizeof(unsigned) * (ntable + 1));
    l.table_tiles = off;  off = align(off + sizeof(unsigned) * (scan_tiles(ntable) + 1));
    l.rec1 = off;         off = align(off + (size_t)l.cap_in * sort_rec1_words(C) * sizeof(unsigned));
  }
  l.total = off;
  return l;
}

// particles -> records {cell-in-bucket, payload[C]} grouped by bucket + start[nbuckets + 1]
template <typename F, int C, bool RHOV>
int sort_into_buckets(vps_ctx* ctx, const F* pos, const float* payload, const float* rho, int64_t np, F lcell,
                      F nsz, const Bricks& b, const DepLayout& l, char* work) {
  unsigned* count = reinterpret_cast<unsigned*>(work + l.count);
  unsigned* start = reinterpret_cast<unsigned*>(work + l.start);
  unsigned* records = reinterpret_cast<unsigned*>(work + l.records);
  vps_launch_timer tm(ctx, VPS_K_DEPOSIT);
  if (np == 0) {
    VPS_HIP_CHECK(ctx, hipMemsetAsync(start, 0, sizeof(unsigned) * (l.nbricks + 1), ctx->stream));
    return VPS_OK;
  }
  if (l.two_level) {
    const SortGeom& g = l.geom;
    unsigned* table = reinterpret_cast<unsigned*>(work + l.table);
    unsigned* table_start = reinterpret_cast<unsigned*>(work + l.table_start);
    unsigned* table_tiles = reinterpret_cast<unsigned*>(work + l.table_tiles);
    unsigned* rec1 = reinterpret_cast<unsigned*>(work + l.rec1);
    const size_t lds1 = sizeof(unsigned) * g.ngroups;
    const size_t lds2 = sizeof(unsigned) * ((1u << g.gshift) + FINE_THREADS / 64);
    const size_t lds_staged = sort_staged_lds(g, C);
    const bool staged = sort_staged() && lds_staged <= ctx->lds_per_cu;
    if (l.recompute && !staged) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "slab-sized sort workspace needs the LDS-staged scatter");
    auto level1 = [&](auto* keys) -> int {
      typedef typename std::remove_pointer<decltype(keys)>::type K;
      long long n_sort = np;                   // particles the sort proper sees
      const float* pay = payload;
      if (l.recompute) {
        // filter first: {key, [rho v, rho]} of the slab's particles, compact
        if constexpr (RHOV) {
          unsigned long long* counter = reinterpret_cast<unsigned long long*>(work + l.counter);
          float4* cpay = reinterpret_cast<float4*>(work + l.cpay);
          VPS_HIP_CHECK(ctx, hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
          const long long nch = (np + (long long)COMPACT_THREADS * COMPACT_ITEMS - 1) / ((long long)COMPACT_THREADS * COMPACT_ITEMS);
          const unsigned cgrid = (unsigned)std::min<long long>(std::min<long long>(nch, (long long)ctx->num_cu * 8), COMPACT_MAXGRID);
          hipLaunchKernelGGL((slab_compact_kernel<F, K>), dim3(cgrid), dim3(COMPACT_THREADS), 0, ctx->stream, pos, payload, rho,
                             (long long)np, lcell, nsz, b, g, keys, cpay, counter, l.cap_in);
          unsigned long long reserved = 0;      // slots handed out: the slab's particles + at most one partly used block per workgroup
          VPS_HIP_CHECK(ctx, hipMemcpyAsync(&reserved, counter, sizeof(reserved), hipMemcpyDeviceToHost, ctx->stream));
          VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
          if ((long long)reserved > l.cap_in)
            return vps_fail(ctx, VPS_ERR_ARG, "more particles lie inside the slab than the workspace was sized for (%lld; vps_count_in_slab)",
                            l.cap);
          n_sort = (long long)reserved;
          pay = reinterpret_cast<const float*>(cpay);
          if (n_sort == 0) {
            VPS_HIP_CHECK(ctx, hipMemsetAsync(start, 0, sizeof(unsigned) * (l.nbricks + 1), ctx->stream));
            return 1;     // (nothing to sort: start[] is all zero)
          }
          hipLaunchKernelGGL((sort_hist_keys_kernel<K>), dim3((unsigned)g.nchunks), dim3(SORT_THREADS), lds1, ctx->stream,
                             (const K*)keys, n_sort, g, table);
        } else {
          return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "slab-sized sort workspace: [rho v, rho] records only");
        }
      } else {
        hipLaunchKernelGGL((sort_hist_kernel<F, K>), dim3((unsigned)g.nchunks), dim3(SORT_THREADS), lds1, ctx->stream, pos,
                           (long long)np, lcell, nsz, b, g, keys, table);
      }
      launch_exclusive_scan(ctx->stream, table, (long long)g.ngroups * g.nchunks, table_tiles, table_start);
      if (staged) {
        const unsigned grid = (unsigned)(8 * ((g.nchunks + 7) / 8));
        auto go = [&](auto kern) -> int {
          if (lds_staged > 64 * 1024)
            VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged));
          hipLaunchKernelGGL(kern, dim3(grid), dim3(SORT_THREADS), lds_staged, ctx->stream, (const K*)keys, pay, rho,
                             n_sort, g, table_start, rec1);
          return VPS_OK;
        };
        int rcs;
        if constexpr (C == 4) rcs = l.recompute ? go(sort_scatter_staged_kernel<4, false, K>) : go(sort_scatter_staged_kernel<C, RHOV, K>);
        else rcs = go(sort_scatter_staged_kernel<C, RHOV, K>);
        if (rcs) return rcs;
      } else {
        hipLaunchKernelGGL((sort_scatter_kernel<C, RHOV, K>), dim3((unsigned)g.nchunks), dim3(SORT_THREADS), lds1,
                           ctx->stream, keys, payload, rho, (long long)np, g, table_start, rec1);
      }
      return VPS_OK;
    };
    // (the keys[] region holds 8 bytes per particle either way: the atomic-rank path's keys are 64-bit)
    const int rc1 = l.wide_keys ? level1(reinterpret_cast<unsigned long long*>(work + l.keys))
                                : level1(reinterpret_cast<unsigned*>(work + l.keys));
    if (rc1 == 1) return VPS_OK;     // (empty slab)
    if (rc1) return rc1;
    hipLaunchKernelGGL(sort_fine_kernel<C>, dim3((unsigned)g.ngroups), dim3(FINE_THREADS), lds2, ctx->stream, rec1,
                       g, table_start, start, records);
  } else {
    unsigned* tiles = reinterpret_cast<unsigned*>(work + l.tiles);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(work + l.keys);
    unsigned* ranks = reinterpret_cast<unsigned*>(work + l.ranks);
    VPS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, sizeof(unsigned) * l.nbricks, ctx->stream));
    const unsigned pblocks = (unsigned)((np + 255) / 256);
    hipLaunchKernelGGL(brick_rank_kernel<F>, dim3(pblocks), dim3(256), 0, ctx->stream, pos, (long long)np, lcell,
                       nsz, b, count, keys, ranks);
    launch_exclusive_scan(ctx->stream, count, l.nbricks, tiles, start);
    hipLaunchKernelGGL((brick_scatter_kernel<C, RHOV>), dim3(pblocks), dim3(256), 0, ctx->stream, keys, ranks,
                       payload, rho, (long long)np, (unsigned)b.cells, start, records);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

template <typename F, int C, bool RHOV, int EPI>
int deposit_run(vps_ctx* ctx, const void* pos_v, const float* payload, const float* rho, int64_t np, int N,
                double Lbox, int x0, int nx, int quantity, int flags, float* grid, void* work_v) {
  const F* pos = reinterpret_cast<const F*>(pos_v);
  // Lcell = Lbox/float(N) in double, then cast to the dtype of pos: what numpy's weak
  // Python-float scalar does in `pos // Lcell`
  const F lcell = (F)(Lbox / (double)N);
  const F nsz = (F)N;
  const Bricks b = make_bricks(N, x0, nx, C);
  const DepLayout l = dep_layout(np, C, b);
  char* work = reinterpret_cast<char*>(work_v);
  unsigned* start = reinterpret_cast<unsigned*>(work + l.start);
  unsigned* records = reinterpret_cast<unsigned*>(work + l.records);
  const float vol = (float)((Lbox / (double)N) * (Lbox / (double)N) * (Lbox / (double)N));
  const int rc = sort_into_buckets<F, C, RHOV>(ctx, pos, payload, rho, np, lcell, nsz, b, l, work);
  if (rc) return rc;
  {
    vps_launch_timer tm(ctx, VPS_K_ALGEBRA);
    const size_t lds = (size_t)C * b.cells * sizeof(float);
    long long per_cu = (long long)(ctx->lds_per_cu / (lds ? lds : 1));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    long long grid_wg = (long long)ctx->num_cu * per_cu;
    if (grid_wg > l.nbricks) grid_wg = l.nbricks;
#define VPS_BRICK(Q)                                                                                      \
  hipLaunchKernelGGL((brick_accumulate_kernel<C, EPI, Q>), dim3((unsigned)grid_wg), dim3(256), lds, ctx->stream, \
                     records, start, b, l.nbricks, flags, vol, grid)
    if (EPI == EPI_RAW || quantity == VPS_VELOCITY) VPS_BRICK(VPS_VELOCITY);
    else if (quantity == VPS_MOMENTUM) VPS_BRICK(VPS_MOMENTUM);
    else if (quantity == VPS_ENERGY) VPS_BRICK(VPS_ENERGY);
    else VPS_BRICK(VPS_VM);
#undef VPS_BRICK
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

// rank -> scan -> scatter of [rho v, rho] records into the buckets `b`; returns the layout used
template <typename F>
int sort_rhov_records(vps_ctx* ctx, const F* pos, const float* vel, const float* rho, int64_t np, int N,
                      double Lbox, const Bricks& b, char* work, DepLayout* lay, int64_t np_cap = -1) {
  const F lcell = (F)(Lbox / (double)N);
  const F nsz = (F)N;
  *lay = dep_layout(np, 4, b, np_cap);
  return sort_into_buckets<F, 4, true>(ctx, pos, vel, rho, np, lcell, nsz, b, *lay, work);
}

// particles whose bit-exact cell lies inside the slab (the rule of `locate`)
template <typename F>
__global__ void __launch_bounds__(256) count_in_slab_kernel(const F* __restrict__ pos, long long np, F lcell, F nsize, Bricks b,
                                                            unsigned long long* __restrict__ out) {
  unsigned n = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (long long)gridDim.x * blockDim.x) {
    unsigned brick, loc;
    n += locate<F>(pos, i, lcell, nsize, b, brick, loc) ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, (unsigned long long)n);
}

int check_deposit_args(vps_ctx* ctx, const char* who, int64_t np, int N, double Lbox, int x0, int nx) {
  if (np < 0 || N < 1 || !(Lbox > 0)) return vps_fail(ctx, VPS_ERR_ARG, "%s: bad np/N/Lbox", who);
  if (x0 < 0 || nx < 1 || x0 + nx > N) return vps_fail(ctx, VPS_ERR_ARG, "%s: slab [%d,%d) outside [0,%d)", who, x0, x0 + nx, N);
  if ((np + 255) / 256 > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "%s: np too large for one launch", who);
  if (np > 0xffffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "%s: np exceeds 32-bit bucket offsets", who);
  return VPS_OK;
}

}  // namespace

extern "C" {

int vps_cell_index(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, int64_t np, int N, double Lbox,
                   int32_t* cell_dev) {
  VPS_ENTER(ctx);
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

size_t vps_deposit_workspace_bytes(int64_t np, int C, int N, int nx) {
  if (np < 0 || C < 1 || N < 1 || nx < 1) return 0;
  const Bricks b = make_bricks(N, 0, nx, C);
  return dep_layout(np, C, b).total;
}

int vps_deposit_ngp(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev,
                    int64_t np, int C, int N, double Lbox, int x0, int nx, float* grid_dev,
                    void* work_dev) {
  VPS_ENTER(ctx);
  int rc = check_deposit_args(ctx, "vps_deposit_ngp", np, N, Lbox, x0, nx);
  if (rc) return rc;
  if (!grid_dev || !work_dev || (np > 0 && (!pos_dev || !payload_dev)))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_ngp: null buffer");
#define VPS_DEP(CC)                                                                                        \
  (pos_is_f64 ? deposit_run<double, CC, false, EPI_RAW>(ctx, pos_dev, payload_dev, nullptr, np, N, Lbox, x0, \
                                                        nx, 0, 0, grid_dev, work_dev)                       \
              : deposit_run<float, CC, false, EPI_RAW>(ctx, pos_dev, payload_dev, nullptr, np, N, Lbox, x0,  \
                                                       nx, 0, 0, grid_dev, work_dev))
  switch (C) {
    case 1: return VPS_DEP(1);
    case 3: return VPS_DEP(3);
    case 4: return VPS_DEP(4);
    default: return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_deposit_ngp: C=%d channels (supported: 1,3,4)", C);
  }
#undef VPS_DEP
}

int vps_deposit_field(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                      const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx, int quantity,
                      int flags, float* fields_dev, void* work_dev) {
  VPS_ENTER(ctx);
  int rc = check_deposit_args(ctx, "vps_deposit_field", np, N, Lbox, x0, nx);
  if (rc) return rc;
  if (quantity < 0 || quantity > 3) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_field: quantity %d", quantity);
  if (flags & VPS_FLAG_INPUT_IS_VM) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_field: VPS_FLAG_INPUT_IS_VM is meaningless here");
  if (!fields_dev || !work_dev || (np > 0 && (!pos_dev || !vel_dev || !rho_dev)))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_field: null buffer");
  return pos_is_f64 ? deposit_run<double, 4, true, EPI_ALGEBRA>(ctx, pos_dev, vel_dev, rho_dev, np, N, Lbox, x0,
                                                               nx, quantity, flags, fields_dev, work_dev)
                    : deposit_run<float, 4, true, EPI_ALGEBRA>(ctx, pos_dev, vel_dev, rho_dev, np, N, Lbox, x0,
                                                              nx, quantity, flags, fields_dev, work_dev);
}

int vps_deposit_fft_zy_supported(vps_ctx* ctx, int N, int quantity) {
  if (!ctx) return 0;
  return (quantity == VPS_VELOCITY || quantity == VPS_MOMENTUM || quantity == VPS_ENERGY) && vps_pencil_supported(ctx, N) ? 1 : 0;
}

size_t vps_deposit_fft_zy_workspace_bytes(int64_t np, int N, int nx) {
  if (np < 0 || N < 16 || nx < 1) return 0;
  const Bricks b = make_pencils(N, 0, nx, vps_pencil_tp(N));
  const size_t sort = dep_layout(np, 4, b).total;
  const size_t images = 3 * ((size_t)nx * (N / 2) * N + (size_t)nx * N) * sizeof(float2);
  return sort + images;
}

size_t vps_deposit_fft_zy_workspace_bytes_shared(int64_t np, int N, int nx) {
  const size_t base = vps_deposit_fft_zy_workspace_bytes(np, N, nx);
  return base ? base + ((size_t)nx * (N / 2) * N + (size_t)nx * N) * sizeof(float2) : 0;
}

static int deposit_fft_impl(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                            const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx, int quantity,
                            int flags, void* spec_dev, void* nyq_dev, void* zimg_dev, void* work_dev, int64_t np_cap = -1);

int vps_deposit_fft_zy(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                       const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx, int quantity,
                       int flags, void* spec_dev, void* nyq_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!spec_dev || !nyq_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_zy: null buffer");
  return deposit_fft_impl(ctx, pos_dev, pos_is_f64, vel_dev, rho_dev, np, N, Lbox, x0, nx, quantity, flags, spec_dev,
                          nyq_dev, nullptr, work_dev);
}

size_t vps_deposit_fft_z_workspace_bytes(int64_t np, int N, int nx) {
  if (np < 0 || N < 16 || nx < 1) return 0;
  return dep_layout(np, 4, make_pencils(N, 0, nx, vps_pencil_tp(N))).total;
}

int vps_deposit_fft_z(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                      const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx, int quantity,
                      int flags, void* zimg_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!zimg_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_z: null buffer");
  return deposit_fft_impl(ctx, pos_dev, pos_is_f64, vel_dev, rho_dev, np, N, Lbox, x0, nx, quantity, flags, nullptr,
                          nullptr, zimg_dev, work_dev);
}

int64_t vps_count_in_slab(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, int64_t np, int N, double Lbox, int x0, int nx) {
  if (!ctx) return VPS_ERR_ARG;
  vps_device_guard guard(ctx);
  int rc = check_deposit_args(ctx, "vps_count_in_slab", np, N, Lbox, x0, nx);
  if (rc) return rc;
  if (np == 0) return 0;
  if (!pos_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_count_in_slab: null buffer");
  const Bricks b = make_pencils(N, x0, nx, 1);
  unsigned long long* d = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&d, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
  const unsigned grid = (unsigned)std::min<long long>((np + 255) / 256, 4096);
  if (e == hipSuccess) {
    if (pos_is_f64)
      hipLaunchKernelGGL(count_in_slab_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const double*>(pos_dev),
                         (long long)np, Lbox / (double)N, (double)N, b, d);
    else
      hipLaunchKernelGGL(count_in_slab_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const float*>(pos_dev),
                         (long long)np, (float)(Lbox / (double)N), (float)N, b, d);
    e = hipGetLastError();
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return vps_fail(ctx, VPS_ERR_HIP, "vps_count_in_slab: %s", hipGetErrorString(e));
  return (int64_t)h;
}

size_t vps_deposit_fft_z_workspace_bytes_slab(int64_t np, int64_t np_slab, int N, int nx) {
  if (np < 0 || np_slab < 0 || N < 16 || nx < 1) return 0;
  return dep_layout(np, 4, make_pencils(N, 0, nx, vps_pencil_tp(N)), np_slab).total;
}

int vps_deposit_fft_z_slab(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                           const float* rho_dev, int64_t np, int64_t np_slab, int N, double Lbox, int x0, int nx, int quantity,
                           int flags, void* zimg_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!zimg_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_z_slab: null buffer");
  if (np_slab < 0) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_z_slab: np_slab < 0");
  return deposit_fft_impl(ctx, pos_dev, pos_is_f64, vel_dev, rho_dev, np, N, Lbox, x0, nx, quantity, flags, nullptr,
                          nullptr, zimg_dev, work_dev, np_slab);
}

// zimg_dev != NULL: stop after the z pass, the images [component][B | BN] go to zimg_dev (work_dev then only holds the sort)
static int deposit_fft_impl(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                            const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx, int quantity,
                            int flags, void* spec_dev, void* nyq_dev, void* zimg_dev, void* work_dev, int64_t np_cap) {
  int rc = check_deposit_args(ctx, "vps_deposit_fft_zy", np, N, Lbox, x0, nx);
  if (rc) return rc;
  if (!vps_deposit_fft_zy_supported(ctx, N, quantity))
    return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_deposit_fft_zy: N=%d quantity=%d not supported by the fused path", N, quantity);
  if (!work_dev || (np > 0 && (!pos_dev || !vel_dev || !rho_dev)))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_zy: null buffer");
  const Bricks b = make_pencils(N, x0, nx, vps_pencil_tp(N));
  char* work = reinterpret_cast<char*>(work_dev);
  DepLayout l;
  if (flags & VPS_FLAG_REUSE_SORT) {
    l = dep_layout(np, 4, b, np_cap);     // the caller vouches that the records of the previous call are still there
  } else {
    rc = pos_is_f64 ? sort_rhov_records<double>(ctx, reinterpret_cast<const double*>(pos_dev), vel_dev, rho_dev, np, N, Lbox, b, work, &l, np_cap)
                    : sort_rhov_records<float>(ctx, reinterpret_cast<const float*>(pos_dev), vel_dev, rho_dev, np, N, Lbox, b, work, &l, np_cap);
    if (rc) return rc;
  }
  const double lc = Lbox / (double)N;
  const int bug = (quantity == VPS_MOMENTUM) && (flags & VPS_FLAG_REFERENCE_MOMENTUM_BUG);
  int chan[3] = {0, bug ? 0 : 1, bug ? 0 : 2};
  int ncomp = 3;
  const int only = (flags & VPS_FLAG_COMPONENT_MASK) >> 4;   // bit c: component c is wanted (VPS_FLAG_COMPONENTS); 0: all
  int with_energy = 0;    // VPS_FLAG_SHARE_ENERGY: 1 the momentum launch that also makes the energy field, 2 the energy call that uses it
  if (flags & VPS_FLAG_SHARE_ENERGY) {
    if (quantity == VPS_MOMENTUM && !bug && !only) with_energy = 1;
    else if (quantity == VPS_ENERGY && (flags & VPS_FLAG_REUSE_SORT)) with_energy = 2;
    else return vps_fail(ctx, VPS_ERR_ARG, "VPS_FLAG_SHARE_ENERGY: a whole momentum field (no component mask, no reference bug), then "
                                           "the energy field with VPS_FLAG_REUSE_SORT");
  }
  if (only) {
    if (quantity == VPS_ENERGY) return vps_fail(ctx, VPS_ERR_ARG, "vps_deposit_fft_zy: VPS_FLAG_COMPONENTS with the (scalar) energy field");
    const int all[3] = {chan[0], chan[1], chan[2]};
    ncomp = 0;
    for (int c = 0; c < 3; ++c)
      if (only & (1 << c)) chan[ncomp++] = all[c];
  }
  // (the sort's rank array -- one word per particle, dead once the records are in place -- is the kernel's per-record scratch)
  return vps_fft_pencil_zy(ctx, N, nx, reinterpret_cast<const unsigned*>(work + l.records),
                           reinterpret_cast<const unsigned*>(work + l.start), reinterpret_cast<float*>(work + l.ranks), ncomp, chan,
                           quantity == VPS_MOMENTUM ? 0 : 1, quantity == VPS_ENERGY ? 1 : 0, (float)(lc * lc * lc),
                           spec_dev, nyq_dev, zimg_dev ? zimg_dev : (void*)(work + l.total), with_energy);
}

int vps_density_velocity_vector(vps_ctx* ctx, const float* vel_dev, const float* rho_dev, int64_t np,
                                float* out_dev) {
  VPS_ENTER(ctx);
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

int vps_assign_expand(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev, int64_t np,
                      int C, int N, double Lbox, int order, float* pos_out_dev, float* payload_out_dev) {
  VPS_ENTER(ctx);
  if (np < 0 || N < 1 || !(Lbox > 0) || C < 1 || C > 4) return vps_fail(ctx, VPS_ERR_ARG, "vps_assign_expand: bad np/N/Lbox/C");
  if (order != 2 && order != 3) return vps_fail(ctx, VPS_ERR_ARG, "vps_assign_expand: order must be 2 (CIC) or 3 (TSC)");
  if (np == 0) return VPS_OK;
  if (!pos_dev || !payload_dev || !pos_out_dev || !payload_out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_assign_expand: null buffer");
  if ((np + 255) / 256 > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "np too large for one launch");
  const double lcell = Lbox / (double)N;
  {
    vps_launch_timer tm(ctx, VPS_K_DEPOSIT);
    const unsigned blocks = (unsigned)((np + 255) / 256);
    if (pos_is_f64)
      hipLaunchKernelGGL(assign_expand_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const double*>(pos_dev), payload_dev, (long long)np, C, N, 1.0 / lcell, lcell,
                         order, pos_out_dev, payload_out_dev);
    else
      hipLaunchKernelGGL(assign_expand_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const float*>(pos_dev), payload_dev, (long long)np, C, N, 1.0 / lcell, lcell,
                         order, pos_out_dev, payload_out_dev);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

int vps_field_algebra_out(vps_ctx* ctx, int quantity, int flags, double Lcell, const float* chans_dev,
                          int64_t ncell, float* out_dev);

int vps_field_algebra(vps_ctx* ctx, int quantity, int flags, double Lcell, float* chans_dev,
                      int64_t ncell) {
  return vps_field_algebra_out(ctx, quantity, flags, Lcell, chans_dev, ncell, nullptr);
}

int vps_field_algebra_out(vps_ctx* ctx, int quantity, int flags, double Lcell, const float* chans_dev,
                          int64_t ncell, float* out_dev) {
  VPS_ENTER(ctx);
  if (quantity < 0 || quantity > 3) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: quantity %d", quantity);
  if (ncell < 0 || (ncell & 3)) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: ncell must be a multiple of 4");
  if (ncell == 0) return VPS_OK;
  if (!chans_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_field_algebra: null buffer");
  const long long nthreads = ncell / 4;
  const unsigned blocks = (unsigned)((nthreads + 255) / 256);
  {
    vps_launch_timer tm(ctx, VPS_K_ALGEBRA);
    hipLaunchKernelGGL(field_algebra_kernel, dim3(blocks), dim3(256), 0, ctx->stream, const_cast<float*>(chans_dev),
                       (long long)ncell, quantity, flags, (float)(Lcell * Lcell * Lcell), out_dev);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

}  // extern "C"
