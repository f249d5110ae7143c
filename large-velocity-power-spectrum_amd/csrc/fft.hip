// 3-D real-to-complex FFT as three batched Stockham line passes, hand-written for
// gfx950 (wave64, 160 KiB LDS/CU), with |F|^2 + spherical-shell binning fused into
// the last pass.  No rocFFT/hipFFT.
//
// Replaces, on the device, the reference's pyFFTW calls and numpy histogram passes:
//   _vector_power / _scalar_power        vpower/interp.py:1372-1387, 1408-1421
//   FFTW_power / FFTW_vector_power       scripts/parallel_optimized.py:92-141
//   _pair_power + _hist_sample           vpower/interp.py:1440-1482
//   pair_power + hist_sample             scripts/parallel_optimized.py:145-190
//
// Data layout (all complex64 unless noted), N = cells per axis, nx = local x rows:
//   input      R[x][y][z]        float32, z contiguous
//   z pass  -> B[x][kz][y]       kz < N/2, y contiguous   (+ Nyquist plane BN[x][y])
//   y pass  -> C[kz][ky][x]      x contiguous             (+ CN[ky][x])
//   x pass  -> lines C[kz][ky][:] transformed in registers and binned (or written).
// Every pass reads whole contiguous lines and writes T-element (T*8 byte) contiguous
// segments of the next layout, so no pass ever does element-strided HBM access.
//
// One line of NC complex points is transformed by L lanes holding RL = NC/L points
// each; radix-8/16 butterflies run in registers and the stages exchange data through
// a padded per-line LDS buffer (Stockham autosort: stage inputs are always read at
// stride NC/R, outputs written at expand(j)+r*Ns).
#include <cstdlib>

#include "vps_internal.h"

// Timing-only build switches (never defined by the product build; results are wrong or incomplete with them -- they exist so
// that the ablations quoted in DESIGN.md can be repeated with tools/build_variant.sh + tools/time_pencil.py / time_xpass.py):
//   VPS_PENCIL_NOSTORE  pencil kernel without its global stores      VPS_ABL_NOZERO / VPS_ABL_NOSCATTER  ... without the
//   VPS_ABL_NOFFT       ... without transform, image and stores      accumulator's zero-fill / the LDS adds
//   VPS_ABL_X_NOATOMIC / VPS_ABL_X_NOBIN   x pass without the LDS shell atomics / without the shell search and binning
#ifndef VPS_Y_EARLY_G1
#define VPS_Y_EARLY_G1 0   // wide y pass: the second group of lines requested ahead of the first group's transform (experiment)
#endif
#ifndef VPS_Y_ST16
#define VPS_Y_ST16 1   // wide y pass: 16-byte stores (2048^3 launch 12.56 -> 12.10 ms; 0 restores the 8-byte epilogue)
#endif
namespace {

typedef float2 cf;
typedef float vps_f4 __attribute__((ext_vector_type(4)));

// (written on whole (re, im) pairs: the compiler then keeps a complex value in one aligned register pair and maps these
// onto v_pk_add / v_pk_mul / v_pk_fma with operand-select modifiers; the component-wise form was re-vectorised across
// DIFFERENT values and paid for it in register moves -- a fifth of the VALU instructions of the 2048-point passes)
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cmul(cf a, cf b) {
  const cf t = a * make_float2(b.x, b.x);
  const cf s = make_float2(a.y, a.x) * make_float2(-b.y, b.y);
  return t + s;
}
// multiply by -i
__device__ __forceinline__ cf cmul_mi(cf a) { return make_float2(a.y, -a.x); }

#define VPS_SQRT1_2 0.70710678118654752440f
#define VPS_COS_PI_8 0.92387953251128675613f
#define VPS_SIN_PI_8 0.38268343236508977173f

// ---- small DFTs on registers, forward sign exp(-2 pi i nk/R), natural order ----
template <int R>
struct Dft;

template <>
struct Dft<1> {
  static __device__ __forceinline__ void run(cf*) {}
};
template <>
struct Dft<2> {
  static __device__ __forceinline__ void run(cf* v) {
    cf a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};
template <>
struct Dft<4> {
  static __device__ __forceinline__ void run(cf* v) {
    cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    cf t2 = cadd(v[1], v[3]), t3 = cmul_mi(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
  }
};
template <>
struct Dft<8> {
  static __device__ __forceinline__ void run(cf* v) {
    cf e[4] = {v[0], v[2], v[4], v[6]};
    cf o[4] = {v[1], v[3], v[5], v[7]};
    Dft<4>::run(e);
    Dft<4>::run(o);
    // o[k] *= w8^k
    cf o1 = make_float2((o[1].x + o[1].y) * VPS_SQRT1_2, (o[1].y - o[1].x) * VPS_SQRT1_2);
    cf o2 = cmul_mi(o[2]);
    cf o3 = make_float2((o[3].y - o[3].x) * VPS_SQRT1_2, -(o[3].x + o[3].y) * VPS_SQRT1_2);
    v[0] = cadd(e[0], o[0]);
    v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);
    v[5] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);
    v[6] = csub(e[2], o2);
    v[3] = cadd(e[3], o3);
    v[7] = csub(e[3], o3);
  }
};
template <>
struct Dft<16> {
  static __device__ __forceinline__ void run(cf* v) {
    // n = 4*n1 + n2, k = k1 + 4*k2
    cf y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      cf t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
      Dft<4>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) y[n2][k1] = t[k1];
    }
    // twiddles w16^(n2*k1)
    const cf w1 = make_float2(VPS_COS_PI_8, -VPS_SIN_PI_8);
    const cf w2 = make_float2(VPS_SQRT1_2, -VPS_SQRT1_2);
    const cf w3 = make_float2(VPS_SIN_PI_8, -VPS_COS_PI_8);
    const cf w6 = make_float2(-VPS_SQRT1_2, -VPS_SQRT1_2);
    const cf w9 = make_float2(-VPS_COS_PI_8, VPS_SIN_PI_8);
    y[1][1] = cmul(y[1][1], w1);
    y[1][2] = cmul(y[1][2], w2);
    y[1][3] = cmul(y[1][3], w3);
    y[2][1] = cmul(y[2][1], w2);
    y[2][2] = cmul_mi(y[2][2]);
    y[2][3] = cmul(y[2][3], w6);
    y[3][1] = cmul(y[3][1], w3);
    y[3][2] = cmul(y[3][2], w6);
    y[3][3] = cmul(y[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      cf t[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
      Dft<4>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = t[k2];
    }
  }
};

// radix 3, 6 = 2 x 3, 12 = 2 x 6 (forward): lengths 3 * 2^a (N = 96, 192, 384, 768)
#define VPS_SQRT3_2 0.86602540378443864676f
template <>
struct Dft<3> {
  static __device__ __forceinline__ void run(cf* v) {
    const cf t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const cf m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    const cf u = make_float2(VPS_SQRT3_2 * d.y, -VPS_SQRT3_2 * d.x);   // -i (sqrt(3)/2) d
    v[0] = cadd(v[0], t);
    v[1] = cadd(m, u);
    v[2] = csub(m, u);
  }
};
template <>
struct Dft<6> {
  static __device__ __forceinline__ void run(cf* v) {
    cf e[3] = {v[0], v[2], v[4]};
    cf o[3] = {v[1], v[3], v[5]};
    Dft<3>::run(e);
    Dft<3>::run(o);
    o[1] = cmul(o[1], make_float2(0.5f, -VPS_SQRT3_2));    // w6
    o[2] = cmul(o[2], make_float2(-0.5f, -VPS_SQRT3_2));   // w6^2
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = cadd(e[k], o[k]);
      v[k + 3] = csub(e[k], o[k]);
    }
  }
};
template <>
struct Dft<12> {
  static __device__ __forceinline__ void run(cf* v) {
    cf e[6] = {v[0], v[2], v[4], v[6], v[8], v[10]};
    cf o[6] = {v[1], v[3], v[5], v[7], v[9], v[11]};
    Dft<6>::run(e);
    Dft<6>::run(o);
    // o[k] *= w12^k, w12 = exp(-i pi/6)
    o[1] = cmul(o[1], make_float2(VPS_SQRT3_2, -0.5f));
    o[2] = cmul(o[2], make_float2(0.5f, -VPS_SQRT3_2));
    o[3] = cmul_mi(o[3]);
    o[4] = cmul(o[4], make_float2(-0.5f, -VPS_SQRT3_2));
    o[5] = cmul(o[5], make_float2(-VPS_SQRT3_2, -0.5f));
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      v[k] = cadd(e[k], o[k]);
      v[k + 6] = csub(e[k], o[k]);
    }
  }
};

// radix 5 (forward, w = exp(-2 pi i/5)) and radix 10 = 2 x 5: lengths 2^a 5^b (N = 250, 500, 1000)
#define VPS_C1_5 0.30901699437494742410f    /* cos(2 pi/5) */
#define VPS_C2_5 (-0.80901699437494742410f) /* cos(4 pi/5) */
#define VPS_S1_5 0.95105651629515357212f    /* sin(2 pi/5) */
#define VPS_S2_5 0.58778525229247312917f    /* sin(4 pi/5) */
template <>
struct Dft<5> {
  static __device__ __forceinline__ void run(cf* v) {
    const cf a = cadd(v[1], v[4]), b = csub(v[1], v[4]);
    const cf c = cadd(v[2], v[3]), d = csub(v[2], v[3]);
    const cf t1 = make_float2(v[0].x + VPS_C1_5 * a.x + VPS_C2_5 * c.x, v[0].y + VPS_C1_5 * a.y + VPS_C2_5 * c.y);
    const cf t2 = make_float2(v[0].x + VPS_C2_5 * a.x + VPS_C1_5 * c.x, v[0].y + VPS_C2_5 * a.y + VPS_C1_5 * c.y);
    // u = -i * (s b +- s' d)
    const cf q1 = make_float2(VPS_S1_5 * b.x + VPS_S2_5 * d.x, VPS_S1_5 * b.y + VPS_S2_5 * d.y);
    const cf q2 = make_float2(VPS_S2_5 * b.x - VPS_S1_5 * d.x, VPS_S2_5 * b.y - VPS_S1_5 * d.y);
    const cf u1 = cmul_mi(q1), u2 = cmul_mi(q2);
    v[0] = make_float2(v[0].x + a.x + c.x, v[0].y + a.y + c.y);
    v[1] = cadd(t1, u1);
    v[4] = csub(t1, u1);
    v[2] = cadd(t2, u2);
    v[3] = csub(t2, u2);
  }
};
template <>
struct Dft<10> {
  static __device__ __forceinline__ void run(cf* v) {
    cf e[5] = {v[0], v[2], v[4], v[6], v[8]};
    cf o[5] = {v[1], v[3], v[5], v[7], v[9]};
    Dft<5>::run(e);
    Dft<5>::run(o);
    // o[k] *= w10^k, w10 = exp(-2 pi i/10): cos(pi/5) = -c2_5, sin(pi/5) = s2_5, cos(2pi/5) = c1_5, sin(2pi/5) = s1_5
    const cf w1 = make_float2(-VPS_C2_5, -VPS_S2_5), w2 = make_float2(VPS_C1_5, -VPS_S1_5);
    const cf w3 = make_float2(-VPS_C1_5, -VPS_S1_5), w4 = make_float2(VPS_C2_5, -VPS_S2_5);
    o[1] = cmul(o[1], w1);
    o[2] = cmul(o[2], w2);
    o[3] = cmul(o[3], w3);
    o[4] = cmul(o[4], w4);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      v[k] = cadd(e[k], o[k]);
      v[k + 5] = csub(e[k], o[k]);
    }
  }
};

// radix 24 = 8 x 3 and radix 20 = 5 x 4 (generated: twiddle constants are cos / sin of -2 pi m / N in double, printed to 17 digits)
template <>
struct Dft<24> {
  static __device__ __forceinline__ void run(cf* v) {
    // n = 3 n1 + n2, k = k1 + 8 k2: 3 x Dft<8>, twiddles w24^(n2 k1), 8 x Dft<3>
    cf y[3][8];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
      cf t[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) t[n1] = v[3 * n1 + n2];
      Dft<8>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) y[n2][k1] = t[k1];
    }
    y[1][1] = cmul(y[1][1], make_float2(0.96592582628906831f, -0.25881904510252074f));
    y[1][2] = cmul(y[1][2], make_float2(0.86602540378443871f, -0.49999999999999994f));
    y[1][3] = cmul(y[1][3], make_float2(0.70710678118654757f, -0.70710678118654746f));
    y[1][4] = cmul(y[1][4], make_float2(0.50000000000000011f, -0.8660254037844386f));
    y[1][5] = cmul(y[1][5], make_float2(0.25881904510252074f, -0.96592582628906831f));
    y[1][6] = cmul_mi(y[1][6]);
    y[1][7] = cmul(y[1][7], make_float2(-0.25881904510252063f, -0.96592582628906831f));
    y[2][1] = cmul(y[2][1], make_float2(0.86602540378443871f, -0.49999999999999994f));
    y[2][2] = cmul(y[2][2], make_float2(0.50000000000000011f, -0.8660254037844386f));
    y[2][3] = cmul_mi(y[2][3]);
    y[2][4] = cmul(y[2][4], make_float2(-0.49999999999999978f, -0.86602540378443871f));
    y[2][5] = cmul(y[2][5], make_float2(-0.86602540378443871f, -0.49999999999999994f));
    y[2][6] = make_float2(-y[2][6].x, -y[2][6].y);
    y[2][7] = cmul(y[2][7], make_float2(-0.86602540378443882f, 0.49999999999999972f));
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      cf t[3];
#pragma unroll
      for (int n2 = 0; n2 < 3; ++n2) t[n2] = y[n2][k1];
      Dft<3>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) v[k1 + 8 * k2] = t[k2];
    }
  }
};
template <>
struct Dft<20> {
  static __device__ __forceinline__ void run(cf* v) {
    // n = 4 n1 + n2, k = k1 + 5 k2: 4 x Dft<5>, twiddles w20^(n2 k1), 5 x Dft<4>
    cf y[4][5];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      cf t[5];
#pragma unroll
      for (int n1 = 0; n1 < 5; ++n1) t[n1] = v[4 * n1 + n2];
      Dft<5>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < 5; ++k1) y[n2][k1] = t[k1];
    }
    y[1][1] = cmul(y[1][1], make_float2(0.95105651629515353f, -0.3090169943749474f));
    y[1][2] = cmul(y[1][2], make_float2(0.80901699437494745f, -0.58778525229247314f));
    y[1][3] = cmul(y[1][3], make_float2(0.58778525229247314f, -0.80901699437494745f));
    y[1][4] = cmul(y[1][4], make_float2(0.30901699437494745f, -0.95105651629515353f));
    y[2][1] = cmul(y[2][1], make_float2(0.80901699437494745f, -0.58778525229247314f));
    y[2][2] = cmul(y[2][2], make_float2(0.30901699437494745f, -0.95105651629515353f));
    y[2][3] = cmul(y[2][3], make_float2(-0.30901699437494734f, -0.95105651629515364f));
    y[2][4] = cmul(y[2][4], make_float2(-0.80901699437494734f, -0.58778525229247325f));
    y[3][1] = cmul(y[3][1], make_float2(0.58778525229247314f, -0.80901699437494745f));
    y[3][2] = cmul(y[3][2], make_float2(-0.30901699437494734f, -0.95105651629515364f));
    y[3][3] = cmul(y[3][3], make_float2(-0.95105651629515353f, -0.30901699437494751f));
    y[3][4] = cmul(y[3][4], make_float2(-0.80901699437494756f, 0.58778525229247303f));
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
      cf t[4];
#pragma unroll
      for (int n2 = 0; n2 < 4; ++n2) t[n2] = y[n2][k1];
      Dft<4>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 5 * k2] = t[k2];
    }
  }
};

// ---- per-length plans: L lanes per line, radices R0*R1*R2 = NC -----------------
template <int NC>
struct Plan;
#define VPS_PLAN(NC_, L_, A_, B_, C_)                          \
  template <>                                                  \
  struct Plan<NC_> {                                           \
    static constexpr int L = L_, R0 = A_, R1 = B_, R2 = C_;    \
  };
VPS_PLAN(8, 1, 8, 1, 1)
VPS_PLAN(16, 1, 16, 1, 1)
VPS_PLAN(32, 4, 8, 4, 1)
VPS_PLAN(64, 8, 8, 8, 1)
VPS_PLAN(128, 8, 16, 8, 1)
VPS_PLAN(256, 16, 16, 16, 1)
VPS_PLAN(512, 32, 8, 8, 8)
VPS_PLAN(1024, 64, 16, 8, 8)
VPS_PLAN(2048, 128, 16, 16, 8)
VPS_PLAN(4096, 256, 16, 16, 16)
// 2^a 5^b lengths: N = 250, 500, 1000 (lines of N/2 packed-real and N complex points)
// 3 * 2^a lengths: N = 96, 192, 384, 768
VPS_PLAN(48, 2, 8, 6, 1)
VPS_PLAN(96, 4, 8, 12, 1)
VPS_PLAN(192, 8, 8, 8, 3)
VPS_PLAN(384, 16, 8, 8, 6)
VPS_PLAN(768, 32, 8, 8, 12)
VPS_PLAN(1536, 64, 8, 8, 24)
VPS_PLAN(125, 25, 5, 5, 5)
VPS_PLAN(250, 25, 10, 5, 5)
VPS_PLAN(500, 25, 10, 10, 5)
VPS_PLAN(1000, 50, 10, 10, 10)
VPS_PLAN(2000, 100, 10, 10, 20)

template <int NC>
struct PlanInfo {
  typedef Plan<NC> P;
  static constexpr int L = P::L;
  static constexpr int RL = NC / L;
  static constexpr int R0 = P::R0, R1 = P::R1, R2 = P::R2;
  static constexpr int NS1 = R0, NS2 = R0 * R1;
  static constexpr int TW1 = (R1 > 1) ? (R1 - 1) * NS1 : 0;
  static constexpr int TW2 = (R2 > 1) ? (R2 - 1) * NS2 : 0;
  static constexpr int TW = TW1 + TW2;  // entries of the per-stage twiddle image
  // the image is staged in LDS except for the longest lines, where the tile itself needs
  // nearly all of the 160 KiB (the table then stays in L2)
  static constexpr bool TWLDS = NC <= 2048;
  static constexpr int TWL = TWLDS ? ((TW + 1) & ~1) : 0;  // LDS entries reserved for it
  // LDS line pitch (complex elements): one pad slot per 32 elements
  static constexpr int PITCH = NC + (NC >> 5) + 1;
  static_assert(R0 * R1 * R2 == NC, "bad plan");
  static_assert(RL % R0 == 0 && RL % R1 == 0 && RL % R2 == 0, "radix must divide RL");
};

// Position of element p inside a line's LDS exchange buffer: one pad slot per 32 elements.
// (Measured alternative: the XOR swizzle p ^ ((p >> A) & 15) makes every exchange access of the
// 512..4096-point plans bank-conflict free on paper, but its per-access address arithmetic
// costs ~30 VGPRs, drops the x pass from 3 to 2 waves/SIMD and ran 20-60 % slower at
// N = 1024/2048, with no measurable gain in the z/y passes: the exchanges are not the limiter.)
template <int NC>
__device__ __forceinline__ int padidx(int p) {
  return p + (p >> 5);
}

// Transposed tile image [k][t] with pitch T and an XOR swizzle of t, so that both the
// column-wise writes (16 consecutive k, one t) and the row-wise reads (one k, all t)
// of ds_write_b64 / ds_read_b64 touch distinct LDS slots.
template <int T>
__device__ __forceinline__ int tridx(int k, int t) {
  constexpr int SH = (T >= 16) ? 0 : ((T == 8) ? 1 : ((T == 4) ? 2 : ((T == 2) ? 3 : 4)));
  return k * T + (t ^ ((k >> SH) & (T - 1)));
}

// Stage s input for lane l: v[m*R + r] = src(l + L*m + r*NC/R)
template <int NC, int L, int RL, int R>
__device__ __forceinline__ void lds_load_stage(cf (&v)[RL], const cf* line, int l) {
  constexpr int NB = RL / R;
  // Where every stride is a multiple of the 32 elements between pad slots, padidx(l + c) = padidx(l) + c + c / 32 exactly: ONE
  // lane-dependent address and immediate offsets.  (Left to the compiler each of the RL addresses cost four VALU operations --
  // or, shift, and, add3 -- a fifth of the instructions of a 2048-point transform.)
  if constexpr ((L % 32 == 0) && ((NC / R) % 32 == 0)) {
    const cf* base = line + padidx<NC>(l);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int c = L * m + r * (NC / R);
        v[m * R + r] = base[c + (c >> 5)];
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < NB; ++m) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[m * R + r] = line[padidx<NC>(l + L * m + r * (NC / R))];
    }
  }
}

template <int NC, int L, int RL, int R, int NS>
__device__ __forceinline__ void twiddle_butterfly(cf (&v)[RL], const cf* tw, int l) {
  constexpr int NB = RL / R;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    if constexpr (NS > 1) {
      const int k = (l + L * m) % NS;   // (NS is a compile-time constant: a mask for powers of two)
#pragma unroll
      for (int r = 1; r < R; ++r) v[m * R + r] = cmul(v[m * R + r], tw[(r - 1) * NS + k]);
    }
    Dft<R>::run(&v[m * R]);
  }
}

// The same with the stage's twiddles in registers, loaded once per kernel (load_stage_twiddles): they depend on the lane alone.
// twr[r - 1] = tw[(r - 1) * NS + l] is the twiddle of the lane's FIRST butterfly (m = 0); for m > 0 the index moves by L m and
// the twiddle by the constant factor exp(-2 pi i r L m / (NS R)) -- a 16th root of unity for the last stage of the 2048-point
// plan (L = 128, NS = 256, R = 8), applied to the value first (rot16: two packed operations, none for r = 4).  Seven values in
// 14 VGPRs instead of 14 KB of LDS that the x pass no longer needs.
template <int Q>   // v * exp(-2 pi i Q / 16)
__device__ __forceinline__ cf rot16(cf v) {
  constexpr int q = ((Q % 16) + 16) % 16;
  if constexpr (q == 0) return v;
  else if constexpr (q == 4) return cmul_mi(v);
  else if constexpr (q == 8) return make_float2(-v.x, -v.y);
  else if constexpr (q == 12) return make_float2(-v.y, v.x);
  else if constexpr (q == 2) return make_float2((v.x + v.y) * VPS_SQRT1_2, (v.y - v.x) * VPS_SQRT1_2);
  else if constexpr (q == 6) return make_float2((v.y - v.x) * VPS_SQRT1_2, -(v.x + v.y) * VPS_SQRT1_2);
  else if constexpr (q == 10) return make_float2(-(v.x + v.y) * VPS_SQRT1_2, (v.x - v.y) * VPS_SQRT1_2);
  else if constexpr (q == 14) return make_float2((v.x - v.y) * VPS_SQRT1_2, (v.x + v.y) * VPS_SQRT1_2);
  else {
    // odd q: cos / sin of q pi / 8 from the pi / 8 pair
    constexpr float c = (q == 1 || q == 15) ? VPS_COS_PI_8 : (q == 3 || q == 13) ? VPS_SIN_PI_8 : (q == 5 || q == 11) ? -VPS_SIN_PI_8 : -VPS_COS_PI_8;
    constexpr float sn = (q == 1 || q == 7) ? VPS_SIN_PI_8 : (q == 3 || q == 5) ? VPS_COS_PI_8 : (q == 9 || q == 15) ? -VPS_SIN_PI_8 : -VPS_COS_PI_8;
    return cmul(v, make_float2(c, -sn));   // exp(-i q pi / 8) = cos - i sin
  }
}
template <int NC, int L, int RL, int R, int NS>
struct RegTwiddles {
  static constexpr int NB = RL / R;
  // SAME: L is a multiple of NS, every butterfly of a lane has the same twiddle index l % NS.  Otherwise the index moves by L
  // per butterfly without wrapping, and the angle by STEP16 sixteenths of a turn per unit of r.
  static constexpr bool SAME = (L % NS == 0);
  static constexpr bool OK = SAME || ((L * (NB - 1) < NS) && ((16 * L) % (NS * R) == 0));
  static constexpr int STEP16 = (OK && !SAME) ? (16 * L) / (NS * R) : 0;
};
template <int NC, int L, int RL, int R, int NS, int M, int RR>
__device__ __forceinline__ void twiddle_reg_apply(cf (&v)[RL], const cf* twr) {
  if constexpr (RR < R) {
    v[M * R + RR] = cmul(rot16<RegTwiddles<NC, L, RL, R, NS>::STEP16 * M * RR>(v[M * R + RR]), twr[RR - 1]);
    twiddle_reg_apply<NC, L, RL, R, NS, M, RR + 1>(v, twr);
  }
}
template <int NC, int L, int RL, int R, int NS, int M = 0>
__device__ __forceinline__ void twiddle_butterfly_reg(cf (&v)[RL], const cf* twr) {
  static_assert(RegTwiddles<NC, L, RL, R, NS>::OK, "register twiddles: constant 16th-root steps between a lane's butterflies");
  if constexpr (M < RL / R) {
    twiddle_reg_apply<NC, L, RL, R, NS, M, 1>(v, twr);
    Dft<R>::run(&v[M * R]);
    twiddle_butterfly_reg<NC, L, RL, R, NS, M + 1>(v, twr);
  }
}
template <int NC, int L, int RL, int R, int NS>
__device__ __forceinline__ void load_stage_twiddles(cf (&twr)[R - 1], const cf* __restrict__ tw_global, int l) {
#pragma unroll
  for (int r = 1; r < R; ++r) twr[r - 1] = tw_global[(r - 1) * NS + l % NS];
}

// Can padidx(j0 + r NS) be formed as padidx(j0) + (r NS + r NS / 32) for every butterfly of a radix-R stage?  j0 = A + k with A a
// multiple of NS R and k < NS; the split is exact iff (j0 mod 32) + (r NS mod 32) < 32 -- checked here for every residue.
template <int R, int NS>
constexpr bool pad_split_ok() {
  int g = NS * R;
  while (32 % g != 0 && g > 1) {   // gcd(NS R, 32) for the power-of-two cases; anything else: no split
    if (g % 2) return false;
    g /= 2;
  }
  if (32 % g != 0) return false;
  for (int a = 0; a < 32; a += g)
    for (int k = 0; k < NS && k < 32; ++k)
      for (int r = 0; r < R; ++r)
        if ((a + k) % 32 + (r * NS) % 32 >= 32) return false;
  return (NS & (NS - 1)) == 0 && (R & (R - 1)) == 0;
}
template <int NC, int L, int RL, int R, int NS>
__device__ __forceinline__ void lds_store_stage(const cf (&v)[RL], cf* line, int l) {
  constexpr int NB = RL / R;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const int j = l + L * m;
    const int k = j % NS;
    const int j0 = (j - k) * R + k;
    if constexpr (pad_split_ok<R, NS>()) {   // one lane-dependent address per butterfly, immediate offsets (see lds_load_stage)
      cf* base = line + padidx<NC>(j0);
#pragma unroll
      for (int r = 0; r < R; ++r) base[r * NS + ((r * NS) >> 5)] = v[m * R + r];
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) line[padidx<NC>(j0 + r * NS)] = v[m * R + r];
    }
  }
}

// Exchange synchronisation.  When a line's L lanes live inside one wave (L <= 64, lines never
// straddle waves) the stage exchanges only need wave-level ordering: a wave's LDS operations
// execute in issue order, so a compiler fence is enough and the waves of a workgroup run
// decoupled.  Otherwise a workgroup barrier.
template <bool WAVE>
__device__ __forceinline__ void exchange_sync() {
  if constexpr (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// Second exchange of the 1024-point plan (64 lanes x 16 points, radices 16 x 8 x 8) WITHOUT LDS.  Between the two radix-8 stages
// element (lane 16 g + c, slot m*8 + 4a + b) of the next stage is element (lane 16 b + c, slot a*8 + 4m + g) of the previous one:
// the column c inside a row of 16 lanes stays, and for every (a, m) the four registers q = 0..3 at slots a*8 + 4m + q are
// transposed against the four lane ROWS.  gfx950's v_permlane32_swap (upper half of vdst <-> lower half of vsrc) and
// v_permlane16_swap (odd rows of vdst <-> even rows of vsrc) do a 4 x 4 row transpose of four registers in four instructions:
// 32 VALU swaps replace 16 ds_write_b64 + 16 ds_read_b64 and a wave-level sync in a kernel whose LDS pipe is the busiest unit
// (pencil kernel at C4: SQ_ACTIVE_INST_LDS x 16 waves ~ 90 % of the CU's cycles).  tools/micro/permlane_swap.hip pins the semantics.
#ifndef VPS_NO_SWAP_EXCHANGE
#define VPS_SWAP_EXCHANGE 1
#else
#define VPS_SWAP_EXCHANGE 0
#endif
template <int NC, int L, bool WAVE>
constexpr bool swap_exchange2() {
  return VPS_SWAP_EXCHANGE && WAVE && NC == 1024 && L == 64 && PlanInfo<NC>::R0 == 16 && PlanInfo<NC>::R1 == 8 && PlanInfo<NC>::R2 == 8;
}
typedef unsigned vps_u2 __attribute__((ext_vector_type(2)));
// y_b at lane row g = x_g at lane row b (rows of 16 lanes), in place
__device__ __forceinline__ void rows_transpose4(float& x0, float& x1, float& x2, float& x3) {
  vps_u2 r;
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x2), false, false);
  x0 = __uint_as_float(r.x); x2 = __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x1), __float_as_uint(x3), false, false);
  x1 = __uint_as_float(r.x); x3 = __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
  x0 = __uint_as_float(r.x); x1 = __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x2), __float_as_uint(x3), false, false);
  x2 = __uint_as_float(r.x); x3 = __uint_as_float(r.y);
}

// Runs stages 1.. (stage 0 inputs already in v).  On return v holds the spectrum:
// v[m*RLAST + r] = F[l + L*m + r*NC/RLAST].  L lanes per line (default: the plan's; the persistent transposing pass
// of the longest lines runs a line on half as many lanes with twice the points each).
// TWREG: bit 0 -- the twiddles of stage 1 come from registers (twr1), bit 1 -- those of stage 2 (twr); see twiddle_butterfly_reg
template <int NC, int L, bool WAVE, int TWREG = 0>
__device__ __forceinline__ void fft_from_regs_l(cf (&v)[NC / L], cf* line, const cf* tw, int l, const cf* twr = nullptr,
                                                const cf* twr1 = nullptr) {
  typedef PlanInfo<NC> PI;
  constexpr int RL = NC / L;
  static_assert(RL % PI::R0 == 0 && RL % PI::R1 == 0 && RL % PI::R2 == 0, "radix must divide RL");
  twiddle_butterfly<NC, L, RL, PI::R0, 1>(v, tw, l);
  if constexpr (PI::R1 > 1) {
    lds_store_stage<NC, L, RL, PI::R0, 1>(v, line, l);
    exchange_sync<WAVE>();
    lds_load_stage<NC, L, RL, PI::R1>(v, line, l);
    if constexpr ((TWREG & 1) != 0)
      twiddle_butterfly_reg<NC, L, RL, PI::R1, PI::NS1>(v, twr1);
    else
      twiddle_butterfly<NC, L, RL, PI::R1, PI::NS1>(v, tw, l);
    if constexpr (PI::R2 > 1) {
      if constexpr (swap_exchange2<NC, L, WAVE>()) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // the four (a, m) groups: slots 4 g4 .. 4 g4 + 3
          rows_transpose4(v[4 * g4].x, v[4 * g4 + 1].x, v[4 * g4 + 2].x, v[4 * g4 + 3].x);
          rows_transpose4(v[4 * g4].y, v[4 * g4 + 1].y, v[4 * g4 + 2].y, v[4 * g4 + 3].y);
        }
        // slot m*8 + 4a + b of the next stage <- transposed slot a*8 + 4m + b: (a, m) = (0, 1) and (1, 0) trade places
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const cf tmp = v[4 + b];
          v[4 + b] = v[8 + b];
          v[8 + b] = tmp;
        }
      } else {
        exchange_sync<WAVE>();
        lds_store_stage<NC, L, RL, PI::R1, PI::NS1>(v, line, l);
        exchange_sync<WAVE>();
        lds_load_stage<NC, L, RL, PI::R2>(v, line, l);
      }
      if constexpr ((TWREG & 2) != 0)
        twiddle_butterfly_reg<NC, L, RL, PI::R2, PI::NS2>(v, twr);
      else
        twiddle_butterfly<NC, L, RL, PI::R2, PI::NS2>(v, tw + PI::TW1, l);
    }
  }
}
template <int NC, bool WAVE = false, int TWREG = 0>
__device__ __forceinline__ void fft_from_regs(cf (&v)[PlanInfo<NC>::RL], cf* line, const cf* tw, int l, const cf* twr = nullptr) {
  fft_from_regs_l<NC, PlanInfo<NC>::L, WAVE, TWREG>(v, line, tw, l, twr);
}

template <int NC>
struct LastRadix {
  typedef PlanInfo<NC> PI;
  static constexpr int R = (PI::R2 > 1) ? PI::R2 : ((PI::R1 > 1) ? PI::R1 : PI::R0);
};

// output index of register slot i = m*R + r after the last stage
template <int NC, int L>
__device__ __forceinline__ int out_index_l(int l, int i) {
  constexpr int R = LastRadix<NC>::R;
  const int m = i / R, r = i % R;
  return l + L * m + r * (NC / R);
}
template <int NC>
__device__ __forceinline__ int out_index(int l, int i) {
  return out_index_l<NC, PlanInfo<NC>::L>(l, i);
}

struct PassParams {
  const void* in;
  const void* in_w;          // REAL passes: optional second real field of the same layout; the line transformed is in * in_w
                             // (momentum p_c = v_c * mass of a gridded field without a separate algebra pass)
  void* out;
  void* out_nyq;
  long long in_sa, in_sb;  // input strides of the tile axis a and batch axis b (elements)
  long long out_ob, out_ok;  // output strides (complex elements): out[b*ob + k*ok + a]
  long long nyq_ob;          // out_nyq[b*nyq_ob + a]
  int A, B;                  // extent of tile axis / batch axis
  const cf* tw_stage;
  const cf* tw_r2c;
  // Chunked slab exchange (vps_fft_y): the B batches of a launch are `bg`-row groups, one per destination rank --
  // launch batch b reads input batch (b / bg) * bg_in + b_off + b % bg and its output starts bg_gap elements later per
  // group (room for the Nyquist rows that ride behind a destination's last chunk).  bg = 0: plain batches.
  int bg, b_off;
  long long bg_in, bg_gap;
  // the Nyquist-plane launch (B = 1): output row k moves by (k / kg) * kg_gap elements (kg = 0: none), so that each
  // destination's rows land behind that destination's block
  int kg;
  long long kg_gap;
  // binning-only consumers (vps_set_bin_only): rows ky with min(ky, NC - ky) > kcut[kz] are not stored -- every mode
  // there lies beyond the last shell edge and the binning x pass does not read them.  kz = kz_fixed, or the input batch
  const int* kcut;
  int kz_fixed;
  // chunked exchange (vps_fft_y): ptab[j] = {first row, kc} of the j-th plane of every destination's block (j = launch batch %
  // bg) -- the plane keeps the 2 kc + 1 rows |ky| <= kc (kc >= its kcut; -1: all NC rows): row ky at position ky, row NC - i at
  // position 2 kc + 1 - i behind the block's first row.  Launch batch b reads input batch
  // b_off + (b / bg) * bg_in + (b % bg) * bg_step.  NULL: plain planes of NC rows, out_ob apart.
  const int2* ptab = nullptr;
  int bg_step = 1;
};

__device__ __forceinline__ int packed_row(int k, int NC, int kc_pack) {
  return (kc_pack < 0 || k <= kc_pack) ? k : k - (NC - 2 * kc_pack - 1);
}

// Non-temporal (streaming) access to one complex value: data that is written once for the next pass
// or read once from the previous one should not displace what the caches hold.
// Measured (512^3 / 1024^3): x-pass loads -12 / -14 %, z-side stores -6 %.
__device__ __forceinline__ cf load_stream(const cf* ptr) {
  const double raw = __builtin_nontemporal_load(reinterpret_cast<const double*>(ptr));
  return *reinterpret_cast<const cf*>(&raw);
}
__device__ __forceinline__ void store_stream(cf* ptr, cf v) {
  __builtin_nontemporal_store(*reinterpret_cast<const double*>(&v), reinterpret_cast<double*>(ptr));
}

// Real-input post-processing of a transposed tile image buf[k][t] (NC packed-complex outputs Z per line):
//   X[k] = 0.5*((Z[k]+conj(Z[NC-k])) - i w^k (Z[k]-conj(Z[NC-k]))),  w = exp(-2 pi i/(2 NC)),
// X[0] and X[NC] (Nyquist, stored apart) real.  Modes k and NC-k share Z[k], Z[NC-k] and, because
// w^(NC-k) = -conj(w^k), the product w^k (Z[k]-conj(Z[NC-k])): one thread writes both.
template <int NC, int T, int NT, bool BOUNDS, bool PLAIN = false>
__device__ __forceinline__ void r2c_store_tile(const cf* buf, int tid, const cf* __restrict__ tw_r2c, cf* out,
                                               long long out_ok, cf* nyq, int nlive) {
  // pair index kp = 0 .. NC/2 (the last one only for odd NC: for even NC the self-paired mode NC/2 rides with kp = 0)
  constexpr int PAIRS = (NC / 2 + (NC & 1)) * T;
  constexpr int IT = (PAIRS + NT - 1) / NT;
#pragma unroll 4
  for (int i = 0; i < IT; ++i) {
    const int idx = tid + i * NT;
    if ((PAIRS % NT) != 0 && idx >= PAIRS) break;
    const int tt = idx % T, k = idx / T;
#ifdef VPS_PENCIL_NOSTORE
    const cf zk = buf[tridx<T>(k, tt)];
    const bool ok = (!BOUNDS || tt < nlive) && (zk.x == 1.2345e30f);   // TIMING ONLY: nothing is stored
#else
    const bool ok = !BOUNDS || tt < nlive;
    const cf zk = buf[tridx<T>(k, tt)];
#endif
    if (k == 0) {
      if (ok) {
        out[tt] = make_float2(zk.x + zk.y, 0.f);
        nyq[tt] = make_float2(zk.x - zk.y, 0.f);
      }
      if constexpr ((NC & 1) == 0) {
        const cf zh = buf[tridx<T>(NC / 2, tt)];
        if (ok) out[(long long)(NC / 2) * out_ok + tt] = make_float2(zh.x, -zh.y);
      }
    } else {
      const cf zn = buf[tridx<T>(NC - k, tt)];
      const cf w = tw_r2c[k];
      const cf sm = make_float2(zk.x + zn.x, zk.y - zn.y);   // Z[k] + conj(Z[NC-k])
      const cf d = make_float2(zk.x - zn.x, zk.y + zn.y);    // Z[k] - conj(Z[NC-k])
      const cf wd = cmul(w, d);
      if (ok) {
        // -i * wd = (wd.y, -wd.x);  for NC-k: conj(sm) and -i * conj(wd) = (-wd.y, -wd.x)
        if constexpr (PLAIN) {
          out[(long long)k * out_ok + tt] = make_float2(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x));
          out[(long long)(NC - k) * out_ok + tt] = make_float2(0.5f * (sm.x - wd.y), 0.5f * (-sm.y - wd.x));
        } else {
        store_stream(&out[(long long)k * out_ok + tt], make_float2(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x)));
        store_stream(&out[(long long)(NC - k) * out_ok + tt], make_float2(0.5f * (sm.x - wd.y), 0.5f * (-sm.y - wd.x)));
        }
      }
    }
  }
}

// The same for 8-line tiles with 16-byte stores: a thread owns the lines (tt, tt + 1) of a mode pair, so a 64-byte output
// segment leaves as four dwordx4 stores instead of eight dwordx2 (half the store instructions of the pencil kernel's epilogue).
// zero_behind (uniform): every image element is cleared once it has been read (each is read by exactly one thread), so that the
// region is an all-zero accumulator again when the epilogue ends -- the next component's dense zero-fill and the barrier behind
// it move off the critical path into this store-issue-bound loop.
template <int NC, int T, int NT, bool PLAIN = false>
__device__ __forceinline__ void r2c_store_tile16(cf* buf, int tid, const cf* __restrict__ tw_r2c, cf* out,
                                                 long long out_ok, cf* nyq, bool zero_behind = false, const cf* wpre = nullptr) {
  static_assert((NC & 1) == 0 && (T & 1) == 0, "pairs of lines");
  constexpr int H = T / 2;
  constexpr int ITEMS = (NC / 2) * H;
  static_assert(ITEMS % NT == 0, "whole rounds");
#pragma unroll 4
  for (int i = 0; i < ITEMS / NT; ++i) {
    const int idx = tid + i * NT;
    const int tt = (idx % H) * 2, k = idx / H;
    const cf zz = make_float2(0.f, 0.f);
    const cf a0 = buf[tridx<T>(k, tt)], a1 = buf[tridx<T>(k, tt + 1)];
    if (zero_behind) {
      buf[tridx<T>(k, tt)] = zz;
      buf[tridx<T>(k, tt + 1)] = zz;
    }
    if (k == 0) {
      *reinterpret_cast<vps_f4*>(&out[tt]) = vps_f4{a0.x + a0.y, 0.f, a1.x + a1.y, 0.f};
      *reinterpret_cast<vps_f4*>(&nyq[tt]) = vps_f4{a0.x - a0.y, 0.f, a1.x - a1.y, 0.f};
      const cf h0 = buf[tridx<T>(NC / 2, tt)], h1 = buf[tridx<T>(NC / 2, tt + 1)];
      if (zero_behind) {
        buf[tridx<T>(NC / 2, tt)] = zz;
        buf[tridx<T>(NC / 2, tt + 1)] = zz;
      }
      *reinterpret_cast<vps_f4*>(&out[(long long)(NC / 2) * out_ok + tt]) = vps_f4{h0.x, -h0.y, h1.x, -h1.y};
    } else {
      const cf n0 = buf[tridx<T>(NC - k, tt)], n1 = buf[tridx<T>(NC - k, tt + 1)];
      if (zero_behind) {
        buf[tridx<T>(NC - k, tt)] = zz;
        buf[tridx<T>(NC - k, tt + 1)] = zz;
      }
      const cf w = wpre ? wpre[i] : tw_r2c[k];   // (wpre: the thread's ITEMS / NT twiddles, loaded once per kernel)
      const cf s0 = make_float2(a0.x + n0.x, a0.y - n0.y), d0 = make_float2(a0.x - n0.x, a0.y + n0.y);
      const cf s1 = make_float2(a1.x + n1.x, a1.y - n1.y), d1 = make_float2(a1.x - n1.x, a1.y + n1.y);
      const cf w0 = cmul(w, d0), w1 = cmul(w, d1);
      const vps_f4 lo = {0.5f * (s0.x + w0.y), 0.5f * (s0.y - w0.x), 0.5f * (s1.x + w1.y), 0.5f * (s1.y - w1.x)};
      const vps_f4 hi = {0.5f * (s0.x - w0.y), 0.5f * (-s0.y - w0.x), 0.5f * (s1.x - w1.y), 0.5f * (-s1.y - w1.x)};
      if constexpr (PLAIN) {
        *reinterpret_cast<vps_f4*>(&out[(long long)k * out_ok + tt]) = lo;
        *reinterpret_cast<vps_f4*>(&out[(long long)(NC - k) * out_ok + tt]) = hi;
      } else {
        __builtin_nontemporal_store(lo, reinterpret_cast<vps_f4*>(&out[(long long)k * out_ok + tt]));
        __builtin_nontemporal_store(hi, reinterpret_cast<vps_f4*>(&out[(long long)(NC - k) * out_ok + tt]));
      }
    }
  }
}

// ------------------------------------------------------------------------------
// Transposing pass (z pass: REAL=true, y pass: REAL=false).
// One workgroup transforms T lines a0..a0+T-1 of batch b and writes, for every
// output index k, the T results to T contiguous complex slots out[b][k][a0..].
// ------------------------------------------------------------------------------
// NTEMP: non-temporal loads and stores (y pass of images up to ~0.75 GB: -4..-9 % there; on larger ones +2..+5 % with
// 64-byte segments, -6 % with full 128-byte lines)
// KG: the Nyquist-plane launch of the chunked exchange (output rows shifted per destination rank, PassParams::kg) -- its own
// instantiation, so that the main passes carry no per-element division
template <int NC, int T, bool REAL, bool NTEMP = false, bool KG = false>
__global__ void __launch_bounds__(T* PlanInfo<NC>::L)
    fft_transpose_pass(const PassParams p) {
  typedef PlanInfo<NC> PI;
  constexpr int L = PI::L, RL = PI::RL, NT = T * L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* tw_lds = reinterpret_cast<cf*>(smem_raw);
  cf* buf = tw_lds + PI::TWL;
  const cf* tw = PI::TWLDS ? tw_lds : p.tw_stage;

  const int tid = threadIdx.x;
  const int t = tid / L, l = tid % L;
  const int tiles = (p.A + T - 1) / T;
  // Tiles narrower than a 128-byte cache line (T < 16): run the 16/T tiles that share output
  // lines on ONE XCD, close in time, so that its L2 merges their partial-line writes before
  // write-back.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8, observed,
  // speed only): hardware block h = 8 s + x handles logical tile G*(8*(s/G) + x) + s%G.
  unsigned bid = blockIdx.x;
  constexpr unsigned G = (T < 16) ? 16 / T : 1;
  if constexpr (G > 1) {
    const unsigned span = 8 * G;
    if (bid / span < gridDim.x / span) {   // whole groups only; the tail keeps the identity map
      const unsigned base = (bid / span) * span, h = bid % span;
      bid = base + G * (h % 8) + (h / 8);
    }
  }
  const int bo = bid / tiles;                 // batch as the output sees it
  const int a0 = (bid % tiles) * T;
  const long long b = p.bg ? (long long)(bo / p.bg) * p.bg_in + p.b_off + (long long)(bo % p.bg) * p.bg_step : bo;   // batch as the input sees it
  const long long ogap = p.bg ? (long long)(bo / p.bg) * p.bg_gap : 0;

  if constexpr (PI::TWLDS)
    for (int i = tid; i < PI::TW; i += NT) tw_lds[i] = p.tw_stage[i];

  cf v[RL];
  {
    const bool live = (a0 + t) < p.A;
    constexpr int R = PI::R0, NB = RL / R;
    if constexpr (REAL) {
      // line of 2*NC floats read as NC packed complex z[j] = x[2j] + i x[2j+1]
      const cf* src = reinterpret_cast<const cf*>(reinterpret_cast<const float*>(p.in) +
                                                   (long long)b * p.in_sb +
                                                   (long long)(a0 + t) * p.in_sa);
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
          v[m * R + r] = live ? load_stream(&src[l + L * m + r * (NC / R)]) : make_float2(0.f, 0.f);
      if (p.in_w) {
        const cf* srcw = reinterpret_cast<const cf*>(reinterpret_cast<const float*>(p.in_w) +
                                                      (long long)b * p.in_sb + (long long)(a0 + t) * p.in_sa);
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const cf w = live ? srcw[l + L * m + r * (NC / R)] : make_float2(0.f, 0.f);
            v[m * R + r] = make_float2(v[m * R + r].x * w.x, v[m * R + r].y * w.y);
          }
      }
    } else {
      const cf* src = reinterpret_cast<const cf*>(p.in) + (long long)b * p.in_sb +
                      (long long)(a0 + t) * p.in_sa;
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if constexpr (NTEMP) {
            v[m * R + r] = live ? load_stream(&src[l + L * m + r * (NC / R)]) : make_float2(0.f, 0.f);
          } else {
            v[m * R + r] = live ? src[l + L * m + r * (NC / R)] : make_float2(0.f, 0.f);
          }
        }
    }
  }
  __syncthreads();  // twiddle image visible
  // a line's lanes sit in one wave when L divides 64: stage exchanges then need no workgroup barrier
  constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
  fft_from_regs<NC, WSYNC>(v, buf + t * PI::PITCH, tw, l);
  __syncthreads();  // every lane done with the per-line buffers
  // transposed image [k][t]
#pragma unroll
  for (int i = 0; i < RL; ++i) buf[tridx<T>(out_index<NC>(l, i), t)] = v[i];
  __syncthreads();

  cf* out = reinterpret_cast<cf*>(p.out) + (long long)bo * p.out_ob + ogap + a0;
  if constexpr (!REAL) {
    const int kc = p.kcut ? p.kcut[p.kz_fixed >= 0 ? p.kz_fixed : (int)b] : NC;
    const int2 pt = p.ptab ? p.ptab[bo % p.bg] : make_int2(0, -1);   // (uniform over the workgroup)
    out += (long long)pt.x * p.out_ok;
#pragma unroll 4
    for (int i = 0; i < RL; ++i) {
      const int idx = tid + i * NT;
      const int tt = idx % T, k = idx / T;
      if (a0 + tt < p.A && min(k, NC - k) <= kc) {
        const cf val = buf[tridx<T>(k, tt)];
        long long o = (long long)packed_row(k, NC, pt.y) * p.out_ok + tt;
        if constexpr (KG) o += (long long)(k / p.kg) * p.kg_gap;
        if constexpr (NTEMP)
          store_stream(&out[o], val);
        else
          out[o] = val;
      }
    }
  } else {
    cf* nyq = reinterpret_cast<cf*>(p.out_nyq) + (long long)b * p.nyq_ob + a0;
    r2c_store_tile<NC, T, NT, true>(buf, tid, p.tw_r2c, out, p.out_ok, nyq, p.A - a0);
  }
}


// ------------------------------------------------------------------------------
// Wide transposing y pass for lines whose 16-line tile does not fit LDS (NC > 1024) or fills it (NC = 1024).
// What bounds the transposing pass of long lines is the WRITE pattern, not the transform: a plain
// transposing copy of a 2048^3 half spectrum (tools/micro/transpose_bw.hip) moves 3.9 TB/s with the
// 64-byte segments of 8-line tiles and 5.5 TB/s with full 128-byte lines -- and fft_transpose_pass<2048, 8>
// ran at exactly the former.  This kernel writes 128-byte segments with the LDS of an 8-line tile:
// a workgroup loads 2 x TG lines, transforms them one TG-line group after the other through the same
// exchange buffers (the first group's spectrum waits in registers), and then stages the transposed
// image in two k-halves [NC/2][2 TG] -- after the last radix-R stage register slot (m, r) holds
// k = l + L m + r NC/R, so r < R/2 is exactly the lower half.
// ------------------------------------------------------------------------------
// (1024-point lines: two 512-thread workgroups fit a CU's LDS, so the kernel is held to the 128 VGPRs of four waves per SIMD)
template <int NC, int TG, int L, bool NTEMP>
__global__ void __launch_bounds__(TG* L, (NC == 1024 ? 4 : 1))
    fft_transpose_pass_wide(const PassParams p) {
  typedef PlanInfo<NC> PI;
  constexpr int RL = NC / L, NT = TG * L, T = 2 * TG;
  constexpr int R = LastRadix<NC>::R;
  static_assert((R & 1) == 0 && (NC & 1) == 0, "the image is staged in two k-halves");
  static_assert(((NC / 2) * T) % NT == 0, "whole store rounds");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cf* tw_lds = reinterpret_cast<cf*>(smem_raw);
  cf* buf = tw_lds + PI::TWL;
  const cf* tw = PI::TWLDS ? tw_lds : p.tw_stage;

  const int tid = threadIdx.x;
  const unsigned tiles = (unsigned)((p.A + T - 1) / T);
  const unsigned ntiles = tiles * (unsigned)p.B;   // (the launcher checks that this fits)

  if constexpr (PI::TWLDS)
    for (int i = tid; i < PI::TW; i += NT) tw_lds[i] = p.tw_stage[i];

  // Register budget: 1024 threads leave 128 VGPRs, and two groups' values are 64 of them.  Left to itself the compiler
  // forms the LDS / global addresses of BOTH transforms and of the store loops once, ahead of the first transform, and
  // keeps ~40 of them live throughout (84 bytes per lane spilled, +4 % run time).  A thread index passed through an
  // empty asm is a new value to it: addresses derived from it are formed where they are used.
  auto fresh = [](int x) {
    asm volatile("" : "+v"(x));
    return x;
  };
  // stage-0 inputs of line `first` + (line of thread `who`) of tile `tile` (who: the thread index, possibly laundered)
  auto load_line = [&](cf (&v)[RL], unsigned tile, int who, int first) {
    constexpr int R0 = PI::R0, NB = RL / R0;
    const int bo_ = (int)(tile / tiles), a0_ = (int)(tile % tiles) * T;
    const long long b_ = p.bg ? (long long)(bo_ / p.bg) * p.bg_in + p.b_off + (long long)(bo_ % p.bg) * p.bg_step : bo_;
    const int tl = who / L, ll = who % L;
    const bool live = (a0_ + first + tl) < p.A;
    const cf* src = reinterpret_cast<const cf*>(p.in) + b_ * p.in_sb + (long long)(a0_ + first + tl) * p.in_sa;
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int r = 0; r < R0; ++r) {
        const int j = ll + L * m + r * (NC / R0);
        if constexpr (NTEMP) {   // (streaming loads: 13.7 against 14.4 ms per 2048^3 launch with plain ones)
          v[m * R0 + r] = live ? load_stream(&src[j]) : make_float2(0.f, 0.f);
        } else {
          v[m * R0 + r] = live ? src[j] : make_float2(0.f, 0.f);
        }
      }
  };
  constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
  // Persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...  ONE 1024-thread workgroup fits a CU, so nothing else
  // would overlap a tile's first loads or its last stores: the next tile's first group is requested as soon as the registers
  // of the current tile's lower half are free -- it flies behind the second half's image and stores.
  // (Placement measured and not kept: each XCD taking gridDim.x / 8 CONSECUTIVE tiles of every group of gridDim.x, so that its L2
  // owns contiguous 4 KB runs of an output row instead of every 8th 128-byte piece: 12.59 against 12.59 ms per 2048^3 launch.)
  cf v0[RL], v1[RL];
  unsigned tile = blockIdx.x;
  if (tile < ntiles) load_line(v0, tile, tid, 0);
  __syncthreads();  // twiddle image visible
  for (; tile < ntiles; tile = (ntiles - tile > gridDim.x) ? tile + gridDim.x : ntiles) {
    const int bo = (int)(tile / tiles);                 // batch as the output sees it
    const int a0 = (int)(tile % tiles) * T;
    const long long b = p.bg ? (long long)(bo / p.bg) * p.bg_in + p.b_off + (long long)(bo % p.bg) * p.bg_step : bo;   // batch as the input sees it
    const long long ogap = p.bg ? (long long)(bo / p.bg) * p.bg_gap : 0;
#if VPS_Y_EARLY_G1
    load_line(v1, tile, fresh(tid), TG);
#endif
    {
      const int tida = fresh(tid);
      fft_from_regs_l<NC, L, WSYNC>(v0, buf + (tida / L) * PI::PITCH, tw, tida % L);
    }
    // (requesting the second group ahead of the first transform: 14.1 against 12.4 ms per 2048^3 launch)
#if !VPS_Y_EARLY_G1
    load_line(v1, tile, fresh(tid), TG);
#endif
    __syncthreads();  // every lane done with the per-line buffers
    {
      const int tidb = fresh(tid);
      fft_from_regs_l<NC, L, WSYNC>(v1, buf + (tidb / L) * PI::PITCH, tw, tidb % L);
    }

    const int2 pt = p.ptab ? p.ptab[bo % p.bg] : make_int2(0, -1);   // (uniform over the workgroup)
    cf* out = reinterpret_cast<cf*>(p.out) + (long long)bo * p.out_ob + ogap + (long long)pt.x * p.out_ok + a0;
    const int kc = p.kcut ? p.kcut[p.kz_fixed >= 0 ? p.kz_fixed : (int)b] : NC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();  // exchange buffers / the previous half image are free
      const int tid2 = fresh(tid);
#pragma unroll
      for (int i = 0; i < RL; ++i) {
        if (((i % R) >= R / 2) == (h == 1)) {
          const int k = out_index_l<NC, L>(tid2 % L, i) - h * (NC / 2);
          buf[tridx<T>(k, tid2 / L)] = v0[i];
          buf[tridx<T>(k, tid2 / L + TG)] = v1[i];
        }
      }
      // the lower halves are in the image, the upper halves just went: v0 takes the next tile's first group
      // (requested after the barrier instead: 12.96 against 12.43 ms)
      if (h == 1 && ntiles - tile > gridDim.x) load_line(v0, tile + gridDim.x, fresh(tid), 0);
      __syncthreads();
      constexpr int IT = (NC / 2) * T / NT;
#if VPS_Y_ST16
      // 16-byte stores: a lane owns the lines (tt, tt + 1) of a row, a 128-byte segment leaves as eight dwordx4 stores instead of
      // sixteen dwordx2 (the pencil kernel's epilogue gained 7 % from the same change: these epilogues are bound by store ISSUE)
      if (((p.out_ok | p.A) & 1) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0) {   // (uniform)
        static_assert(IT % 2 == 0, "whole rounds of line pairs");
#pragma unroll 4
        for (int i = 0; i < IT / 2; ++i) {
          const int idx = tid2 + i * NT;
          const int tt = (idx % (T / 2)) * 2, kk = idx / (T / 2), k = kk + h * (NC / 2);
          if (a0 + tt < p.A && min(k, NC - k) <= kc) {
            const cf va = buf[tridx<T>(kk, tt)], vb = buf[tridx<T>(kk, tt + 1)];
            const long long o = (long long)packed_row(k, NC, pt.y) * p.out_ok + tt;
            const vps_f4 val = {va.x, va.y, vb.x, vb.y};
            if constexpr (NTEMP)
              __builtin_nontemporal_store(val, reinterpret_cast<vps_f4*>(&out[o]));
            else
              *reinterpret_cast<vps_f4*>(&out[o]) = val;
          }
        }
        continue;
      }
#endif
#pragma unroll 4
      for (int i = 0; i < IT; ++i) {
        const int idx = tid2 + i * NT;
        const int tt = idx % T, kk = idx / T, k = kk + h * (NC / 2);
        if (a0 + tt < p.A && min(k, NC - k) <= kc) {
          const cf val = buf[tridx<T>(kk, tt)];
          const long long o = (long long)packed_row(k, NC, pt.y) * p.out_ok + tt;
          if constexpr (NTEMP)
            store_stream(&out[o], val);
          else
            out[o] = val;
        }
      }
    }
    __syncthreads();  // the image is consumed before the next tile's exchanges overwrite it
  }
}

// ------------------------------------------------------------------------------
// Fused deposit + field algebra + z pass ("pencil" kernel).  A pencil is the TP z-lines
// (x, y0..y0+TP-1, all z) that one z-pass tile transforms.  The deposit stage has sorted
// the particle records {cell-in-pencil, rho vx, rho vy, rho vz, rho} by pencil
// (deposit.hip), so one workgroup accumulates rho of its pencil in LDS (vps_lds_add), keeps
// 1/rho of its own stage-0 cells in registers, then per component accumulates rho*v in the same
// LDS region, forms v = rho v / rho (or p = rho v Lcell^3) as it loads the FFT's stage-0 inputs,
// transforms and writes B[x][kz][y] -- the real-space grid is never written to or read from
// HBM, and the accumulator, exchange buffers and transposed image all share one LDS region.
// ------------------------------------------------------------------------------
struct PencilParams {
  const unsigned* records;   // (1 + 4) 32-bit words per particle, sorted by pencil
  const unsigned* start;     // [npencils + 1]
  int N, nx, nby;            // grid, slab rows, pencils per x row (N / TP)
  unsigned npencils;         // pencils of the launch (the persistent form walks them: slot, slot + gridDim.x, ...)
  int ncomp;
  int chan[3];               // record channel (0..2) feeding component c
  int divide;                // 1: v = q / rho (0 where rho == 0);  0: p = q * vol
  int energy;                // 1: ONE output field E = vol * sum_c q_c^2 / rho (= mass * |v|^2, interp.py:546)
  int with_energy;           // momentum launch (divide = 0, three components) that ALSO writes the energy field as out[3]: the
                             // cell totals of rho v_c its rounds accumulate are what E is made of -- one more round (rho) and
                             // one more transform instead of a launch of its own with four rounds and a start of its own
  float vol;
  float* side;               // [records] one float per record, for the records a workgroup cannot keep in registers
  cf* out[4];                // B_c[x][kz][y]
  cf* nyq[4];                // BN_c[x][y]
  const cf* tw_stage;
  const cf* tw_r2c;
};

// tuning knobs (measured at 512^3 / 1024^3 / 2048^3): 4 waves/SIMD needs <= 128 VGPRs
#ifndef VPS_PENCIL_MINW
#define VPS_PENCIL_MINW 4
#endif
#ifndef VPS_PENCIL_MINW_LONG
#define VPS_PENCIL_MINW_LONG 4
#endif
template <int NC>
constexpr int pencil_min_waves() {
  return NC >= 1024 ? VPS_PENCIL_MINW_LONG : VPS_PENCIL_MINW;
}

// Lanes per line: the plan's.  (While the kernel kept 1/rho -- and for energy the sums of squares -- per CELL in
// registers, 2048-cell lines did not fit the 128 VGPRs of four waves per SIMD: it ran them on half the lanes, two waves
// per SIMD, 83 ms per C4 step of which 65 ms were on-chip work that nothing overlapped.)
#ifndef VPS_PENCIL_HALF_LANES_LONG
#define VPS_PENCIL_HALF_LANES_LONG 0
#endif
template <int NC>
constexpr int pencil_lanes() {
  return (NC >= 1024 && VPS_PENCIL_HALF_LANES_LONG) ? PlanInfo<NC>::L / 2 : PlanInfo<NC>::L;
}

// Epilogue of 8-line pencils (2048- and 4096-cell lines; measured at C4, ms per launch, vector / energy): 8-byte streaming
// stores 29.5 / 11.5 (rounds 2-3); 8-byte plain 30.6 / 10.95; 16-byte streaming 27.4 / 12.4; 16-byte plain 27.0 / 10.2 -- the
// epilogue is bound by store ISSUE (half the instructions with dwordx4); the energy launch, whose workgroups live for one
// component only, merges its half lines in L2 with the partner pencil's when they are not marked streaming (PMC: 44.9 -> 35.1 GB
// written for 34.4 GB of output).
#ifndef VPS_ST16_MODE
#define VPS_ST16_MODE 3   // bit 0: vector launches, bit 1: the energy launch
#endif
#ifndef VPS_PLAIN_MODE
#define VPS_PLAIN_MODE 2   // (vector launches with plain 16-byte stores: 27.0 ms on one box, 30.7 on two others, against 26.5 - 27.6 with
#endif                     //  streaming stores everywhere; PMC 113.9 against 117.2 GB written for 103.2 GB of output: not worth the risk)
#ifndef VPS_SHARED_E_PLAIN
#define VPS_SHARED_E_PLAIN 1
#endif
#ifndef VPS_ST16_ALL
#define VPS_ST16_ALL 0    // 1: 16-byte stores for 16-line pencils too (experiment)
#endif
// 1024-point lines on the plan's 64 lanes (the 2048^3 grid): the twiddles of both later stages depend on the lane alone and
// CAN stay in registers for the whole kernel (7 + 7 values; the second butterfly of stage 2 by constant 16th-root rotations:
// 28 of the 32 LDS reads per lane and transform that are not data) -- measured slower here, unlike in the x pass, see below.
#ifndef VPS_PENCIL_R2C_PRELOAD
#define VPS_PENCIL_R2C_PRELOAD 0
#endif
#ifndef VPS_PENCIL_ZERO_BEHIND
#define VPS_PENCIL_ZERO_BEHIND 0   // the epilogue of a component clears the image behind itself: no dense zero-fill for the next one
                                   // (measured at C4: vector launch 28.3 against 27.2 ms -- the epilogue is where the kernel is slowest)
#endif
#ifndef VPS_PENCIL_PERSIST
#define VPS_PENCIL_PERSIST 0   // 1: persistent workgroups with the next pencil's records prefetched (see the kernel).  Measured at C4,
                               // twice (rounds 2 and 4, the second time with the bounds, cells and first values of the next pencil
                               // requested behind the last transform and the twiddles staged once): vector launch 27.3 against 26.5 ms,
                               // energy 13.2 against 10.3 -- with or without holding every other workgroup back.  What the hardware
                               // dispatcher gives for free -- workgroups of a CU out of step with each other -- is worth more than the
                               // two dependent trips to HBM a fresh workgroup starts with.
#endif
#ifndef VPS_PENCIL_STAGGER
#define VPS_PENCIL_STAGGER 0   // PERSIST: s_sleep(127) rounds by which the second workgroup of every CU starts late
#endif
#ifndef VPS_PENCIL_TWREG
#define VPS_PENCIL_TWREG 0     // bit 0: stage 1, bit 1: stage 2.  Measured at C4 (vector / energy launch, ms): 0: 26.5 / 10.4;
                               // 1: 27.8 / 11.8; 2: 27.2 / 12.1; 3 (7 registers spilled): 27.1 / 13.3 -- the LDS reads they save
                               // cost less than the registers they take (80 -> 104..128 VGPRs): off
#endif
template <int NC>
constexpr int pencil_twreg() {
  typedef PlanInfo<NC> PI;
  return (NC == 1024 && pencil_lanes<NC>() == 64 && PI::R1 == 8 && PI::R2 == 8) ? (VPS_PENCIL_TWREG) : 0;
}
template <int NC, int TP, bool ENERGY = false>
__global__ void __launch_bounds__(TP* pencil_lanes<NC>(), pencil_min_waves<NC>()) pencil_fft_z_kernel(const PencilParams p) {
  typedef PlanInfo<NC> PI;
  constexpr int L = pencil_lanes<NC>(), RL = NC / L, NT = TP * L, N = 2 * NC;
  constexpr int ACC = TP * N;                       // floats of one accumulator
  constexpr int LINES = TP * PI::PITCH * 2;         // floats of the exchange buffers
  constexpr int SHARED = (ACC > LINES ? ACC : LINES);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // ONE LDS region serves, in turn, as the rho accumulator, each rho*v accumulator and the FFT's
  // exchange / transposed-image buffer.
  float* acc = reinterpret_cast<float*>(smem_raw);
  cf* buf = reinterpret_cast<cf*>(acc);
  cf* tw_lds = reinterpret_cast<cf*>(acc + SHARED);
  constexpr int TWR = pencil_twreg<NC>();
#ifdef VPS_PENCIL_TW_GLOBAL
  constexpr bool TWL = false;
#else
  constexpr bool TWL = PI::TWLDS && TWR != 3;
#endif
  const cf* tw = TWL ? tw_lds : p.tw_stage;

  const int tid = threadIdx.x;
  const int t = tid / L, l = tid % L;
  // Pencils narrower than a 128-byte output line (TP < 16): the 16/TP pencils that share output lines are consecutive
  // pencil numbers; run them on ONE XCD, close in time, so that its L2 merges their partial-line writes (same map as
  // fft_transpose_pass: hardware block h = 8 s + x handles logical pencil G*(8*(s/G) + x) + s%G)
  constexpr unsigned GP = (TP < 16) ? 16 / TP : 1;
  auto map_slot = [&](unsigned slot) -> unsigned {
    if constexpr (GP > 1) {
      const unsigned span = 8 * GP;
      if (slot / span < p.npencils / span) {   // whole groups only; the tail keeps the identity map
        const unsigned base = (slot / span) * span, h = slot % span;
        return base + GP * (h % 8) + (h / 8);
      }
    }
    return slot;
  };
  // PERSIST (VPS_PENCIL_PERSIST): the workgroup walks pencil slots blockIdx.x, + gridDim.x, ... -- the twiddles are staged
  // once, and the bucket bounds, cells and first values of the NEXT pencil are requested behind the last component's
  // transform, so that a pencil no longer begins with two dependent trips to HBM (bounds, then records) and a table copy.
  constexpr bool PERSIST = VPS_PENCIL_PERSIST != 0;
  unsigned slot = blockIdx.x;
  unsigned pencil = map_slot(slot);
  unsigned s = p.start[pencil], e = p.start[pencil + 1];
  unsigned s_n = 0, e_n = 0, pencil_n = 0;
  if constexpr (TWL)
    for (int i = tid; i < PI::TW; i += NT) tw_lds[i] = p.tw_stage[i];
  // the real-to-complex twiddles of this thread's rows in the 16-byte epilogue: k = tid / (TP / 2) + i NT / (TP / 2), the same
  // for every component (and pencil) -- loaded once instead of one dependent global load per component ahead of the stores
  constexpr int WHO0 = ENERGY ? 2 : 1;
  constexpr bool ST16_0 = (VPS_ST16_MODE & WHO0) && TP >= 8 && TP <= 16 && (VPS_ST16_ALL || TP == 8) && (NC & 1) == 0 && ((NC / 2) * (TP / 2)) % NT == 0;
  constexpr int NW = (ST16_0 && VPS_PENCIL_R2C_PRELOAD) ? ((NC / 2) * (TP / 2)) / NT : 0;
  cf wr2c[NW > 0 ? NW : 1];
  if constexpr (NW > 0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) wr2c[i] = p.tw_r2c[(tid + i * NT) / (TP / 2)];
  }
  cf twr1[(TWR & 1) ? (PI::R1 - 1) : 1], twr2[(TWR & 2) ? (PI::R2 - 1) : 1];
  if constexpr ((TWR & 1) != 0) load_stage_twiddles<NC, L, RL, PI::R1, PI::NS1>(twr1, p.tw_stage, l);
  if constexpr ((TWR & 2) != 0) load_stage_twiddles<NC, L, RL, PI::R2, PI::NS2>(twr2, p.tw_stage + PI::TW1, l);
  // What a CELL needs besides the sums -- 1/rho (velocity), the running sum of (rho v_c)^2 (energy) -- is kept per RECORD:
  // every record of a cell reads the cell's total from the accumulator and carries the same value.  A per-cell table would
  // be RL register pairs per lane next to the RL transform registers (it was: the kernel then fit four waves per SIMD only
  // up to 1024-cell lines, and not at all for energy); per record it is KR registers, whatever the line length.
  // The cells of the first KR*NT records of the bucket stay in registers; the value each round adds (rho, then rho v_c) is
  // fetched one round ahead, so the loads fly behind the previous round's FFT.  Only unusually full pencils read records
  // inside a round (tail loops below); their per-record value lives in p.side[record].
  // register-resident record groups: enough for a typical bucket (~1.2 x mean occupancy at the bench
  // densities) -- 3 x 256 threads at 512^3, 2 x 512 at 1024^3 (measured optimum each)
#ifdef VPS_PENCIL_KR
  constexpr int KR = VPS_PENCIL_KR;
#else
  constexpr int KR = NT >= 512 ? 2 : (NT >= 256 ? 3 : 4);
#endif
  unsigned rloc[KR];
  float rval[KR];
  float rrec[KR];   // 1/rho of the record's cell (velocity) / sum over components of (rho v_c)^2 of its cell (ENERGY)
  unsigned rloc_n[PERSIST ? KR : 1];
  float rval_n[PERSIST ? KR : 1];
  auto fetch = [&](int word) {   // record word 1..3: rho v_c, 4: rho
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const unsigned j = s + tid + k * NT;
      if (j < e) rval[k] = __uint_as_float(p.records[(size_t)j * 5 + word]);
    }
  };
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const unsigned j = s + tid + k * NT;
    rloc[k] = (j < e) ? p.records[(size_t)j * 5] : 0xffffffffu;
  }
  const bool divide = !ENERGY && p.divide;
  fetch(divide ? 4 : 1 + p.chan[0]);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int R0 = PI::R0, NB0 = RL / R0;
#if VPS_PENCIL_STAGGER
  // the two workgroups of a CU start together and would stay in step (both accumulating, both storing): hold every other one back
  if (PERSIST && (blockIdx.x / (gridDim.x / 2)) != 0)
    for (int i = 0; i < VPS_PENCIL_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
 for (;;) {   // (one pass per pencil; a single pass unless PERSIST)
  if constexpr (PERSIST) {
    __syncthreads();   // the previous pencil's transposed image is consumed
    const unsigned nxt = slot + gridDim.x;
    if (nxt < p.npencils) {     // bounds of the next pencil: needed at the END of this pass
      pencil_n = map_slot(nxt);
      s_n = p.start[pencil_n];
      e_n = p.start[pencil_n + 1];
    }
  }
  const int x = pencil / p.nby, y0 = (pencil % p.nby) * TP;
  // more than two particles per cell on average: hot cells are likely, take the native atomics (vps_lds_add)
#ifdef VPS_PENCIL_NATIVE_ADD
  const bool crowded = true;
#else
  const bool crowded = (e - s) > 2u * (unsigned)ACC;
#endif
#pragma unroll
  for (int k = 0; k < KR; ++k) rrec[k] = 1.f;
  const unsigned tail0 = s + tid + KR * NT;   // this thread's first record beyond the register-resident ones
  if (divide) {
    for (int i = tid; i < ACC / 4; i += NT) reinterpret_cast<float4*>(acc)[i] = zero4;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (rloc[k] != 0xffffffffu) vps_lds_add(&acc[rloc[k]], rval[k], crowded);
    fetch(1 + p.chan[0]);
    for (unsigned j = tail0; j < e; j += NT) {
      const unsigned* rec = p.records + (size_t)j * 5;
      vps_lds_add(&acc[rec[0]], __uint_as_float(rec[4]), crowded);
    }
    __syncthreads();
    // one reciprocal per record; empty-rho cells give 0 (the NaN->0 rule of interp.py:329-331)
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (rloc[k] != 0xffffffffu) {
        const float r = acc[rloc[k]];
        rrec[k] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
      }
    for (unsigned j = tail0; j < e; j += NT) {
      const float r = acc[p.records[(size_t)j * 5]];
      p.side[j] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
    }
  }

  // ENERGY: the three rho*v_c rounds add up q_c^2 per record, a FOURTH round accumulates rho and finishes
  // E = vol * sum / rho in the cells that hold records (all others stay 0)
  const bool we = !ENERGY && p.with_energy;     // (uniform) momentum launch that also produces the energy field
  const int nround = (ENERGY || we) ? p.ncomp + 1 : p.ncomp;
  for (int c = 0; c < nround; ++c) {
    // opaque copies: keeps the compiler from hoisting ~50 loop-invariant LDS addresses out of the
    // component loop (they cost more registers than they save instructions, and an occupancy step)
    int lc = l, tc = t, tidc = tid;
#ifndef VPS_PENCIL_NO_OPAQUE
    asm volatile("" : "+v"(lc), "+v"(tc), "+v"(tidc));
#endif
    lc &= L - 1;   // give the value ranges back to the compiler (address folding needs them)
    tc &= TP - 1;
    tidc &= NT - 1;
    __syncthreads();   // rho / previous component's transposed image fully consumed
    // The accumulator still holds the previous round's totals -- rho ahead of the first velocity component, a component ahead
    // of the next one (ENERGY) -- and is zero in every cell without a record: the records clear their own cells (a few hundred
    // 4-byte writes) instead of a dense fill of the whole region (64 - 128 KB at the LDS write rate: 830 - 1500 clk per round).
    // After a transform the region is FFT scratch and needs the dense fill.
    const bool sparse_clear = ENERGY ? (c > 0) : (divide && c == 0);
    // ZB: the previous component's epilogue left the region all zero behind itself (r2c_store_tile16: zero_behind)
    constexpr int WHO_ = ENERGY ? 2 : 1;
    constexpr bool ZB = VPS_PENCIL_ZERO_BEHIND && !ENERGY && (VPS_ST16_MODE & WHO_) && TP == 8 && (NC & 1) == 0 &&
                        ((NC / 2) * (TP / 2)) % NT == 0 && ACC * 4 <= NC * TP * 8;
    const bool prezeroed = ZB && c > 0;
    if (prezeroed) {
    } else if (sparse_clear) {
#pragma unroll
      for (int k = 0; k < KR; ++k)
        if (rloc[k] != 0xffffffffu) acc[rloc[k]] = 0.f;
      for (unsigned j = tail0; j < e; j += NT) acc[p.records[(size_t)j * 5]] = 0.f;
    } else {
#ifndef VPS_ABL_NOZERO
      for (int i = tid; i < ACC / 4; i += NT) reinterpret_cast<float4*>(acc)[i] = zero4;
#endif
    }
    if (!prezeroed) __syncthreads();
    const bool rho_round = (ENERGY || we) && c == p.ncomp;
    const int word = rho_round ? 4 : 1 + p.chan[c < p.ncomp ? c : 0];
    // velocity: each term is divided by its cell's rho as it is added -- sum_k (q_k / rho) for the reference's
    // (sum_k q_k) / rho: a different rounding of the same value, within the float32 accumulation noise of the sums
#ifndef VPS_ABL_NOSCATTER
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (rloc[k] != 0xffffffffu) vps_lds_add(&acc[rloc[k]], divide ? rval[k] * rrec[k] : rval[k], crowded);
#endif
    if (c + 1 < nround) fetch(((ENERGY || we) && c + 1 == p.ncomp) ? 4 : 1 + p.chan[c + 1 < p.ncomp ? c + 1 : 0]);
    for (unsigned j = tail0; j < e; j += NT) {
      const unsigned* rec = p.records + (size_t)j * 5;
      const float val = __uint_as_float(rec[word]);
      vps_lds_add(&acc[rec[0]], divide ? val * p.side[j] : val, crowded);
    }
    __syncthreads();
    if (ENERGY || we) {
      if (!rho_round) {
        // the cell totals of this component, squared, per record
#pragma unroll
        for (int k = 0; k < KR; ++k)
          if (rloc[k] != 0xffffffffu) {
            const float q = acc[rloc[k]];
            rrec[k] = (c == 0) ? q * q : rrec[k] + q * q;
          }
        for (unsigned j = tail0; j < e; j += NT) {
          const float q = acc[p.records[(size_t)j * 5]];
          p.side[j] = (c == 0) ? q * q : p.side[j] + q * q;
        }
        if constexpr (ENERGY) continue;   // (the barrier at the top of the loop protects the accumulator)
        // (with_energy: the accumulator goes on into this component's own transform)
      } else {
        // the accumulator holds rho: E = sum * (1 / rho) * vol where there is mass, 0 elsewhere -- every record writes its cell's
        // value (records of one cell write the same bits), after everybody has read rho
#pragma unroll
        for (int k = 0; k < KR; ++k)
          if (rloc[k] != 0xffffffffu) {
            const float r = acc[rloc[k]];
            rrec[k] = r != 0.f ? rrec[k] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
          }
        for (unsigned j = tail0; j < e; j += NT) {
          const float r = acc[p.records[(size_t)j * 5]];
          p.side[j] = r != 0.f ? p.side[j] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KR; ++k)
          if (rloc[k] != 0xffffffffu) acc[rloc[k]] = rrec[k];
        for (unsigned j = tail0; j < e; j += NT) acc[p.records[(size_t)j * 5]] = p.side[j];
        __syncthreads();
      }
    }
    // stage-0 inputs straight from the accumulator: z[j] = f[2j] + i f[2j+1]
    cf v[RL];
    {
      const float* q = acc + tc * N;
      const float sc = (ENERGY || divide || rho_round) ? 1.f : p.vol;
#pragma unroll
      for (int m = 0; m < NB0; ++m)
#pragma unroll
        for (int rr = 0; rr < R0; ++rr) {
          const int j = lc + L * m + rr * (NC / R0);
          const float2 qq = *reinterpret_cast<const float2*>(q + 2 * j);
          v[m * R0 + rr] = make_float2(qq.x * sc, qq.y * sc);
        }
    }
    __syncthreads();   // accumulator of this component consumed: its memory becomes FFT scratch
    constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
#ifdef VPS_ABL_NOFFT
#pragma unroll
    for (int i = 0; i < RL; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));
    if (v[0].x != 1.2345e30f) continue;
#endif
    if constexpr (PERSIST) {
      // last round: the records of this pencil are done with -- request the next pencil's cells and first values now, they
      // land behind this transform and its stores
      if (c + 1 == nround && slot + gridDim.x < p.npencils) {
        const int w0 = divide ? 4 : 1 + p.chan[0];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const unsigned j = s_n + tid + k * NT;
          rloc_n[k] = (j < e_n) ? p.records[(size_t)j * 5] : 0xffffffffu;
          if (j < e_n) rval_n[k] = __uint_as_float(p.records[(size_t)j * 5 + w0]);
        }
      }
    }
    fft_from_regs_l<NC, L, WSYNC, TWR>(v, buf + tc * PI::PITCH, tw, lc, twr2, twr1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RL; ++i) buf[tridx<TP>(out_index_l<NC, L>(lc, i), tc)] = v[i];
    __syncthreads();
    const int oc = ENERGY ? 0 : c;
    cf* out = p.out[oc] + (long long)x * NC * N + y0;
    cf* nyq = p.nyq[oc] + (long long)x * N + y0;
    // epilogue flavour of 8-line pencils (64-byte output segments): 16-byte stores and / or plain instead of streaming stores,
    // per kernel (bit 0: vector launches, bit 1: the energy launch) -- measured at C4, DESIGN.md section 7
    constexpr int WHO = ENERGY ? 2 : 1;
    constexpr bool ST16 = (VPS_ST16_MODE & WHO) && TP >= 8 && TP <= 16 && (VPS_ST16_ALL || TP == 8) && (NC & 1) == 0 && ((NC / 2) * (TP / 2)) % NT == 0;
    constexpr bool PLAIN = (VPS_PLAIN_MODE & WHO) && TP < 16;
    if constexpr (ST16) {
      // (the energy field of a with_energy launch is the workgroup's LAST component, like the one field of an energy launch:
      //  plain stores there too, so that the half lines of partner pencils meet in L2 -- VPS_SHARED_E_PLAIN)
      if (VPS_SHARED_E_PLAIN && !ENERGY && !PLAIN && rho_round)
        r2c_store_tile16<NC, TP, NT, true>(buf, tidc, p.tw_r2c, out, N, nyq, false, nullptr);
      else
        r2c_store_tile16<NC, TP, NT, PLAIN>(buf, tidc, p.tw_r2c, out, N, nyq, ZB && c + 1 < nround, NW > 0 ? wr2c : nullptr);
    } else
      r2c_store_tile<NC, TP, NT, false, PLAIN>(buf, tidc, p.tw_r2c, out, N, nyq, TP);
  }
  if constexpr (!PERSIST) break;
  slot += gridDim.x;
  if (slot >= p.npencils) break;
  pencil = pencil_n;
  s = s_n;
  e = e_n;
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    rloc[k] = rloc_n[PERSIST ? k : 0];
    rval[k] = rval_n[PERSIST ? k : 0];
  }
 }
}

// ------------------------------------------------------------------------------
// Epilogue of the kernels that transform a pencil of TP lines in two halves of TP/2 lines (split pencil, pencil pair): thread
// (t, l) holds the spectrum of line t in `vpark` and of line t + TP/2 in `v`.  The transposed image of all TP lines does not
// fit next to nothing else: it is staged in two halves of the mode PAIRS -- the real-to-complex step needs Z[k] with Z[NC - k]:
// after the last radix-R stage register slot (m, r) holds k = l + L m + r NC/R, so r < R/4 and r >= 3R/4 are exactly the modes
// k < NC/4 and k >= 3 NC/4 (the pairs of the first half), the middle slots the second.  Z[3 NC/4] pairs with Z[NC/4] of the
// OTHER half image: the first phase leaves it in `edge` (TP values).  Row kk of a half image: k < NC/4 -> kk = k;
// k >= 3NC/4 -> kk = k - NC/2 (first half); NC/4 <= k < 3NC/4 -> kk = k - NC/4 (second half).  16-byte stores: two lines per lane.
// ------------------------------------------------------------------------------
template <int NC, int TP, int L, bool STREAM>
__device__ __forceinline__ void two_half_image_store(cf* buf, cf* edge, const cf (&vpark)[NC / L], const cf (&v)[NC / L], int l, int t,
                                                     int tid, cf* out, cf* nyq, const cf* __restrict__ tw_r2c) {
  constexpr int RL = NC / L, TH = TP / 2, NT = TH * L, N = 2 * NC;
  constexpr int R = LastRadix<NC>::R;
  static_assert((R % 4) == 0 && (NC % 4) == 0, "image in two halves of the mode pairs");
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int lc = l, tc = t, tidc = tid;
    asm volatile("" : "+v"(lc), "+v"(tc), "+v"(tidc));
    lc &= L - 1;
    tc &= TH - 1;
    tidc &= NT - 1;
    __syncthreads();   // exchange buffers / the previous half image are free
#pragma unroll
    for (int i = 0; i < RL; ++i) {
      const int r = i % R;
      const bool outer = (r < R / 4) || (r >= 3 * R / 4);
      if (outer == (g == 0)) {
        const int k = out_index_l<NC, L>(lc, i);
        const int kk = (g == 0) ? ((r < R / 4) ? k : k - NC / 2) : k - NC / 4;
        buf[tridx<TP>(kk, tc)] = vpark[i];
        buf[tridx<TP>(kk, tc + TH)] = v[i];
        if (g == 0 && i == 3 * R / 4 && lc == 0) {   // slot (m = 0, r = 3R/4) of lane 0: mode 3 NC/4
          edge[tc] = vpark[i];
          edge[tc + TH] = v[i];
        }
      }
    }
    __syncthreads();
    // pairs of this half: k = 0 .. NC/4 - 1 with NC - k (g = 0; k = 0: the self-paired modes 0 and NC/2 -- NC/2 sits in
    // the OTHER half image, so g = 0 writes X[0] and the Nyquist plane, g = 1 writes X[NC/2] from its own row);
    // k = NC/4 .. NC/2 - 1 with NC - k, and the self-paired NC/2 (g = 1)
    constexpr int H2 = TP / 2, ITEMS = (NC / 4) * H2;
    static_assert(ITEMS % NT == 0, "whole rounds");
#pragma unroll 4
    for (int i = 0; i < ITEMS / NT; ++i) {
      const int idx = tidc + i * NT;
      const int tt = (idx % H2) * 2, kq = idx / H2;           // kq = 0 .. NC/4 - 1
      const int k = (g == 0) ? kq : kq + NC / 4;              // the smaller mode of the pair
      const int rowk = (g == 0) ? k : k - NC / 4;             // its row in this half image
      const int rown = (g == 0) ? (NC - k) - NC / 2 : (NC - k) - NC / 4;   // row of NC - k (k > 0)
      const cf a0 = buf[tridx<TP>(rowk, tt)], a1 = buf[tridx<TP>(rowk, tt + 1)];
      if (g == 0 && k == 0) {
        *reinterpret_cast<vps_f4*>(&out[tt]) = vps_f4{a0.x + a0.y, 0.f, a1.x + a1.y, 0.f};
        *reinterpret_cast<vps_f4*>(&nyq[tt]) = vps_f4{a0.x - a0.y, 0.f, a1.x - a1.y, 0.f};
      } else {
        if (g == 1 && k == NC / 4) {
          // (row NC/2 - NC/4 of this half image holds mode NC/2, which pairs with itself)
          const cf h0 = buf[tridx<TP>(NC / 2 - NC / 4, tt)], h1 = buf[tridx<TP>(NC / 2 - NC / 4, tt + 1)];
          *reinterpret_cast<vps_f4*>(&out[(long long)(NC / 2) * N + tt]) = vps_f4{h0.x, -h0.y, h1.x, -h1.y};
        }
        const bool at_edge = (g == 1 && k == NC / 4);      // partner 3 NC/4 was left in `edge` by the first phase
        const cf n0 = at_edge ? edge[tt] : buf[tridx<TP>(rown, tt)], n1 = at_edge ? edge[tt + 1] : buf[tridx<TP>(rown, tt + 1)];
        const cf w = tw_r2c[k];
        const cf s0 = make_float2(a0.x + n0.x, a0.y - n0.y), d0 = make_float2(a0.x - n0.x, a0.y + n0.y);
        const cf s1 = make_float2(a1.x + n1.x, a1.y - n1.y), d1 = make_float2(a1.x - n1.x, a1.y + n1.y);
        const cf w0 = cmul(w, d0), w1 = cmul(w, d1);
        const vps_f4 lo = {0.5f * (s0.x + w0.y), 0.5f * (s0.y - w0.x), 0.5f * (s1.x + w1.y), 0.5f * (s1.y - w1.x)};
        const vps_f4 hi = {0.5f * (s0.x - w0.y), 0.5f * (-s0.y - w0.x), 0.5f * (s1.x - w1.y), 0.5f * (-s1.y - w1.x)};
        if constexpr (!STREAM) {
          *reinterpret_cast<vps_f4*>(&out[(long long)k * N + tt]) = lo;
          *reinterpret_cast<vps_f4*>(&out[(long long)(NC - k) * N + tt]) = hi;
        } else {
          __builtin_nontemporal_store(lo, reinterpret_cast<vps_f4*>(&out[(long long)k * N + tt]));
          __builtin_nontemporal_store(hi, reinterpret_cast<vps_f4*>(&out[(long long)(NC - k) * N + tt]));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------
// Split pencil: the same pencil (TP = 8 y-lines) on HALF the threads -- a workgroup runs the deposit rounds and the transform of
// lines 0..3, parks their spectrum in registers (RL values per lane), does the same for lines 4..7, and only then stages the
// transposed image of all eight lines and stores it.  The LDS region is that of FOUR lines: at 4096-cell lines 67.6 KB instead
// of 135 KB (+ 16 KB of twiddles, now read through L2 as the 4096-point y pass does), so TWO workgroups of 512 threads share a
// CU where one of 1024 ran alone with every phase exposed (pencil kernel at 0.26 of peak at N = 4096).
// The image of eight lines no longer fits either: it is staged in two halves of the mode PAIRS -- the real-to-complex step needs
// Z[k] with Z[NC - k]: after the last radix-R stage register slot (m, r) holds k = l + L m + r NC/R, so r < R/4 and
// r >= 3R/4 are exactly the modes k < NC/4 and k >= 3 NC/4 (the pairs of the first half), the middle slots the second.
// Every record is looked at once per half (a record belongs to the half that holds its line); everything else -- per-record
// 1/rho and energy sums, sparse clears, CAS-first LDS adds, the 16-byte epilogue -- is the pencil kernel's.
// ------------------------------------------------------------------------------
template <int NC, int TP, bool ENERGY>
__global__ void __launch_bounds__((TP / 2) * pencil_lanes<NC>(), 4) pencil_split_fft_z_kernel(const PencilParams p) {
  typedef PlanInfo<NC> PI;
  constexpr int L = pencil_lanes<NC>(), RL = NC / L, TH = TP / 2, NT = TH * L, N = 2 * NC;
  constexpr int R = LastRadix<NC>::R;
  static_assert(TP == 8 && (R % 4) == 0 && (NC % 4) == 0, "eight lines in two halves, image in two halves of the mode pairs");
  constexpr int ACC = TH * N;                       // floats of one accumulator (four lines)
  constexpr int LINES = TH * PI::PITCH * 2;         // floats of the exchange buffers
  constexpr int IMG = (NC / 2) * TP * 2;            // floats of half an image: NC/2 modes x 8 lines
  constexpr int SHARED = (ACC > LINES ? (ACC > IMG ? ACC : IMG) : (LINES > IMG ? LINES : IMG));
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* acc = reinterpret_cast<float*>(smem_raw);
  cf* buf = reinterpret_cast<cf*>(acc);
  // Z[3 NC/4] of the eight lines: its partner Z[NC/4] sits in the OTHER half image (slot r = R/4 against r = 3R/4), so the first
  // image phase leaves it here for the second
  cf* edge = reinterpret_cast<cf*>(acc + SHARED);
  const cf* tw = p.tw_stage;                        // (through L2: the region is the whole of the workgroup's LDS)

  const int tid = threadIdx.x;
  const int t = tid / L, l = tid % L;
  unsigned pencil = blockIdx.x;
  {   // the two pencils that complete a 128-byte output line on ONE XCD, close in time (as in pencil_fft_z_kernel)
    constexpr unsigned GP = 16 / TP, span = 8 * GP;
    if (pencil / span < p.npencils / span) {
      const unsigned base = (pencil / span) * span, h = pencil % span;
      pencil = base + GP * (h % 8) + (h / 8);
    }
  }
  const int x = pencil / p.nby, y0 = (pencil % p.nby) * TP;
  const unsigned s = p.start[pencil], e = p.start[pencil + 1];
  const bool crowded = (e - s) > 4u * (unsigned)ACC;
  constexpr int KR = 2;      // register-resident record groups: 2 x 256 / 2 x 512 threads hold a typical bucket of EIGHT lines
  constexpr int NW_ = ENERGY ? 4 : 3;
  unsigned rloc[KR];         // cell inside the pencil: line * N + z
  float rw[KR][NW_];         // the record's values [rho v_x, rho v_y, rho v_z (, rho)]: every round of both halves reads them here
  float rrec[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const unsigned j = s + tid + k * NT;
    rloc[k] = 0xffffffffu;
    rrec[k] = 1.f;
    if (j < e) {
      const unsigned* rec = p.records + (size_t)j * 5;
      rloc[k] = rec[0];
#pragma unroll
      for (int w_ = 0; w_ < NW_; ++w_) rw[k][w_] = __uint_as_float(rec[1 + w_]);
    }
  }
  const bool divide = !ENERGY && p.divide;
  const unsigned tail0 = s + tid + KR * NT;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int R0 = PI::R0, NB0 = RL / R0;
  constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
  // cell of a record inside the half that is being worked on, or 0xffffffff: not this half's
  auto local = [&](unsigned cell, int h) -> unsigned {
    const unsigned lo = (unsigned)h * (unsigned)ACC;
    return (cell - lo) < (unsigned)ACC ? cell - lo : 0xffffffffu;     // (0xffffffff - lo wraps far beyond ACC)
  };
  auto dense_clear = [&]() {
    for (int i = tid; i < ACC / 4; i += NT) reinterpret_cast<float4*>(acc)[i] = zero4;
  };
  // one accumulation round of half h: adds `word` of every record of the half (times its 1/rho when dividing)
  auto scatter = [&](int h, int word, bool times_rrec) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const unsigned c_ = local(rloc[k], h);
      if (c_ != 0xffffffffu) {
        float v_;
        if constexpr (ENERGY) {
          v_ = word == 1 ? rw[k][0] : (word == 2 ? rw[k][1] : (word == 3 ? rw[k][2] : rw[k][NW_ - 1]));
        } else {   // (vector launches keep three words; rho -- the dividing launches' first rounds -- is read where it is used)
          v_ = word == 1 ? rw[k][0] : (word == 2 ? rw[k][1] : (word == 3 ? rw[k][2]
                                                                          : __uint_as_float(p.records[(size_t)(s + tid + k * NT) * 5 + 4])));
        }
        vps_lds_add(&acc[c_], times_rrec ? v_ * rrec[k] : v_, crowded);
      }
    }
    for (unsigned j = tail0; j < e; j += NT) {
      const unsigned* rec = p.records + (size_t)j * 5;
      const unsigned c_ = local(rec[0], h);
      if (c_ != 0xffffffffu) {
        const float v_ = __uint_as_float(rec[word]);
        vps_lds_add(&acc[c_], times_rrec ? v_ * p.side[j] : v_, crowded);
      }
    }
  };
  auto sparse_clear = [&](int h) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const unsigned c_ = local(rloc[k], h);
      if (c_ != 0xffffffffu) acc[c_] = 0.f;
    }
    for (unsigned j = tail0; j < e; j += NT) {
      const unsigned c_ = local(p.records[(size_t)j * 5], h);
      if (c_ != 0xffffffffu) acc[c_] = 0.f;
    }
  };

  // ---- velocity: 1 / rho of every record's cell, half by half ----
  if (divide) {
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      dense_clear();
      __syncthreads();
      scatter(h, 4, false);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const unsigned c_ = local(rloc[k], h);
        if (c_ != 0xffffffffu) {
          const float r = acc[c_];
          rrec[k] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
        }
      }
      for (unsigned j = tail0; j < e; j += NT) {
        const unsigned c_ = local(p.records[(size_t)j * 5], h);
        if (c_ != 0xffffffffu) {
          const float r = acc[c_];
          p.side[j] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
        }
      }
    }
  }
  // ---- ENERGY: per record the sum over components of (cell total of rho v_c)^2, half by half and component by component ----
  if constexpr (ENERGY) {
    for (int h = 0; h < 2; ++h)
      for (int c = 0; c < p.ncomp; ++c) {
        __syncthreads();
        if (c == 0) dense_clear(); else sparse_clear(h);
        __syncthreads();
        scatter(h, 1 + p.chan[c], false);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const unsigned c_ = local(rloc[k], h);
          if (c_ != 0xffffffffu) {
            const float q = acc[c_];
            rrec[k] = (c == 0) ? q * q : rrec[k] + q * q;
          }
        }
        for (unsigned j = tail0; j < e; j += NT) {
          const unsigned c_ = local(p.records[(size_t)j * 5], h);
          if (c_ != 0xffffffffu) {
            const float q = acc[c_];
            p.side[j] = (c == 0) ? q * q : p.side[j] + q * q;
          }
        }
      }
  }

  // ---- per output field: both halves accumulated and transformed, then the image of all eight lines in two halves ----
  const int nfields = ENERGY ? 1 : p.ncomp;
  for (int c = 0; c < nfields; ++c) {
    cf vpark[RL], v[RL];
    for (int h = 0; h < 2; ++h) {
      int lc = l, tc = t;
      asm volatile("" : "+v"(lc), "+v"(tc));   // (keeps the LDS addresses of the two halves from being formed up front)
      lc &= L - 1;
      tc &= TH - 1;
      __syncthreads();   // previous image / previous half's exchange buffers consumed
      dense_clear();
      __syncthreads();
      if constexpr (ENERGY) {
        scatter(h, 4, false);              // rho
        __syncthreads();
        // E = vol * sum / rho where there is mass: every record writes its cell's value once all have read rho
        float ev[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const unsigned c_ = local(rloc[k], h);
          ev[k] = 0.f;
          if (c_ != 0xffffffffu) {
            const float r = acc[c_];
            ev[k] = r != 0.f ? rrec[k] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
          }
        }
        for (unsigned j = tail0; j < e; j += NT) {
          const unsigned c_ = local(p.records[(size_t)j * 5], h);
          if (c_ != 0xffffffffu) {
            const float r = acc[c_];
            p.side[j] = r != 0.f ? p.side[j] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const unsigned c_ = local(rloc[k], h);
          if (c_ != 0xffffffffu) acc[c_] = ev[k];
        }
        for (unsigned j = tail0; j < e; j += NT) {
          const unsigned c_ = local(p.records[(size_t)j * 5], h);
          if (c_ != 0xffffffffu) acc[c_] = p.side[j];
        }
      } else {
        scatter(h, 1 + p.chan[c], divide);
      }
      __syncthreads();
      {
        const float* q = acc + tc * N;
        const float sc = (ENERGY || divide) ? 1.f : p.vol;
#pragma unroll
        for (int m = 0; m < NB0; ++m)
#pragma unroll
          for (int rr = 0; rr < R0; ++rr) {
            const int j = lc + L * m + rr * (NC / R0);
            const float2 qq = *reinterpret_cast<const float2*>(q + 2 * j);
            v[m * R0 + rr] = make_float2(qq.x * sc, qq.y * sc);
          }
      }
      __syncthreads();   // accumulator consumed: its memory becomes FFT scratch
      fft_from_regs_l<NC, L, WSYNC>(v, buf + tc * PI::PITCH, tw, lc);
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < RL; ++i) vpark[i] = v[i];
      }
    }
    // ---- image halves: pairs (k, NC - k) with min(k, NC - k) < NC/4 first, the rest second ----
    cf* out = p.out[c] + (long long)x * NC * N + y0;
    cf* nyq = p.nyq[c] + (long long)x * N + y0;
    two_half_image_store<NC, TP, L, !ENERGY>(buf, edge, vpark, v, l, t, tid, out, nyq, p.tw_r2c);
  }
}

// ------------------------------------------------------------------------------
// Pencil PAIR: two neighbouring 8-line pencils (y0 .. y0 + 7 and y0 + 8 .. y0 + 15 -- consecutive buckets of the sort) in ONE
// workgroup of the 8-line kernel's shape.  The halves run one after the other through the 8-line kernel's rounds and
// transform (same LDS region, same threads; each half scans only its own bucket); the first half's spectrum waits in registers,
// and the epilogue stores the 16 lines together: whole 128-byte lines with 16-byte stores, where two separate pencils wrote 64-byte
// halves that only sometimes met in L2 (PMC: 117 GB written for 103 GB of output in a vector launch at 2048^3).
// ------------------------------------------------------------------------------
template <int NC, bool ENERGY>
__global__ void __launch_bounds__(8 * pencil_lanes<NC>(), 4) pencil_pair_fft_z_kernel(const PencilParams p) {
  typedef PlanInfo<NC> PI;
  constexpr int TH = 8, TP = 16;
  constexpr int L = pencil_lanes<NC>(), RL = NC / L, NT = TH * L, N = 2 * NC;
  constexpr int ACC = TH * N;                       // floats of one accumulator (eight lines)
  constexpr int LINES = TH * PI::PITCH * 2;         // floats of the exchange buffers
  constexpr int IMG = (NC / 2) * TP * 2;            // floats of half an image: NC/2 modes x 16 lines
  constexpr int SHARED = (ACC > LINES ? (ACC > IMG ? ACC : IMG) : (LINES > IMG ? LINES : IMG));
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* acc = reinterpret_cast<float*>(smem_raw);
  cf* buf = reinterpret_cast<cf*>(acc);
  cf* edge = reinterpret_cast<cf*>(acc + SHARED);
  cf* tw_lds = edge + TP;
  constexpr bool TWL = PI::TWLDS;
  const cf* tw = TWL ? tw_lds : p.tw_stage;

  const int tid = threadIdx.x;
  const int t = tid / L, l = tid % L;
  const unsigned nby2 = (unsigned)p.nby / 2u;
  const unsigned pair = blockIdx.x;
  const int x = pair / nby2, y0 = (pair % nby2) * TP;
  const unsigned p0 = (unsigned)x * (unsigned)p.nby + 2u * (pair % nby2);
  const unsigned sA = p.start[p0], sB = p.start[p0 + 1], eB = p.start[p0 + 2];
  if constexpr (TWL)
    for (int i = tid; i < PI::TW; i += NT) tw_lds[i] = p.tw_stage[i];
  // per half one register-resident record per thread (a bucket of 8 lines holds ~190 records at the bench density, 512
  // threads); fuller buckets read the rest inside the rounds (tail loops; their per-record value lives in p.side[record])
  unsigned rloc[2];
  float rval[2], rrec[2];
  const unsigned hs[2] = {sA, sB}, he[2] = {sB, eB};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned j = hs[h] + tid;
    rloc[h] = (j < he[h]) ? p.records[(size_t)j * 5] : 0xffffffffu;
    rrec[h] = 1.f;
  }
  const bool divide = !ENERGY && p.divide;
  auto fetch = [&](int h, int word) {   // record word 1..3: rho v_c, 4: rho
    const unsigned j = hs[h] + tid;
    if (j < he[h]) rval[h] = __uint_as_float(p.records[(size_t)j * 5 + word]);
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int R0 = PI::R0, NB0 = RL / R0;
  constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
  auto dense_clear = [&]() {
    for (int i = tid; i < ACC / 4; i += NT) reinterpret_cast<float4*>(acc)[i] = zero4;
  };
  auto sparse_clear = [&](int h) {
    if (rloc[h] != 0xffffffffu) acc[rloc[h]] = 0.f;
    for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) acc[p.records[(size_t)j * 5]] = 0.f;
  };
  // one accumulation round of half h: the value in rval[h] (word `word` of the tail records), times 1/rho when dividing
  auto scatter = [&](int h, int word, bool times_rrec, bool crowded) {
    if (rloc[h] != 0xffffffffu) vps_lds_add(&acc[rloc[h]], times_rrec ? rval[h] * rrec[h] : rval[h], crowded);
    for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) {
      const unsigned* rec = p.records + (size_t)j * 5;
      const float val = __uint_as_float(rec[word]);
      vps_lds_add(&acc[rec[0]], times_rrec ? val * p.side[j] : val, crowded);
    }
  };
  const bool crowdedh[2] = {(sB - sA) > 2u * (unsigned)ACC, (eB - sB) > 2u * (unsigned)ACC};

  // ---- velocity: 1 / rho of every record's cell; the second half first, so that the accumulator holds the FIRST half's rho
  // when its first component starts (sparse clear there) ----
  if (divide) {
    fetch(1, 4);
    fetch(0, 4);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {   // (unrolled: h indexes register arrays)
      const int h = 1 - hh;
      __syncthreads();
      dense_clear();
      __syncthreads();
      scatter(h, 4, false, crowdedh[h]);
      __syncthreads();
      if (rloc[h] != 0xffffffffu) {
        const float r = acc[rloc[h]];
        rrec[h] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
      }
      for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) {
        const float r = acc[p.records[(size_t)j * 5]];
        p.side[j] = r != 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
      }
    }
  }

  const int nfields = ENERGY ? 1 : p.ncomp;
  for (int c = 0; c < nfields; ++c) {
    cf vpark[RL], v[RL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // (unrolled: h indexes register arrays)
      int lc = l, tc = t;
      asm volatile("" : "+v"(lc), "+v"(tc));   // (keeps the LDS addresses of the two halves from being formed up front)
      lc &= L - 1;
      tc &= TH - 1;
      if constexpr (ENERGY) {
        // the three rho v_c rounds add up q_c^2 per record, a fourth accumulates rho and finishes E = vol * sum / rho
        fetch(h, 1 + p.chan[0]);
        for (int cc = 0; cc < p.ncomp; ++cc) {
          __syncthreads();
          if (cc == 0) dense_clear(); else sparse_clear(h);
          __syncthreads();
          scatter(h, 1 + p.chan[cc], false, crowdedh[h]);
          fetch(h, cc + 1 < p.ncomp ? 1 + p.chan[cc + 1] : 4);
          __syncthreads();
          if (rloc[h] != 0xffffffffu) {
            const float q = acc[rloc[h]];
            rrec[h] = (cc == 0) ? q * q : rrec[h] + q * q;
          }
          for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) {
            const float q = acc[p.records[(size_t)j * 5]];
            p.side[j] = (cc == 0) ? q * q : p.side[j] + q * q;
          }
        }
        __syncthreads();
        sparse_clear(h);
        __syncthreads();
        scatter(h, 4, false, crowdedh[h]);
        __syncthreads();
        if (rloc[h] != 0xffffffffu) {
          const float r = acc[rloc[h]];
          rrec[h] = r != 0.f ? rrec[h] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
        }
        for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) {
          const float r = acc[p.records[(size_t)j * 5]];
          p.side[j] = r != 0.f ? p.side[j] * __builtin_amdgcn_rcpf(r) * p.vol : 0.f;
        }
        __syncthreads();
        if (rloc[h] != 0xffffffffu) acc[rloc[h]] = rrec[h];
        for (unsigned j = hs[h] + tid + NT; j < he[h]; j += NT) acc[p.records[(size_t)j * 5]] = p.side[j];
      } else {
        if (c == 0 && !divide) fetch(h, 1 + p.chan[0]);
        if (c == 0 && divide && h == 0) {
          fetch(0, 1 + p.chan[0]);
          fetch(1, 1 + p.chan[0]);
        }
        __syncthreads();   // previous image / the other half's exchange buffers consumed
        if (divide && c == 0 && h == 0) sparse_clear(0); else dense_clear();
        __syncthreads();
        scatter(h, 1 + p.chan[c], divide, crowdedh[h]);
        if (c + 1 < nfields) fetch(h, 1 + p.chan[c + 1]);
      }
      __syncthreads();
      {
        const float* q = acc + tc * N;
        const float sc = (ENERGY || divide) ? 1.f : p.vol;
#pragma unroll
        for (int m = 0; m < NB0; ++m)
#pragma unroll
          for (int rr = 0; rr < R0; ++rr) {
            const int j = lc + L * m + rr * (NC / R0);
            const float2 qq = *reinterpret_cast<const float2*>(q + 2 * j);
            v[m * R0 + rr] = make_float2(qq.x * sc, qq.y * sc);
          }
      }
      __syncthreads();   // accumulator consumed: its memory becomes FFT scratch
      fft_from_regs_l<NC, L, WSYNC>(v, buf + tc * PI::PITCH, tw, lc);
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < RL; ++i) vpark[i] = v[i];
      }
    }
    cf* out = p.out[c] + (long long)x * NC * N + y0;
    cf* nyq = p.nyq[c] + (long long)x * N + y0;
    two_half_image_store<NC, TP, L, !ENERGY>(buf, edge, vpark, v, l, t, tid, out, nyq, p.tw_r2c);
  }
}

// (A wave-private form -- every wave scans all records of the pencil, keeps those of its own line(s) and runs zero-fill, LDS
// adds, read-back, stage-0 loads and the exchanges inside its own LDS region with wave-level ordering only, three workgroup
// barriers per component instead of seven -- measured at C4: 97 against 75 ms per step of pencil launches, bit-identical
// results.  The barriers are not what the kernel waits for; eight-fold record scans and 64-lane zero-fills cost more.)
// (A persistent form of the pencil kernel -- workgroups walking pencil slots, the next pencil's bucket bounds, cells and
// first values requested behind the last component's stores -- measured slower: 32.5 against 30.6 ms per C4 vector launch,
// 16.1 against 13.0 for energy, 0.48 against 0.39 ms at C2.  Unlike the y pass's tiles, pencils differ in work; the
// hardware dispatcher balances them, a fixed stride does not.)
// y-lines per pencil: 16 (128-byte output segments); 8 for 2048-cell lines, on half the plan's lanes (pencil_lanes): 256
// threads and 76 KB of LDS per workgroup, so TWO workgroups share a CU and one's stores overlap the other's LDS work.  The two
// pencils that complete a 128-byte output line are placed on one XCD (remap in the kernel); 38 % of the lines still
// reach HBM as two halves (PMC WRITE_SIZE 142 GB for 103 GB of output), which costs less than the lost overlap.
// Measured at C4 (2048^3, ms per step of 7 fields): 16 lines x 64 lanes, spilling 111; 8 x 64 spilling 118, not spilling
// (two waves per SIMD) 106; 16 x 32 (one workgroup per CU, full-line writes) 88; 8 x 32 83; 4 x 64 (32-byte segments) 228.
#ifndef VPS_PENCIL_TP_LONG
#define VPS_PENCIL_TP_LONG 8
#endif
template <int NC>
constexpr int pencil_tp() {
  return NC >= 1024 ? VPS_PENCIL_TP_LONG : 16;
}

template <int NC>
size_t pencil_lds_bytes() {
  typedef PlanInfo<NC> PI;
  constexpr int PENCIL_TP = pencil_tp<NC>();
  constexpr int ACC = PENCIL_TP * 2 * NC, LINES = PENCIL_TP * PI::PITCH * 2;
  return (size_t)(ACC > LINES ? ACC : LINES) * sizeof(float) + (size_t)(pencil_twreg<NC>() == 3 ? 0 : PI::TWL) * sizeof(cf);
}

// which line lengths run the split form (pencil_split_fft_z_kernel): bit 0 -- 1024 packed points (2048-cell lines), bit 1 -- 2048
// (4096-cell lines, where the whole pencil leaves one workgroup per CU)
#ifndef VPS_PENCIL_SPLIT
#define VPS_PENCIL_SPLIT 2
#endif
template <int NC>
constexpr bool pencil_split() {
  return pencil_tp<NC>() == 8 && (LastRadix<NC>::R % 4) == 0 &&
         ((NC == 1024 && (VPS_PENCIL_SPLIT & 1)) || (NC == 2048 && (VPS_PENCIL_SPLIT & 2)));
}
template <int NC>
size_t pencil_split_lds_bytes() {
  typedef PlanInfo<NC> PI;
  constexpr int TH = pencil_tp<NC>() / 2;
  constexpr size_t ACC = (size_t)TH * 2 * NC, LINES = (size_t)TH * PI::PITCH * 2, IMG = (size_t)(NC / 2) * pencil_tp<NC>() * 2;
  const size_t shared = ACC > LINES ? (ACC > IMG ? ACC : IMG) : (LINES > IMG ? LINES : IMG);
  return shared * sizeof(float) + (size_t)pencil_tp<NC>() * sizeof(cf);
}

// which launches of 2048-cell lines run as pencil pairs (pencil_pair_fft_z_kernel): bit 0 -- vector launches, bit 1 -- energy
#ifndef VPS_PENCIL_PAIR
#define VPS_PENCIL_PAIR 0
#endif
template <int NC>
constexpr int pencil_pair() {
  return (NC == 1024 && pencil_tp<NC>() == 8 && (LastRadix<NC>::R % 4) == 0) ? (VPS_PENCIL_PAIR) : 0;
}
template <int NC>
size_t pencil_pair_lds_bytes() {
  typedef PlanInfo<NC> PI;
  constexpr size_t ACC = (size_t)8 * 2 * NC, LINES = (size_t)8 * PI::PITCH * 2, IMG = (size_t)(NC / 2) * 16 * 2;
  const size_t shared = ACC > LINES ? (ACC > IMG ? ACC : IMG) : (LINES > IMG ? LINES : IMG);
  return shared * sizeof(float) + (size_t)16 * sizeof(cf) + (size_t)(PI::TWLDS ? PI::TW : 0) * sizeof(cf);
}

template <int NC>
int launch_pencil(vps_ctx* ctx, const PencilParams& p, long long npencils) {
  typedef PlanInfo<NC> PI;
  if constexpr (pencil_pair<NC>() != 0) if ((pencil_pair<NC>() & (p.energy ? 2 : 1)) && !p.with_energy && (p.nby % 2) == 0 && npencils % 2 == 0) {
    const size_t lds2 = pencil_pair_lds_bytes<NC>();
    auto kern2 = p.energy ? pencil_pair_fft_z_kernel<NC, true> : pencil_pair_fft_z_kernel<NC, false>;
    if (lds2 > 64 * 1024)
      VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern2),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    PencilParams pp2 = p;
    pp2.npencils = (unsigned)npencils;
    {
      vps_launch_timer tm(ctx, VPS_K_FFT_Z);
      hipLaunchKernelGGL(kern2, dim3((unsigned)(npencils / 2)), dim3(8 * pencil_lanes<NC>()), lds2, ctx->stream, pp2);
    }
    VPS_HIP_CHECK(ctx, hipGetLastError());
    return VPS_OK;
  }
  if constexpr (pencil_split<NC>()) if (p.energy) {   // (the vector form of the split kernel does not fit 128 VGPRs: whole pencils there)
    const size_t lds2 = pencil_split_lds_bytes<NC>();
    constexpr int TP2 = pencil_tp<NC>();
    auto kern2 = pencil_split_fft_z_kernel<NC, TP2, true>;
    if (lds2 > 64 * 1024)
      VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern2),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    PencilParams pp2 = p;
    pp2.npencils = (unsigned)npencils;
    {
      vps_launch_timer tm(ctx, VPS_K_FFT_Z);
      hipLaunchKernelGGL(kern2, dim3((unsigned)npencils), dim3((TP2 / 2) * pencil_lanes<NC>()), lds2, ctx->stream, pp2);
    }
    VPS_HIP_CHECK(ctx, hipGetLastError());
    return VPS_OK;
  }
  const size_t lds = pencil_lds_bytes<NC>();
  constexpr int PENCIL_TP = pencil_tp<NC>();
  if (lds > ctx->lds_per_cu) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "pencil kernel needs %zu B LDS", lds);
  auto kern = p.energy ? pencil_fft_z_kernel<NC, PENCIL_TP, true> : pencil_fft_z_kernel<NC, PENCIL_TP, false>;
  if (lds > 64 * 1024)
    VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  PencilParams pp = p;
  pp.npencils = (unsigned)npencils;
  long long grid = npencils;
  if (VPS_PENCIL_PERSIST) {
    long long per_cu = (long long)(ctx->lds_per_cu / (lds ? lds : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const long long want = (long long)ctx->num_cu * per_cu;
    if (grid > want) grid = want;
  }
  {
    vps_launch_timer tm(ctx, VPS_K_FFT_Z);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PENCIL_TP * pencil_lanes<NC>()), lds, ctx->stream, pp);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

// ------------------------------------------------------------------------------
// x pass: lines are contiguous (or made of nseg contiguous segments); the result is
// binned straight from registers (MODE 0) or written in place order (MODE 1).
// Persistent workgroups loop over tiles of T lines.
// ------------------------------------------------------------------------------
struct XParams {
  const cf* in;            // component 0
  const cf* in1;           // components 1, 2 of a vector field (binning modes, ncomp > 1)
  const cf* in2;
  int ncomp;
  cf* out;
  long long nlines, line0;
  int N, kz0;
  int seglen, seg_shift;  // segment length and its log2 (-1: not a power of two)
  long long seg_stride;
  const cf* tw_stage;
  const double* k2;
  const double* thr;
  const float* win;        // 1 / W^2 per axis index (vps_set_window) or NULL
  int nbins;
  float edge0, inv_spacing;
  double* psum;
  unsigned long long* nsample;
  const int2* ptab;   // chunked exchange (vps_fft_x_bin_chunk; NULL otherwise): plane i of the launch starts at row ptab[i].x of a
                      // block and holds the rows |ky| <= ptab[i].y (-1: all N), see PassParams::ptab; its kz is kz0 + i * kz_step
  int kz_step;
  int pair;  // FAST path: tiles pair ky with N-ky (needs whole ky ranges: line0, nlines multiples of N)
  // integer shells (FAST == 2; vps_set_binning has checked that they reproduce the float64 comparison bit for bit): a mode
  // belongs to shell b iff nthr[b] <= ix^2 + iy^2 + iz^2 < nthr[b + 1] (signed mode indices); nmax = nthr[nbins]; kf = 2 pi / L
  const unsigned* nthr;
  unsigned nmax;
  float kf;
  double* part_sum;    // [grid][nbins] per-workgroup partial shell sums
  unsigned* part_cnt;  // [grid][nbins] per-workgroup partial shell counts
};

// FAST (MODE 0 only): the k^2 table is symmetric (k2[N-i] == k2[i]) and non-decreasing on
// [0, N/2] -- true for 2 pi fftfreq -- so kx and -kx share one s and one shell: a lane
// then walks RL/2 values of |kx|, adds the two mirrored |F|^2 and issues one LDS atomic
// per |kx|, with no per-element branches (the host checks the table, vps_set_binning).
// FASTMODE 2 = FAST with INTEGER shells: in units of (2 pi / L)^2 the reference's s = (kx^2 + ky^2) + kz^2 is the integer
// n = ix^2 + iy^2 + iz^2 up to float64 rounding, and where no shell threshold lies within 1e-9 (relative) of an integer --
// both reference flavours: their edges are half-integer multiples of 2 pi / L -- the float64 comparison thr[b] <= s < thr[b+1]
// and the integer comparison nthr[b] <= n < nthr[b+1] (nthr[b] = ceil(thr[b] / k2[1])) decide every mode alike.  The kernel
// then needs no k^2 table at all: no float64 registers or adds per mode, 4-byte thresholds, and nothing to load per tile or
// per line (tile_beyond_shells / locate_line are arithmetic on the tile index).  vps_set_binning checks the condition and
// falls back to FASTMODE 1 where it fails (custom k ranges with edges on integer multiples).
// With integer shells the 2048-point plan also keeps its last stage's twiddles in registers (x_twreg): thresholds 4 KB + sums
// 8 KB + counts 4 KB + stage-1 twiddles 2 KB + two lines 34 KB = 52 KB, so THREE workgroups share a CU instead of two
// (70.6 KB before).  Measured at C4: persistent workgroups per CU 1 -> 2: 35.7 -> 22.7 ms per vector launch.
// (Lines made of segments -- the received blocks of a slab exchange -- take the same form: their loads are a scalar base and
//  one lane offset, load_line.  While the kernel also carried the general segment addressing, a 64-bit offset per element for
//  segments shorter than a line's lanes, that path alone cost 60 VGPRs and the register-twiddle variant spilled 97.)
template <int NC, int FASTMODE, bool SEG>
constexpr bool x_twreg() {
  return FASTMODE == 2 && NC == 2048 && PlanInfo<NC>::R2 > 1;
}
template <int NC, int T, int MODE, bool SEG, bool COUNT, int FASTMODE>
#ifndef VPS_X_MIN_WAVES
#define VPS_X_MIN_WAVES 1
#endif
__global__ void __launch_bounds__(T* PlanInfo<NC>::L, (x_twreg<NC, FASTMODE, SEG>() ? 3 : VPS_X_MIN_WAVES)) fft_x_pass(const XParams p) {
  typedef PlanInfo<NC> PI;
  constexpr bool FAST = FASTMODE != 0, INTB = FASTMODE == 2, TWREG = x_twreg<NC, FASTMODE, SEG>();
  constexpr int L = PI::L, RL = PI::RL, NT = T * L;
  constexpr int H = RL / 2;   // |kx| values per lane on the FAST path
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // carve: thr (double, nbins+2 with a +inf sentinel; INTB: unsigned, 0xffffffff sentinel) | hsum (double) | tw | line buffers | hcnt
  double* thr = reinterpret_cast<double*>(smem_raw);
  unsigned* nthr = reinterpret_cast<unsigned*>(smem_raw);
  double* hsum = INTB ? reinterpret_cast<double*>(nthr + ((p.nbins + 3) & ~1)) : thr + (MODE == 0 ? (p.nbins + 2) : 0);
  cf* tw_lds = reinterpret_cast<cf*>(hsum + (MODE == 0 ? p.nbins : 0));
  constexpr int TWCOPY = TWREG ? PI::TW1 : PI::TW;                      // twiddle entries staged in LDS
  constexpr int TWRES = TWREG ? ((PI::TW1 + 1) & ~1) : PI::TWL;         // ... and reserved for them
  cf* buf = tw_lds + TWRES;
  const cf* tw = PI::TWLDS ? tw_lds : p.tw_stage;
  unsigned* hcnt = reinterpret_cast<unsigned*>(buf + T * PI::PITCH);
  float* wl = reinterpret_cast<float*>(hcnt + ((MODE == 0 && COUNT) ? p.nbins : 0));   // [NC] window factors (MODE 0 with p.win)

  const int tid = threadIdx.x;
  const int t = tid / L, l = tid % L;
  if constexpr (PI::TWLDS)
    for (int i = tid; i < TWCOPY; i += NT) tw_lds[i] = p.tw_stage[i];
  cf twr[TWREG ? (PI::R2 - 1) : 1];
  if constexpr (TWREG) load_stage_twiddles<NC, L, RL, PI::R2, PI::NS2>(twr, p.tw_stage + PI::TW1, l);
  // fl(kx*kx) of this lane's contiguous chunk of RL kx values, ordered by non-decreasing
  // |kx|: the chunks of the negative-frequency half (index >= NC/2) are walked backwards
  double k2x[(MODE == 0 && !INTB) ? (FAST ? H : RL) : 1];
  const bool rev = (l * RL) >= NC / 2;
  if constexpr (MODE == 0) {
    if constexpr (INTB) {
      for (int i = tid; i <= p.nbins + 1; i += NT) nthr[i] = (i <= p.nbins) ? p.nthr[i] : 0xffffffffu;
    } else {
      for (int i = tid; i <= p.nbins + 1; i += NT) thr[i] = (i <= p.nbins) ? p.thr[i] : INFINITY;
    }
    for (int i = tid; i < p.nbins; i += NT) {
      hsum[i] = 0.0;
      if constexpr (COUNT) hcnt[i] = 0u;
    }
    if (p.win)
      for (int i = tid; i < NC; i += NT) wl[i] = p.win[i];
    if constexpr (INTB) {
    } else if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < H; ++i) k2x[i] = p.k2[l * H + i];
    } else {
#pragma unroll
      for (int i = 0; i < RL; ++i) k2x[i] = p.k2[l * RL + (rev ? RL - 1 - i : i)];
    }
  }
  __syncthreads();

  cf* line = buf + t * PI::PITCH;
  const int segmask = p.seglen - 1;
  const long long ntiles = (p.nlines + T - 1) / T;
  // PAIR (FAST path on whole ky ranges): a tile holds T/2 lines ky and their mirrors N-ky,
  // which share every s with them, so one lane bins four modes (+-kx, +-ky) per step.
  // wave-level synchronisation of everything line-local (exchanges, |F|^2 image)
  constexpr bool WSYNC = (L <= 64) && (64 % L == 0);
  // pairs are adjacent lines (t, t^1): with at least two lines per wave a pair never leaves its wave
  constexpr bool CANPAIR = FAST && (T >= 2) && (NC >= T) && (NC % T == 0) && (!WSYNC || L <= 32);
  const bool pair = CANPAIR && p.pair;
  constexpr int TH = (T >= 2) ? T / 2 : 1;
  // A PAIR tile holds TH consecutive |ky|, aligned to TH.  The y passes of a binning-only scope store the rows |ky| <= kcut[kz],
  // kcut = the exact cut rounded UP to 16 k + 15 (vps_set_binning, api.hip).  Where TH divides 16 (every power-of-two grid from
  // 128 on) a tile that is transformed lies entirely inside the stored rows.  Elsewhere (TH = 32 on the small grids, TH = 6 on
  // 3 2^a) a transformed tile may read rows the y pass left unwritten: every mode of such a row has fl(ky^2 + kz^2) >=
  // thr[nbins], the shell search below puts it at bin == nbins, and the `bin < nbins` guard keeps whatever the row held
  // (NaN included) out of every sum and count -- tests/test_gpu_configs.py runs the exchange buffers NaN-prefilled for this.
  // tile -> this lane's line, and the loads of its stage-0 inputs
  long long li = 0, lrow = 0;   // line index inside the launch's range, and the row of the input it is read from
  bool live = false, mirrored = false, has_partner = false;
  double k2y = 0.0, k2z = 0.0;   // fl(ky*ky), fl(kz*kz) of the line (MODE 0)
  unsigned nyz = 0u;             // INTB: iy^2 + iz^2 of the line
  float wyz = 1.f;               // window factor of the line's (ky, kz)
  unsigned wz = 1u;              // Hermitian multiplicity of its kz plane
  double k2half = 0.0;
  if constexpr (MODE == 0 && !INTB) k2half = p.k2[NC / 2];
  cf v[RL];
  // PAIR tiles whose smallest |ky| already puts every mode of the tile beyond the last shell edge -- fl(ky^2 + kz^2) >=
  // thr[nbins], and s = (kx^2 + ky^2) + kz^2 can only be larger -- are neither loaded nor transformed: with the default
  // k range (kmax = Nyquist) that is the quarter of each kz plane outside the inscribed circle (1 - pi/4 = 21 %).
  // (The |ky| order of a plane's tiles is rotated from plane to plane: a persistent workgroup takes every gridDim-th
  // tile, and without the rotation the same workgroups would always draw the skipped outer |ky| and the others never.)
  constexpr long long tiles_per_plane = (NC / T) > 0 ? (NC / T) : 1;
  auto tile_q = [&](long long tile) -> int {   // which group of TH |ky| values PAIR tile `tile` holds
    const long long plane = tile / tiles_per_plane;
    return (int)((tile % tiles_per_plane + plane * 619) % tiles_per_plane);
  };
  auto tile_beyond_shells = [&](long long tile) -> bool {
    if constexpr (MODE == 0 && FAST) {
      if (pair) {
        const int kz = p.kz0 + ((int)(p.line0 / NC) + (int)(tile / tiles_per_plane)) * p.kz_step;
        if constexpr (INTB) {
          const unsigned ka = (unsigned)(tile_q(tile) * TH);
          return ka * ka + (unsigned)kz * (unsigned)kz >= p.nmax;
        } else {
          return (p.k2[tile_q(tile) * TH] + p.k2[kz]) >= p.thr[p.nbins];
        }
      }
    }
    return false;
  };
  auto locate_line = [&](long long tile) {
    li = tile * T + t;       // local line index
    mirrored = false;        // this line is the N-ky partner of line t - 1
    has_partner = false;     // line t + 1 holds this line's N-ky partner
    if (pair) {
      const long long plane = tile / tiles_per_plane;
      const int q = tile_q(tile);
      const int ky_a = q * TH + (t >> 1);
      const int ky = ((t & 1) == 0) ? ky_a : (ky_a == 0 ? NC / 2 : NC - ky_a);
      li = plane * NC + ky;
      mirrored = ((t & 1) == 1) && (ky_a != 0);
      has_partner = ((t & 1) == 0) && (ky_a != 0);
    }
    live = li < p.nlines && !tile_beyond_shells(tile);
    lrow = li;
    if (p.ptab) {   // chunked exchange (line0 = 0): every mode of a row that was not sent lies beyond the last shell edge
      const int ky = (int)(li % NC);
      const int2 pt = p.ptab[li / NC];
      live = live && (pt.y < 0 || ky <= pt.y || ky >= NC - pt.y);
      lrow = pt.x + packed_row(ky, NC, pt.y);
    }
    if constexpr (MODE == 0) {
      if (live) {
        const long long g = p.line0 + li;
        // (lines of the x pass have N = NC points: a compile-time divisor instead of a 64-bit division per tile and thread)
        const int kz = p.kz0 + (int)(g / NC) * p.kz_step;
        if constexpr (INTB) {
          const int kyi = (int)(g % NC);
          const unsigned iy = (unsigned)(2 * kyi <= NC ? kyi : NC - kyi);
          nyz = iy * iy + (unsigned)kz * (unsigned)kz;
        } else {
          k2y = p.k2[(int)(g % NC)];
          k2z = p.k2[kz];
        }
        wz = (kz == 0 || 2 * kz == NC) ? 1u : 2u;
        if (p.win) wyz = p.win[(int)(g % NC)] * p.win[kz];
      }
    }
  };
  auto load_line = [&](cf (&v)[RL], int c, int l) {   // l: the lane index (callers inside loops pass an opaque copy)
    constexpr int R = PI::R0, NB = RL / R;
    const cf* base = (c == 0 ? p.in : (c == 1 ? p.in1 : p.in2)) + lrow * p.seglen;
    if constexpr (!SEG && (L % 64 == 0)) {
      // A wave holds lanes of ONE line: whether the line is read at all is wave-uniform -- one scalar branch around the RL loads
      // instead of an exec-mask branch around every pair of them (which is what `live ? load : 0` per element compiles to).
      if (__builtin_amdgcn_readfirstlane((int)live)) {
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double raw = __builtin_nontemporal_load(reinterpret_cast<const double*>(&base[l + L * m + r * (NC / R)]));
            v[m * R + r] = *reinterpret_cast<const cf*>(&raw);
          }
      } else {
#pragma unroll
        for (int i = 0; i < RL; ++i) v[i] = make_float2(0.f, 0.f);
      }
      return;
    }
    if constexpr (SEG && (L % 64 == 0)) {
      // A wave holds lanes of ONE line here, so the line's base is wave-uniform; and with power-of-two segments of at least L
      // elements, element x = l + xr (xr = L m + r NC/R, a multiple of L) sits in segment xr >> seg_shift at offset
      // (xr & segmask) + l with no carry.  Everything but l is then scalar: the loads take an SGPR base and one 32-bit lane
      // offset instead of a 64-bit multiply-add per element (x pass of a received 2048^3 chunk, 8 segments: 4.22 -> see DESIGN).
      // (power-of-two lines: launch_x refuses segments shorter than L -- more than NC / L ranks -- so this is the only path the
      //  kernel carries for them; the general form below, with a 64-bit offset per element, costs it ~60 VGPRs)
      constexpr bool ONLY_SCALAR = (NC & (NC - 1)) == 0;
      if (ONLY_SCALAR || (p.seg_shift >= 0 && p.seglen >= L)) {
        const unsigned long long ub = reinterpret_cast<unsigned long long>(base);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)ub), bhi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
        const cf* sbase = reinterpret_cast<const cf*>(((unsigned long long)bhi << 32) | blo);
        // (the shift through an empty asm: the RL segment offsets are loop-invariant, and hoisted out of the tile loop they
        //  sit in ~2 RL VGPRs for the whole kernel -- the scalar file is full -- which spills the register-twiddle variant;
        //  formed where they are used they are a handful of scalar operations per load)
        int sh = p.seg_shift;
        asm volatile("" : "+s"(sh));
        const int smask = (1 << sh) - 1;
        if (__builtin_amdgcn_readfirstlane((int)live)) {   // (wave-uniform: one scalar branch around all the loads)
#pragma unroll
          for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int xr = L * m + r * (NC / R);
              const long long off = (long long)(xr >> sh) * p.seg_stride + (xr & smask);
              // (uniform 64-bit base + zero-extended 32-bit lane offset: the saddr + voffset form of global_load, one VGPR of
              //  address for all loads instead of a 64-bit VGPR pair each)
              const char* sb = reinterpret_cast<const char*>(sbase + off);
              const double raw = __builtin_nontemporal_load(reinterpret_cast<const double*>(sb + (unsigned)l * 8u));
              v[m * R + r] = *reinterpret_cast<const cf*>(&raw);
            }
        } else {
#pragma unroll
          for (int i = 0; i < RL; ++i) v[i] = make_float2(0.f, 0.f);
        }
        return;
      }
    }
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int x = l + L * m + r * (NC / R);
        if constexpr (SEG) {
          // element x of the line sits in segment x >> seg_shift at offset x & segmask
          // (this exact form -- a conditional double load, reinterpreted -- keeps every load a streaming one
          // with one base register and immediate offsets; going through load_stream() did not)
          // (seg_shift < 0: segment length not a power of two -- the 2^a 5^b grids on several ranks)
          // (a power-of-two line divides into power-of-two segments only: no division path to compile, or to keep registers for)
          constexpr bool POW2 = (NC & (NC - 1)) == 0;
          const int sg = (POW2 || p.seg_shift >= 0) ? (x >> p.seg_shift) : x / p.seglen;
          const int so = (POW2 || p.seg_shift >= 0) ? (x & segmask) : x - sg * p.seglen;
          const double raw = live ? __builtin_nontemporal_load(reinterpret_cast<const double*>(
                                        &base[(long long)sg * p.seg_stride + so]))
                                  : 0.0;
          v[m * R + r] = *reinterpret_cast<const cf*>(&raw);
        } else {
          const double raw = live ? __builtin_nontemporal_load(reinterpret_cast<const double*>(&base[x])) : 0.0;
          v[m * R + r] = *reinterpret_cast<const cf*>(&raw);
        }
      }
  };
  if ((long long)blockIdx.x < ntiles) {
    locate_line(blockIdx.x);
    load_line(v, 0, l);
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // (carrying the flag over from the prefetch's locate_line instead of testing again: -2 % at 2048, +30 % at 512 -- not kept)
    if (tile_beyond_shells(tile)) {   // (uniform over the workgroup) nothing to bin here: only keep the prefetch chain going
      if (tile + gridDim.x < ntiles) {
        locate_line(tile + gridDim.x);
        load_line(v, 0, l);
      }
      continue;
    }
    // line bookkeeping of THIS tile (v already holds, or is receiving, its inputs)
    const long long li_cur = li;
    const bool live_cur = live, mirrored_cur = mirrored, partner_cur = has_partner;
    const double k2y_cur = k2y, k2z_cur = k2z;
    const unsigned nyz_cur = nyz;
    const unsigned wz_cur = wz;
    const float wyz_cur = wyz;
    if constexpr (MODE != 0) {
      exchange_sync<WSYNC>();  // previous tile's readers are done with the line buffers
      fft_from_regs<NC, WSYNC, (TWREG ? 2 : 0)>(v, line, tw, l, twr);
    }
    if constexpr (MODE == 1) {
      if (live_cur) {
        cf* o = p.out + li_cur * (long long)NC;
#pragma unroll
        for (int i = 0; i < RL; ++i) o[out_index<NC>(l, i)] = v[i];
      }
    } else if constexpr (MODE == 4) {
      // transform only (timing aid): keep the results alive without storing them
#pragma unroll
      for (int i = 0; i < RL; ++i) asm volatile("" ::"v"(v[i].x), "v"(v[i].y));
    } else if constexpr (MODE == 2) {
      if (live_cur) {
        float* o = reinterpret_cast<float*>(p.out) + li_cur * (long long)NC;
#pragma unroll
        for (int i = 0; i < RL; ++i) o[out_index<NC>(l, i)] += v[i].x * v[i].x + v[i].y * v[i].y;
      }
    } else {
      // Shell sums.  |F|^2 goes through LDS once so that every lane gets a CONTIGUOUS chunk
      // of RL kx values: along kx the shell index moves monotonically (per half line), so a
      // lane accumulates each run of equal bins in registers and issues one LDS float64
      // atomic per run, and the lanes of a wave-instruction hit different bins.
      // The components of a vector field are transformed one after the other and their |F|^2 summed in
      // registers (the reference bins the SUM, interp.py:1372-1421), so the shell search and the LDS
      // atomics below run once per line, not once per component.  As soon as a transform has been
      // squared its registers take the next loads: the next component of this tile, or the first of
      // the next tile, which then fly while this tile is binned.
      float pacc[RL];
      for (int c = 0; c < p.ncomp; ++c) {
        // opaque copy of the lane index: keeps the LDS / global addresses of this loop from being
        // hoisted out of it as ~50 extra live registers (an occupancy step)
        int lc = l;
#ifndef VPS_X_NO_OPAQUE
        asm volatile("" : "+v"(lc));
#endif
        lc = (L & (L - 1)) == 0 ? (lc & (L - 1)) : lc % L;   // give the value range back to the compiler (address folding needs it)
#ifdef VPS_X_PREFETCH
        // second register set: the NEXT line set (next component of this tile, or the first of the next tile) is requested
        // before this one is transformed, so that its loads fly behind the whole transform
        cf w[RL];
        if (c + 1 < p.ncomp) {
          load_line(w, c + 1, lc);
        } else if (tile + gridDim.x < ntiles) {
          locate_line(tile + gridDim.x);
          load_line(w, 0, lc);
        }
#endif
        exchange_sync<WSYNC>();  // previous readers are done with the line buffers
        fft_from_regs<NC, WSYNC, (TWREG ? 2 : 0)>(v, line, tw, lc, twr);
#pragma unroll
        for (int i = 0; i < RL; ++i) {
          const float a = v[i].x * v[i].x + v[i].y * v[i].y;
          pacc[i] = (c == 0) ? a : pacc[i] + a;
        }
#ifdef VPS_X_PREFETCH
#pragma unroll
        for (int i = 0; i < RL; ++i) v[i] = w[i];
        continue;
#endif
        // (Requesting the next line BEFORE this transform -- a second register set, affordable at 2048 where LDS limits
        // the kernel to two waves per SIMD -- was measured at 2048^3: 73.2 against 72.5 ms per step; twiddles and shell
        // thresholds left in L2 instead of LDS (46 KB: three workgroups per CU instead of two): 73 ms.  Not kept.)
        if (c + 1 < p.ncomp) {
          load_line(v, c + 1, lc);
        } else if (tile + gridDim.x < ntiles) {
          locate_line(tile + gridDim.x);
          load_line(v, 0, lc);
        }
      }
      float* pw = reinterpret_cast<float*>(line);
      if constexpr (PI::R1 > 1) exchange_sync<WSYNC>();  // last exchange fully consumed
      constexpr int CH = FAST ? H : RL;              // chunk length; one pad word per chunk
      {
        int lw = l;   // opaque again: 16 store addresses recomputed per tile instead of kept live
#ifndef VPS_X_NO_OPAQUE
        asm volatile("" : "+v"(lw));
#endif
        lw = (L & (L - 1)) == 0 ? (lw & (L - 1)) : lw % L;
#pragma unroll
        for (int i = 0; i < RL; ++i) {
          // k = lw + c with c a multiple of CH wherever the plan's lanes are (power-of-two plans): k / CH = lw / CH + c / CH
          constexpr int RLAST = LastRadix<NC>::R;
          const int c = L * (i / RLAST) + (i % RLAST) * (NC / RLAST);
          if constexpr ((L % CH == 0) && ((NC / RLAST) % CH == 0)) {
            pw[(lw + lw / CH) + c + c / CH] = pacc[i];
          } else {
            const int k = out_index<NC>(lw, i);
            pw[k + k / CH] = pacc[i];
          }
        }
      }
      exchange_sync<WSYNC>();
      if constexpr (FAST) {
        if (live_cur && !mirrored_cur) {
          const unsigned w = wz_cur * (partner_cur ? 2u : 1u);
          const float wf = (float)wz_cur * wyz_cur;   // Hermitian multiplicity x window factor of (ky, kz)
          // the partner line's |F|^2 image is the next line buffer
          constexpr int POFF = PI::PITCH * 2;
          const float* mine = pw + l * (H + 1);                        // kx = l*H + i
          const float* mirr = pw + (NC + NC / H - 1) - l * (H + 1);    // NC-kx for i >= 1 at mirr[-i]
          // Every |kx| is binned independently (no chain between the LDS reads): the edges
          // are uniform (checked by vps_set_binning), so the float guess is off by at most
          // one shell and is corrected against the exact float64 thresholds.
          auto bin_one = [&](int i, double k2xi) {
            int bin;
            if constexpr (INTB) {
              // n = ix^2 + iy^2 + iz^2: the float guess is within one shell, the integer thresholds decide
              const unsigned ix = (unsigned)(l * H + i);
              const unsigned n = ix * ix + nyz_cur;
              int g = (int)((sqrtf((float)n) * p.kf - p.edge0) * p.inv_spacing);
              g = min(max(g, 0), p.nbins - 1);
              const unsigned lo = nthr[g], hi = nthr[g + 1];
              bin = g - ((n < lo) ? 1 : 0) + ((n >= hi) ? 1 : 0);
            } else {
            // s = (kx*kx + ky*ky) + kz*kz with numpy's rounding (the table holds fl(k*k))
            const double s = (k2xi + k2y_cur) + k2z_cur;
            int g = (int)((sqrtf((float)s) - p.edge0) * p.inv_spacing);
            g = min(max(g, 0), p.nbins - 1);
            const double lo = thr[g], hi = thr[g + 1];
            bin = g - ((s < lo) ? 1 : 0) + ((s >= hi) ? 1 : 0);
            }
            float pv = mine[i];
            if (partner_cur) pv += mine[i + POFF];
            unsigned c = 1u;
            if (i > 0) {
              pv += mirr[-i];
              if (partner_cur) pv += mirr[POFF - i];
              c = 2u;
            } else if (l > 0) {
              pv += mirr[1];
              if (partner_cur) pv += mirr[POFF + 1];
              c = 2u;
            }
#ifdef VPS_ABL_X_NOATOMIC
            if ((unsigned)bin < (unsigned)p.nbins && pv == 1.2345e30f) {   // TIMING ONLY
#else
            if ((unsigned)bin < (unsigned)p.nbins) {
#endif
              if (p.win) pv *= wl[l * H + i];
              atomicAdd(&hsum[bin], (double)(pv * wf));
              if constexpr (COUNT) atomicAdd(&hcnt[bin], c * w);
            }
          };
#ifndef VPS_ABL_X_NOBIN
#pragma unroll
          for (int i = 0; i < H; ++i) bin_one(i, INTB ? 0.0 : k2x[INTB ? 0 : i]);
#endif
          if (l == L - 1) {   // the unpaired kx = NC/2 mode
            int bin;
            if constexpr (INTB) {
              const unsigned n = (unsigned)(NC / 2) * (unsigned)(NC / 2) + nyz_cur;
              int g = (int)((sqrtf((float)n) * p.kf - p.edge0) * p.inv_spacing);
              g = min(max(g, 0), p.nbins - 1);
              bin = g - ((n < nthr[g]) ? 1 : 0) + ((n >= nthr[g + 1]) ? 1 : 0);
            } else {
            const double s = (k2half + k2y_cur) + k2z_cur;
            int g = (int)((sqrtf((float)s) - p.edge0) * p.inv_spacing);
            g = min(max(g, 0), p.nbins - 1);
            bin = g - ((s < thr[g]) ? 1 : 0) + ((s >= thr[g + 1]) ? 1 : 0);
            }
            if ((unsigned)bin < (unsigned)p.nbins) {
              float pv = pw[NC / 2 + L];
              if (partner_cur) pv += pw[NC / 2 + L + POFF];
              if (p.win) pv *= wl[NC / 2];
              atomicAdd(&hsum[bin], (double)(pv * wf));
              if constexpr (COUNT) atomicAdd(&hcnt[bin], w);
            }
          }
        }
      } else if (live_cur) {
        const double k2y = k2y_cur, k2z = k2z_cur;
        const unsigned w = wz_cur;
        const double wd = (double)w * (double)wyz_cur;
        const int kx0 = l * RL + (rev ? RL - 1 : 0);     // kx index of the chunk's first element in walking order
        // walk the chunk in the direction of non-decreasing |kx| (k2x was loaded that way)
        const float* mine = pw + l * (RL + 1) + (rev ? RL - 1 : 0);
        const int step = rev ? -1 : 1;
        // shell of the first element: thr[cur] <= s < thr[cur+1], cur = -1 / nbins outside
        double s = (k2x[0] + k2y) + k2z;
        int cur = (int)((sqrtf((float)s) - p.edge0) * p.inv_spacing);
        cur = min(max(cur, 0), p.nbins - 1);
        while (cur > 0 && s < thr[cur]) --cur;
        while (cur < p.nbins - 1 && s >= thr[cur + 1]) ++cur;
        if (s < thr[cur]) cur = -1;
        else if (s >= thr[cur + 1]) cur = p.nbins;
        double hi = (cur < p.nbins) ? thr[cur + 1] : INFINITY;
        double lo = (cur >= 0) ? thr[cur] : -INFINITY;
        double acc = 0.0;
        unsigned cnt = 0;
        auto flush = [&]() {   // one LDS atomic per (lane, shell) run
          if (cur >= 0 && cur < p.nbins) {
            atomicAdd(&hsum[cur], acc * wd);
            if constexpr (COUNT) atomicAdd(&hcnt[cur], cnt * w);
          }
          acc = 0.0;
          cnt = 0;
        };
#pragma unroll
        for (int i = 0; i < RL; ++i) {
          // s = (kx*kx + ky*ky) + kz*kz with numpy's rounding (the table holds fl(k*k))
          s = (k2x[i] + k2y) + k2z;
          while (s >= hi) {   // moved outwards to the next shell (the usual direction)
            flush();
            ++cur;
            lo = hi;
            hi = (cur < p.nbins) ? thr[cur + 1] : INFINITY;
          }
          while (s < lo) {    // only for chunks that are not monotone in |kx| (N = 16)
            flush();
            --cur;
            hi = lo;
            lo = (cur >= 0) ? thr[cur] : -INFINITY;
          }
          acc += (double)(p.win ? mine[i * step] * wl[kx0 + i * step] : mine[i * step]);
          ++cnt;
        }
        flush();
      }
    }
    if constexpr (MODE != 0) {
      if (tile + gridDim.x < ntiles) {
        locate_line(tile + gridDim.x);
        load_line(v, 0, l);
      }
    }
  }
  if constexpr (MODE == 0) {
    // Per-workgroup partial shell sums go out with plain stores; reduce_partials adds them
    // up.  (Hundreds of workgroups doing float64 atomics on the same few cache lines of
    // psum serialise at the memory side and cost as much as the whole transform.)
    __syncthreads();
    double* ps = p.part_sum + (size_t)blockIdx.x * p.nbins;
    unsigned* pc = p.part_cnt + (size_t)blockIdx.x * p.nbins;
    for (int i = tid; i < p.nbins; i += NT) {
      ps[i] = hsum[i];
      if constexpr (COUNT) pc[i] = hcnt[i];
    }
  }
}

__global__ void __launch_bounds__(256)
    reduce_partials(const double* __restrict__ part_sum, const unsigned* __restrict__ part_cnt, int nparts,
                    int nbins, double* __restrict__ psum, unsigned long long* __restrict__ nsample) {
  // grid = (bins / 256, groups): block (., g) sums a slice of the partials for 256 bins
  // and adds it with one atomic per bin (only gridDim.y adders per address)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbins) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int w0 = blockIdx.y * per, w1 = min(w0 + per, nparts);
  double s = 0.0;
  unsigned long long c = 0;
  for (int w = w0; w < w1; ++w) {
    s += part_sum[(size_t)w * nbins + i];
    if (part_cnt) c += part_cnt[(size_t)w * nbins + i];
  }
  if (s != 0.0) atomicAdd(&psum[i], s);
  if (part_cnt && c) atomicAdd(&nsample[i], c);
}

template <int NC>
constexpr int transpose_T() {
  // tile width: 16 lines (128-byte output segments) while the LDS image fits,
  // at least one full wave of threads for short lines
  if (NC <= 16) return 64;
  if (NC <= 512) return 16;
  if (NC <= 2048) return 8;
  return 4;
}
// (4096-point lines, L = 256: ONE line per 256-thread workgroup; two lines -- (ky, N-ky) pairs binned together, 512 threads -- measured
// slower on the C5 rank share: 17.0 against 13.6 ms)
#ifndef VPS_X_T_LONG
#define VPS_X_T_LONG 1
#endif
template <int NC>
constexpr int xpass_T() {
  constexpr int L = Plan<NC>::L;
  constexpr int t = (256 / L) > 1 ? (256 / L) : VPS_X_T_LONG;   // (L = 256: 4096-point lines)
  return (t > 1 && (t & 1)) ? t - 1 : t;   // even, so that a tile holds whole (ky, N-ky) pairs (L = 50 -> 4)
}

// lines whose y pass runs the wide kernel (two 8-line groups per workgroup)
template <int NC>
constexpr bool wide_transpose() {
#ifdef VPS_Y_NO_WIDE
  return false;
#else
  // 1024: the 16-line tile fits LDS as ONE 1024-thread workgroup per CU (1.80 ms per 1024^3 launch); as two groups of 8 it is
  // two persistent 512-thread workgroups per CU: 1.58 ms.  512 (32 lines, 256-byte segments): 0.20 -> 0.21 ms, not taken.
  // 2000: 800 threads x 2 x 20 points do not fit 128 VGPRs.
  return NC == 1024 || NC == 1536 || NC == 2048 || NC == 4096;
#endif
}

template <int NC, int T>
size_t transpose_lds_bytes() {
  typedef PlanInfo<NC> PI;
  size_t lines = (size_t)T * PI::PITCH;
  size_t tr = (size_t)NC * T;
  return (PI::TWL + (lines > tr ? lines : tr)) * sizeof(cf);
}

template <int NC, bool REAL>
int launch_transpose(vps_ctx* ctx, const PassParams& p, int kind) {
  // 1024-point complex lines: 16-line tiles (128-byte segments, one 1024-thread workgroup per
  // CU) measured 13 % faster than 8-line tiles; for the packed-real 1024-point z pass the
  // 8-line tiles with XCD-paired placement are faster.
  // (4-line tiles with four-way XCD grouping were slower at 2048: 2.59 vs 2.25 ms.)
  // (Persistent workgroups that request the next tile's lines before transforming the current one -- to overlap load,
  // transform and store where only ONE workgroup fits a CU -- measured at 2048^3: at the plan's 1024 threads the 32
  // prefetch registers spill (128-VGPR cap); on half the lanes per line (512 threads, 256 VGPRs, wave-level exchanges)
  // still 124 bytes per lane of scratch and 164 ms per step of y passes against 113 ms for this kernel.  This kernel itself
  // on half the lanes (512 threads, 197 VGPRs, no spill, wave-level exchanges): 121.5 against 115.6 ms.  The persistent
  // form at the plan's 1024 threads with the lane indices re-materialised per tile (117 VGPRs, no spill): 137.9 ms.  Not kept.)
  typedef PlanInfo<NC> PI;
  constexpr int T = (NC == 1024 && !REAL) ? 16 : transpose_T<NC>();
  // (single planes -- the Nyquist plane's own launch -- stay with the narrow kernel: nothing to gain there, and the
  // profiler's per-kernel averages then describe the main launches only)
  if constexpr (!REAL && wide_transpose<NC>()) if (p.B > 1) {
    // 2 x T lines per workgroup, 128-byte output segments (fft_transpose_pass_wide)
    constexpr int TGW = transpose_T<NC>();   // lines per group
    const size_t lds = transpose_lds_bytes<NC, TGW>();
    // full 128-byte lines written once: non-temporal stores (2048^3: 3.93 against 4.56 ms per 512-row slab; with the 64-byte
    // segments of the 8-line kernel they were a loss on images this large).  On half the plan's lanes per line (512 threads)
    // the kernel spills 47 registers.
    constexpr int LW = PI::L;
    auto kern = fft_transpose_pass_wide<NC, TGW, LW, true>;
    if (lds > 64 * 1024)
      VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long tiles = (p.A + 2 * TGW - 1) / (2 * TGW);
    long long grid = tiles * p.B;
    if (grid <= 0 || grid > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_ARG, "fft grid out of range");
    // as many workgroups as fit the chip at once (one per CU at 2048), each walking its share of the tiles: 2048^3 launch
    // 13.4 -> 12.4 ms, a 256-row slab 1.80 -> 1.63 ms against one workgroup per tile
    const long long per_cu = (long long)(ctx->lds_per_cu / lds) > 0 ? (long long)(ctx->lds_per_cu / lds) : 1;
    if (grid > (long long)ctx->num_cu * per_cu) grid = (long long)ctx->num_cu * per_cu;
    {
      vps_launch_timer tm(ctx, kind);
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TGW * LW), lds, ctx->stream, p);
    }
    VPS_HIP_CHECK(ctx, hipGetLastError());
    return VPS_OK;
  }
  const size_t lds = transpose_lds_bytes<NC, T>();
  auto kern = fft_transpose_pass<NC, T, REAL>;
  if constexpr (!REAL) {
    const double image_bytes = 8.0 * (double)p.A * (double)p.B * (double)NC;
    // non-temporal loads and stores: always with full 128-byte segments (T >= 16: 1024^3 y pass 1.79 against 1.90 ms);
    // with narrower tiles only while the image is small (partial lines have to meet in L2 first)
    if (image_bytes <= 768.0 * 1048576.0 || T >= 16) kern = fft_transpose_pass<NC, T, REAL, true>;
    if (p.kg) kern = fft_transpose_pass<NC, T, REAL, true, true>;   // Nyquist plane of a chunked exchange (one small image)
  }
  if (lds > 64 * 1024)
    VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long tiles = (p.A + T - 1) / T;
  const long long grid = tiles * p.B;
  if (grid <= 0 || grid > 0x7fffffffLL) return vps_fail(ctx, VPS_ERR_ARG, "fft grid out of range");
  {
    vps_launch_timer tm(ctx, kind);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T * PI::L), lds, ctx->stream, p);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

template <int NC, int MODE, bool COUNT = false>
int launch_x(vps_ctx* ctx, const XParams& p_in, int fast = 0) {   // fast: 0 general shell walk, 1 mirrored kx (float64), 2 integer shells
  XParams p = p_in;
  constexpr int T = xpass_T<NC>();
  typedef PlanInfo<NC> PI;
  if (!(MODE == 0 && NC >= 32)) fast = 0;
  const bool seg = p.seglen != NC;
  const bool twreg = fast == 2 && x_twreg<NC, 2, false>();
  size_t lds = ((twreg ? ((PI::TW1 + 1) & ~1) : PI::TWL) + (size_t)T * PI::PITCH) * sizeof(cf);
  if (MODE == 0) lds += (fast == 2 ? (size_t)((p.nbins + 3) & ~1) * sizeof(unsigned) : (size_t)(p.nbins + 2) * sizeof(double)) +
                        (size_t)p.nbins * sizeof(double) + (COUNT ? (size_t)p.nbins * sizeof(unsigned) : 0) +
                        (p.win ? (size_t)NC * sizeof(float) : 0);
  if (lds > ctx->lds_per_cu) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "x pass needs %zu B LDS", lds);
  if (seg && (NC & (NC - 1)) == 0 && PI::L % 64 == 0 && (p.seg_shift < 0 || p.seglen < PI::L))
    return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "x pass: lines of %d points in segments of %d (more than %d ranks)", NC, p.seglen, NC / PI::L);
  auto kern = seg ? fft_x_pass<NC, T, MODE, true, COUNT, 0> : fft_x_pass<NC, T, MODE, false, COUNT, 0>;
  if constexpr (MODE == 0 && NC >= 32) {
    if (fast == 1) kern = seg ? fft_x_pass<NC, T, MODE, true, COUNT, 1> : fft_x_pass<NC, T, MODE, false, COUNT, 1>;
    if (fast == 2) kern = seg ? fft_x_pass<NC, T, MODE, true, COUNT, 2> : fft_x_pass<NC, T, MODE, false, COUNT, 2>;
  }
  if (lds > 64 * 1024)
    VPS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long ntiles = (p.nlines + T - 1) / T;
  long long per_cu = (long long)(ctx->lds_per_cu / (lds ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  {
    const long long want = (long long)vps_option("x_wg_per_cu", 0);
    if (want >= 1 && want < per_cu) per_cu = want;
  }
  long long grid = (long long)ctx->num_cu * per_cu;
  if (grid > ntiles) grid = ntiles;
  if (grid < 1) return VPS_OK;
  if (MODE == 0) {
    const size_t need = (size_t)grid * p.nbins * (sizeof(double) + sizeof(unsigned));
    if (need > ctx->xpart_cap) {
      VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_xpart) VPS_HIP_CHECK(ctx, hipFree(ctx->d_xpart));
      ctx->d_xpart = nullptr;
      ctx->xpart_cap = 0;
      VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_xpart, need));
      ctx->xpart_cap = need;
    }
    p.part_sum = reinterpret_cast<double*>(ctx->d_xpart);
    p.part_cnt = reinterpret_cast<unsigned*>(p.part_sum + (size_t)grid * p.nbins);
  }
  {
    vps_launch_timer tm(ctx, VPS_K_FFT_X);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T * PI::L), lds, ctx->stream, p);
    if (MODE == 0)
      hipLaunchKernelGGL(reduce_partials, dim3((unsigned)((p.nbins + 255) / 256), grid >= 64 ? 32u : 1u),
                         dim3(256), 0, ctx->stream,
                         p.part_sum, COUNT ? p.part_cnt : nullptr, (int)grid, p.nbins, p.psum, p.nsample);
  }
  VPS_HIP_CHECK(ctx, hipGetLastError());
  return VPS_OK;
}

// ---- translation-unit split ---------------------------------------------------------------
// Every line length instantiates ~20 kernels, and 21 lengths in one translation unit take minutes to
// compile.  The build therefore compiles this file once per length FAMILY (-DVPS_FFT_PART=0..3): each
// part instantiates the launchers of its own lengths behind the five vps_fftpart_* entry points below,
// and part 0 also holds the API, the tables and the routing.  Without the macro (VPS_FFT_PART = -1:
// tools/build_variant.sh, single-file builds) one unit holds everything.
#ifndef VPS_FFT_PART
#define VPS_FFT_PART -1
#endif
#define VPS_PART_NOT_MINE (-12345)
#define VPS_CAT2(a, b) a##b
#define VPS_CAT(a, b) VPS_CAT2(a, b)
#if VPS_FFT_PART == -1
#define VPS_PARTFN(name) VPS_CAT(name, _all)
#else
#define VPS_PARTFN(name) VPS_CAT(VPS_CAT(name, _p), VPS_FFT_PART)
#endif

#if VPS_FFT_PART == -1 || VPS_FFT_PART == 0
#define VPS_FAMILY_0(CALL) case 8: { constexpr int NC_ = 8; CALL; } break; case 16: { constexpr int NC_ = 16; CALL; } break; case 32: { constexpr int NC_ = 32; CALL; } break; case 64: { constexpr int NC_ = 64; CALL; } break; case 128: { constexpr int NC_ = 128; CALL; } break; case 256: { constexpr int NC_ = 256; CALL; } break;
#else
#define VPS_FAMILY_0(CALL)
#endif
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 1
#define VPS_FAMILY_1(CALL) case 512: { constexpr int NC_ = 512; CALL; } break; case 1024: { constexpr int NC_ = 1024; CALL; } break; case 2048: { constexpr int NC_ = 2048; CALL; } break; case 4096: { constexpr int NC_ = 4096; CALL; } break;
#else
#define VPS_FAMILY_1(CALL)
#endif
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 2
#define VPS_FAMILY_2(CALL) case 48: { constexpr int NC_ = 48; CALL; } break; case 96: { constexpr int NC_ = 96; CALL; } break; case 192: { constexpr int NC_ = 192; CALL; } break; case 384: { constexpr int NC_ = 384; CALL; } break; case 768: { constexpr int NC_ = 768; CALL; } break; case 1536: { constexpr int NC_ = 1536; CALL; } break;
#else
#define VPS_FAMILY_2(CALL)
#endif
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 3
#define VPS_FAMILY_3(CALL) case 125: { constexpr int NC_ = 125; CALL; } break; case 250: { constexpr int NC_ = 250; CALL; } break; case 500: { constexpr int NC_ = 500; CALL; } break; case 1000: { constexpr int NC_ = 1000; CALL; } break; case 2000: { constexpr int NC_ = 2000; CALL; } break;
#else
#define VPS_FAMILY_3(CALL)
#endif

#define VPS_DISPATCH_NC(NCVAL, CALL)                                  \
  switch (NCVAL) {                                                    \
    VPS_FAMILY_0(CALL)                                               \
    VPS_FAMILY_1(CALL)                                               \
    VPS_FAMILY_2(CALL)                                               \
    VPS_FAMILY_3(CALL)                                               \
    default: rc = VPS_PART_NOT_MINE;                                   \
  }


// host-side twiddle image for complex length NC (double precision, rounded once)
template <int NC>
void build_stage_tw(std::vector<cf>& out) {
  typedef PlanInfo<NC> PI;
  out.assign(PI::TW > 0 ? PI::TW : 1, make_float2(1.f, 0.f));
  const double tp = -2.0 * 3.14159265358979323846;
  if (PI::R1 > 1)
    for (int r = 1; r < PI::R1; ++r)
      for (int k = 0; k < PI::NS1; ++k) {
        double a = tp * (double)r * (double)k / (double)(PI::NS1 * PI::R1);
        out[(r - 1) * PI::NS1 + k] = make_float2((float)cos(a), (float)sin(a));
      }
  if (PI::R2 > 1)
    for (int r = 1; r < PI::R2; ++r)
      for (int k = 0; k < PI::NS2; ++k) {
        double a = tp * (double)r * (double)k / (double)(PI::NS2 * PI::R2);
        out[PI::TW1 + (r - 1) * PI::NS2 + k] = make_float2((float)cos(a), (float)sin(a));
      }
}

}  // namespace

// ---- per-part entry points (this unit's line lengths; VPS_PART_NOT_MINE for the others) ----
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 0
#define VPS_PENCIL_FAMILY_0(CALL) case 32: { constexpr int NC_ = 32; CALL; } break; case 64: { constexpr int NC_ = 64; CALL; } break; case 128: { constexpr int NC_ = 128; CALL; } break; case 256: { constexpr int NC_ = 256; CALL; } break;
#else
#define VPS_PENCIL_FAMILY_0(CALL)
#endif
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 1
#define VPS_PENCIL_FAMILY_1(CALL) case 512: { constexpr int NC_ = 512; CALL; } break; case 1024: { constexpr int NC_ = 1024; CALL; } break; case 2048: { constexpr int NC_ = 2048; CALL; } break;
#else
#define VPS_PENCIL_FAMILY_1(CALL)
#endif
#if VPS_FFT_PART == -1 || VPS_FFT_PART == 2
#define VPS_PENCIL_FAMILY_2(CALL) case 96: { constexpr int NC_ = 96; CALL; } break; case 192: { constexpr int NC_ = 192; CALL; } break; case 384: { constexpr int NC_ = 384; CALL; } break; case 768: { constexpr int NC_ = 768; CALL; } break;
#else
#define VPS_PENCIL_FAMILY_2(CALL)
#endif
#define VPS_DISPATCH_PENCIL(NCVAL, CALL) \
  switch (NCVAL) {                       \
    VPS_PENCIL_FAMILY_0(CALL)            \
    VPS_PENCIL_FAMILY_1(CALL)            \
    VPS_PENCIL_FAMILY_2(CALL)            \
    default: rc = VPS_PART_NOT_MINE;     \
  }

int VPS_PARTFN(vps_fftpart_tw)(int NC, std::vector<cf>* st) {
  int rc = VPS_OK;
  VPS_DISPATCH_NC(NC, build_stage_tw<NC_>(*st));
  return rc;
}

int VPS_PARTFN(vps_fftpart_transpose)(vps_ctx* ctx, int NC, int real, const void* params, int kind) {
  const PassParams& p = *static_cast<const PassParams*>(params);
  int rc = VPS_OK;
  if (real) {
    VPS_DISPATCH_NC(NC, (rc = launch_transpose<NC_, true>(ctx, p, kind)));
  } else {
    VPS_DISPATCH_NC(NC, (rc = launch_transpose<NC_, false>(ctx, p, kind)));
  }
  return rc;
}

int VPS_PARTFN(vps_fftpart_x)(vps_ctx* ctx, int NC, int mode, int count, const void* params, int fast) {
  const XParams& p = *static_cast<const XParams*>(params);
  int rc = VPS_OK;
  if (mode == 0 && count) {
    VPS_DISPATCH_NC(NC, (rc = launch_x<NC_, 0, true>(ctx, p, fast)));
  } else if (mode == 0) {
    VPS_DISPATCH_NC(NC, (rc = launch_x<NC_, 0, false>(ctx, p, fast)));
  } else if (mode == 1) {
    VPS_DISPATCH_NC(NC, (rc = launch_x<NC_, 1>(ctx, p)));
  } else if (mode == 2) {
    VPS_DISPATCH_NC(NC, (rc = launch_x<NC_, 2>(ctx, p)));
  } else {
    VPS_DISPATCH_NC(NC, (rc = launch_x<NC_, 4>(ctx, p)));
  }
  return rc;
}

long long VPS_PARTFN(vps_fftpart_pencil_lds)(int NC) {
  long long rc = VPS_PART_NOT_MINE;
  long long lds = -1;
  VPS_DISPATCH_PENCIL(NC, (lds = (long long)pencil_lds_bytes<NC_>(), rc = 0));
  return rc == 0 ? lds : (long long)VPS_PART_NOT_MINE;
}

int VPS_PARTFN(vps_fftpart_pencil)(vps_ctx* ctx, int NC, const void* params, long long npencils) {
  const PencilParams& p = *static_cast<const PencilParams*>(params);
  int rc = VPS_OK;
  VPS_DISPATCH_PENCIL(NC, (rc = launch_pencil<NC_>(ctx, p, npencils)));
  return rc;
}

#if VPS_FFT_PART <= 0   // ---- API, tables and routing: part 0 (or the single-unit build) only ----
#if VPS_FFT_PART == -1
#define VPS_FOR_PARTS(X) X(_all)
#else
#define VPS_FOR_PARTS(X) X(_p0) X(_p1) X(_p2) X(_p3)
#endif
#define VPS_DECL_PART(sfx)                                                                         \
  int VPS_CAT(vps_fftpart_tw, sfx)(int, std::vector<cf>*);                                          \
  int VPS_CAT(vps_fftpart_transpose, sfx)(vps_ctx*, int, int, const void*, int);                    \
  int VPS_CAT(vps_fftpart_x, sfx)(vps_ctx*, int, int, int, const void*, int);                       \
  long long VPS_CAT(vps_fftpart_pencil_lds, sfx)(int);                                              \
  int VPS_CAT(vps_fftpart_pencil, sfx)(vps_ctx*, int, const void*, long long);
VPS_FOR_PARTS(VPS_DECL_PART)

static int route_tw(vps_ctx* ctx, int NC, std::vector<cf>* st) {
  int rc;
#define VPS_TRY(sfx) if ((rc = VPS_CAT(vps_fftpart_tw, sfx)(NC, st)) != VPS_PART_NOT_MINE) return rc;
  VPS_FOR_PARTS(VPS_TRY)
#undef VPS_TRY
  return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "unsupported FFT length %d", NC);
}
static int route_transpose(vps_ctx* ctx, int NC, int real, const PassParams& p, int kind) {
  int rc;
#define VPS_TRY(sfx) if ((rc = VPS_CAT(vps_fftpart_transpose, sfx)(ctx, NC, real, &p, kind)) != VPS_PART_NOT_MINE) return rc;
  VPS_FOR_PARTS(VPS_TRY)
#undef VPS_TRY
  return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "unsupported FFT length %d", NC);
}
static int route_x(vps_ctx* ctx, int NC, int mode, int count, const XParams& p, int fast) {
  int rc;
#define VPS_TRY(sfx) if ((rc = VPS_CAT(vps_fftpart_x, sfx)(ctx, NC, mode, count, &p, fast)) != VPS_PART_NOT_MINE) return rc;
  VPS_FOR_PARTS(VPS_TRY)
#undef VPS_TRY
  return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "unsupported FFT length %d", NC);
}
static long long route_pencil_lds(int NC) {
  long long r;
#define VPS_TRY(sfx) if ((r = VPS_CAT(vps_fftpart_pencil_lds, sfx)(NC)) != VPS_PART_NOT_MINE) return r;
  VPS_FOR_PARTS(VPS_TRY)
#undef VPS_TRY
  return -1;
}
static int route_pencil(vps_ctx* ctx, int NC, const PencilParams& p, long long npencils) {
  int rc;
#define VPS_TRY(sfx) if ((rc = VPS_CAT(vps_fftpart_pencil, sfx)(ctx, NC, &p, npencils)) != VPS_PART_NOT_MINE) return rc;
  VPS_FOR_PARTS(VPS_TRY)
#undef VPS_TRY
  return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "pencil path: no kernel for lines of %d points", NC);
}

int vps_fft_get_tables(vps_ctx* ctx, int NC, vps_fft_tables* out) {
  auto it = ctx->fft_tables.find(NC);
  if (it != ctx->fft_tables.end()) {
    *out = it->second;
    return VPS_OK;
  }
  std::vector<cf> st;
  int rc = route_tw(ctx, NC, &st);
  if (rc) return rc;
  std::vector<cf> r2c(NC);
  for (int k = 0; k < NC; ++k) {
    double a = -2.0 * 3.14159265358979323846 * (double)k / (double)(2 * NC);
    r2c[k] = make_float2((float)cos(a), (float)sin(a));
  }
  vps_fft_tables t;
  VPS_HIP_CHECK(ctx, hipMalloc(&t.tw_stage, st.size() * sizeof(cf)));
  VPS_HIP_CHECK(ctx, hipMalloc(&t.tw_r2c, r2c.size() * sizeof(cf)));
  VPS_HIP_CHECK(ctx, hipMemcpy(t.tw_stage, st.data(), st.size() * sizeof(cf), hipMemcpyHostToDevice));
  VPS_HIP_CHECK(ctx, hipMemcpy(t.tw_r2c, r2c.data(), r2c.size() * sizeof(cf), hipMemcpyHostToDevice));
  ctx->fft_tables[NC] = t;
  *out = t;
  return VPS_OK;
}

void vps_fft_free_tables(vps_ctx* ctx) {
  for (auto& kv : ctx->fft_tables) {
    (void)hipFree(kv.second.tw_stage);
    (void)hipFree(kv.second.tw_r2c);
  }
  ctx->fft_tables.clear();
}

static int fft_y_of(vps_ctx* ctx, int N, int nx, const cf* B, const cf* BN, void* spec_dev, void* nyq_dev);

extern "C" {

int vps_fft_supported(int N) {
  if (N == 250 || N == 500 || N == 1000 || N == 2000) return 1;   // 2^a 5^b plans (radix 5 / 10 / 20)
  if (N == 96 || N == 192 || N == 384 || N == 768 || N == 1536) return 1;   // 3 * 2^a plans (radix 3 / 6 / 12 / 24)
  return (N >= 16 && N <= 4096 && (N & (N - 1)) == 0) ? 1 : 0;
}

size_t vps_fft_workspace_bytes(int N, int nx) {
  // B[x][kz][y] (kz < N/2) + Nyquist plane BN[x][y]
  return ((size_t)nx * (size_t)(N / 2) * (size_t)N + (size_t)nx * (size_t)N) * sizeof(cf);
}

size_t vps_power_workspace_bytes(int N) {
  // z-pass image + y-pass image (each with its Nyquist plane)
  return 2 * vps_fft_workspace_bytes(N, N);
}

int vps_fft_zy_weighted(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev,
                        void* spec_dev, void* nyq_dev, void* work_dev);

int vps_fft_zy(vps_ctx* ctx, int N, int nx, const float* field_dev, void* spec_dev, void* nyq_dev,
               void* work_dev) {
  return vps_fft_zy_weighted(ctx, N, nx, field_dev, nullptr, spec_dev, nyq_dev, work_dev);
}

static int fft_z_of(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev, cf* B, cf* BN) {
  const int NH = N / 2;
  vps_fft_tables tz;
  int rc = vps_fft_get_tables(ctx, NH, &tz);
  if (rc) return rc;
  // z pass: lines (a = y, b = x) of R[x][y][:] -> B[x][kz][y], BN[x][y]
  PassParams pz{};
  pz.in = field_dev;
  pz.in_w = weight_dev;
  pz.out = B;
  pz.out_nyq = BN;
  pz.in_sa = N;
  pz.in_sb = (long long)N * N;
  pz.out_ob = (long long)NH * N;
  pz.out_ok = N;
  pz.nyq_ob = N;
  pz.A = N;
  pz.B = nx;
  pz.tw_stage = tz.tw_stage;
  pz.tw_r2c = tz.tw_r2c;
  return route_transpose(ctx, NH, 1, pz, VPS_K_FFT_Z);
}

int vps_fft_zy_weighted(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev,
                        void* spec_dev, void* nyq_dev, void* work_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "N=%d: need a power of two in [16,4096], 96, 192, 384, 768, 1536, 250, 500, 1000 or 2000", N);
  if (nx < 1 || nx > N) return vps_fail(ctx, VPS_ERR_ARG, "nx=%d out of range", nx);
  if (!field_dev || !spec_dev || !nyq_dev || !work_dev) return vps_fail(ctx, VPS_ERR_ARG, "null buffer");
  cf* B = reinterpret_cast<cf*>(work_dev);
  cf* BN = B + (size_t)nx * (N / 2) * N;
  int rc = fft_z_of(ctx, N, nx, field_dev, weight_dev, B, BN);
  if (rc) return rc;
  // y pass: lines (a = x, b = kz) of B[x][kz][:] -> C[kz][ky][x]; Nyquist plane BN[x][:] -> CN[ky][x]
  return fft_y_of(ctx, N, nx, B, BN, spec_dev, nyq_dev);
}

// ---- the split form for the chunked slab exchange: z pass into an image, y pass one kz chunk at a time ----
size_t vps_fft_zimage_bytes(int N, int nx) { return vps_fft_workspace_bytes(N, nx); }

int vps_fft_z(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev, void* zimg_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_fft_z: unsupported N=%d", N);
  if (nx < 1 || nx > N) return vps_fail(ctx, VPS_ERR_ARG, "nx=%d out of range", nx);
  if (!field_dev || !zimg_dev) return vps_fail(ctx, VPS_ERR_ARG, "null buffer");
  cf* B = reinterpret_cast<cf*>(zimg_dev);
  return fft_z_of(ctx, N, nx, field_dev, weight_dev, B, B + (size_t)nx * (N / 2) * N);
}

int64_t vps_fft_y_chunk_elems(int N, int nx, int G, int nchunks, int chunk) {
  if (G < 1 || nchunks < 1 || chunk < 0 || chunk >= nchunks || N % G || (N / 2) % (G * nchunks)) return -1;
  const int64_t nkc = N / 2 / G / nchunks;
  return (int64_t)G * (nkc * N * nx + (chunk == nchunks - 1 ? (int64_t)(N / G) * nx : 0));
}

// ---- layout of the chunked slab exchange --------------------------------------------------------------------------------
// The kz < N/2 planes are cut into `nchunks` bands of G * nkc planes (nkc = N/2/G/nchunks); inside band c the planes are dealt
// out round-robin: slot j of rank h is plane c*G*nkc + j*G + h.  So (a) every rank holds planes of every |kz| -- the same
// share of rows its x passes can skip -- and (b) the j-th planes of all ranks are neighbours in kz and need the same rows to
// within one step of the cut: packed, every destination's block carries, for slot j, the 2 kc_j + 1 rows |ky| <= kc_j with
// kc_j = the largest kcut of the G planes of slot j.  Equal blocks for all destinations, a table of nkc entries per chunk.
static bool ypack_wanted(const vps_ctx* ctx, int N) {
  return ctx->bin_only && ctx->bin_N == N && (int)ctx->h_kcut.size() == N / 2 + 1;
}

// plane table of chunk `chunk` (device) and the rows of one destination's block; (re)built for (N, G, nchunks, packed)
static int ypack_get(vps_ctx* ctx, int N, int G, int nchunks, int chunk, bool packed, const int2** tab, long long* rows) {
  if (G < 1 || nchunks < 1 || N < 2 || (N / 2) % (G * nchunks) || chunk < 0 || chunk >= nchunks)
    return vps_fail(ctx, VPS_ERR_ARG, "chunked exchange: G=%d ranks x %d chunks must divide N/2=%d", G, nchunks, N / 2);
  if (packed && (int)ctx->h_kcut.size() != N / 2 + 1)
    return vps_fail(ctx, VPS_ERR_ARG, "chunked exchange: packed rows need the row cut of vps_set_binning(N=%d)", N);
  const int nkc = N / 2 / G / nchunks;
  auto& y = ctx->ypack;
  if (!(y.N == N && y.G == G && y.C == nchunks && y.packed == (int)packed && y.d_tab)) {
    std::vector<int2> t((size_t)nchunks * nkc);
    y.rows.assign(nchunks, 0);
    for (int c = 0; c < nchunks; ++c) {
      long long r = 0;
      for (int j = 0; j < nkc; ++j) {
        int kc = -1;
        if (packed) {
          kc = 0;
          for (int h = 0; h < G; ++h) kc = std::max(kc, ctx->h_kcut[c * G * nkc + j * G + h]);
          if (2 * kc + 1 >= N) kc = -1;
        }
        t[(size_t)c * nkc + j] = make_int2((int)r, kc);
        r += kc < 0 ? N : 2 * kc + 1;
      }
      y.rows[c] = r;
    }
    VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // (a launch may still read the old table)
    if (y.d_tab) VPS_HIP_CHECK(ctx, hipFree(y.d_tab));
    y.d_tab = nullptr;
    y.N = 0;
    VPS_HIP_CHECK(ctx, hipMalloc(&y.d_tab, t.size() * sizeof(int2)));
    VPS_HIP_CHECK(ctx, hipMemcpy(y.d_tab, t.data(), t.size() * sizeof(int2), hipMemcpyHostToDevice));
    y.N = N; y.G = G; y.C = nchunks; y.packed = (int)packed;
  }
  if (tab) *tab = reinterpret_cast<const int2*>(y.d_tab) + (size_t)chunk * nkc;
  if (rows) *rows = y.rows[chunk];
  return VPS_OK;
}

int vps_fft_y_packed(vps_ctx* ctx, int N) { return ctx && ypack_wanted(ctx, N) ? 1 : 0; }

// A pure size query: the rows of a block follow from the host copy of the row cut; no device table is built or touched
// (ypack_get, which the y / x passes call, keeps its one-entry cache for the launch that follows).
// Returns -1: bad arguments, -2: G x nchunks does not divide N/2 (or G does not divide N), -3: packed without a row cut.
int64_t vps_fft_y_chunk_block(vps_ctx* ctx, int N, int nx, int G, int nchunks, int chunk, int packed) {
  if (!ctx || nx < 1 || N < 2) return -1;
  if (G < 1 || nchunks < 1 || chunk < 0 || chunk >= nchunks) return -1;
  if (N % G || (N / 2) % (G * nchunks)) {
    vps_fail(ctx, VPS_ERR_ARG, "chunked exchange: G=%d ranks x %d chunks must divide N/2=%d (and G divide N)", G, nchunks, N / 2);
    return -2;
  }
  if (packed && (int)ctx->h_kcut.size() != N / 2 + 1) {
    vps_fail(ctx, VPS_ERR_ARG, "chunked exchange: packed rows need the row cut of vps_set_binning(N=%d)", N);
    return -3;
  }
  const int nkc = N / 2 / G / nchunks;
  long long rows = 0;
  for (int j = 0; j < nkc; ++j) {
    int kc = -1;
    if (packed) {
      kc = 0;
      for (int h = 0; h < G; ++h) kc = std::max(kc, ctx->h_kcut[chunk * G * nkc + j * G + h]);
      if (2 * kc + 1 >= N) kc = -1;
    }
    rows += kc < 0 ? N : 2 * kc + 1;
  }
  return rows * nx + (chunk == nchunks - 1 ? (long long)(N / G) * nx : 0);
}

int vps_fft_y(vps_ctx* ctx, int N, int nx, const void* zimg_dev, int G, int nchunks, int chunk, void* out_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "vps_fft_y: unsupported N=%d", N);
  if (nx < 1 || nx > N || !zimg_dev || !out_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_y: bad nx / null buffer");
  if (vps_fft_y_chunk_elems(N, nx, G, nchunks, chunk) < 0)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_y: G=%d ranks x %d chunks must divide N/2=%d (and G divide N)", G, nchunks, N / 2);
  const int NH = N / 2, nkz = NH / G, nkc = nkz / nchunks, nky = N / G;
  const bool last = chunk == nchunks - 1;
  vps_fft_tables ty;
  int rc = vps_fft_get_tables(ctx, N, &ty);
  if (rc) return rc;
  const cf* B = reinterpret_cast<const cf*>(zimg_dev);
  const cf* BN = B + (size_t)nx * NH * N;
  const int2* tab = nullptr;
  long long rows = 0;   // rows of one destination's block (N per plane when not packed)
  rc = ypack_get(ctx, N, G, nchunks, chunk, ypack_wanted(ctx, N), &tab, &rows);
  if (rc) return rc;
  const long long blk = rows * nx + (last ? (long long)nky * nx : 0);   // one destination's block
  (void)nkz;
  // launch batch b = h nkc + j (destination h, slot j) reads plane chunk G nkc + j G + h and writes the rows of slot j
  PassParams py{};
  py.in = B;
  py.out = out_dev;
  py.in_sa = (long long)NH * N;
  py.in_sb = N;
  py.out_ob = 0;
  py.out_ok = nx;
  py.A = nx;
  py.B = G * nkc;
  py.tw_stage = ty.tw_stage;
  py.bg = nkc;
  py.b_off = chunk * G * nkc;
  py.bg_in = 1;
  py.bg_step = G;
  py.bg_gap = blk;
  py.ptab = tab;
  py.kcut = (ctx->bin_only && ctx->bin_N == N) ? ctx->d_kcut : nullptr;
  py.kz_fixed = -1;
  rc = route_transpose(ctx, N, 0, py, VPS_K_FFT_Y);
  if (rc || !last) return rc;
  // Nyquist plane: ky rows of destination h go behind that destination's kz rows
  PassParams pn{};
  pn.in = BN;
  pn.out = reinterpret_cast<cf*>(out_dev) + rows * nx;
  pn.in_sa = N;
  pn.in_sb = 0;
  pn.out_ob = 0;
  pn.out_ok = nx;
  pn.A = nx;
  pn.B = 1;
  pn.tw_stage = ty.tw_stage;
  pn.kg = nky;
  pn.kg_gap = blk - (long long)nky * nx;
  pn.kcut = py.kcut;
  pn.kz_fixed = NH;
  return route_transpose(ctx, N, 0, pn, VPS_K_FFT_Y);
}

}  // extern "C"

// y pass of one field: B[x][kz][y] (+BN[x][y]) -> spec[kz][ky][x] (+nyq[ky][x])
static int fft_y_of(vps_ctx* ctx, int N, int nx, const cf* B, const cf* BN, void* spec_dev, void* nyq_dev) {
  const int NH = N / 2;
  vps_fft_tables ty;
  int rc = vps_fft_get_tables(ctx, N, &ty);
  if (rc) return rc;
  PassParams py{};
  py.in = B;
  py.out = spec_dev;
  py.in_sa = (long long)NH * N;
  py.in_sb = N;
  py.out_ob = (long long)N * nx;
  py.out_ok = nx;
  py.A = nx;
  py.B = NH;
  py.tw_stage = ty.tw_stage;
  py.kcut = (ctx->bin_only && ctx->bin_N == N) ? ctx->d_kcut : nullptr;
  py.kz_fixed = -1;
  rc = route_transpose(ctx, N, 0, py, VPS_K_FFT_Y);
  if (rc) return rc;
  PassParams pn = py;
  pn.kz_fixed = NH;
  pn.in = BN;
  pn.out = nyq_dev;
  pn.in_sa = N;
  pn.in_sb = 0;
  pn.out_ob = 0;
  pn.B = 1;
  rc = route_transpose(ctx, N, 0, pn, VPS_K_FFT_Y);
  return rc;
}

int vps_pencil_tp(int N) { return N / 2 >= 1024 ? VPS_PENCIL_TP_LONG : 16; }   // = pencil_tp<N/2>()

bool vps_pencil_supported(vps_ctx* ctx, int N) {
  if (!vps_fft_supported(N) || N < 64 || N > 4096) return false;   // (4096: 8-line pencils of 2048 packed points on 1024 threads, 148 KB of LDS)
  const long long lds = route_pencil_lds(N / 2);
  if (lds < 0) return false;
  return (size_t)lds <= ctx->lds_per_cu;
}

// records sorted by pencil -> ncomp half spectra after the z and y passes (spec_dev == NULL: z pass only, the
// z images [component][B | BN] stay in bwork_dev)
int vps_fft_pencil_zy(vps_ctx* ctx, int N, int nx, const unsigned* records, const unsigned* start, float* side,
                      int ncomp, const int* chan, int divide, int energy, float vol, void* spec_dev, void* nyq_dev,
                      void* bwork_dev, int with_energy) {
  // with_energy = 1: a momentum launch (three components) that leaves the energy field's z image as component 3 of bwork_dev;
  // with_energy = 2: no launch -- the y pass of that component 3 (the energy quantity of a step whose momentum launch made it)
  if (with_energy && (energy != (with_energy == 2) || divide != (with_energy == 2 ? 1 : 0) || (with_energy == 1 && ncomp != 3)))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_pencil_zy: the shared energy field needs an undivided three-component momentum launch");
  const int NH = N / 2;
  vps_fft_tables tz;
  int rc = vps_fft_get_tables(ctx, NH, &tz);
  if (rc) return rc;
  const size_t bfield = (size_t)nx * NH * N, bnyq = (size_t)nx * N;
  cf* Bbase = reinterpret_cast<cf*>(bwork_dev);
  PencilParams p{};
  p.records = records;
  p.start = start;
  p.side = side;
  p.N = N;
  p.nx = nx;
  p.nby = N / vps_pencil_tp(N);
  p.ncomp = ncomp;
  for (int c = 0; c < 3; ++c) p.chan[c] = chan[c < ncomp ? c : 0];
  for (int c = 0; c < 4; ++c) {
    p.out[c] = Bbase + (size_t)c * (bfield + bnyq);
    p.nyq[c] = p.out[c] + bfield;
  }
  p.divide = divide;
  p.energy = energy;
  p.with_energy = with_energy == 1;
  p.vol = vol;
  p.tw_stage = tz.tw_stage;
  p.tw_r2c = tz.tw_r2c;
  const long long npencils = (long long)nx * p.nby;
  if (with_energy != 2) {
    rc = route_pencil(ctx, NH, p, npencils);
    if (rc) return rc;
  }
  if (!spec_dev) return VPS_OK;
  cf* spec = reinterpret_cast<cf*>(spec_dev);
  cf* nyq = reinterpret_cast<cf*>(nyq_dev);
  if (with_energy == 2)      // the z image the momentum launch left as component 3
    return fft_y_of(ctx, N, nx, p.out[3], p.nyq[3], spec, nyq);
  const int nout = energy ? 1 : ncomp;
  for (int c = 0; c < nout; ++c) {
    rc = fft_y_of(ctx, N, nx, p.out[c], p.nyq[c], spec + (size_t)c * NH * N * nx, nyq + (size_t)c * N * nx);
    if (rc) return rc;
  }
  return VPS_OK;
}

extern "C" {

static int fft_x_impl(vps_ctx* ctx, int N, int64_t nlines, int64_t line0, int kz0, const void* in_dev,
                      const void* in1_dev, const void* in2_dev, int ncomp, int nseg, int64_t seg_stride,
                      int mode, double* psum_dev, unsigned long long* nsample_dev, void* out_dev,
                      const int2* ptab = nullptr, int kz_step = 1) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "N=%d: need a power of two in [16,4096], 96, 192, 384, 768, 1536, 250, 500, 1000 or 2000", N);
  if (nlines < 0 || !in_dev) return vps_fail(ctx, VPS_ERR_ARG, "bad line count / null input");
  if (nseg < 1 || N % nseg) return vps_fail(ctx, VPS_ERR_ARG, "nseg=%d must divide N", nseg);
  if (nlines == 0) return VPS_OK;
  vps_fft_tables tx;
  int rc = vps_fft_get_tables(ctx, N, &tx);
  if (rc) return rc;
  XParams p{};
  p.in = reinterpret_cast<const cf*>(in_dev);
  p.in1 = reinterpret_cast<const cf*>(in1_dev);
  p.in2 = reinterpret_cast<const cf*>(in2_dev);
  p.ncomp = ncomp;
  p.out = reinterpret_cast<cf*>(out_dev);
  p.nlines = nlines;
  p.line0 = line0;
  p.N = N;
  p.kz0 = kz0;
  p.seglen = N / nseg;
  p.seg_shift = 0;
  while ((1 << p.seg_shift) < p.seglen) ++p.seg_shift;
  if ((1 << p.seg_shift) != p.seglen) p.seg_shift = -1;   // not a power of two: the kernel divides
  p.seg_stride = seg_stride;
  p.tw_stage = tx.tw_stage;
  p.ptab = ptab;
  p.kz_step = kz_step;
  if (ptab && (line0 != 0 || nlines % N || (mode != 0 && mode != 3)))
    return vps_fail(ctx, VPS_ERR_ARG, "chunk planes: whole planes from line 0, binning modes only");
  if (mode == 0 || mode == 3) {
    if (ctx->bin_N != N || !ctx->d_k2) return vps_fail(ctx, VPS_ERR_ARG, "vps_set_binning(N=%d) has not been called", N);
    if (!psum_dev || (mode == 0 && !nsample_dev)) return vps_fail(ctx, VPS_ERR_ARG, "null accumulator");
    const long long maxline = line0 + nlines - 1;
    if (kz0 + (int)(maxline / N) * kz_step > N / 2) return vps_fail(ctx, VPS_ERR_ARG, "kz range exceeds N/2");
    p.k2 = ctx->d_k2;
    p.thr = ctx->d_thr;
    if (ctx->d_win && ctx->win_N != N) return vps_fail(ctx, VPS_ERR_ARG, "vps_set_window was called for N=%d, not %d", ctx->win_N, N);
    p.win = ctx->d_win;
    p.nbins = ctx->nbins;
    p.edge0 = (float)ctx->edge0;
    p.inv_spacing = (float)ctx->inv_spacing;
    p.psum = psum_dev;
    p.nsample = nsample_dev;
    p.pair = (ctx->bin_fast && line0 % N == 0 && nlines % N == 0 && vps_option("no_pair_binning", 0) == 0) ? 1 : 0;
    p.nthr = ctx->d_nthr;
    p.nmax = ctx->bin_nmax;
    p.kf = ctx->bin_kf;
    const int fastmode = !ctx->bin_fast ? 0 : (ctx->bin_int && ctx->d_nthr && vps_option("no_int_binning", 0) == 0) ? 2 : 1;
    if (mode == 0) {
      rc = route_x(ctx, N, 0, 1, p, fastmode);
    } else {
      rc = route_x(ctx, N, 0, 0, p, fastmode);
    }
  } else if (mode == 1) {
    if (!out_dev) return vps_fail(ctx, VPS_ERR_ARG, "null output");
    rc = route_x(ctx, N, 1, 0, p, 0);
  } else if (mode == 2) {
    if (!out_dev) return vps_fail(ctx, VPS_ERR_ARG, "null output");
    rc = route_x(ctx, N, 2, 0, p, 0);
  } else if (mode == 4) {
    rc = route_x(ctx, N, 4, 0, p, 0);
  } else {
    return vps_fail(ctx, VPS_ERR_ARG, "mode must be 0..4");
  }
  return rc;
}

int vps_fft_x(vps_ctx* ctx, int N, int64_t nlines, int64_t line0, int kz0, const void* in_dev,
              int nseg, int64_t seg_stride, int mode, double* psum_dev,
              unsigned long long* nsample_dev, void* out_dev) {
  return fft_x_impl(ctx, N, nlines, line0, kz0, in_dev, in_dev, in_dev, 1, nseg, seg_stride, mode, psum_dev,
                    nsample_dev, out_dev);
}

int vps_fft_x_bin(vps_ctx* ctx, int N, int64_t nlines, int64_t line0, int kz0, const void* const* in_devs,
                  int ncomp, int nseg, int64_t seg_stride, int count, double* psum_dev,
                  unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  if (ncomp < 1 || ncomp > 3 || !in_devs) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_x_bin: ncomp must be 1..3");
  for (int c = 0; c < ncomp; ++c)
    if (!in_devs[c]) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_x_bin: null component %d", c);
  return fft_x_impl(ctx, N, nlines, line0, kz0, in_devs[0], in_devs[ncomp > 1 ? 1 : 0], in_devs[ncomp > 2 ? 2 : 0],
                    ncomp, nseg, seg_stride, count ? 0 : 3, psum_dev, nsample_dev, nullptr);
}

// Binning x pass of ONE received chunk of the slab exchange: in_devs[c] = the G blocks this rank received for component c
// (vps_fft_y's layout, x running over the senders' slabs), `packed` as the senders' vps_fft_y_packed said.  Planes of slot j
// are kz = chunk*G*nkc + j*G + rank; the last chunk's blocks end with this rank's Nyquist-plane rows, binned here too.
int vps_fft_x_bin_chunk(vps_ctx* ctx, int N, int nx, int G, int nchunks, int chunk, int rank, int packed,
                        const void* const* in_devs, int ncomp, int count, double* psum_dev,
                        unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  if (ncomp < 1 || ncomp > 3 || !in_devs) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_x_bin_chunk: ncomp must be 1..3");
  for (int c = 0; c < ncomp; ++c)
    if (!in_devs[c]) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_x_bin_chunk: null component %d", c);
  if (nx < 1 || nx * G != N || rank < 0 || rank >= G) return vps_fail(ctx, VPS_ERR_ARG, "vps_fft_x_bin_chunk: nx * G must be N, 0 <= rank < G");
  const int2* tab = nullptr;
  long long rows = 0;
  int rc = ypack_get(ctx, N, G, nchunks, chunk, packed != 0, &tab, &rows);
  if (rc) return rc;
  const int nkc = N / 2 / G / nchunks, nky = N / G;
  const bool last = chunk == nchunks - 1;
  const long long blk = rows * nx + (last ? (long long)nky * nx : 0);
  const int mode = count ? 0 : 3;
  rc = fft_x_impl(ctx, N, (int64_t)nkc * N, 0, chunk * G * nkc + rank, in_devs[0], in_devs[ncomp > 1 ? 1 : 0],
                  in_devs[ncomp > 2 ? 2 : 0], ncomp, G, blk, mode, psum_dev, nsample_dev, nullptr, tab, G);
  if (rc || !last) return rc;
  const cf* nq[3];
  for (int c = 0; c < 3; ++c) nq[c] = reinterpret_cast<const cf*>(in_devs[c < ncomp ? c : 0]) + rows * nx;
  return fft_x_impl(ctx, N, nky, (int64_t)rank * nky, N / 2, nq[0], nq[1], nq[2], ncomp, G, blk, mode, psum_dev, nsample_dev,
                    nullptr);
}

int vps_power_bin(vps_ctx* ctx, int N, const float* field_dev, void* work_dev, double* psum_dev,
                  unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "N=%d: need a power of two in [16,4096], 96, 192, 384, 768, 1536, 250, 500, 1000 or 2000", N);
  const size_t half = vps_fft_workspace_bytes(N, N);
  char* w = reinterpret_cast<char*>(work_dev);
  cf* spec = reinterpret_cast<cf*>(w + half);
  cf* nyq = spec + (size_t)(N / 2) * N * N;
  const bool keep = ctx->bin_only;   // binned right away: modes beyond the last shell edge need not be stored
  ctx->bin_only = true;
  int rc = vps_fft_zy(ctx, N, N, field_dev, spec, nyq, w);
  ctx->bin_only = keep;
  if (rc) return rc;
  rc = vps_fft_x(ctx, N, (int64_t)(N / 2) * N, 0, 0, spec, 1, 0, 0, psum_dev, nsample_dev, nullptr);
  if (rc) return rc;
  return vps_fft_x(ctx, N, N, 0, N / 2, nyq, 1, 0, 0, psum_dev, nsample_dev, nullptr);
}

int vps_rfft3(vps_ctx* ctx, int N, const float* field_dev, void* work_dev, void* out_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "N=%d: need a power of two in [16,4096], 96, 192, 384, 768, 1536, 250, 500, 1000 or 2000", N);
  const size_t half = vps_fft_workspace_bytes(N, N);
  char* w = reinterpret_cast<char*>(work_dev);
  cf* spec = reinterpret_cast<cf*>(w + half);
  cf* nyq = spec + (size_t)(N / 2) * N * N;
  const bool keep = ctx->bin_only;   // every mode is wanted here: no store cut in the y pass
  ctx->bin_only = false;
  int rc = vps_fft_zy(ctx, N, N, field_dev, spec, nyq, w);
  ctx->bin_only = keep;
  if (rc) return rc;
  cf* out = reinterpret_cast<cf*>(out_dev);
  rc = vps_fft_x(ctx, N, (int64_t)(N / 2) * N, 0, 0, spec, 1, 0, 1, nullptr, nullptr, out);
  if (rc) return rc;
  return vps_fft_x(ctx, N, N, 0, N / 2, nyq, 1, 0, 1, nullptr, nullptr, out + (size_t)(N / 2) * N * N);
}

int vps_power_grid(vps_ctx* ctx, int N, const float* field_dev, void* work_dev, float* power_dev) {
  VPS_ENTER(ctx);
  if (!vps_fft_supported(N)) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "N=%d: need a power of two in [16,4096], 96, 192, 384, 768, 1536, 250, 500, 1000 or 2000", N);
  const size_t half = vps_fft_workspace_bytes(N, N);
  char* w = reinterpret_cast<char*>(work_dev);
  cf* spec = reinterpret_cast<cf*>(w + half);
  cf* nyq = spec + (size_t)(N / 2) * N * N;
  const bool keep = ctx->bin_only;   // every mode is wanted here: no store cut in the y pass
  ctx->bin_only = false;
  int rc = vps_fft_zy(ctx, N, N, field_dev, spec, nyq, w);
  ctx->bin_only = keep;
  if (rc) return rc;
  rc = vps_fft_x(ctx, N, (int64_t)(N / 2) * N, 0, 0, spec, 1, 0, 2, nullptr, nullptr, power_dev);
  if (rc) return rc;
  return vps_fft_x(ctx, N, N, 0, N / 2, nyq, 1, 0, 2, nullptr, nullptr, power_dev + (size_t)(N / 2) * N * N);
}

}  // extern "C"

#endif   // VPS_FFT_PART <= 0
