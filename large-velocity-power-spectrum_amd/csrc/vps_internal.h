// Internal declarations shared by the translation units of libvps_hip.so.
// Not part of the ABI (that is include/vps_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/vps_hip.h"

struct vps_timed_launch {
  int kind;
  hipEvent_t start, stop;
};

struct vps_fft_tables {
  float2* tw_stage = nullptr;  // per-stage LDS twiddle image for complex length NC
  float2* tw_r2c = nullptr;    // exp(-2 pi i k / (2 NC)), k < NC  (real length 2 NC)
};

struct vps_comm;   // comm.hip: RCCL communicator, communication stream, events

struct vps_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  int num_cu = 256;
  size_t lds_per_cu = 160 * 1024;

  std::map<int, vps_fft_tables> fft_tables;  // keyed by complex length NC

  // binning tables (vps_set_binning)
  int bin_N = 0;
  int nbins = 0;
  double* d_k2 = nullptr;   // [N]
  double* d_thr = nullptr;  // [nbins+1]
  double edge0 = 0.0, inv_spacing = 0.0;
  bool bin_fast = false;    // k2 table symmetric and monotone: mirrored-kx binning is valid
  // integer shells (fft.hip: FASTMODE 2): k2[i] = i^2 k2[1] to rounding and no threshold within 1e-9 of an integer multiple of
  // k2[1] -- then nthr[b] = ceil(thr[b] / k2[1]) decides every mode exactly like the float64 comparison
  bool bin_int = false;
  unsigned* d_nthr = nullptr;   // [nbins + 1]
  unsigned bin_nmax = 0;        // nthr[nbins]
  float bin_kf = 0.f;           // sqrt(k2[1])
  std::vector<double> h_k2, h_thr;  // host copies, to skip re-uploading identical tables

  // y-pass store cut for binning-only consumers (vps_set_bin_only): kcut[kz] = largest |ky| index whose modes can still reach
  // a shell, rounded up to the x pass's tile granularity; -1: none
  int* d_kcut = nullptr;
  std::vector<int> h_kcut;   // host copy (row packing of the chunked exchange)
  // plane tables of the chunked exchange for one (N, G, nchunks, packed): per chunk and plane slot j {first row, kc} (int2 on
  // the device) and the rows per destination block of every chunk; rebuilt on demand, dropped by vps_set_binning
  struct {
    int N = 0, G = 0, C = 0, packed = -1;
    void* d_tab = nullptr;
    std::vector<long long> rows;
  } ypack;
  bool bin_only = false;

  // 1 / W(k)^2 per axis index of the mass-assignment window (vps_set_window); NULL: no deconvolution
  float* d_win = nullptr;
  int win_N = 0;
  std::vector<float> h_win;

  // per-workgroup partial shell sums of the x pass
  void* d_xpart = nullptr;
  size_t xpart_cap = 0;

  // small device scratch for the NN lattice axes
  double* d_axes = nullptr;
  size_t axes_cap = 0;

  vps_comm* comm = nullptr;      // vps_comm_create

  unsigned nn_open_points = 0;   // diagnostics: lattice points the last NN scatter pass left to the exact fallback

  bool timing = false;
  std::vector<vps_timed_launch> launches;
  std::vector<hipEvent_t> event_pool;
};

int vps_fail(vps_ctx* ctx, int code, const char* fmt, ...);
// Tuning / test switches set by the host through vps_set_option (process-wide; the library never reads the environment).
double vps_option(const char* name, double dflt);

// Every entry point of the C ABI runs with the context's device current and restores the caller's
// on return: the library allocates (tables, partial sums, lattice axes) and launches on streams
// that belong to ctx->device, whatever device the calling thread had selected.
struct vps_device_guard {
  int prev = -1;
  explicit vps_device_guard(const vps_ctx* ctx) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != ctx->device && hipSetDevice(ctx->device) == hipSuccess) prev = cur;
  }
  ~vps_device_guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define VPS_ENTER(ctx)                 \
  if (!(ctx)) return VPS_ERR_ARG;      \
  vps_device_guard vps_guard_(ctx)

#define VPS_HIP_CHECK(ctx, expr)                                                   \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return vps_fail((ctx), VPS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,          \
                      hipGetErrorString(_e), __FILE__, __LINE__);                  \
  } while (0)

// Brackets a kernel launch with events when ctx->timing is on.
struct vps_launch_timer {
  vps_ctx* ctx;
  int idx = -1;
  hipStream_t stream;
  vps_launch_timer(vps_ctx* c, int kind, hipStream_t on = nullptr);   // on: the stream the interval lives on (default: the context's)
  ~vps_launch_timer();
};

// fft.hip
int vps_fft_get_tables(vps_ctx* ctx, int NC, vps_fft_tables* out);
void vps_fft_free_tables(vps_ctx* ctx);
// fused deposit -> z pass ("pencil" path): records sorted by pencil -> ncomp half spectra
int vps_pencil_tp(int N);
bool vps_pencil_supported(vps_ctx* ctx, int N);
// side: one float per record (scratch of the kernel: what it keeps per record when a bucket outgrows its registers)
int vps_fft_pencil_zy(vps_ctx* ctx, int N, int nx, const unsigned* records, const unsigned* start, float* side,
                      int ncomp, const int* chan, int divide, int energy, float vol, void* spec_dev, void* nyq_dev,
                      void* bwork_dev, int with_energy = 0);

// ---- LDS floating-point accumulation ---------------------------------------------------------
// gfx950 executes ds_add_f32 far below the LDS rate (measured: about one lane every two clocks per
// CU), while its integer LDS atomics run at bank speed.  So a float add is first tried as ONE
// compare-and-swap on the bit pattern (plain read, add, ds_cmpst_rtn): it succeeds unless another
// lane or wave changed the word in between, which is rare when the lanes of a wave hit different
// words.  Only the losers fall back to the native atomic -- many particles in one cell therefore
// cost what they cost before, never a retry storm.  Same sums (in some order) either way.
// Pencil kernel 0.57 -> 0.49 ms at 512^3 / 1e7 particles.
#if defined(__HIPCC__)
// `crowded` (uniform over the caller's workgroup): expect many adds per word -- skip the attempt.
__device__ __forceinline__ void vps_lds_add(float* addr, float v, bool crowded = false) {
  if (crowded) {
    atomicAdd(addr, v);
    return;
  }
  unsigned* a = reinterpret_cast<unsigned*>(addr);
  const unsigned assumed = *a;
  if (atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v)) != assumed) atomicAdd(addr, v);
}
#endif
