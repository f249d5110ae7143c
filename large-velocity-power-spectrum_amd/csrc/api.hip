// Context, memory, timing and binning-table entry points of the C ABI.
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>

#include "vps_internal.h"

static char g_last_error[512] = "no error";

int vps_fail(vps_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    strncpy(ctx->err, buf, sizeof(ctx->err) - 1);
    ctx->err[sizeof(ctx->err) - 1] = 0;
  }
  strncpy(g_last_error, buf, sizeof(g_last_error) - 1);
  g_last_error[sizeof(g_last_error) - 1] = 0;
  return code;
}

vps_launch_timer::vps_launch_timer(vps_ctx* c, int kind, hipStream_t on) : ctx(c), stream(on ? on : c->stream) {
  if (!ctx->timing) return;
  vps_timed_launch tl;
  tl.kind = kind;
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i) {
    if (!ctx->event_pool.empty()) {
      ev[i] = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else if (hipEventCreate(&ev[i]) != hipSuccess) {
      return;
    }
  }
  tl.start = ev[0];
  tl.stop = ev[1];
  (void)hipEventRecord(tl.start, stream);
  ctx->launches.push_back(tl);
  idx = (int)ctx->launches.size() - 1;
}

vps_launch_timer::~vps_launch_timer() {
  if (idx >= 0) (void)hipEventRecord(ctx->launches[idx].stop, stream);
}

// ---- options ---------------------------------------------------------------------------------------------------------------
static const char* const k_option_names[] = {
    "no_fast_binning",    // 1: the general (shell-walk) binning path even for symmetric tables / uniform edges (tests)
    "no_pair_binning",    // 1: the x pass does not pair line ky with N - ky (tests)
    "nn_query_centric",   // 1: exact NN by the query-centric ring search even on uniform lattices (tests)
    "nn_kappa",           // scatter radius factor of the particle-centric NN search (tuning; default 1.15)
    "nn_column",          // exact NN on uniform lattices: 1 force the column-register search, 0 never (default: by density)
    "nn_build_atomic",    // 1: NN cell list by the counting sort with global atomics instead of the two-level LDS sort (tests)
    "nn_stats",           // 1: count (and print) the lattice points the NN scatter pass leaves to the exact fallback
    "sort_groups",        // level-2 workgroups the two-level bucket sort aims for (tuning; default 512)
    "sort_staged",        // 0: level-1 records scattered directly instead of LDS-staged runs (tests)
    "sort_atomic",        // 1: one returning global atomic per particle instead of the two-level sort (tests)
    "nn_ablate",          // timing-only builds (-DVPS_TIMING_VARIANTS): ignored otherwise
    "no_int_binning",     // 1: the mirrored-kx x pass compares float64 k^2 sums even where integer shells are exact (tests)
    "x_wg_per_cu",        // persistent x pass: workgroups per CU (tuning / occupancy experiments; default: what LDS admits)
    "comm_fail_send",     // n >= 1: the n-th ncclSend from now on fails without being issued (error-path tests; clears itself)
};
static std::map<std::string, double>& option_map() {
  static std::map<std::string, double> m;
  return m;
}
static std::mutex& option_lock() {   // (contexts of several host threads share the options)
  static std::mutex mu;
  return mu;
}
double vps_option(const char* name, double dflt) {
  std::lock_guard<std::mutex> hold(option_lock());
  auto& m = option_map();
  auto it = m.find(name);
  return it == m.end() ? dflt : it->second;
}

extern "C" {

int vps_version(void) { return VPS_ABI_VERSION; }

int vps_set_option(const char* name, double value) {
  if (!name) return vps_fail(nullptr, VPS_ERR_ARG, "vps_set_option: null name");
  for (const char* n : k_option_names)
    if (!strcmp(n, name)) {
      std::lock_guard<std::mutex> hold(option_lock());
      if (value != value) option_map().erase(name);   // NaN: back to the default
      else option_map()[name] = value;
      return VPS_OK;
    }
  return vps_fail(nullptr, VPS_ERR_ARG, "vps_set_option: unknown option '%s'", name);
}

double vps_get_option(const char* name, double dflt) { return name ? vps_option(name, dflt) : dflt; }

int vps_create(vps_ctx** out, int device_id) {
  if (!out) return vps_fail(nullptr, VPS_ERR_ARG, "vps_create: null out pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return vps_fail(nullptr, VPS_ERR_HIP, "vps_create: no HIP device visible (%s)",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return vps_fail(nullptr, VPS_ERR_ARG, "vps_create: device %d of %d", device_id, ndev);
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return vps_fail(nullptr, VPS_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  vps_ctx* ctx = new (std::nothrow) vps_ctx();
  if (!ctx) return vps_fail(nullptr, VPS_ERR_NOMEM, "vps_create: out of host memory");
  ctx->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor > 0) ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
  }
  strcpy(ctx->err, "no error");
  *out = ctx;
  return VPS_OK;
}

int vps_destroy(vps_ctx* ctx) {
  if (!ctx) return VPS_OK;
  vps_device_guard guard(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) (void)vps_comm_destroy(ctx);
  vps_fft_free_tables(ctx);
  if (ctx->d_k2) (void)hipFree(ctx->d_k2);
  if (ctx->d_thr) (void)hipFree(ctx->d_thr);
  if (ctx->d_nthr) (void)hipFree(ctx->d_nthr);
  if (ctx->d_axes) (void)hipFree(ctx->d_axes);
  if (ctx->d_xpart) (void)hipFree(ctx->d_xpart);
  if (ctx->d_win) (void)hipFree(ctx->d_win);
  if (ctx->d_kcut) (void)hipFree(ctx->d_kcut);
  if (ctx->ypack.d_tab) (void)hipFree(ctx->ypack.d_tab);
  for (auto& l : ctx->launches) {
    (void)hipEventDestroy(l.start);
    (void)hipEventDestroy(l.stop);
  }
  for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
  delete ctx;
  return VPS_OK;
}

const char* vps_last_error(const vps_ctx* ctx) { return ctx ? ctx->err : g_last_error; }

int vps_set_stream(vps_ctx* ctx, void* hip_stream) {
  VPS_ENTER(ctx);
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return VPS_OK;
}

int vps_sync(vps_ctx* ctx) {
  VPS_ENTER(ctx);
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return VPS_OK;
}

int vps_device_info(vps_ctx* ctx, int64_t out[4]) {
  VPS_ENTER(ctx);
  if (!out) return VPS_ERR_ARG;
  hipDeviceProp_t prop;
  VPS_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  out[0] = prop.multiProcessorCount;
  out[1] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
  out[2] = prop.warpSize;
  out[3] = (int64_t)(prop.totalGlobalMem >> 20);
  return VPS_OK;
}

int vps_malloc(vps_ctx* ctx, void** dev, size_t bytes) {
  VPS_ENTER(ctx);
  if (!dev) return VPS_ERR_ARG;
  *dev = nullptr;
  hipError_t e = hipMalloc(dev, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return vps_fail(ctx, VPS_ERR_NOMEM, "hipMalloc(%zu) out of memory", bytes);
  VPS_HIP_CHECK(ctx, e);
  return VPS_OK;
}

int vps_free(vps_ctx* ctx, void* dev) {
  VPS_ENTER(ctx);
  if (dev) VPS_HIP_CHECK(ctx, hipFree(dev));
  return VPS_OK;
}

int vps_memset(vps_ctx* ctx, void* dev, int value, size_t bytes) {
  VPS_ENTER(ctx);
  if (!dev && bytes) return VPS_ERR_ARG;
  if (bytes) VPS_HIP_CHECK(ctx, hipMemsetAsync(dev, value, bytes, ctx->stream));
  return VPS_OK;
}

int vps_memcpy_h2d(vps_ctx* ctx, void* dev, const void* host, size_t bytes) {
  VPS_ENTER(ctx);
  if ((!dev || !host) && bytes) return VPS_ERR_ARG;
  if (bytes) VPS_HIP_CHECK(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return VPS_OK;
}

int vps_memcpy_d2h(vps_ctx* ctx, void* host, const void* dev, size_t bytes) {
  VPS_ENTER(ctx);
  if ((!dev || !host) && bytes) return VPS_ERR_ARG;
  if (bytes) VPS_HIP_CHECK(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return VPS_OK;
}

int vps_timing_enable(vps_ctx* ctx, int on) {
  VPS_ENTER(ctx);
  ctx->timing = on != 0;
  return VPS_OK;
}

int vps_timing_reset(vps_ctx* ctx) {
  VPS_ENTER(ctx);
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& l : ctx->launches) {
    ctx->event_pool.push_back(l.start);
    ctx->event_pool.push_back(l.stop);
  }
  ctx->launches.clear();
  return VPS_OK;
}

int vps_timing_get(vps_ctx* ctx, int kind, int64_t* launches, double* total_ms) {
  VPS_ENTER(ctx);
  if (kind < 0 || kind >= VPS_K_COUNT) return VPS_ERR_ARG;
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  double ms = 0.0;
  for (auto& l : ctx->launches) {
    if (l.kind != kind) continue;
    float t = 0.f;
    VPS_HIP_CHECK(ctx, hipEventElapsedTime(&t, l.start, l.stop));
    ms += t;
    ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  return VPS_OK;
}

int vps_timing_list(vps_ctx* ctx, int kind, double* ms_out, int64_t cap, int64_t* n_out) {
  VPS_ENTER(ctx);
  if (kind < 0 || kind >= VPS_K_COUNT || cap < 0 || (cap && !ms_out)) return VPS_ERR_ARG;
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  for (auto& l : ctx->launches) {
    if (l.kind != kind) continue;
    if (n < cap) {
      float t = 0.f;
      VPS_HIP_CHECK(ctx, hipEventElapsedTime(&t, l.start, l.stop));
      ms_out[n] = t;
    }
    ++n;
  }
  if (n_out) *n_out = n;
  return VPS_OK;
}

int vps_set_binning(vps_ctx* ctx, int N, const double* k2_axis_host, const double* thr_host,
                    int nbins, double edge0, double inv_spacing) {
  VPS_ENTER(ctx);
  if (N < 2 || nbins < 1 || !k2_axis_host || !thr_host)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_set_binning: bad arguments (N=%d nbins=%d)", N, nbins);
  if (nbins > 8192) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "nbins=%d > 8192", nbins);
  for (int i = 0; i < nbins; ++i)
    if (!(thr_host[i] <= thr_host[i + 1]))
      return vps_fail(ctx, VPS_ERR_ARG, "vps_set_binning: thresholds must be non-decreasing (i=%d)", i);
  // mirrored-kx fast path of the x pass: k2[N-i] == k2[i] and k2 non-decreasing on [0, N/2]
  // (option no_fast_binning forces the general path, for tests)
  bool fast = (N % 2 == 0) && vps_option("no_fast_binning", 0) == 0;
  for (int i = 1; fast && i < N / 2; ++i) fast = (k2_axis_host[N - i] == k2_axis_host[i]);
  for (int i = 0; fast && i < N / 2; ++i) fast = (k2_axis_host[i] <= k2_axis_host[i + 1]);
  // ... and uniform edges, so that the float guess (sqrt(s)-edge0)*inv_spacing is within one
  // shell of the truth (both reference flavours, np.arange and np.linspace, are uniform)
  for (int i = 0; fast && i <= nbins; ++i)
    fast = fabs((sqrt(thr_host[i]) - edge0) * inv_spacing - (double)i) < 1e-3;
  // unchanged tables (the usual case inside a loop over fields/steps): nothing to do
  if (ctx->d_k2 && ctx->bin_fast == fast && ctx->bin_N == N && ctx->nbins == nbins && ctx->edge0 == edge0 &&
      ctx->inv_spacing == inv_spacing && (int)ctx->h_k2.size() == N &&
      memcmp(ctx->h_k2.data(), k2_axis_host, sizeof(double) * N) == 0 &&
      memcmp(ctx->h_thr.data(), thr_host, sizeof(double) * (nbins + 1)) == 0)
    return VPS_OK;
  // the x pass may still be reading the old tables
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_k2) VPS_HIP_CHECK(ctx, hipFree(ctx->d_k2));
  if (ctx->d_thr) VPS_HIP_CHECK(ctx, hipFree(ctx->d_thr));
  ctx->d_k2 = ctx->d_thr = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_k2, sizeof(double) * N));
  VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_thr, sizeof(double) * (nbins + 1)));
  VPS_HIP_CHECK(ctx, hipMemcpy(ctx->d_k2, k2_axis_host, sizeof(double) * N, hipMemcpyHostToDevice));
  VPS_HIP_CHECK(ctx, hipMemcpy(ctx->d_thr, thr_host, sizeof(double) * (nbins + 1), hipMemcpyHostToDevice));
  // per kz plane: the largest |ky| index that the binning x pass can still read (a mode is beyond the last shell edge when
  // fl(ky^2 + kz^2) >= thr[nbins]; x-pass tiles hold up to 16 consecutive |ky|, hence the round-up)
  if (ctx->d_kcut) VPS_HIP_CHECK(ctx, hipFree(ctx->d_kcut));
  ctx->d_kcut = nullptr;
  ctx->h_kcut.clear();
  ctx->ypack.N = 0;   // (the exchange's plane tables follow the cut: rebuilt on the next use)
  if (fast && N >= 128) {
    std::vector<int> kcut(N / 2 + 1);
    for (int kz = 0; kz <= N / 2; ++kz) {
      int kc = -1;
      for (int ky = 0; ky <= N / 2; ++ky)
        if (!(k2_axis_host[ky] + k2_axis_host[kz] >= thr_host[nbins])) kc = ky;
      kcut[kz] = kc < 0 ? -1 : ((kc | 15) < N / 2 ? (kc | 15) : N / 2);
    }
    VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_kcut, sizeof(int) * kcut.size()));
    VPS_HIP_CHECK(ctx, hipMemcpy(ctx->d_kcut, kcut.data(), sizeof(int) * kcut.size(), hipMemcpyHostToDevice));
    ctx->h_kcut = kcut;
  }
  // integer shells: exact where k2[i] = i^2 k2[1] (to float64 rounding) and no threshold comes within 1e-9 (relative) of an
  // integer multiple of k2[1] -- a mode's s = (kx^2 + ky^2) + kz^2 deviates from k2[1] n by a few 1e-16 n, so every mode of an
  // integer n then falls on the same side of every threshold, whatever its float64 rounding
  if (ctx->d_nthr) VPS_HIP_CHECK(ctx, hipFree(ctx->d_nthr));
  ctx->d_nthr = nullptr;
  ctx->bin_int = false;
  if (fast && N >= 4 && N <= 32768 && k2_axis_host[1] > 0.0) {
    const double c2 = k2_axis_host[1];
    bool ok = k2_axis_host[0] == 0.0;
    for (int i = 1; ok && i <= N / 2; ++i) ok = fabs(k2_axis_host[i] - (double)i * (double)i * c2) <= 1e-12 * (double)i * (double)i * c2;
    std::vector<unsigned> nthr(nbins + 1);
    for (int b = 0; ok && b <= nbins; ++b) {
      const double q = thr_host[b] / c2;
      if (!(q == q) || q >= 4.0e9) {
        ok = false;
      } else if (q <= 0.0) {
        nthr[b] = 0u;
      } else {
        const double r = nearbyint(q);
        if (fabs(q - r) <= 1e-9 * (q > 1.0 ? q : 1.0)) ok = false;   // a threshold (numerically) ON an integer: rounding decides
        nthr[b] = (unsigned)ceil(q);
      }
    }
    if (ok) {
      VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_nthr, sizeof(unsigned) * (nbins + 1)));
      VPS_HIP_CHECK(ctx, hipMemcpy(ctx->d_nthr, nthr.data(), sizeof(unsigned) * (nbins + 1), hipMemcpyHostToDevice));
      ctx->bin_int = true;
      ctx->bin_nmax = nthr[nbins];
      ctx->bin_kf = (float)sqrt(c2);
    }
  }
  ctx->h_k2.assign(k2_axis_host, k2_axis_host + N);
  ctx->h_thr.assign(thr_host, thr_host + nbins + 1);
  ctx->bin_fast = fast;
  ctx->bin_N = N;
  ctx->nbins = nbins;
  ctx->edge0 = edge0;
  ctx->inv_spacing = inv_spacing;
  return VPS_OK;
}

int vps_binning_mode(vps_ctx* ctx) {
  VPS_ENTER(ctx);
  if (!ctx->d_k2) return -1;
  if (!ctx->bin_fast) return 0;
  return (ctx->bin_int && ctx->d_nthr && vps_option("no_int_binning", 0) == 0) ? 2 : 1;
}

int vps_set_bin_only(vps_ctx* ctx, int on) {
  VPS_ENTER(ctx);
  ctx->bin_only = on != 0;
  return VPS_OK;
}

int vps_set_window(vps_ctx* ctx, int N, const float* inv_w2_axis_host) {
  VPS_ENTER(ctx);
  if (!inv_w2_axis_host) {                       // back to no deconvolution
    if (ctx->d_win) {
      VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      VPS_HIP_CHECK(ctx, hipFree(ctx->d_win));
    }
    ctx->d_win = nullptr;
    ctx->win_N = 0;
    ctx->h_win.clear();
    return VPS_OK;
  }
  if (N < 2) return vps_fail(ctx, VPS_ERR_ARG, "vps_set_window: N=%d", N);
  for (int i = 1; i < N / 2; ++i)
    if (inv_w2_axis_host[i] != inv_w2_axis_host[N - i])
      return vps_fail(ctx, VPS_ERR_ARG, "vps_set_window: the table must be even in k (entry %d != entry %d)", i, N - i);
  if (ctx->d_win && ctx->win_N == N && memcmp(ctx->h_win.data(), inv_w2_axis_host, sizeof(float) * N) == 0) return VPS_OK;
  VPS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_win) VPS_HIP_CHECK(ctx, hipFree(ctx->d_win));
  ctx->d_win = nullptr;
  VPS_HIP_CHECK(ctx, hipMalloc(&ctx->d_win, sizeof(float) * N));
  VPS_HIP_CHECK(ctx, hipMemcpy(ctx->d_win, inv_w2_axis_host, sizeof(float) * N, hipMemcpyHostToDevice));
  ctx->h_win.assign(inv_w2_axis_host, inv_w2_axis_host + N);
  ctx->win_N = N;
  return VPS_OK;
}

}  // extern "C"
