// Slab exchange inside the library: RCCL over xGMI behind the C ABI.
//
// Replaces, for hosts that are not Python, what vpower/device.py drives through torch.distributed -- and, in the
// reference, the four comm.allgather per buffer flush and the two comm.Reduce of scripts/parallel_optimized.py:365-368,
// 455-456: per scalar field ONE message per pair of ranks (cut into kz chunks only to pipeline it), then one all-reduce of
// the (nbins,) shell sums and counts.
//
//   vps_comm_create      ncclCommInitRank on the context's device (one process per GPU; the host moves the 128-byte id)
//   vps_spectrum_zimages per kz chunk: y pass of every component into the send buffer (context stream) -> event ->
//                        ncclSend / ncclRecv to every rank inside one group (communication stream) -> event -> x pass with
//                        fused shell binning of the received blocks (context stream).  All y passes are enqueued first, so
//                        chunk c travels while chunk c + 1 is transformed, and chunk c is binned while c + 1 travels.
//   vps_allreduce_shells ncclAllReduce of psum (float64) and nsample (uint64)
//
// RCCL is resolved at run time (dlopen): the library keeps linking against libamdhip64 only and loads on hosts
// without RCCL; a process that already holds an RCCL (torch's) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

#include "comm_group.h"
#include "vps_internal.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.handle || api.ok) return api;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)   // an RCCL the process already holds (torch's) first
    if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!api.handle) return api;
#define VPS_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name))
  VPS_SYM(GetUniqueId, "ncclGetUniqueId");
  VPS_SYM(CommInitRank, "ncclCommInitRank");
  VPS_SYM(CommDestroy, "ncclCommDestroy");
  VPS_SYM(GroupStart, "ncclGroupStart");
  VPS_SYM(GroupEnd, "ncclGroupEnd");
  VPS_SYM(Send, "ncclSend");
  VPS_SYM(Recv, "ncclRecv");
  VPS_SYM(AllReduce, "ncclAllReduce");
  VPS_SYM(GetErrorString, "ncclGetErrorString");
#undef VPS_SYM
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv &&
           api.AllReduce && api.GetErrorString;
  return api;
}

#define VPS_NCCL_CHECK(ctx, expr)                                                                     \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess)                                                                            \
      return vps_fail((ctx), VPS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

}  // namespace

struct vps_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;                 // communication stream
  std::vector<hipEvent_t> ev_y, ev_a, ev_x;     // per chunk: send buffer written / blocks arrived / received blocks binned
};

// Error injection for the tests of the error paths (tests/test_gpu_distributed.py): option "comm_fail_send" = n >= 1 makes the
// n-th ncclSend from now on fail WITHOUT being issued (and clears itself).  Never set by the product path.
static bool inject_send_failure() {
  const double n = vps_option("comm_fail_send", 0.0);
  if (!(n >= 1.0)) return false;
  vps_set_option("comm_fail_send", n - 1.0);
  return n < 2.0;
}

// the adapter vps_exchange_chunk_group drives (comm_group.h)
struct RcclGroupApi {
  typedef ncclResult_t result_t;
  RcclApi& api;
  ncclComm_t comm;
  hipStream_t stream;
  static ncclResult_t success() { return ncclSuccess; }
  ncclResult_t GroupStart() { return api.GroupStart(); }
  ncclResult_t GroupEnd() { return api.GroupEnd(); }
  ncclResult_t Send(const float2* p, size_t nfloats, int peer) {
    if (inject_send_failure()) return ncclInternalError;
    return api.Send(p, nfloats, ncclFloat32, peer, comm, stream);
  }
  ncclResult_t Recv(float2* p, size_t nfloats, int peer) { return api.Recv(p, nfloats, ncclFloat32, peer, comm, stream); }
};

extern "C" {

int vps_comm_unique_id(char* id128) {
  if (!id128) return vps_fail(nullptr, VPS_ERR_ARG, "vps_comm_unique_id: null buffer");
  RcclApi& api = rccl();
  if (!api.ok) {
    const char* why = dlerror();   // (NULL when the library loaded but lacks a symbol)
    return vps_fail(nullptr, VPS_ERR_UNSUPPORTED, "RCCL (librccl.so) cannot be loaded: %s", why ? why : "missing symbols");
  }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = api.GetUniqueId(&id);
  if (r != ncclSuccess) return vps_fail(nullptr, VPS_ERR_HIP, "ncclGetUniqueId: %s", api.GetErrorString(r));
  memcpy(id128, &id, sizeof(id));
  return VPS_OK;
}

int vps_comm_destroy(vps_ctx* ctx) {
  VPS_ENTER(ctx);
  vps_comm* c = ctx->comm;
  if (!c) return VPS_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  for (auto e : c->ev_y) (void)hipEventDestroy(e);
  for (auto e : c->ev_a) (void)hipEventDestroy(e);
  for (auto e : c->ev_x) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  ctx->comm = nullptr;
  return VPS_OK;
}

int vps_comm_create(vps_ctx* ctx, int rank, int world, const char* id128) {
  VPS_ENTER(ctx);
  if (world < 1 || rank < 0 || rank >= world || !id128) return vps_fail(ctx, VPS_ERR_ARG, "vps_comm_create: rank %d of %d / null id", rank, world);
  RcclApi& api = rccl();
  if (!api.ok) return vps_fail(ctx, VPS_ERR_UNSUPPORTED, "RCCL (librccl.so) cannot be loaded");
  if (ctx->comm) {
    int rc = vps_comm_destroy(ctx);
    if (rc) return rc;
  }
  vps_comm* c = new (std::nothrow) vps_comm();
  if (!c) return vps_fail(ctx, VPS_ERR_NOMEM, "vps_comm_create: out of host memory");
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);     // (collective: every rank of the job calls it)
  if (r != ncclSuccess) {
    delete c;
    return vps_fail(ctx, VPS_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, api.GetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    (void)api.CommDestroy(c->comm);
    delete c;
    return vps_fail(ctx, VPS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  ctx->comm = c;
  return VPS_OK;
}

int vps_comm_info(vps_ctx* ctx, int* rank, int* world) {
  VPS_ENTER(ctx);
  if (!ctx->comm) return vps_fail(ctx, VPS_ERR_ARG, "no communicator (vps_comm_create)");
  if (rank) *rank = ctx->comm->rank;
  if (world) *world = ctx->comm->world;
  return VPS_OK;
}

// Chunk buffers: TWO slots, whatever the chunk count -- chunk c uses slot c & 1, and is enqueued only behind the exchange
// (send half) and the binning x pass (receive half) of chunk c - 2.  Each slot holds, per component, G send blocks and G
// receive blocks of the largest chunk.
size_t vps_spectrum_zimages_workspace_bytes(int N, int nx, int G, int nchunks, int ncomp) {
  if (ncomp < 1 || ncomp > 3 || nchunks < 1) return 0;
  size_t big = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int64_t e = vps_fft_y_chunk_elems(N, nx, G, nchunks, c);   // upper bound (all rows)
    if (e < 0) return 0;
    if ((size_t)e > big) big = (size_t)e;
  }
  const int slots = nchunks < 2 ? nchunks : 2;
  return 2 * (size_t)slots * (size_t)ncomp * big * sizeof(float2);     // send + receive
}

int vps_spectrum_zimages(vps_ctx* ctx, int N, int nx, const void* const* zimg_devs, int ncomp, int nchunks, void* xwork_dev,
                         int count, double* psum_dev, unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  vps_comm* cm = ctx->comm;
  if (!cm) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: no communicator (vps_comm_create)");
  if (ncomp < 1 || ncomp > 3 || !zimg_devs || !xwork_dev || !psum_dev || (count && !nsample_dev))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: bad ncomp / null buffer");
  for (int k = 0; k < ncomp; ++k)
    if (!zimg_devs[k]) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: null z image %d", k);
  const int G = cm->world, r = cm->rank;
  if (nx * G != N) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: nx=%d x %d ranks != N=%d", nx, G, N);
  if (nchunks < 1 || vps_fft_y_chunk_elems(N, nx, G, nchunks, 0) < 0)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: %d ranks x %d chunks must divide N/2=%d", G, nchunks, N / 2);
  RcclApi& api = rccl();
  while ((int)cm->ev_y.size() < nchunks) {
    hipEvent_t a, b, c;
    VPS_HIP_CHECK(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
    VPS_HIP_CHECK(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
    VPS_HIP_CHECK(ctx, hipEventCreateWithFlags(&c, hipEventDisableTiming));
    cm->ev_y.push_back(a);
    cm->ev_a.push_back(b);
    cm->ev_x.push_back(c);
  }
  // inside a binning-only scope the blocks carry only the rows a shell can reach
  struct BinOnlyScope {
    vps_ctx* c;
    bool prev;
    explicit BinOnlyScope(vps_ctx* c_) : c(c_), prev(c_->bin_only) { c->bin_only = true; }
    ~BinOnlyScope() { c->bin_only = prev; }
  } scope(ctx);
  const int packed = vps_fft_y_packed(ctx, N);
  // every chunk's geometry is checked BEFORE anything is enqueued: a rank that failed in the middle of the pipeline would
  // leave its peers inside exchanges it never joins
  std::vector<int64_t> blk(nchunks);
  int64_t big = 0;
  for (int c = 0; c < nchunks; ++c) {
    blk[c] = vps_fft_y_chunk_block(ctx, N, nx, G, nchunks, c, packed);
    if (blk[c] < 0) return VPS_ERR_ARG;
    const int64_t e = vps_fft_y_chunk_elems(N, nx, G, nchunks, c);
    if (e > big) big = e;
    if ((int64_t)G * blk[c] > e) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: chunk %d larger than its bound", c);
  }
  // slot s: [component]{send G blocks | recv G blocks}, sized for the largest chunk
  const int slots = nchunks < 2 ? nchunks : 2;
  float2* sendp[2][3];
  float2* recvp[2][3];
  {
    float2* p = reinterpret_cast<float2*>(xwork_dev);
    for (int s = 0; s < slots; ++s)
      for (int k = 0; k < ncomp; ++k) {
        sendp[s][k] = p; p += (size_t)big;
        recvp[s][k] = p; p += (size_t)big;
      }
  }
  // y passes of chunk c into slot c & 1 (context stream) -> ev_y -> grouped exchange (communication stream) -> ev_a
  // (the previous CALL's x passes read these buffers on the context stream too: ordered ahead of the y passes below, and
  // its exchanges were complete before those x passes ran, so the communication stream is idle here)
  auto start_chunk = [&](int c) -> int {
    const int s = c & 1;
    // send half of slot s: free once the exchange of chunk c - 2 has completed -- the context stream already waited for
    // ev_a[c - 2] ahead of that chunk's x pass, which was enqueued before this call
    for (int k = 0; k < ncomp; ++k) {
      const int rc = vps_fft_y(ctx, N, nx, zimg_devs[k], G, nchunks, c, sendp[s][k]);
      if (rc) return rc;
    }
    VPS_HIP_CHECK(ctx, hipEventRecord(cm->ev_y[c], ctx->stream));
    VPS_HIP_CHECK(ctx, hipStreamWaitEvent(cm->stream, cm->ev_y[c], 0));
    // receive half of slot s: free once chunk c - 2 has been binned
    if (c >= 2) VPS_HIP_CHECK(ctx, hipStreamWaitEvent(cm->stream, cm->ev_x[c - 2], 0));
    RcclGroupApi g{api, cm->comm, cm->stream};
    const char* what = nullptr;
    ncclResult_t nr;
    {
      vps_launch_timer tm(ctx, VPS_K_EXCHANGE, cm->stream);
      nr = vps_exchange_chunk_group(g, sendp[s], recvp[s], ncomp, G, (size_t)blk[c], (size_t)2, &what);
    }
    if (nr != ncclSuccess)
      return vps_fail(ctx, VPS_ERR_HIP, "vps_spectrum_zimages: %s failed in chunk %d: %s (group closed)", what ? what : "RCCL",
                      c, api.GetErrorString(nr));
    VPS_HIP_CHECK(ctx, hipEventRecord(cm->ev_a[c], cm->stream));
    return VPS_OK;
  };
  int rc = start_chunk(0);
  for (int c = 0; c < nchunks && !rc; ++c) {
    // chunk c + 1 is transformed while c travels, and travels while c is binned
    if (c + 1 < nchunks && (rc = start_chunk(c + 1))) break;
    {
      vps_launch_timer tm(ctx, VPS_K_EXCHANGE_WAIT);   // (two events around a wait: the time the stream stood still)
      VPS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, cm->ev_a[c], 0));
    }
    const void* ins[3];
    for (int k = 0; k < ncomp; ++k) ins[k] = recvp[c & 1][k];
    rc = vps_fft_x_bin_chunk(ctx, N, nx, G, nchunks, c, r, packed, ins, ncomp, count, psum_dev, nsample_dev);
    if (!rc) VPS_HIP_CHECK(ctx, hipEventRecord(cm->ev_x[c], ctx->stream));
  }
  if (rc) {
    // whatever was enqueued still runs to completion on both streams; leave them drained so that the buffers can be freed
    (void)hipStreamSynchronize(cm->stream);
    (void)hipStreamSynchronize(ctx->stream);
  }
  return rc;
}

int vps_allreduce_shells(vps_ctx* ctx, double* psum_dev, unsigned long long* nsample_dev, int nbins) {
  VPS_ENTER(ctx);
  vps_comm* cm = ctx->comm;
  if (!cm) return vps_fail(ctx, VPS_ERR_ARG, "vps_allreduce_shells: no communicator (vps_comm_create)");
  if (nbins < 1 || !psum_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_allreduce_shells: bad arguments");
  RcclApi& api = rccl();
  // on the context's stream: ordered behind the x passes that filled the accumulators
  VPS_NCCL_CHECK(ctx, api.GroupStart());
  ncclResult_t r1 = api.AllReduce(psum_dev, psum_dev, (size_t)nbins, ncclFloat64, ncclSum, cm->comm, ctx->stream);
  if (r1 == ncclSuccess && nsample_dev)
    r1 = api.AllReduce(nsample_dev, nsample_dev, (size_t)nbins, ncclUint64, ncclSum, cm->comm, ctx->stream);
  const ncclResult_t r2 = api.GroupEnd();   // on every path (comm_group.h)
  if (r1 != ncclSuccess) return vps_fail(ctx, VPS_ERR_HIP, "ncclAllReduce failed: %s (group closed)", api.GetErrorString(r1));
  VPS_NCCL_CHECK(ctx, r2);
  return VPS_OK;
}

}  // extern "C"
