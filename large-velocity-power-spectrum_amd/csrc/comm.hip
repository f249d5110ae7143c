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
  hipStream_t stream = nullptr;           // communication stream
  std::vector<hipEvent_t> ev_y, ev_a;     // per chunk: send buffer written / blocks arrived
};

extern "C" {

int vps_comm_unique_id(char* id128) {
  if (!id128) return vps_fail(nullptr, VPS_ERR_ARG, "vps_comm_unique_id: null buffer");
  RcclApi& api = rccl();
  if (!api.ok) return vps_fail(nullptr, VPS_ERR_UNSUPPORTED, "RCCL (librccl.so) cannot be loaded: %s", dlerror());
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

size_t vps_spectrum_zimages_workspace_bytes(int N, int nx, int G, int nchunks, int ncomp) {
  if (ncomp < 1 || ncomp > 3) return 0;
  size_t tot = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int64_t e = vps_fft_y_chunk_elems(N, nx, G, nchunks, c);   // upper bound (all rows)
    if (e < 0) return 0;
    tot += (size_t)e;
  }
  return 2 * (size_t)ncomp * tot * sizeof(float2);     // send + receive, every component, every chunk in flight at once
}

int vps_spectrum_zimages(vps_ctx* ctx, int N, int nx, const void* const* zimg_devs, int ncomp, int nchunks, void* xwork_dev,
                         int count, double* psum_dev, unsigned long long* nsample_dev) {
  VPS_ENTER(ctx);
  vps_comm* cm = ctx->comm;
  if (!cm) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: no communicator (vps_comm_create)");
  if (ncomp < 1 || ncomp > 3 || !zimg_devs || !xwork_dev || !psum_dev || (count && !nsample_dev))
    return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: bad ncomp / null buffer");
  const int G = cm->world, r = cm->rank;
  if (nx * G != N) return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: nx=%d x %d ranks != N=%d", nx, G, N);
  if (vps_fft_y_chunk_elems(N, nx, G, nchunks, 0) < 0)
    return vps_fail(ctx, VPS_ERR_ARG, "vps_spectrum_zimages: %d ranks x %d chunks must divide N/2=%d", G, nchunks, N / 2);
  RcclApi& api = rccl();
  while ((int)cm->ev_y.size() < nchunks) {
    hipEvent_t a, b;
    VPS_HIP_CHECK(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
    VPS_HIP_CHECK(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
    cm->ev_y.push_back(a);
    cm->ev_a.push_back(b);
  }
  // inside a binning-only scope the blocks carry only the rows a shell can reach
  struct BinOnlyScope {
    vps_ctx* c;
    bool prev;
    explicit BinOnlyScope(vps_ctx* c_) : c(c_), prev(c_->bin_only) { c->bin_only = true; }
    ~BinOnlyScope() { c->bin_only = prev; }
  } scope(ctx);
  int rc = VPS_OK;
  const int packed = vps_fft_y_packed(ctx, N);
  // buffers: [chunk][component]{send G blocks | recv G blocks}
  std::vector<float2*> sendp((size_t)nchunks * ncomp), recvp((size_t)nchunks * ncomp);
  std::vector<int64_t> blk(nchunks);
  {
    float2* p = reinterpret_cast<float2*>(xwork_dev);
    for (int c = 0; c < nchunks; ++c) {
      blk[c] = vps_fft_y_chunk_block(ctx, N, nx, G, nchunks, c, packed);
      if (blk[c] < 0) return VPS_ERR_ARG;
      for (int k = 0; k < ncomp; ++k) {
        sendp[(size_t)c * ncomp + k] = p; p += (size_t)G * blk[c];
        recvp[(size_t)c * ncomp + k] = p; p += (size_t)G * blk[c];
      }
    }
  }
  // the buffers may still be read by the previous call's x passes (same stream: ordered) -- and its exchanges are
  // complete before those x passes ran, so the communication stream is idle here
  for (int c = 0; c < nchunks && !rc; ++c) {
    for (int k = 0; k < ncomp && !rc; ++k) rc = vps_fft_y(ctx, N, nx, zimg_devs[k], G, nchunks, c, sendp[(size_t)c * ncomp + k]);
    if (rc) break;
    VPS_HIP_CHECK(ctx, hipEventRecord(cm->ev_y[c], ctx->stream));
    VPS_HIP_CHECK(ctx, hipStreamWaitEvent(cm->stream, cm->ev_y[c], 0));
    VPS_NCCL_CHECK(ctx, api.GroupStart());
    for (int k = 0; k < ncomp; ++k)
      for (int h = 0; h < G; ++h) {
        const size_t nfl = (size_t)blk[c] * 2;   // complex64 as two floats
        VPS_NCCL_CHECK(ctx, api.Send(sendp[(size_t)c * ncomp + k] + (size_t)h * blk[c], nfl, ncclFloat32, h, cm->comm, cm->stream));
        VPS_NCCL_CHECK(ctx, api.Recv(recvp[(size_t)c * ncomp + k] + (size_t)h * blk[c], nfl, ncclFloat32, h, cm->comm, cm->stream));
      }
    VPS_NCCL_CHECK(ctx, api.GroupEnd());
    VPS_HIP_CHECK(ctx, hipEventRecord(cm->ev_a[c], cm->stream));
  }
  if (rc) return rc;
  for (int c = 0; c < nchunks; ++c) {
    VPS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, cm->ev_a[c], 0));
    const void* ins[3];
    for (int k = 0; k < ncomp; ++k) ins[k] = recvp[(size_t)c * ncomp + k];
    rc = vps_fft_x_bin_chunk(ctx, N, nx, G, nchunks, c, r, packed, ins, ncomp, count, psum_dev, nsample_dev);
    if (rc) return rc;
  }
  return VPS_OK;
}

int vps_allreduce_shells(vps_ctx* ctx, double* psum_dev, unsigned long long* nsample_dev, int nbins) {
  VPS_ENTER(ctx);
  vps_comm* cm = ctx->comm;
  if (!cm) return vps_fail(ctx, VPS_ERR_ARG, "vps_allreduce_shells: no communicator (vps_comm_create)");
  if (nbins < 1 || !psum_dev) return vps_fail(ctx, VPS_ERR_ARG, "vps_allreduce_shells: bad arguments");
  RcclApi& api = rccl();
  // on the context's stream: ordered behind the x passes that filled the accumulators
  VPS_NCCL_CHECK(ctx, api.GroupStart());
  VPS_NCCL_CHECK(ctx, api.AllReduce(psum_dev, psum_dev, (size_t)nbins, ncclFloat64, ncclSum, cm->comm, ctx->stream));
  if (nsample_dev)
    VPS_NCCL_CHECK(ctx, api.AllReduce(nsample_dev, nsample_dev, (size_t)nbins, ncclUint64, ncclSum, cm->comm, ctx->stream));
  VPS_NCCL_CHECK(ctx, api.GroupEnd());
  return VPS_OK;
}

}  // extern "C"
