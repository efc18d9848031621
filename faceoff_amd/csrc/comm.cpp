// fo_comm_*: the data-parallel gradient exchange behind the C-ABI ("Collectives: fo_comm_{init,allreduce_async,wait,
// destroy} over RCCL, ncclUniqueId exchanged through the existing TCP dist_url").  Replaces what nn.parallel.DistributedDataParallel's
// reducer does for the reference (train_faceoff_perceptual.py:164-169; process group from distributed/launch.py:61-66) for a host that does
// not go through torch.distributed: one communicator per process (= per GPU), in-place fp32 SUM all-reduces of slices of the flat gradient
// arena on the communicator's OWN stream, ordered against the compute streams by events -- nothing here synchronises the host.
// RCCL is opened with dlopen at fo_comm_unique_id / fo_comm_init, so the library loads (and every other entry point works) where librccl is absent.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
bool g_ok = false;
std::string g_why;          // why librccl could not be used: dlerror() captured right where dlopen / dlsym failed (it is cleared by being read)

bool load_rccl() {
  std::call_once(g_once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
      const char* e = dlerror();
      g_why += std::string(g_why.empty() ? "" : "; ") + (e ? e : n);
    }
    if (!g_rccl.lib) return;
    g_why.clear();
#define FO_SYM(field, name)                                                                    \
  do {                                                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name));          \
    if (!g_rccl.field) {                                                                       \
      const char* e = dlerror();                                                               \
      g_why += std::string(g_why.empty() ? "" : "; ") + (e ? e : "missing symbol " name);      \
    }                                                                                          \
  } while (0)
    FO_SYM(GetUniqueId, "ncclGetUniqueId");
    FO_SYM(CommInitRank, "ncclCommInitRank");
    FO_SYM(AllReduce, "ncclAllReduce");
    FO_SYM(Broadcast, "ncclBroadcast");
    FO_SYM(CommDestroy, "ncclCommDestroy");
    FO_SYM(GetErrorString, "ncclGetErrorString");
#undef FO_SYM
    g_ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.CommDestroy && g_rccl.GetErrorString;
  });
  return g_ok;
}

}  // namespace

struct fo_comm {
  ncclComm_t comm;
  hipStream_t stream;      // every collective of this communicator runs here, in issue order
  bool own_stream;         // created here (destroyed with the communicator) -- false after fo_comm_set_stream
  hipEvent_t ev;           // re-recorded per call (ordering only)
  int rank, world, device;
  long long issued;        // all-reduces enqueued so far
};

#define FO_NCCL(call)                                                                  \
  do {                                                                                 \
    ncclResult_t r__ = (call);                                                         \
    if (r__ != ncclSuccess) {                                                          \
      fo_set_error("%s:%d: RCCL: %s", __FILE__, __LINE__, g_rccl.GetErrorString(r__)); \
      return FO_E_HIP;                                                                 \
    }                                                                                  \
  } while (0)
#define FO_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      fo_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__));           \
      return FO_E_HIP;                                                                 \
    }                                                                                  \
  } while (0)

extern "C" {

int fo_comm_unique_id(void* id128) {
  FO_REQUIRE(id128, FO_E_SHAPE, "comm_unique_id: null buffer");
  FO_REQUIRE(load_rccl(), FO_E_HIP, "comm: librccl.so cannot be used: %s", g_why.c_str());
  static_assert(sizeof(ncclUniqueId) == 128, "the C-ABI hands the RCCL id around as 128 opaque bytes");
  ncclUniqueId id;
  FO_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return FO_OK;
}

int fo_comm_init(fo_comm** out, int rank, int world, const void* id128, int device) {
  FO_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, FO_E_SHAPE, "comm_init: bad rank / world / id");
  FO_REQUIRE(load_rccl(), FO_E_HIP, "comm: librccl.so cannot be used: %s", g_why.c_str());
  FO_HIP(hipSetDevice(device));
  fo_comm* c = new fo_comm();
  c->rank = rank; c->world = world; c->device = device; c->issued = 0;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    fo_set_error("comm_init: ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
    delete c;
    return FO_E_HIP;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) {
    fo_set_error("comm_init: cannot create the communicator's stream / event");
    g_rccl.CommDestroy(c->comm);
    delete c;
    return FO_E_HIP;
  }
  c->own_stream = true;
  *out = c;
  return FO_OK;
}

/* The communicator's collectives run on `stream` (the caller's, kept alive by the caller) from now on instead of on the stream fo_comm_init created: HIP gives a
 * stream its hardware queue at first use, and a caller that can tell queues apart (faceoff_amd.engine.streams_by_queue) hands in one that does not share the compute
 * stream's -- on a shared queue every all-reduce lines up behind the kernels it is meant to run beside.  Everything issued so far is waited for first. */
int fo_comm_set_stream(fo_comm* c, void* stream) {
  FO_REQUIRE(c && stream, FO_E_SHAPE, "comm_set_stream: null communicator / stream");
  FO_HIP(hipStreamSynchronize(c->stream));
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)stream;
  c->own_stream = false;
  return FO_OK;
}

int fo_comm_rank(const fo_comm* c) { return c ? c->rank : -1; }
int fo_comm_world(const fo_comm* c) { return c ? c->world : -1; }
int64_t fo_comm_issued(const fo_comm* c) { return c ? c->issued : -1; }

/* buf[0..count) := sum over ranks of buf (fp32, in place), enqueued on the communicator's stream BEHIND everything enqueued so far on
 * `after_stream` (the stream that produced buf).  Returns at once. */
int fo_comm_allreduce_async(fo_comm* c, float* buf, int64_t count, void* after_stream) {
  FO_REQUIRE(c && buf && count > 0, FO_E_SHAPE, "comm_allreduce: null communicator / buffer or empty message");
  FO_HIP(hipEventRecord(c->ev, (hipStream_t)after_stream));
  FO_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
  FO_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, c->stream));
  ++c->issued;
  return FO_OK;
}

/* buf[0..count) on every rank := rank `root`'s buf (fp32, in place), ordered like fo_comm_allreduce_async.  What DistributedDataParallel's
 * broadcast_buffers does for module buffers that are NOT summed over ranks (the discriminators' InstanceNorm running statistics,
 * mocoganhd_video_disc.py:139-147): rank 0's copy is the one every rank carries on.  Not counted by fo_comm_issued. */
int fo_comm_broadcast_async(fo_comm* c, float* buf, int64_t count, int root, void* after_stream) {
  FO_REQUIRE(c && buf && count > 0 && root >= 0 && root < c->world, FO_E_SHAPE, "comm_broadcast: null communicator / buffer, empty message or bad root");
  FO_HIP(hipEventRecord(c->ev, (hipStream_t)after_stream));
  FO_HIP(hipStreamWaitEvent(c->stream, c->ev, 0));
  FO_NCCL(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat, root, c->comm, c->stream));
  return FO_OK;
}

/* `stream` waits (on the device) for every collective issued so far. */
int fo_comm_wait(fo_comm* c, void* stream) {
  FO_REQUIRE(c, FO_E_SHAPE, "comm_wait: null communicator");
  FO_HIP(hipEventRecord(c->ev, c->stream));
  FO_HIP(hipStreamWaitEvent((hipStream_t)stream, c->ev, 0));
  return FO_OK;
}

int fo_comm_destroy(fo_comm* c) {
  if (!c) return FO_OK;
  (void)hipStreamSynchronize(c->stream);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  (void)hipEventDestroy(c->ev);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  if (r != ncclSuccess) {
    fo_set_error("comm_destroy: %s", g_rccl.GetErrorString(r));
    return FO_E_HIP;
  }
  return FO_OK;
}

}  // extern "C"
