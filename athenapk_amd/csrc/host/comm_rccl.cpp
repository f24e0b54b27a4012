// comm_rccl.cpp -- the native multi-GPU transport of the standalone driver: RCCL called from the C++
// host, no callback into Python on the exchange path (BASELINE north_star: "ghost-cell halo exchange
// carried by RCCL send/recv over xGMI overlapped with interior flux computation on a second HIP
// stream").  It stands where the reference calls Parthenon's boundary communication
// (src/hydro/hydro_driver.cpp:506, 567-568) and MPI_Allreduce (src/hydro/hydro.cpp:127-128).
//
//  * ONE grouped ncclSend / ncclRecv pair per peer rank and exchange (a periodic 2 x 2 x 2 rank grid
//    has 7 peers), on a dedicated halo stream: exchange_begin makes that stream wait for the pack
//    kernel (event on the sim's stream) and posts the group; exchange_end makes the sim's stream wait
//    for the group's completion event -- the host never blocks, so the next stage's x3 sweep / x1
//    sweep runs while the messages fly.
//  * the tiny per-cycle reductions (dt, c_h, history sums) go through a SECOND communicator on its
//    own stream: on the halo communicator they would queue behind messages still in flight across
//    the cycle boundary and the host would wait for them.
//  * librccl is opened with dlopen (the copy torch has loaded is reused when there is one), so the
//    library also loads on hosts without RCCL and single-GPU runs never touch it.
//
// Bootstrap: rank 0 creates two ncclUniqueIds (apk_rccl_unique_ids), the launcher hands them to
// every rank by whatever means it has (the Python driver: torch.distributed.broadcast_object_list;
// an MPI launcher: MPI_Bcast), every rank calls apk_sim_comm_rccl before apk_sim_initialize.
#include <dlfcn.h>

#include <cstring>

// The handful of RCCL types and constants this file passes through function pointers, declared here so
// that the build does not need RCCL's headers either (values: rccl.h of ROCm 7.2 = NCCL's public ABI;
// ncclUniqueId is 128 opaque bytes, checked against APK_RCCL_ID_BYTES below).
extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
}

#include "sim_internal.hpp"

namespace apk {
namespace host {
namespace {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl_api(std::string *why) {
  static RcclApi api;
  static bool tried = false, ok = false;
  static std::string err;
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      const char *e = dlerror();  // (a second call returns NULL: the first one clears the error)
      err = std::string("librccl not found: ") + (e ? e : "");
    } else {
      auto sym = [&](const char *n) {
        void *p = dlsym(api.lib, n);
        if (!p && err.empty()) err = std::string("librccl lacks ") + n;
        return p;
      };
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
      api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
      api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
      ok = err.empty();
    }
  }
  if (!ok && why) *why = err;
  return ok ? &api : nullptr;
}

}  // namespace

struct RcclTransport {
  RcclApi *api = nullptr;
  ncclComm_t halo = nullptr, red = nullptr;
  hipStream_t s_halo = nullptr, s_red = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  double *d_red = nullptr;  // device staging of the reductions
  double *h_red = nullptr;  // pinned
  static constexpr int kRedMax = 64;
  int rank = 0, nranks = 1;
  long long exchanges = 0, reductions = 0;
  long long loopback_bytes = 0;  // loopback transport: bytes "sent" so far
  std::string err;
};

namespace {

bool nccl_ok(RcclTransport *t, ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return true;
  t->err = std::string(what) + ": " + (t->api->GetErrorString ? t->api->GetErrorString(r) : "nccl error");
  return false;
}
bool hip_ok(RcclTransport *t, hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  t->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

// post the current message set (apk_sim_peer: the uniform mesh's halo buffers or, on refined meshes,
// whichever set the driver made current) as one group on the halo stream
int rccl_exchange_begin(void *user) {
  apk_sim *s = static_cast<apk_sim *>(user);
  RcclTransport *t = s->rccl;
  hipStream_t sim_stream = hs(s);
  // everything enqueued so far (pack kernel; the unpack of the previous exchange, which read the
  // receive buffers) must be done before the messages move
  if (!hip_ok(t, hipEventRecord(t->ev_ready, sim_stream), "hipEventRecord")) return 1;
  if (!hip_ok(t, hipStreamWaitEvent(t->s_halo, t->ev_ready, 0), "hipStreamWaitEvent")) return 1;
  const int np = apk_sim_num_peers(s);
  if (!nccl_ok(t, t->api->GroupStart(), "ncclGroupStart")) return 1;
  bool ok = true;
  for (int p = 0; p < np && ok; ++p) {
    apk_peer_info pi;
    if (apk_sim_peer(s, p, &pi) != APK_OK) {
      t->err = "apk_sim_peer failed";
      ok = false;
      break;
    }
    if (pi.send_count > 0)
      ok = ok && nccl_ok(t, t->api->Send(pi.send_buf, (size_t)pi.send_count, ncclDouble, pi.rank, t->halo, t->s_halo), "ncclSend");
    if (pi.recv_count > 0)
      ok = ok && nccl_ok(t, t->api->Recv(pi.recv_buf, (size_t)pi.recv_count, ncclDouble, pi.rank, t->halo, t->s_halo), "ncclRecv");
  }
  if (!nccl_ok(t, t->api->GroupEnd(), "ncclGroupEnd") || !ok) return 1;
  if (!hip_ok(t, hipEventRecord(t->ev_done, t->s_halo), "hipEventRecord")) return 1;
  t->exchanges += 1;
  return 0;
}

// Loopback (one-GPU rehearsal of a rank of the 2 x 2 x 2 run, mesh.hpp "rehearse"): the same ordering as above --
// the halo stream waits for the pack kernel, the sim's stream later waits for the halo stream -- with one
// device-to-device copy per peer standing in for the ncclSend / ncclRecv pair: the message the periodic image of this
// rank would send is this rank's own send buffer.  Everything but the wire (xGMI) is as in the 8-GPU run.
// (ONE kernel for the whole group, as RCCL launches one kernel per ncclGroupEnd: seven hipMemcpyAsync calls were seven
// blit launches of 5 - 25 us each in a row)
constexpr int kLoopbackSpans = 16;
struct LoopbackSpans {
  const double *src[kLoopbackSpans];
  double *dst[kLoopbackSpans];
  long long n[kLoopbackSpans];
};
__global__ void __launch_bounds__(256) loopback_spans_kernel(LoopbackSpans sp) {
  const double *src = sp.src[blockIdx.y];
  double *dst = sp.dst[blockIdx.y];
  const long long n = sp.n[blockIdx.y];
  const long long stride = (long long)gridDim.x * 256;
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const long long n2 = n >> 1;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (long long q = t; q < n2; q += stride) d2[q] = s2[q];
    if ((n & 1) && t == 0) dst[n - 1] = src[n - 1];
  } else {
    for (long long q = t; q < n; q += stride) dst[q] = src[q];
  }
}

int loopback_exchange_begin(void *user) {
  apk_sim *s = static_cast<apk_sim *>(user);
  RcclTransport *t = s->rccl;
  if (!hip_ok(t, hipEventRecord(t->ev_ready, hs(s)), "hipEventRecord")) return 1;
  if (!hip_ok(t, hipStreamWaitEvent(t->s_halo, t->ev_ready, 0), "hipStreamWaitEvent")) return 1;
  const int np = apk_sim_num_peers(s);
  LoopbackSpans sp{};
  int nsp = 0;
  long long longest = 0;
  for (int p = 0; p < np; ++p) {
    apk_peer_info pi;
    if (apk_sim_peer(s, p, &pi) != APK_OK || pi.send_count != pi.recv_count) {
      t->err = "loopback: a peer's send and receive sizes differ";
      return 1;
    }
    if (pi.send_count <= 0) continue;
    t->loopback_bytes += (long long)sizeof(double) * pi.send_count;
    if (np > kLoopbackSpans) {
      if (!hip_ok(t, hipMemcpyAsync(pi.recv_buf, pi.send_buf, sizeof(double) * (size_t)pi.send_count, hipMemcpyDeviceToDevice, t->s_halo),
                  "hipMemcpyAsync"))
        return 1;
      continue;
    }
    sp.src[nsp] = static_cast<const double *>(pi.send_buf);
    sp.dst[nsp] = static_cast<double *>(pi.recv_buf);
    sp.n[nsp] = pi.send_count;
    if (pi.send_count > longest) longest = pi.send_count;
    ++nsp;
  }
  if (nsp > 0) {
    long long gx = (longest / 2 + 256 * 8 - 1) / (256 * 8);  // eight double2 per thread of the longest message
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(loopback_spans_kernel, dim3((unsigned)gx, (unsigned)nsp), dim3(256), 0, t->s_halo, sp);
    if (!hip_ok(t, hipGetLastError(), "loopback_spans_kernel")) return 1;
  }
  if (!hip_ok(t, hipEventRecord(t->ev_done, t->s_halo), "hipEventRecord")) return 1;
  t->exchanges += 1;
  return 0;
}

// work enqueued on the sim's stream from now on (unpack kernel, boundary conditions) waits for the receives
int rccl_exchange_end(void *user) {
  apk_sim *s = static_cast<apk_sim *>(user);
  RcclTransport *t = s->rccl;
  return hip_ok(t, hipStreamWaitEvent(hs(s), t->ev_done, 0), "hipStreamWaitEvent") ? 0 : 1;
}

int rccl_exchange(void *user) {
  if (rccl_exchange_begin(user) != 0) return 1;
  return rccl_exchange_end(user);
}
int loopback_exchange(void *user) {
  if (loopback_exchange_begin(user) != 0) return 1;
  return rccl_exchange_end(user);
}
// (one rank: a reduction over the ranks is the identity)
int loopback_allreduce(void *user, double *, int) {
  static_cast<apk_sim *>(user)->rccl->reductions += 1;
  return 0;
}

// vals live on the host (the driver has synchronised its stream to read them): stage them through
// pinned memory, reduce on the reduction communicator's own stream, wait for that stream only
int rccl_allreduce(void *user, double *vals, int n, ncclRedOp_t op) {
  apk_sim *s = static_cast<apk_sim *>(user);
  RcclTransport *t = s->rccl;
  for (int off = 0; off < n; off += RcclTransport::kRedMax) {
    const int m = (n - off < RcclTransport::kRedMax) ? n - off : RcclTransport::kRedMax;
    std::memcpy(t->h_red, vals + off, sizeof(double) * m);
    if (!hip_ok(t, hipMemcpyAsync(t->d_red, t->h_red, sizeof(double) * m, hipMemcpyHostToDevice, t->s_red), "hipMemcpyAsync")) return 1;
    if (!nccl_ok(t, t->api->AllReduce(t->d_red, t->d_red, (size_t)m, ncclDouble, op, t->red, t->s_red), "ncclAllReduce")) return 1;
    if (!hip_ok(t, hipMemcpyAsync(t->h_red, t->d_red, sizeof(double) * m, hipMemcpyDeviceToHost, t->s_red), "hipMemcpyAsync")) return 1;
    if (!hip_ok(t, hipStreamSynchronize(t->s_red), "hipStreamSynchronize")) return 1;
    std::memcpy(vals + off, t->h_red, sizeof(double) * m);
  }
  t->reductions += 1;
  return 0;
}
int rccl_allreduce_min(void *user, double *vals, int n) { return rccl_allreduce(user, vals, n, ncclMin); }
int rccl_allreduce_sum(void *user, double *vals, int n) { return rccl_allreduce(user, vals, n, ncclSum); }

}  // namespace

void rccl_transport_destroy(RcclTransport *t) {
  if (!t) return;
  if (t->s_halo) (void)hipStreamSynchronize(t->s_halo);
  if (t->s_red) (void)hipStreamSynchronize(t->s_red);
  if (t->api) {
    if (t->halo) (void)t->api->CommDestroy(t->halo);
    if (t->red) (void)t->api->CommDestroy(t->red);
  }
  if (t->ev_ready) (void)hipEventDestroy(t->ev_ready);
  if (t->ev_done) (void)hipEventDestroy(t->ev_done);
  if (t->s_halo) (void)hipStreamDestroy(t->s_halo);
  if (t->s_red) (void)hipStreamDestroy(t->s_red);
  if (t->d_red) (void)hipFree(t->d_red);
  if (t->h_red) (void)hipHostFree(t->h_red);
  delete t;
}

int comm_loopback_attach(apk_sim *s) {
  if (!s || s->rccl || s->have_comm) return fail(s, APK_ERR_INVALID, "loopback transport: the sim has a transport already");
  RcclTransport *t = new RcclTransport();
  const bool ok = hip_ok(t, hipStreamCreateWithFlags(&t->s_halo, hipStreamNonBlocking), "hipStreamCreate") &&
                  hip_ok(t, hipEventCreateWithFlags(&t->ev_ready, hipEventDisableTiming), "hipEventCreate") &&
                  hip_ok(t, hipEventCreateWithFlags(&t->ev_done, hipEventDisableTiming), "hipEventCreate");
  if (!ok) {
    const std::string why = t->err;
    rccl_transport_destroy(t);
    return fail(s, APK_ERR_DEVICE, "loopback transport: " + why);
  }
  s->rccl = t;
  s->comm.user = s;
  s->comm.exchange = loopback_exchange;
  s->comm.exchange_begin = loopback_exchange_begin;
  s->comm.exchange_end = rccl_exchange_end;
  s->comm.allreduce_min = loopback_allreduce;
  s->comm.allreduce_sum = loopback_allreduce;
  s->have_comm = true;
  return APK_OK;
}

RcclTransport *rccl_transport_create(const char *ids, int rank, int nranks, std::string *why) {
  RcclApi *api = rccl_api(why);
  if (!api) return nullptr;
  RcclTransport *t = new RcclTransport();
  t->api = api;
  t->rank = rank;
  t->nranks = nranks;
  ncclUniqueId id_halo, id_red;
  static_assert(sizeof(ncclUniqueId) == APK_RCCL_ID_BYTES, "ncclUniqueId size");
  std::memcpy(&id_halo, ids, sizeof(ncclUniqueId));
  std::memcpy(&id_red, ids + sizeof(ncclUniqueId), sizeof(ncclUniqueId));
  bool ok = hip_ok(t, hipStreamCreateWithFlags(&t->s_halo, hipStreamNonBlocking), "hipStreamCreate") &&
            hip_ok(t, hipStreamCreateWithFlags(&t->s_red, hipStreamNonBlocking), "hipStreamCreate") &&
            hip_ok(t, hipEventCreateWithFlags(&t->ev_ready, hipEventDisableTiming), "hipEventCreate") &&
            hip_ok(t, hipEventCreateWithFlags(&t->ev_done, hipEventDisableTiming), "hipEventCreate") &&
            hip_ok(t, hipMalloc(&t->d_red, sizeof(double) * RcclTransport::kRedMax), "hipMalloc") &&
            hip_ok(t, hipHostMalloc(&t->h_red, sizeof(double) * RcclTransport::kRedMax, hipHostMallocDefault), "hipHostMalloc") &&
            nccl_ok(t, api->CommInitRank(&t->halo, nranks, id_halo, rank), "ncclCommInitRank (halo)") &&
            nccl_ok(t, api->CommInitRank(&t->red, nranks, id_red, rank), "ncclCommInitRank (reductions)");
  if (!ok) {
    if (why) *why = t->err;
    rccl_transport_destroy(t);
    return nullptr;
  }
  return t;
}

}  // namespace host
}  // namespace apk

using namespace apk::host;

extern "C" {

int apk_rccl_unique_ids(char *ids, size_t len) {
  if (!ids || len < 2 * APK_RCCL_ID_BYTES) return APK_ERR_INVALID;
  std::string why;
  RcclApi *api = rccl_api(&why);
  if (!api) return APK_ERR_UNSUPPORTED;
  for (int q = 0; q < 2; ++q) {
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return APK_ERR_DEVICE;
    std::memcpy(ids + q * APK_RCCL_ID_BYTES, &id, sizeof(id));
  }
  return APK_OK;
}

int apk_sim_comm_rccl(apk_sim *s, const char *ids, size_t len) {
  if (!s || !ids || len < 2 * APK_RCCL_ID_BYTES || s->host_only) return APK_ERR_INVALID;
  if (s->rccl) return fail(s, APK_ERR_INVALID, "apk_sim_comm_rccl: the transport exists already");
  std::string why;
  RcclTransport *t = rccl_transport_create(ids, s->rank, s->nranks, &why);
  if (!t) return fail(s, APK_ERR_DEVICE, "RCCL transport: " + why);
  s->rccl = t;
  s->comm.user = s;
  s->comm.exchange = rccl_exchange;
  s->comm.exchange_begin = rccl_exchange_begin;
  s->comm.exchange_end = rccl_exchange_end;
  s->comm.allreduce_min = rccl_allreduce_min;
  s->comm.allreduce_sum = rccl_allreduce_sum;
  s->have_comm = true;
  return APK_OK;
}

int apk_sim_comm_stats(const apk_sim *s, long long *exchanges, long long *reductions) {
  if (!s || !s->rccl) return APK_ERR_INVALID;
  if (exchanges) *exchanges = s->rccl->exchanges;
  if (reductions) *reductions = s->rccl->reductions;
  return APK_OK;
}

const char *apk_sim_comm_error(const apk_sim *s) { return (s && s->rccl) ? s->rccl->err.c_str() : ""; }

// One-rank exercise of everything the transport does (library lookup, two communicators, a grouped
// send/recv pair on the halo stream ordered against a "sim" stream by events, min / sum reductions),
// for boxes with a single GPU: rank 0 sends `n` doubles to itself.
int apk_rccl_selftest(int n, char *msg, size_t len) {
  auto say = [&](const std::string &m) {
    if (msg && len) std::snprintf(msg, len, "%s", m.c_str());
  };
  if (n < 1) return APK_ERR_INVALID;
  char ids[2 * APK_RCCL_ID_BYTES];
  int rc = apk_rccl_unique_ids(ids, sizeof(ids));
  if (rc != APK_OK) {
    say("apk_rccl_unique_ids failed (librccl missing?)");
    return rc;
  }
  std::string why;
  RcclTransport *t = rccl_transport_create(ids, 0, 1, &why);
  if (!t) {
    say("transport: " + why);
    return APK_ERR_DEVICE;
  }
  apk_sim fake;  // only the fields the transport touches
  fake.host_only = false;
  fake.rccl = t;
  hipStream_t sim_stream = nullptr;
  double *d_send = nullptr, *d_recv = nullptr;
  std::vector<double> h((size_t)n), back((size_t)n, -1.0);
  for (int q = 0; q < n; ++q) h[q] = 0.5 * q + 1.0;
  bool ok = hipStreamCreate(&sim_stream) == hipSuccess && hipMalloc(&d_send, sizeof(double) * n) == hipSuccess &&
            hipMalloc(&d_recv, sizeof(double) * n) == hipSuccess;
  fake.stream = reinterpret_cast<apk_stream_t>(sim_stream);
  if (ok) {
    fake.mesh.peers.resize(1);
    fake.mesh.peers[0].rank = 0;
    fake.mesh.peers[0].send_count = fake.mesh.peers[0].recv_count = n;
    fake.send_buf = {d_send};
    fake.recv_buf = {d_recv};
    ok = hipMemcpyAsync(d_send, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, sim_stream) == hipSuccess &&
         hipMemsetAsync(d_recv, 0, sizeof(double) * n, sim_stream) == hipSuccess;
    ok = ok && rccl_exchange_begin(&fake) == 0 && rccl_exchange_end(&fake) == 0;
    ok = ok && hipMemcpyAsync(back.data(), d_recv, sizeof(double) * n, hipMemcpyDeviceToHost, sim_stream) == hipSuccess &&
         hipStreamSynchronize(sim_stream) == hipSuccess;
    if (ok && back != h) {
      t->err = "self send/recv returned different data";
      ok = false;
    }
    double v[3] = {3.0, -1.5, 7.25}, w[3] = {3.0, -1.5, 7.25};
    ok = ok && rccl_allreduce_min(&fake, v, 3) == 0 && rccl_allreduce_sum(&fake, w, 3) == 0;
    if (ok && (v[0] != 3.0 || v[1] != -1.5 || v[2] != 7.25 || w[0] != 3.0 || w[1] != -1.5 || w[2] != 7.25)) {
      t->err = "one-rank reduction changed the values";
      ok = false;
    }
  } else {
    t->err = "hip allocation failed";
  }
  say(ok ? "ok" : t->err);
  if (d_send) (void)hipFree(d_send);
  if (d_recv) (void)hipFree(d_recv);
  if (sim_stream) (void)hipStreamDestroy(sim_stream);
  fake.rccl = nullptr;
  rccl_transport_destroy(t);
  return ok ? APK_OK : APK_ERR_DEVICE;
}

}  // extern "C"
