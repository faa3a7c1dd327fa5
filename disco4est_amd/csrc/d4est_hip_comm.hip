// RCCL transport of the ghost exchange, in C behind the plan's communication hooks.
//
// Replaces d4est_ghost_data_exchange (src/Mesh/d4est_ghost_data.c:143-256: per-peer MPI_Irecv / pack / MPI_Isend / Waitall of whole
// mirror elements) by: ONE pack kernel (face-trace blocks of all peers into one send buffer) -> ncclGroupStart; ncclRecv / ncclSend
// per neighbouring rank; ncclGroupEnd on a communication stream -> ONE unpack kernel into the ghost trace buffer.  Posted right after
// the trace kernel, completed right before the flux kernel, so the transfers run beside the volume kernel (which needs no ghost data).
// sc_allreduce of the CG / Lanczos scalars (src/Solver/d4est_solver_cg_eigs.c:181-243) becomes ncclAllReduce.
// EVERY operation of a communicator -- the grouped send / receive rounds, the all-reduces, the stand-alone entries -- is issued on the
// communicator's OWN stream, with an event in from and an event out to the caller's stream: one communicator, one stream, so the order
// in which RCCL sees the operations is the host's issue order on every rank, whatever streams the plans run on.
// No Python, no host synchronisation in the per-apply path.
//
// librccl is opened at run time (dlopen), so the library loads -- and every single-GPU entry point works -- on a box without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "d4est_hip_internal.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.handle) return api;
  // a library already mapped into the process first (a torch process carries its own librccl next to its own HIP runtime; mixing
  // that runtime with another ROCm's RCCL is the one combination to avoid), then the system one
  const char* env = std::getenv("D4EST_HIP_RCCL_LIB");
  const char* names[] = {env, "librccl.so", "librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.handle; ++pass)
    for (const char* n : names) {
      if (!n) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.handle) break;
    }
  if (!api.handle) D4EST_HIP_ABORT("RCCL: cannot open librccl (%s); set D4EST_HIP_RCCL_LIB", dlerror());
#define LOAD(field, sym)                                                     \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
  if (!api.field) D4EST_HIP_ABORT("RCCL: symbol %s not found", sym)
  LOAD(GetUniqueId, "ncclGetUniqueId");
  LOAD(CommInitRank, "ncclCommInitRank");
  LOAD(CommDestroy, "ncclCommDestroy");
  LOAD(CommCount, "ncclCommCount");
  LOAD(GroupStart, "ncclGroupStart");
  LOAD(GroupEnd, "ncclGroupEnd");
  LOAD(Send, "ncclSend");
  LOAD(Recv, "ncclRecv");
  LOAD(AllReduce, "ncclAllReduce");
  LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
  return api;
}

#define RCCL_CHECK(expr)                                                                                   \
  do {                                                                                                     \
    ncclResult_t _r = (expr);                                                                              \
    if (_r != ncclSuccess) D4EST_HIP_ABORT("%s failed: %s", #expr, rccl().GetErrorString(_r));             \
  } while (0)

}  // namespace

struct d4est_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;               // the one stream this communicator's operations run on
  hipEvent_t ev_in = nullptr, ev_out = nullptr;   // hand-over from / to a caller's stream (stand-alone entries, all-reduce)
};

// per plan: the exchange lists and buffers
struct d4est_hip_rccl_exchange {
  d4est_hip_comm* comm = nullptr;
  d4est_hip_plan* plan = nullptr;
  std::vector<int> peer;
  std::vector<long long> send_first, recv_first;   // per peer: first double of its range in the packed send / receive buffer
  long long send_total = 0, recv_total = 0;
  int n_send_blocks = 0, n_recv_blocks = 0;
  double *d_send = nullptr, *d_recv = nullptr;
  long long *d_s_src = nullptr, *d_s_dst = nullptr, *d_r_src = nullptr, *d_r_dst = nullptr;
  int *d_s_len = nullptr, *d_r_len = nullptr;
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;   // the exchange's own hand-over events (the stream is the communicator's)
  long long n_exchanges = 0;
};

namespace {

template <typename T>
T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

void exchange_hook(void* ctx, int phase, const double* trace_dev, double* ghost_trace_dev) {
  auto* x = static_cast<d4est_hip_rccl_exchange*>(ctx);
  RcclApi& api = rccl();
  hipStream_t s = x->plan->stream;
  if (phase == 0) {
    // pack on the plan's stream (behind the trace kernel), transfers on the communication stream
    d4est_hip::launch_copy_blocks(s, x->n_send_blocks, trace_dev, x->d_s_src, x->d_send, x->d_s_dst, x->d_s_len);
    hipStream_t cs = x->comm->stream;
    HIP_CHECK(hipEventRecord(x->ev_packed, s));
    HIP_CHECK(hipStreamWaitEvent(cs, x->ev_packed, 0));
    RCCL_CHECK(api.GroupStart());
    for (size_t p = 0; p < x->peer.size(); ++p) {
      const long long nr = x->recv_first[p + 1] - x->recv_first[p], ns = x->send_first[p + 1] - x->send_first[p];
      if (nr > 0) RCCL_CHECK(api.Recv(x->d_recv + x->recv_first[p], (size_t)nr, ncclDouble, x->peer[p], x->comm->comm, cs));
      if (ns > 0) RCCL_CHECK(api.Send(x->d_send + x->send_first[p], (size_t)ns, ncclDouble, x->peer[p], x->comm->comm, cs));
    }
    RCCL_CHECK(api.GroupEnd());
    HIP_CHECK(hipEventRecord(x->ev_arrived, cs));
    x->n_exchanges++;
  } else {
    HIP_CHECK(hipStreamWaitEvent(s, x->ev_arrived, 0));
    d4est_hip::launch_copy_blocks(s, x->n_recv_blocks, x->d_recv, x->d_r_src, ghost_trace_dev, x->d_r_dst, x->d_r_len);
  }
}

// run fn(communicator stream) ordered after what `user` has queued, and make `user` wait for it
template <typename F>
void on_comm_stream(d4est_hip_comm* c, hipStream_t user, F fn) {
  HIP_CHECK(hipEventRecord(c->ev_in, user));
  HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  fn(c->stream);
  HIP_CHECK(hipEventRecord(c->ev_out, c->stream));
  HIP_CHECK(hipStreamWaitEvent(user, c->ev_out, 0));
}

void allreduce_hook(void* ctx, double* scalars_dev, int n) {
  auto* x = static_cast<d4est_hip_rccl_exchange*>(ctx);
  if (x->comm->world == 1) return;
  on_comm_stream(x->comm, x->plan->stream, [&](hipStream_t cs) {
    RCCL_CHECK(rccl().AllReduce(scalars_dev, scalars_dev, (size_t)n, ncclDouble, ncclSum, x->comm->comm, cs));
  });
}

void comm_streams_create(d4est_hip_comm* c) {
  HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
}

}  // namespace

extern "C" {

int d4est_hip_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

void d4est_hip_comm_get_unique_id(void* id_out) {
  if (!id_out) D4EST_HIP_ABORT("comm_get_unique_id: NULL");
  ncclUniqueId id;
  RCCL_CHECK(rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
}

d4est_hip_comm_t* d4est_hip_comm_create(const void* unique_id, int rank, int world) {
  if (!unique_id || world < 1 || rank < 0 || rank >= world) D4EST_HIP_ABORT("comm_create: rank %d of %d", rank, world);
  auto* c = new d4est_hip_comm();
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  RCCL_CHECK(rccl().CommInitRank(&c->comm, world, id, rank));
  comm_streams_create(c);
  return c;
}

// same, but a failing ncclCommInitRank (e.g. two ranks on one GPU) returns NULL instead of aborting: lets a host fall back
d4est_hip_comm_t* d4est_hip_comm_try_create(const void* unique_id, int rank, int world) {
  if (!unique_id || world < 1 || rank < 0 || rank >= world) return nullptr;
  auto* c = new d4est_hip_comm();
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  const ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    std::fprintf(stderr, "[d4est_hip] ncclCommInitRank failed on rank %d of %d: %s\n", rank, world, rccl().GetErrorString(r));
    delete c;
    return nullptr;
  }
  comm_streams_create(c);
  return c;
}

void d4est_hip_comm_destroy(d4est_hip_comm_t* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->stream) { (void)hipStreamDestroy(c->stream); (void)hipEventDestroy(c->ev_in); (void)hipEventDestroy(c->ev_out); }
  delete c;
}

int d4est_hip_comm_rank(const d4est_hip_comm_t* c) { return c ? c->rank : -1; }
// the number of ranks RCCL itself reports for the communicator (ncclCommCount), not what the host passed in
int d4est_hip_comm_nccl_count(const d4est_hip_comm_t* c) {
  if (!c || !c->comm) return 0;
  int n = 0;
  RCCL_CHECK(rccl().CommCount(c->comm, &n));
  return n;
}
int d4est_hip_comm_size(const d4est_hip_comm_t* c) { return c ? c->world : 0; }

d4est_hip_rccl_exchange_t* d4est_hip_plan_set_rccl_exchange(d4est_hip_plan_t* plan, d4est_hip_comm_t* comm, int n_peers, const int* peer_rank,
                                                            const int* send_first, const long long* send_off, const int* send_len,
                                                            const int* recv_first, const long long* recv_off, const int* recv_len) {
  if (!plan || !comm) D4EST_HIP_ABORT("plan_set_rccl_exchange: NULL plan / communicator");
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_rccl_exchange: call plan_set_faces first");
  if (n_peers < 0 || (n_peers > 0 && (!peer_rank || !send_first || !recv_first))) D4EST_HIP_ABORT("plan_set_rccl_exchange: bad peer lists");
  if (n_peers > 0 && ((send_first[n_peers] > send_first[0] && (!send_off || !send_len)) || (recv_first[n_peers] > recv_first[0] && (!recv_off || !recv_len))))
    D4EST_HIP_ABORT("plan_set_rccl_exchange: block lists are NULL");
  auto* x = new d4est_hip_rccl_exchange();
  x->comm = comm;
  x->plan = plan;
  std::vector<long long> s_src, s_dst, r_src, r_dst;
  std::vector<int> s_len, r_len;
  x->send_first.assign(1, 0);
  x->recv_first.assign(1, 0);
  for (int p = 0; p < n_peers; ++p) {
    if (peer_rank[p] < 0 || peer_rank[p] >= comm->world) D4EST_HIP_ABORT("plan_set_rccl_exchange: peer %d outside the communicator", peer_rank[p]);
    x->peer.push_back(peer_rank[p]);
    for (int b = send_first[p]; b < send_first[p + 1]; ++b) {
      if (send_off[b] < 0 || send_len[b] < 0 || send_off[b] + send_len[b] > plan->local_trace_doubles) D4EST_HIP_ABORT("plan_set_rccl_exchange: send block %d outside the trace buffer", b);
      s_src.push_back(send_off[b]); s_dst.push_back(x->send_total); s_len.push_back(send_len[b]);
      x->send_total += send_len[b];
    }
    for (int b = recv_first[p]; b < recv_first[p + 1]; ++b) {
      if (recv_off[b] < 0 || recv_len[b] < 0 || recv_off[b] + recv_len[b] > plan->ghost_trace_doubles) D4EST_HIP_ABORT("plan_set_rccl_exchange: receive block %d outside the ghost trace buffer", b);
      r_src.push_back(x->recv_total); r_dst.push_back(recv_off[b]); r_len.push_back(recv_len[b]);
      x->recv_total += recv_len[b];
    }
    x->send_first.push_back(x->send_total);
    x->recv_first.push_back(x->recv_total);
  }
  x->n_send_blocks = (int)s_len.size();
  x->n_recv_blocks = (int)r_len.size();
  HIP_CHECK(hipMalloc(&x->d_send, std::max<size_t>((size_t)x->send_total, 1) * sizeof(double)));
  HIP_CHECK(hipMalloc(&x->d_recv, std::max<size_t>((size_t)x->recv_total, 1) * sizeof(double)));
  x->d_s_src = to_device(s_src); x->d_s_dst = to_device(s_dst); x->d_s_len = to_device(s_len);
  x->d_r_src = to_device(r_src); x->d_r_dst = to_device(r_dst); x->d_r_len = to_device(r_len);
  HIP_CHECK(hipEventCreateWithFlags(&x->ev_packed, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&x->ev_arrived, hipEventDisableTiming));
  d4est_hip_plan_set_comm(plan, plan->ghost_trace_doubles > 0 ? exchange_hook : nullptr, allreduce_hook, x);
  return x;
}

void d4est_hip_rccl_exchange_destroy(d4est_hip_rccl_exchange_t* x) {
  if (!x) return;
  // (a plan destroyed before its exchange clears this back-pointer: d4est_hip_rccl_exchange_detach_plan, called by plan_destroy)
  if (x->plan && x->plan->comm_ctx == x) d4est_hip_plan_set_comm(x->plan, nullptr, nullptr, nullptr);
  if (x->comm && x->comm->stream) (void)hipStreamSynchronize(x->comm->stream);
  (void)hipFree(x->d_send); (void)hipFree(x->d_recv);
  (void)hipFree(x->d_s_src); (void)hipFree(x->d_s_dst); (void)hipFree(x->d_s_len);
  (void)hipFree(x->d_r_src); (void)hipFree(x->d_r_dst); (void)hipFree(x->d_r_len);
  (void)hipEventDestroy(x->ev_packed); (void)hipEventDestroy(x->ev_arrived);
  delete x;
}

// plan_destroy: an exchange object that outlives its plan must not touch it again
void d4est_hip_rccl_exchange_detach_plan(d4est_hip_plan_t* plan) {
  if (!plan || !plan->comm_ctx || (plan->exchange_fn != exchange_hook && plan->allreduce_fn != allreduce_hook)) return;
  static_cast<d4est_hip_rccl_exchange*>(plan->comm_ctx)->plan = nullptr;
}

long long d4est_hip_rccl_exchange_count(const d4est_hip_rccl_exchange_t* x) { return x ? x->n_exchanges : 0; }
long long d4est_hip_rccl_exchange_send_doubles(const d4est_hip_rccl_exchange_t* x) { return x ? x->send_total : 0; }
long long d4est_hip_rccl_exchange_recv_doubles(const d4est_hip_rccl_exchange_t* x) { return x ? x->recv_total : 0; }

// stand-alone form for tests / other payloads (whole-element exchanges of the Schwarz smoother): one grouped point-to-point round,
// buffers already packed; ordered on `plan`'s stream
void d4est_hip_comm_sendrecv(d4est_hip_comm_t* comm, d4est_hip_plan_t* plan, int n_peers, const int* peer_rank, const double* send_dev,
                             const long long* send_first, double* recv_dev, const long long* recv_first) {
  if (!comm || !plan) D4EST_HIP_ABORT("comm_sendrecv: NULL argument");
  if (n_peers < 0 || (n_peers > 0 && (!peer_rank || !send_first || !recv_first))) D4EST_HIP_ABORT("comm_sendrecv: bad peer lists");
  for (int p = 0; p < n_peers; ++p)
    if (peer_rank[p] < 0 || peer_rank[p] >= comm->world) D4EST_HIP_ABORT("comm_sendrecv: peer %d outside the communicator of %d ranks", peer_rank[p], comm->world);
  RcclApi& api = rccl();
  on_comm_stream(comm, plan->stream, [&](hipStream_t cs) {
    RCCL_CHECK(api.GroupStart());
    for (int p = 0; p < n_peers; ++p) {
      const long long nr = recv_first[p + 1] - recv_first[p], ns = send_first[p + 1] - send_first[p];
      if (nr > 0) RCCL_CHECK(api.Recv(recv_dev + recv_first[p], (size_t)nr, ncclDouble, peer_rank[p], comm->comm, cs));
      if (ns > 0) RCCL_CHECK(api.Send(send_dev + send_first[p], (size_t)ns, ncclDouble, peer_rank[p], comm->comm, cs));
    }
    RCCL_CHECK(api.GroupEnd());
  });
}

void d4est_hip_comm_allreduce_sum(d4est_hip_comm_t* comm, d4est_hip_plan_t* plan, double* scalars_dev, int n) {
  if (!comm || !plan) D4EST_HIP_ABORT("comm_allreduce_sum: NULL argument");
  on_comm_stream(comm, plan->stream, [&](hipStream_t cs) {
    RCCL_CHECK(rccl().AllReduce(scalars_dev, scalars_dev, (size_t)n, ncclDouble, ncclSum, comm->comm, cs));
  });
}

}  // extern "C"
