// TEST / MEASUREMENT INFRASTRUCTURE ONLY.  A second in-process stand-in for the ten RCCL entry points libemap_hip.so resolves with
// dlopen (emap_comm_init) -- like fake_rccl.cpp its "ranks" are THREADS of one process that drive strip contexts on ONE GPU (RCCL
// refuses two ranks on one device), but unlike it every collective is STREAM ORDERED, the way RCCL's are: nothing synchronises a
// stream with the host.  A collective records an event on the caller's stream, the ranks meet at a HOST-ONLY barrier (they exchange
// pointers and event handles, a few microseconds), and each rank then makes its stream wait for the other ranks' events and
// enqueues the data movement itself (a small reduction kernel, device-to-device copies); a second event per rank tells the
// producers when their buffers may be overwritten.  The frame of a strip therefore runs with the same stream structure as under
// RCCL -- all-reduce on the strip's stream between count and fuse, halo send / recv on the second stream next to the interior
// stencils, the wait before the boundary tiles -- which is what tools/strip_emulation.py times and what the large strip parity
// tests drive.  What it cannot show is the wire: the bytes move through HBM, not over xGMI.
//
// STREAM_RCCL_LOOPBACK=1 (tools/strip_emulation.py, per-rank timing): ONE rank of an n-rank communicator runs alone -- no rendezvous, its
// all-reduce returns its own values, every receive copies the bytes of the group's send of the same position (same size, same
// streams, same event hand-offs inside the library; the contents are meaningless).  Timing only, never for a parity check.
//
// Semantics kept from RCCL: ncclCommInitRank is a rendezvous; ncclAllReduce (sum / max of f64 / f32) gives every rank the same
// result (ranks summed in rank order); grouped ncclSend / ncclRecv pair up per (source, destination) in issue order; every rank
// must issue the same sequence of collectives.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
#define SR_MAX_RANKS 16

struct SendOp { const void* buf; size_t bytes; int peer; };
struct Shared {
  std::mutex m; std::condition_variable cv;
  int nranks = 0, joined = 0;
  int bar_count = 0; unsigned long bar_gen = 0;
  hipEvent_t ready[SR_MAX_RANKS], done[SR_MAX_RANKS];
  void* slot[SR_MAX_RANKS]; size_t slot_cap[SR_MAX_RANKS];       // staging copy of every rank's all-reduce input (in-place calls are legal)
  std::vector<SendOp> sends[SR_MAX_RANKS];
  bool failed = false, loopback = false;
};
struct Comm { Shared* sh; int rank, nranks; };
typedef Comm* ncclComm_t;
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
static std::mutex g_m;
static std::map<std::string, Shared*> g_reg;
static int g_next = 1;
static thread_local int t_depth = 0;
static thread_local std::vector<Op> t_ops;

// host-only barrier of the rank threads; false after 120 s (a rank died: fail the collective instead of hanging the test)
static bool barrier(Shared* sh) {
  if (sh->loopback) return true;
  std::unique_lock<std::mutex> l(sh->m);
  if (sh->failed) return false;
  const unsigned long gen = sh->bar_gen;
  if (++sh->bar_count == sh->nranks) { sh->bar_count = 0; sh->bar_gen++; sh->cv.notify_all(); return true; }
  if (!sh->cv.wait_for(l, std::chrono::seconds(120), [&] { return sh->bar_gen != gen || sh->failed; })) { sh->failed = true; sh->cv.notify_all(); return false; }
  return !sh->failed;
}

struct SrcTab { const void* p[SR_MAX_RANKS]; };
template <class T, bool MAX>
__global__ void k_allreduce(SrcTab src, int n, T* __restrict__ dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T acc = static_cast<const T*>(src.p[0])[i];
    for (int r = 1; r < n; ++r) { const T x = static_cast<const T*>(src.p[r])[i]; acc = MAX ? (x > acc ? x : acc) : acc + x; }
    dst[i] = acc;
  }
}

template <class T>
static void launch_allreduce(const SrcTab& tab, int nsrc, void* recv, size_t count, int op, hipStream_t stream) {
  const unsigned int blocks = (unsigned int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
  if (op == 0) hipLaunchKernelGGL((k_allreduce<T, false>), dim3(blocks), dim3(256), 0, stream, tab, nsrc, (T*)recv, count);
  else hipLaunchKernelGGL((k_allreduce<T, true>), dim3(blocks), dim3(256), 0, stream, tab, nsrc, (T*)recv, count);
}
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> l(g_m);
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "stream-rccl-%d", g_next++);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > SR_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Shared* sh;
  { std::lock_guard<std::mutex> l(g_m);
    std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    auto it = g_reg.find(key);
    if (it == g_reg.end()) { sh = new Shared(); sh->nranks = nranks; memset(sh->slot, 0, sizeof sh->slot); memset(sh->slot_cap, 0, sizeof sh->slot_cap); g_reg[key] = sh; }
    else sh = it->second; }
  if (sh->nranks != nranks) return ncclInvalidArgument;
  if (const char* e = getenv("STREAM_RCCL_LOOPBACK")) sh->loopback = atoi(e) != 0;
  if (hipEventCreateWithFlags(&sh->ready[rank], hipEventDisableTiming) != hipSuccess) return ncclInternalError;
  if (hipEventCreateWithFlags(&sh->done[rank], hipEventDisableTiming) != hipSuccess) return ncclInternalError;
  if (sh->loopback) { *comm = new Comm{sh, rank, nranks}; return ncclSuccess; }
  std::unique_lock<std::mutex> l(sh->m);
  sh->joined++;
  sh->cv.notify_all();
  if (!sh->cv.wait_for(l, std::chrono::seconds(120), [&] { return sh->joined >= sh->nranks; })) return ncclInternalError;   // rendezvous, like the real call
  *comm = new Comm{sh, rank, nranks};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }       // (the shared block stays: other ranks may still be inside a collective)
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = comm->nranks; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "stream rccl stand-in: error (a rank failed or timed out?)"; }

#define HK(call) do { if ((call) != hipSuccess) { std::lock_guard<std::mutex> l_(sh->m); sh->failed = true; sh->cv.notify_all(); return ncclInternalError; } } while (0)

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t stream) {
  // ncclFloat64 (8) / ncclFloat32 (7) / ncclInt64 (4) / ncclUint32 (3) with ncclSum (0) / ncclMax (2): all the path uses
  if ((dtype != 8 && dtype != 7 && dtype != 4 && dtype != 3) || (op != 0 && op != 2)) return ncclInvalidArgument;
  Shared* sh = c->sh; const int me = c->rank, n = c->nranks;
  const size_t bytes = count * ((dtype == 8 || dtype == 4) ? 8 : 4);
  if (bytes > sh->slot_cap[me]) {                       // (growing the staging slot synchronises the device: only the first call of a size does)
    if (sh->slot[me]) HK(hipFree(sh->slot[me]));
    sh->slot[me] = nullptr; sh->slot_cap[me] = 0;
    HK(hipMalloc(&sh->slot[me], bytes));
    sh->slot_cap[me] = bytes;
  }
  HK(hipMemcpyAsync(sh->slot[me], send, bytes, hipMemcpyDeviceToDevice, stream));
  HK(hipEventRecord(sh->ready[me], stream));
  if (!barrier(sh)) return ncclInternalError;            // every rank's input is on its way, every slot pointer is final
  SrcTab tab; memset(&tab, 0, sizeof tab);
  if (sh->loopback) tab.p[0] = sh->slot[me];
  else for (int r = 0; r < n; ++r) { tab.p[r] = sh->slot[r]; if (r != me) HK(hipStreamWaitEvent(stream, sh->ready[r], 0)); }
  const int nsrc = sh->loopback ? 1 : n;
  if (dtype == 8) launch_allreduce<double>(tab, nsrc, recv, count, op, stream);
  else if (dtype == 7) launch_allreduce<float>(tab, nsrc, recv, count, op, stream);
  else if (dtype == 4) launch_allreduce<long long>(tab, nsrc, recv, count, op, stream);
  else launch_allreduce<unsigned int>(tab, nsrc, recv, count, op, stream);
  HK(hipGetLastError());
  HK(hipEventRecord(sh->done[me], stream));
  if (!barrier(sh)) return ncclInternalError;
  if (!sh->loopback) for (int r = 0; r < n; ++r) if (r != me) HK(hipStreamWaitEvent(stream, sh->done[r], 0));      // my slot is free again only when every rank has read it
  return ncclSuccess;
}

static ncclResult_t run_ops(std::vector<Op>& ops) {
  if (ops.empty()) return ncclSuccess;
  Comm* c = ops[0].comm; Shared* sh = c->sh; const int me = c->rank, n = c->nranks;
  hipStream_t stream = ops[0].stream;
  for (auto& o : ops) if (o.comm != c || o.stream != stream) { ops.clear(); return ncclInvalidArgument; }   // one communicator, one stream per group: all the path uses
  sh->sends[me].clear();
  for (auto& o : ops) if (o.send) sh->sends[me].push_back(SendOp{o.buf, o.bytes, o.peer});
  HK(hipEventRecord(sh->ready[me], stream));             // my send buffers are final once this event has passed
  if (!barrier(sh)) { ops.clear(); return ncclInternalError; }
  int taken[SR_MAX_RANKS]; memset(taken, 0, sizeof taken);   // receives pair with the peer's sends to me in issue order
  bool waited[SR_MAX_RANKS]; memset(waited, 0, sizeof waited);
  int nrecv = 0;
  for (auto& o : ops) if (!o.send) {
    const int p = o.peer;
    if (p < 0 || p >= n) { ops.clear(); return ncclInvalidArgument; }
    const SendOp* match = nullptr; int k = 0;
    if (sh->loopback) { if ((size_t)nrecv < sh->sends[me].size()) match = &sh->sends[me][nrecv]; nrecv++; }
    else for (auto& s : sh->sends[p]) if (s.peer == me && k++ == taken[p]) { match = &s; break; }
    if (!match || match->bytes != o.bytes) { std::lock_guard<std::mutex> l(sh->m); sh->failed = true; sh->cv.notify_all(); ops.clear(); return ncclInvalidArgument; }
    taken[p]++;
    if (!sh->loopback && p != me && !waited[p]) { HK(hipStreamWaitEvent(stream, sh->ready[p], 0)); waited[p] = true; }
    HK(hipMemcpyAsync(o.buf, match->buf, o.bytes, hipMemcpyDeviceToDevice, stream));
  }
  HK(hipEventRecord(sh->done[me], stream));
  if (!barrier(sh)) { ops.clear(); return ncclInternalError; }
  memset(waited, 0, sizeof waited);
  if (!sh->loopback) for (auto& o : ops) if (o.send && o.peer != me && !waited[o.peer]) { HK(hipStreamWaitEvent(stream, sh->done[o.peer], 0)); waited[o.peer] = true; }   // a send completes when its payload has been consumed
  ops.clear();
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--t_depth == 0) return run_ops(t_ops); return ncclSuccess; }
static ncclResult_t p2p(bool send, void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  if (dtype != 0) return ncclInvalidArgument;                          // ncclChar: byte counts
  t_ops.push_back(Op{send, buf, count, peer, c, s});
  if (t_depth == 0) return run_ops(t_ops);
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) { return p2p(true, const_cast<void*>(buf), count, dtype, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) { return p2p(false, buf, count, dtype, peer, c, s); }
}
