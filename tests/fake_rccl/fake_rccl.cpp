// TEST INFRASTRUCTURE ONLY.  An in-process stand-in for the ten RCCL entry points libemap_hip.so resolves with dlopen
// (emap_comm_init): several "ranks" are THREADS of one process driving strip contexts on ONE GPU -- RCCL itself refuses two ranks
// on one device, and the test box has one.  Semantics kept from RCCL: ncclCommInitRank is a rendezvous of all ranks;
// ncclAllReduce returns the sum to every rank; grouped ncclSend / ncclRecv pair up by (source, destination) in issue order and a
// group only completes when its receives have arrived AND its sends have been consumed.  Everything is done with blocking host
// synchronisation and copies (performance is irrelevant here); stream order is respected by synchronising the caller's stream.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct Msg { const void* src; size_t bytes; bool taken; };
struct Shared {
  std::mutex m; std::condition_variable cv;
  int nranks = 0, joined = 0;
  int ar_gen = 0, ar_arrived = 0; std::vector<const void*> ar_src; std::vector<void*> ar_dst; size_t ar_count = 0;
  std::map<std::pair<int, int>, std::deque<Msg*>> box;
};
struct Comm { Shared* sh; int rank, nranks; };
typedef Comm* ncclComm_t;
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
static std::mutex g_m;
static std::map<std::string, Shared*> g_reg;
static int g_next = 1;
static thread_local int t_depth = 0;
static thread_local std::vector<Op> t_ops;

template <class T>
static void host_reduce(const std::vector<std::vector<char>>& in, std::vector<char>& out, size_t count, int op) {
  T* o = reinterpret_cast<T*>(out.data());
  for (size_t k = 0; k < count; ++k) {
    T acc = reinterpret_cast<const T*>(in[0].data())[k];
    for (size_t r = 1; r < in.size(); ++r) { const T x = reinterpret_cast<const T*>(in[r].data())[k]; acc = (op == 0) ? (T)(acc + x) : (x > acc ? x : acc); }
    o[k] = acc;
  }
}
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> l(g_m);
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", g_next++);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  Shared* sh;
  { std::lock_guard<std::mutex> l(g_m);
    std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    auto it = g_reg.find(key);
    if (it == g_reg.end()) { sh = new Shared(); sh->nranks = nranks; sh->ar_src.resize(nranks); sh->ar_dst.resize(nranks); g_reg[key] = sh; }
    else sh = it->second; }
  if (sh->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::unique_lock<std::mutex> l(sh->m);
  sh->joined++;
  sh->cv.notify_all();
  sh->cv.wait(l, [&] { return sh->joined >= sh->nranks; });          // rendezvous, like the real call
  *comm = new Comm{sh, rank, nranks};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = comm->nranks; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t stream) {
  // ncclFloat64 (8) / ncclFloat32 (7) / ncclInt64 (4) / ncclUint32 (3) with ncclSum (0) / ncclMax (2): all the path uses (ranks reduced in rank order)
  if ((dtype != 8 && dtype != 7 && dtype != 4 && dtype != 3) || (op != 0 && op != 2)) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclInternalError;
  const size_t esz = (dtype == 8 || dtype == 4) ? 8 : 4;
  Shared* sh = c->sh;
  std::unique_lock<std::mutex> l(sh->m);
  const int gen = sh->ar_gen;
  sh->ar_src[c->rank] = send; sh->ar_dst[c->rank] = recv; sh->ar_count = count;
  if (++sh->ar_arrived == sh->nranks) {
    std::vector<std::vector<char>> in(sh->nranks, std::vector<char>(count * esz));
    std::vector<char> out(count * esz);
    for (int r = 0; r < sh->nranks; ++r)
      if (hipMemcpy(in[r].data(), sh->ar_src[r], count * esz, hipMemcpyDeviceToHost) != hipSuccess) return ncclInternalError;
    if (dtype == 8) host_reduce<double>(in, out, count, op);
    else if (dtype == 7) host_reduce<float>(in, out, count, op);
    else if (dtype == 4) host_reduce<long long>(in, out, count, op);
    else host_reduce<unsigned int>(in, out, count, op);
    for (int r = 0; r < sh->nranks; ++r)
      if (hipMemcpy(sh->ar_dst[r], out.data(), count * esz, hipMemcpyHostToDevice) != hipSuccess) return ncclInternalError;
    sh->ar_arrived = 0; sh->ar_gen++;
    sh->cv.notify_all();
  } else sh->cv.wait(l, [&] { return sh->ar_gen != gen; });
  return ncclSuccess;
}

static ncclResult_t run_ops(std::vector<Op>& ops) {
  if (ops.empty()) return ncclSuccess;
  for (auto& o : ops) if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclInternalError;   // everything issued before the group
  std::vector<Msg*> mine;
  for (auto& o : ops) if (o.send) {                                   // 1. post the sends
    Msg* m = new Msg{o.buf, o.bytes, false};
    std::lock_guard<std::mutex> l(o.comm->sh->m);
    o.comm->sh->box[{o.comm->rank, o.peer}].push_back(m);
    o.comm->sh->cv.notify_all();
    mine.push_back(m);
  }
  for (auto& o : ops) if (!o.send) {                                  // 2. take the receives, in issue order per (source, destination)
    Shared* sh = o.comm->sh;
    Msg* m = nullptr;
    { std::unique_lock<std::mutex> l(sh->m);
      auto& q = sh->box[{o.peer, o.comm->rank}];
      sh->cv.wait(l, [&] { return !q.empty(); });
      m = q.front(); q.pop_front(); }
    if (m->bytes != o.bytes) return ncclInvalidArgument;
    if (hipMemcpy(o.buf, m->src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclInternalError;
    if (hipDeviceSynchronize() != hipSuccess) return ncclInternalError;
    { std::lock_guard<std::mutex> l(sh->m); m->taken = true; sh->cv.notify_all(); }
  }
  for (size_t k = 0, j = 0; k < ops.size(); ++k) if (ops[k].send) {   // 3. a send completes when its payload has been consumed
    Shared* sh = ops[k].comm->sh;
    Msg* m = mine[j++];
    std::unique_lock<std::mutex> l(sh->m);
    sh->cv.wait(l, [&] { return m->taken; });
    delete m;
  }
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
