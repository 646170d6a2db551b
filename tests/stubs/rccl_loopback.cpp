// tests/stubs/rccl_loopback.cpp -- TEST INFRASTRUCTURE: a loop-back stand-in for librccl, for boxes with fewer devices than ranks.
//
// libsjgpu opens RCCL with dlopen on the first sjgpu_comm_* call (SJGPU_RCCL_LIB names the library); pointed at THIS one, the ranks of a
// communicator are THREADS of one process and every transfer is a copy -- so sjgpu_comm_gather_indices' own code (the all-gather of
// counts, the growth round of the root's staging array, the grouped exact-count sends and receives, the widening) runs with a world of two
// and three on a box with one GPU (tests/test_gpu_comm.py) or with none (tests/test_comm_emu.py: sjgpu_comm.hip compiled against
// tests/host/emu, "device" memory = host memory).  It implements the ten entry points that file binds, with NCCL's semantics where
// the caller could tell the difference:
//   * ncclCommInitRank returns when all `nranks` ranks of the id have joined (as the real one does);
//   * ncclAllGather is a collective of all ranks of the communicator;
//   * ncclSend / ncclRecv match in posting order per (sender, receiver) pair; inside a group everything is posted first and completed by
//     ncclGroupEnd, a group without operations completes at once and waits for nobody; counts and types of a matched pair must agree
//     (the real library would hang or corrupt memory: here the receiver gets ncclInvalidArgument);
//   * data is taken from / delivered to the stream's view of memory: the sender's stream is drained before its buffer is offered, the
//     receiver's copy is enqueued on its stream and waited for.
// Two builds of this one file (simdjson_amd/build.py: build_rccl_loopback): with hipcc against the real headers (GPU tier), with g++
// against tests/host/emu (CPU tier).
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

size_t type_bytes(ncclDataType_t t) {
  switch (int(t)) {
  case 0: case 1: return 1;  // int8 / uint8
  case 2: case 3: return 4;  // int32 / uint32
  case 4: case 5: return 8;  // int64 / uint64
  default: return 0;
  }
}

struct posted_send {
  const void *buf;
  size_t count;
  ncclDataType_t type;
  bool taken = false; // the receiver has copied it
  bool bad = false;   // ... or refused it (count / type mismatch)
};

struct world {
  int nranks = 0;
  std::mutex m;
  std::condition_variable cv;
  int joined = 0, left = 0;
  // barrier of the collectives
  int bar_count = 0;
  uint64_t bar_gen = 0;
  std::vector<const void *> gather_src;
  // mail[src][dst]: sends posted and not yet taken, oldest first
  std::vector<std::vector<std::deque<std::shared_ptr<posted_send>>>> mail;
  uint64_t sends_completed = 0, collectives = 0; // counters a test can read (loopback_stats)
};

std::mutex g_registry_m;
std::map<std::string, std::shared_ptr<world>> g_registry;
uint64_t g_next_id = 1;

constexpr int WAIT_SECONDS = 60; // a rank that never arrives fails the others instead of hanging the test run

bool barrier(world &w, std::unique_lock<std::mutex> &lk) {
  const uint64_t gen = w.bar_gen;
  if (++w.bar_count == w.nranks) {
    w.bar_count = 0;
    w.bar_gen++;
    w.cv.notify_all();
    return true;
  }
  return w.cv.wait_for(lk, std::chrono::seconds(WAIT_SECONDS), [&] { return w.bar_gen != gen; });
}

struct pending_op {
  bool is_send;
  void *buf;
  size_t count;
  ncclDataType_t type;
  int peer;
  ncclComm *comm;
  hipStream_t stream;
};
thread_local int t_group_depth = 0;
thread_local std::vector<pending_op> t_group_ops;

} // namespace

struct ncclComm {
  std::shared_ptr<world> w;
  int rank = 0;
};

namespace {

ncclResult_t run_ops(std::vector<pending_op> &ops) {
  ncclResult_t first = ncclSuccess;
  // 1. post every send (its data as the sender's stream leaves it)
  std::vector<std::pair<world *, std::shared_ptr<posted_send>>> mine;
  for (pending_op &op : ops) {
    if (!op.is_send) { continue; }
    if (hipStreamSynchronize(op.stream) != hipSuccess) { first = ncclUnhandledCudaError; continue; }
    world &w = *op.comm->w;
    auto ps = std::make_shared<posted_send>();
    ps->buf = op.buf;
    ps->count = op.count;
    ps->type = op.type;
    {
      std::lock_guard<std::mutex> lk(w.m);
      w.mail[size_t(op.comm->rank)][size_t(op.peer)].push_back(ps);
    }
    w.cv.notify_all();
    mine.emplace_back(&w, ps);
  }
  // 2. complete every receive in posting order
  for (pending_op &op : ops) {
    if (op.is_send) { continue; }
    world &w = *op.comm->w;
    std::shared_ptr<posted_send> ps;
    {
      std::unique_lock<std::mutex> lk(w.m);
      auto &q = w.mail[size_t(op.peer)][size_t(op.comm->rank)];
      if (!w.cv.wait_for(lk, std::chrono::seconds(WAIT_SECONDS), [&] { return !q.empty(); })) {
        if (first == ncclSuccess) { first = ncclRemoteError; }
        continue;
      }
      ps = q.front();
      q.pop_front();
    }
    bool ok = ps->count == op.count && ps->type == op.type;
    if (ok && op.count) {
      ok = hipMemcpyAsync(op.buf, ps->buf, op.count * type_bytes(op.type), hipMemcpyDefault, op.stream) == hipSuccess && hipStreamSynchronize(op.stream) == hipSuccess;
    }
    {
      std::lock_guard<std::mutex> lk(w.m);
      ps->taken = true;
      ps->bad = !ok;
      w.sends_completed++;
    }
    w.cv.notify_all();
    if (!ok && first == ncclSuccess) { first = ncclInvalidArgument; }
  }
  // 3. my sends are complete when their receivers have taken them (the buffer may be reused from then on)
  for (auto &pr : mine) {
    std::unique_lock<std::mutex> lk(pr.first->m);
    if (!pr.first->cv.wait_for(lk, std::chrono::seconds(WAIT_SECONDS), [&] { return pr.second->taken; })) {
      if (first == ncclSuccess) { first = ncclRemoteError; }
    } else if (pr.second->bad && first == ncclSuccess) {
      first = ncclInvalidArgument;
    }
  }
  ops.clear();
  return first;
}

ncclResult_t post(bool is_send, const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || peer < 0 || peer >= comm->w->nranks || type_bytes(type) == 0 || (count && !buf)) { return ncclInvalidArgument; }
  if (peer == comm->rank) { return ncclInvalidUsage; } // (real NCCL allows self send/recv pairs inside a group; nothing here needs them)
  t_group_ops.push_back(pending_op{is_send, const_cast<void *>(buf), count, type, peer, comm, stream});
  if (t_group_depth == 0) { return run_ops(t_group_ops); }
  return ncclSuccess;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) { return ncclInvalidArgument; }
  std::lock_guard<std::mutex> lk(g_registry_m);
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "sjgpu-loopback-%llu", static_cast<unsigned long long>(g_next_id++));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) { return ncclInvalidArgument; }
  std::shared_ptr<world> w;
  {
    std::lock_guard<std::mutex> lk(g_registry_m);
    const std::string key(id.internal, sizeof id.internal);
    auto it = g_registry.find(key);
    if (it == g_registry.end()) {
      w = std::make_shared<world>();
      w->nranks = nranks;
      w->gather_src.assign(size_t(nranks), nullptr);
      w->mail.assign(size_t(nranks), std::vector<std::deque<std::shared_ptr<posted_send>>>(size_t(nranks)));
      g_registry[key] = w;
    } else {
      w = it->second;
    }
  }
  if (w->nranks != nranks) { return ncclInvalidArgument; }
  {
    std::unique_lock<std::mutex> lk(w->m);
    w->joined++;
    w->cv.notify_all();
    if (!w->cv.wait_for(lk, std::chrono::seconds(WAIT_SECONDS), [&] { return w->joined >= nranks; })) { return ncclRemoteError; }
  }
  ncclComm *c = new ncclComm();
  c->w = w;
  c->rank = rank;
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) { return ncclInvalidArgument; }
  {
    std::lock_guard<std::mutex> lk(comm->w->m);
    comm->w->left++;
  }
  delete comm; // the world itself stays in the registry (a few hundred bytes per communicator a test made)
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  if (!comm || !count) { return ncclInvalidArgument; }
  *count = comm->w->nranks;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  const size_t bytes = sendcount * type_bytes(datatype);
  if (!comm || !sendbuff || !recvbuff || type_bytes(datatype) == 0) { return ncclInvalidArgument; }
  if (hipStreamSynchronize(stream) != hipSuccess) { return ncclUnhandledCudaError; }
  world &w = *comm->w;
  {
    std::unique_lock<std::mutex> lk(w.m);
    w.gather_src[size_t(comm->rank)] = sendbuff;
    if (!barrier(w, lk)) { return ncclRemoteError; }
  }
  bool ok = true;
  for (int r = 0; r < w.nranks && ok; r++) {
    ok = hipMemcpyAsync(static_cast<char *>(recvbuff) + size_t(r) * bytes, w.gather_src[size_t(r)], bytes, hipMemcpyDefault, stream) == hipSuccess;
  }
  ok = ok && hipStreamSynchronize(stream) == hipSuccess;
  {
    std::unique_lock<std::mutex> lk(w.m); // nobody overwrites its send buffer before everybody has read it
    if (comm->rank == 0) { w.collectives++; }
    if (!barrier(w, lk)) { return ncclRemoteError; }
  }
  return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(true, sendbuff, count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclGroupStart() {
  t_group_depth++;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (t_group_depth == 0) { return ncclInvalidUsage; }
  if (--t_group_depth) { return ncclSuccess; }
  return run_ops(t_group_ops);
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (int(r)) {
  case 0: return "no error (loop-back stub)";
  case 1: return "unhandled HIP error (loop-back stub)";
  case 4: return "invalid argument: a matched send / receive pair disagrees on count or type (loop-back stub)";
  case 5: return "invalid usage (loop-back stub)";
  case 6: return "a peer never arrived (loop-back stub)";
  default: return "error (loop-back stub)";
  }
}

// what a test can ask the stub: transfers completed and collectives run since the process started, over all communicators
void sjgpu_loopback_stats(uint64_t *sends_completed, uint64_t *collectives) {
  uint64_t s = 0, c = 0;
  std::lock_guard<std::mutex> lk(g_registry_m);
  for (auto &kv : g_registry) {
    std::lock_guard<std::mutex> lk2(kv.second->m);
    s += kv.second->sends_completed;
    c += kv.second->collectives;
  }
  if (sends_completed) { *sends_completed = s; }
  if (collectives) { *collectives = c; }
}

} // extern "C"
