// simdjson_amd/csrc/sjgpu_comm.hip -- the ONE exchange of the multi-GPU path, below the C-ABI: the variable-length gather of the
// per-GPU structural lists to a consumer rank over RCCL / xGMI (BASELINE.json north_star: "RCCL over xGMI only for the
// final index concatenation", host code stays C++).
//
// One process per GPU.  A shard's offsets are relative to the shard (u32); the global position of a structural is the shard's byte
// base + its offset -- the reference's own convention for batches, batch_start + structural_indexes[i]
// (/root/reference/include/simdjson/dom/document_stream-inl.h:250).  sjgpu_comm_gather_indices:
//   1. ncclAllGather of (n, base, room) -- 24 bytes per rank -- so that every rank knows every count, the root the bases, and all of
//      them whether the root's staging array has room (if not, the root allocates and a one-word all-gather spreads its verdict
//      BEFORE anybody posts a send: a root that cannot allocate never leaves a sender waiting);
//   2. grouped ncclSend / ncclRecv of EXACT counts: every rank but the root sends its n u32 offsets once, the root receives each
//      into its slot of a staging array (no padding to the longest shard, no copy on ranks that are not the consumer);
//   3. the root widens them to 64-bit global positions, base of the sending rank added (k_widen_all), into the caller's array.
// xGMI is point to point, so (2) uses one link per sender, all at once.  Everything is enqueued on the caller's stream except
// the 16 * world bytes of counts, which the host needs to size the receives (one wait).
// The Python twin (simdjson_amd/sharded.py: gather_to_root over torch.distributed) stays for the gloo tests of the CPU tier.
#include "sjgpu.h"
#include "sjgpu_internal.h"

#include <rccl/rccl.h> // types and prototypes only: the library itself is opened on first use (below)

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

// RCCL is needed by sjgpu_comm_* and by nothing else in libsjgpu: it is not linked but opened when the first communicator call
// arrives, so that single-GPU users (stage 1 / 2, the plug-in, the in-tree tests) build, load and run on a box without it.
// Search order: SJGPU_RCCL_LIB (a path), then the loader's own (librccl.so.1, librccl.so), then $ROCM_PATH/lib and /opt/rocm/lib.
namespace {
struct rccl_api {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  char why[200] = {0};
};
rccl_api g_rccl;
std::once_flag g_rccl_once;
const rccl_api *rccl() {
  std::call_once(g_rccl_once, [] {
    std::vector<std::string> names;
    if (const char *p = std::getenv("SJGPU_RCCL_LIB")) { names.emplace_back(p); }
    names.emplace_back("librccl.so.1");
    names.emplace_back("librccl.so");
    if (const char *r = std::getenv("ROCM_PATH")) { names.emplace_back(std::string(r) + "/lib/librccl.so"); }
    names.emplace_back("/opt/rocm/lib/librccl.so");
    for (const std::string &n : names) {
      g_rccl.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) { break; }
    }
    if (!g_rccl.handle) { std::snprintf(g_rccl.why, sizeof g_rccl.why, "librccl not found (%s)", dlerror()); return; }
    bool ok = true;
#define SJ_SYM(field, name)                                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));  \
  ok = ok && g_rccl.field != nullptr
    SJ_SYM(GetUniqueId, "ncclGetUniqueId");
    SJ_SYM(CommInitRank, "ncclCommInitRank");
    SJ_SYM(CommDestroy, "ncclCommDestroy");
    SJ_SYM(CommCount, "ncclCommCount");
    SJ_SYM(AllGather, "ncclAllGather");
    SJ_SYM(GroupStart, "ncclGroupStart");
    SJ_SYM(GroupEnd, "ncclGroupEnd");
    SJ_SYM(Send, "ncclSend");
    SJ_SYM(Recv, "ncclRecv");
    SJ_SYM(GetErrorString, "ncclGetErrorString");
#undef SJ_SYM
    if (!ok) { std::snprintf(g_rccl.why, sizeof g_rccl.why, "librccl lacks a symbol the gather needs"); dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
  });
  return g_rccl.handle ? &g_rccl : nullptr;
}
} // namespace

struct sjgpu_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  uint64_t *d_meta = nullptr; // [world][3]: n, base, room in the staging array (the root's counts); behind it this rank's triple, then the widening table
  uint64_t *h_meta = nullptr; // pinned copy
  uint32_t *d_stage = nullptr; // root: the received u32 offsets, rank after rank
  size_t stage_words = 0;
  char err[256] = {0};
};

namespace sjgpu {
namespace {

// out[k] = base of the rank whose slot holds k + the u32 offset received for k; table = [world + 1 slot starts][world bases]
__global__ __launch_bounds__(256) void k_widen_all(const uint32_t *__restrict__ stage, uint64_t *__restrict__ out, uint64_t total, int world,
                                                   const uint64_t *__restrict__ table) {
  const uint64_t k = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (k >= total) { return; }
  int r = 0;
  while (r + 1 < world && table[r + 1] <= k) { r++; } // world <= a few dozen: a linear walk over words that sit in the scalar cache
  out[k] = table[world + 1 + r] + stage[k];
}

} // namespace
} // namespace sjgpu

using namespace sjgpu;

namespace {
int comm_fail(sjgpu_comm *c, const char *what, const char *detail) {
  if (c) { std::snprintf(c->err, sizeof c->err, "%s: %s", what, detail); }
  return SJGPU_E_HIP;
}
} // namespace

#define SJ_NCCL(c, call)                                                                 \
  do {                                                                                   \
    ncclResult_t r_ = (call);                                                            \
    if (r_ != ncclSuccess) { return comm_fail((c), #call, R->GetErrorString(r_)); }       \
  } while (0)
#define SJ_HIPC(c, call)                                                                 \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) { return comm_fail((c), #call, hipGetErrorString(e_)); }        \
  } while (0)

extern "C" {

int sjgpu_comm_unique_id(void *id_out, size_t id_bytes) {
  if (!id_out || id_bytes < sizeof(ncclUniqueId)) { return SJGPU_E_BADARG; }
  const rccl_api *R = rccl();
  if (!R) { return SJGPU_E_HIP; }
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) { return SJGPU_E_HIP; }
  std::memset(id_out, 0, id_bytes);
  std::memcpy(id_out, &id, sizeof id);
  return 0;
}

int sjgpu_comm_create(int rank, int world, const void *id, size_t id_bytes, int device, sjgpu_comm **out) {
  if (!out || !id || id_bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) { return SJGPU_E_BADARG; }
  *out = nullptr;
  const rccl_api *R = rccl();
  if (!R) { return SJGPU_E_HIP; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { return SJGPU_E_NO_DEVICE; }
  sjgpu_comm *c = new (std::nothrow) sjgpu_comm();
  if (!c) { return SJGPU_E_NOMEM; }
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  int rc = 0;
  const size_t meta_words = size_t(world) * 5 + 8; // [world][3] + own triple + table (2 * world + 1)
  if (hipSetDevice(device) != hipSuccess) { rc = SJGPU_E_HIP; }
  if (!rc && R->CommInitRank(&c->comm, world, uid, rank) != ncclSuccess) { rc = SJGPU_E_HIP; }
  if (!rc && hipMalloc(reinterpret_cast<void **>(&c->d_meta), meta_words * sizeof(uint64_t)) != hipSuccess) { rc = SJGPU_E_NOMEM; }
  if (!rc && hipHostMalloc(reinterpret_cast<void **>(&c->h_meta), meta_words * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { rc = SJGPU_E_NOMEM; }
  if (rc) {
    sjgpu_comm_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

void sjgpu_comm_destroy(sjgpu_comm *c) {
  if (!c) { return; }
  (void)hipSetDevice(c->device);
  if (c->comm && rccl()) { (void)rccl()->CommDestroy(c->comm); }
  if (c->d_meta) { (void)hipFree(c->d_meta); }
  if (c->d_stage) { (void)hipFree(c->d_stage); }
  if (c->h_meta) { (void)hipHostFree(c->h_meta); }
  delete c;
}

const char *sjgpu_comm_last_error(const sjgpu_comm *c) { return c ? c->err : (rccl() ? "" : g_rccl.why); }

int sjgpu_comm_ranks(const sjgpu_comm *c) { // what RCCL itself says the communicator spans
  const rccl_api *R = rccl();
  int n = 0;
  if (!c || !R || R->CommCount(c->comm, &n) != ncclSuccess) { return -1; }
  return n;
}

int sjgpu_comm_gather_indices(sjgpu_comm *c, const void *idx_dev, uint32_t n, uint64_t base, int root, void *out_dev, size_t out_cap_words,
                              uint64_t *total_out, uint64_t *counts_out, void *stream) {
  if (!c || root < 0 || root >= c->world || (n && !idx_dev)) { return SJGPU_E_BADARG; }
  const rccl_api *R = rccl();
  if (!R) { return SJGPU_E_HIP; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  SJ_HIPC(c, hipSetDevice(c->device));
  const size_t W = size_t(c->world);
  const bool is_root = c->rank == root;
  const bool fail_alloc = std::getenv("SJGPU_DEBUG_COMM_FAIL_STAGING") != nullptr; // test hook: the root's staging allocation fails
  // 1. everybody learns (n, base) of everybody -- and how much room the root's staging array has, so that EVERY rank knows from the
  //    same numbers whether the root has to allocate before it can receive
  uint64_t mine[3] = {uint64_t(n), base, is_root ? uint64_t(c->stage_words) : 0u};
  uint64_t *d_mine = c->d_meta + W * 3; // 24 spare bytes behind the table
  SJ_HIPC(c, hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, s));
  SJ_NCCL(c, R->AllGather(d_mine, c->d_meta, 3, ncclUint64, c->comm, s));
  SJ_HIPC(c, hipMemcpyAsync(c->h_meta, c->d_meta, W * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  SJ_HIPC(c, hipStreamSynchronize(s));
  uint64_t total = 0;
  for (int r = 0; r < c->world; r++) {
    if (counts_out) { counts_out[r] = c->h_meta[3 * r]; }
    total += c->h_meta[3 * r];
  }
  if (total_out) { *total_out = total; }
  // 1b. The root cannot refuse AFTER the senders have posted.  When its staging array is too small (every rank sees that: total against
  //     the room the root announced) the root allocates FIRST and all ranks meet in a second, one-word all-gather of its verdict; only
  //     then does anybody post.  A root that could not allocate makes every rank return -- SJGPU_E_NOMEM there, SJGPU_E_PEER elsewhere --
  //     with nothing in flight.  Steady state (the array has grown to the job's size): no second round.
  if (total > c->h_meta[3 * root + 2]) {
    uint64_t verdict = 1;
    if (is_root) {
      if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; c->stage_words = 0; }
      const size_t want = size_t(total) + size_t(total) / 4 + 1024;
      if (fail_alloc || hipMalloc(reinterpret_cast<void **>(&c->d_stage), want * sizeof(uint32_t)) != hipSuccess) { c->d_stage = nullptr; verdict = 0; }
      else { c->stage_words = want; }
    }
    SJ_HIPC(c, hipMemcpyAsync(d_mine, &verdict, sizeof verdict, hipMemcpyHostToDevice, s));
    SJ_NCCL(c, R->AllGather(d_mine, c->d_meta, 1, ncclUint64, c->comm, s));
    SJ_HIPC(c, hipMemcpyAsync(c->h_meta + W * 3, c->d_meta, W * sizeof(uint64_t), hipMemcpyDeviceToHost, s)); // (the counts above stay where they are)
    SJ_HIPC(c, hipStreamSynchronize(s));
    if (c->h_meta[W * 3 + size_t(root)] == 0) {
      comm_fail(c, "hipMalloc", is_root ? "staging array of the gather" : "the root could not allocate its staging array");
      return is_root ? SJGPU_E_NOMEM : SJGPU_E_PEER;
    }
  }
  // 2. exact-count sends to the root; the root's own offsets are copied.  An error inside the group still closes the group: what has
  //    been posted completes, the first error is reported afterwards.
  ncclResult_t first = ncclSuccess;
  const char *where = "";
  SJ_NCCL(c, R->GroupStart());
  if (is_root) {
    uint64_t at = 0;
    for (int r = 0; r < c->world; r++) {
      const uint64_t cnt = c->h_meta[3 * r];
      if (r != root && cnt) {
        const ncclResult_t e = R->Recv(c->d_stage + at, cnt, ncclUint32, r, c->comm, s);
        if (e != ncclSuccess && first == ncclSuccess) { first = e; where = "ncclRecv"; }
      }
      at += cnt;
    }
  } else if (n) {
    first = R->Send(idx_dev, n, ncclUint32, root, c->comm, s);
    where = "ncclSend";
  }
  {
    const ncclResult_t e = R->GroupEnd();
    if (e != ncclSuccess && first == ncclSuccess) { first = e; where = "ncclGroupEnd"; }
  }
  if (first != ncclSuccess) { return comm_fail(c, where, R->GetErrorString(first)); }
  if (!is_root) { return 0; }
  // 3. widen: table = slot starts (world + 1) and bases (world), behind the gathered triples in d_meta
  uint64_t *table_h = c->h_meta + W * 3 + 3, *table_d = c->d_meta + W * 3 + 3;
  uint64_t at = 0;
  for (int r = 0; r < c->world; r++) {
    table_h[r] = at;
    table_h[c->world + 1 + r] = c->h_meta[3 * r + 1];
    if (r == root && c->h_meta[3 * r]) { SJ_HIPC(c, hipMemcpyAsync(c->d_stage + at, idx_dev, c->h_meta[3 * r] * sizeof(uint32_t), hipMemcpyDeviceToDevice, s)); }
    at += c->h_meta[3 * r];
  }
  table_h[c->world] = at;
  if (!out_dev || total > out_cap_words) { return SJGPU_E_OVERFLOW; }
  SJ_HIPC(c, hipMemcpyAsync(table_d, table_h, (W * 2 + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  if (total) {
    hipLaunchKernelGGL(k_widen_all, dim3(unsigned((total + 255) / 256)), dim3(256), 0, s, c->d_stage, static_cast<uint64_t *>(out_dev), total, c->world, table_d);
    SJ_HIPC(c, hipGetLastError());
  }
  return 0;
}

} // extern "C"
