// simdjson_amd/csrc/sjgpu_comm.hip -- the ONE exchange of the multi-GPU path, below the C-ABI: the variable-length gather of the
// per-GPU structural lists to a consumer rank over RCCL / xGMI (BASELINE.json north_star: "RCCL over xGMI only for the
// final index concatenation", host code stays C++).
//
// One process per GPU.  A shard's offsets are relative to the shard (u32); the global position of a structural is the shard's byte
// base + its offset -- the reference's own convention for batches, batch_start + structural_indexes[i]
// (/root/reference/include/simdjson/dom/document_stream-inl.h:250).  sjgpu_comm_gather_indices:
//   1. ncclAllGather of (n, base) -- 16 bytes per rank -- so that every rank knows every count (and the root the bases);
//   2. grouped ncclSend / ncclRecv of EXACT counts: every rank but the root sends its n u32 offsets once, the root receives each
//      into its slot of a staging array (no padding to the longest shard, no copy on ranks that are not the consumer);
//   3. the root widens them to 64-bit global positions, base of the sending rank added (k_widen_all), into the caller's array.
// xGMI is point to point, so (2) uses one link per sender, all at once.  Everything is enqueued on the caller's stream except
// the 16 * world bytes of counts, which the host needs to size the receives (one wait).
// The Python twin (simdjson_amd/sharded.py: gather_to_root over torch.distributed) stays for the gloo tests of the CPU tier.
#include "sjgpu.h"
#include "sjgpu_internal.h"

#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <vector>

struct sjgpu_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  uint64_t *d_meta = nullptr; // [world][2]: n, base; behind it this rank's pair, then the widening table
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
    if (r_ != ncclSuccess) { return comm_fail((c), #call, ncclGetErrorString(r_)); }      \
  } while (0)
#define SJ_HIPC(c, call)                                                                 \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) { return comm_fail((c), #call, hipGetErrorString(e_)); }        \
  } while (0)

extern "C" {

int sjgpu_comm_unique_id(void *id_out, size_t id_bytes) {
  if (!id_out || id_bytes < sizeof(ncclUniqueId)) { return SJGPU_E_BADARG; }
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) { return SJGPU_E_HIP; }
  std::memset(id_out, 0, id_bytes);
  std::memcpy(id_out, &id, sizeof id);
  return 0;
}

int sjgpu_comm_create(int rank, int world, const void *id, size_t id_bytes, int device, sjgpu_comm **out) {
  if (!out || !id || id_bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) { return SJGPU_E_BADARG; }
  *out = nullptr;
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
  if (hipSetDevice(device) != hipSuccess) { rc = SJGPU_E_HIP; }
  if (!rc && ncclCommInitRank(&c->comm, world, uid, rank) != ncclSuccess) { rc = SJGPU_E_HIP; }
  if (!rc && hipMalloc(reinterpret_cast<void **>(&c->d_meta), (size_t(world) * 4 + 8) * sizeof(uint64_t)) != hipSuccess) { rc = SJGPU_E_NOMEM; }
  if (!rc && hipHostMalloc(reinterpret_cast<void **>(&c->h_meta), (size_t(world) * 4 + 8) * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { rc = SJGPU_E_NOMEM; }
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
  if (c->comm) { (void)ncclCommDestroy(c->comm); }
  if (c->d_meta) { (void)hipFree(c->d_meta); }
  if (c->d_stage) { (void)hipFree(c->d_stage); }
  if (c->h_meta) { (void)hipHostFree(c->h_meta); }
  delete c;
}

const char *sjgpu_comm_last_error(const sjgpu_comm *c) { return c ? c->err : ""; }

int sjgpu_comm_gather_indices(sjgpu_comm *c, const void *idx_dev, uint32_t n, uint64_t base, int root, void *out_dev, size_t out_cap_words,
                              uint64_t *total_out, uint64_t *counts_out, void *stream) {
  if (!c || root < 0 || root >= c->world || (n && !idx_dev)) { return SJGPU_E_BADARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  SJ_HIPC(c, hipSetDevice(c->device));
  // 1. everybody learns (n, base) of everybody
  uint64_t mine[2] = {uint64_t(n), base};
  uint64_t *d_mine = c->d_meta + size_t(c->world) * 2; // 16 spare bytes behind the table
  SJ_HIPC(c, hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, s));
  SJ_NCCL(c, ncclAllGather(d_mine, c->d_meta, 2, ncclUint64, c->comm, s));
  SJ_HIPC(c, hipMemcpyAsync(c->h_meta, c->d_meta, size_t(c->world) * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  SJ_HIPC(c, hipStreamSynchronize(s));
  uint64_t total = 0;
  for (int r = 0; r < c->world; r++) {
    if (counts_out) { counts_out[r] = c->h_meta[2 * r]; }
    total += c->h_meta[2 * r];
  }
  if (total_out) { *total_out = total; }
  // The root cannot refuse AFTER the senders have posted: every rank takes the same decision from the same numbers.  A root whose
  // array is too small therefore still receives (into its staging array) and reports SJGPU_E_OVERFLOW afterwards.
  const bool is_root = c->rank == root;
  if (is_root && total > c->stage_words) {
    if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; c->stage_words = 0; }
    const size_t want = size_t(total) + size_t(total) / 4 + 1024;
    if (hipMalloc(reinterpret_cast<void **>(&c->d_stage), want * sizeof(uint32_t)) != hipSuccess) { return comm_fail(c, "hipMalloc", "staging array of the gather"), SJGPU_E_NOMEM; }
    c->stage_words = want;
  }
  // 2. exact-count sends to the root; the root's own offsets are copied
  SJ_NCCL(c, ncclGroupStart());
  if (is_root) {
    uint64_t at = 0;
    for (int r = 0; r < c->world; r++) {
      const uint64_t cnt = c->h_meta[2 * r];
      if (r != root && cnt) { SJ_NCCL(c, ncclRecv(c->d_stage + at, cnt, ncclUint32, r, c->comm, s)); }
      at += cnt;
    }
  } else if (n) {
    SJ_NCCL(c, ncclSend(idx_dev, n, ncclUint32, root, c->comm, s));
  }
  SJ_NCCL(c, ncclGroupEnd());
  if (!is_root) { return 0; }
  // 3. widen: table = slot starts (world + 1) and bases (world), behind the gathered pairs in d_meta
  uint64_t *table_h = c->h_meta + size_t(c->world) * 2 + 2, *table_d = c->d_meta + size_t(c->world) * 2 + 2;
  uint64_t at = 0;
  for (int r = 0; r < c->world; r++) {
    table_h[r] = at;
    table_h[c->world + 1 + r] = c->h_meta[2 * r + 1];
    if (r == root && c->h_meta[2 * r]) { SJ_HIPC(c, hipMemcpyAsync(c->d_stage + at, idx_dev, c->h_meta[2 * r] * sizeof(uint32_t), hipMemcpyDeviceToDevice, s)); }
    at += c->h_meta[2 * r];
  }
  table_h[c->world] = at;
  if (!out_dev || total > out_cap_words) { return SJGPU_E_OVERFLOW; }
  SJ_HIPC(c, hipMemcpyAsync(table_d, table_h, (size_t(c->world) * 2 + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  if (total) {
    hipLaunchKernelGGL(k_widen_all, dim3(unsigned((total + 255) / 256)), dim3(256), 0, s, c->d_stage, static_cast<uint64_t *>(out_dev), total, c->world, table_d);
    SJ_HIPC(c, hipGetLastError());
  }
  return 0;
}

} // extern "C"
