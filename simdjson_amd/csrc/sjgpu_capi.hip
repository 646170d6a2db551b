// simdjson_amd/csrc/sjgpu_capi.hip -- the C-ABI of include/sjgpu.h, first part: contexts, the pool, the device-resident entry points (the others:
// sjgpu_capi_host.hip, sjgpu_capi_stage2.hip; what they share: sjgpu_ctx.h).  No CPU compute path exists here: if HIP is unusable every entry point returns a
// negative code.
#include "sjgpu_ctx.h"


extern "C" {

int sjgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { return 0; }
  return n;
}

void *sjgpu_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { return nullptr; }
  return p;
}
void sjgpu_host_free(void *p) {
  if (p) { (void)hipHostFree(p); }
}
int sjgpu_host_register(void *p, size_t bytes) {
  if (!p || bytes == 0) { return SJGPU_E_BADARG; }
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  return e == hipSuccess ? 0 : fail(nullptr, e, "hipHostRegister");
}
int sjgpu_host_unregister(void *p) {
  if (!p) { return SJGPU_E_BADARG; }
  const hipError_t e = hipHostUnregister(p);
  return e == hipSuccess ? 0 : fail(nullptr, e, "hipHostUnregister");
}

} // extern "C"

// Contexts are recycled: creating one costs a stream, two page-locked blocks and the escape table (~0.5 ms, 3 ms with
// the workspace the first call allocates), and the reference creates a dom_parser_implementation per parser object --
// its document_stream tests make 106 000 of them.  sjgpu_ctx_destroy parks the context (with whatever small workspace
// it has grown) and sjgpu_ctx_create takes a parked one of the same device; large workspaces are released on parking.
namespace {
constexpr size_t POOL_MAX = 64;                          // parked contexts per process
constexpr size_t POOL_KEEP_BYTES = size_t(32) << 20;     // workspace / staging for documents beyond this is freed on parking
// The pool and its lock are heap objects that are never destroyed: a context may be parked from a static destructor of the
// caller, or after this library's own statics would have been torn down.
struct ctx_pool {
  std::mutex m;
  std::vector<sjgpu_ctx *> parked;
};
ctx_pool &pool() {
  static ctx_pool *p = new ctx_pool();
  return *p;
}

void apply_environment(sjgpu_ctx *ctx) {
  ctx->pipeline = 2;
  ctx->stream_from = size_t(64) << 20;
  ctx->stream_chunk = size_t(16) << 20;
  ctx->copy_threads = 1;
  if (const char *pl = std::getenv("SJGPU_PIPELINE")) {
    ctx->pipeline = std::strcmp(pl, "split") == 0 ? 0 : (std::strcmp(pl, "fused") == 0 ? 1 : 2);
  }
  if (const char *v = std::getenv("SJGPU_STREAM_FROM_MB")) { ctx->stream_from = size_t(std::strtoull(v, nullptr, 10)) << 20; }
  if (const char *v = std::getenv("SJGPU_COPY_THREADS")) {
    const size_t t = size_t(std::strtoull(v, nullptr, 10));
    if (t >= 1 && t <= 8) { ctx->copy_threads = t; }
  }
  if (const char *v = std::getenv("SJGPU_STREAM_CHUNK_MB")) {
    const size_t mb = size_t(std::strtoull(v, nullptr, 10));
    if (mb >= 1 && mb <= 1024) { ctx->stream_chunk = mb << 20; }
  }
  ctx->small_docs = true;
  if (const char *v = std::getenv("SJGPU_SMALL_DOCS")) { ctx->small_docs = v[0] != '0'; }
  ctx->device_finish = 1; // streaming-mode finish on the device for documents beyond the small-document path
  if (const char *v = std::getenv("SJGPU_FINISH")) { ctx->device_finish = std::strcmp(v, "host") == 0 ? 0 : (std::strcmp(v, "device") == 0 ? 2 : 1); }
}

void really_destroy(sjgpu_ctx *ctx) {
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
  for (std::vector<copy_worker *> *ws : {&ctx->up, &ctx->down}) {
    for (copy_worker *w : *ws) {
      w->shutdown();
      delete w;
    }
    ws->clear();
  }
  for (hipEvent_t ev : ctx->ev_in) { (void)hipEventDestroy(ev); }
  ctx->ev_in.clear();
  release_scan_workspace(ctx);
  release_staging(ctx);
  drop_events(ctx);
  if (ctx->h_result) { (void)hipHostFree(ctx->h_result); }
  if (ctx->h_small) { (void)hipHostFree(ctx->h_small); }
  for (sjgpu_ctx::span_slot &sl : ctx->la) {
    if (sl.h_idx) { (void)hipHostFree(sl.h_idx); }
    if (sl.h_res) { (void)hipHostFree(sl.h_res); }
    if (sl.d_in) { (void)hipFree(sl.d_in); }
    if (sl.d_idx) { (void)hipFree(sl.d_idx); }
    if (sl.ev) { (void)hipEventDestroy(sl.ev); }
  }
  dev_free(ctx->esc_tab);
  dev_free(ctx->d_tmp);
  dev_free(ctx->d_stage2);
  dev_free(ctx->d_doc);
  if (ctx->stream) { (void)hipStreamDestroy(ctx->stream); }
  delete ctx;
}
} // namespace

extern "C" int sjgpu_ctx_create(int device, size_t capacity, sjgpu_ctx **out) {
  if (!out) { return SJGPU_E_BADARG; }
  *out = nullptr;
  if (capacity > 0xFFFFFFFFull) { return SJGPU_E_BADARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { return SJGPU_E_NO_DEVICE; }
  sjgpu_ctx *ctx = nullptr;
  {
    ctx_pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.m);
    for (size_t i = pl.parked.size(); i-- > 0;) {
      if (pl.parked[i]->device == device) {
        ctx = pl.parked[i];
        pl.parked.erase(pl.parked.begin() + long(i));
        break;
      }
    }
  }
  if (ctx) { // a parked context: same stream, same page-locked blocks, whatever workspace it kept
    apply_environment(ctx);
    ctx->capacity = capacity;
    ctx->density_permille = 1000;
    ctx->pending_scan_bytes = 0;
    ctx->last_pipeline = 0;
    ctx->last_kernel = "";
    ctx->err[0] = 0;
    *out = ctx;
    return 0;
  }
  ctx = new (std::nothrow) sjgpu_ctx();
  if (!ctx) { return SJGPU_E_NOMEM; }
  ctx->device = device;
  apply_environment(ctx);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) {
      ctx->max_workgroups = uint32_t(cus) * 8u; // more than can be resident; surplus workgroups just start later
    }
    if (const char *v = std::getenv("SJGPU_MAX_WORKGROUPS")) { // A/B switch: the grid of the single-pass kernels
      const unsigned long g = std::strtoul(v, nullptr, 10);
      if (g >= 64 && g <= 65536) { ctx->max_workgroups = uint32_t(g); }
    }
  }
  if (e == hipSuccess) { e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking); }
  if (e == hipSuccess) { e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_result), 256, hipHostMallocDefault); } // [0] the scan's result, [64] / [128] stage 2's two results
  if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void **>(&ctx->esc_tab), SEGMENT_BYTES_TABLE); }
  if (e == hipSuccess) { e = hipMemset(ctx->esc_tab, 0, SEGMENT_BYTES_TABLE); } // entry 0 and the pass flag start at zero
  if (e != hipSuccess) {
    int rc = fail(nullptr, e, "ctx_create");
    really_destroy(ctx);
    return rc;
  }
  ctx->capacity = capacity;
  *out = ctx;
  return 0;
}

extern "C" void sjgpu_ctx_destroy(sjgpu_ctx *ctx) {
  if (!ctx) { return; }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
  drop_events(ctx);
  ctx->profile = false;
  if (ctx->ws_capacity > POOL_KEEP_BYTES) { release_scan_workspace(ctx); }
  if (ctx->d_in_bytes > POOL_KEEP_BYTES || ctx->d_idx_words * sizeof(uint32_t) > 4 * POOL_KEEP_BYTES || ctx->d_out_bytes > POOL_KEEP_BYTES) {
    release_staging(ctx);
  }
  if (ctx->d_tmp_bytes > 4 * POOL_KEEP_BYTES) { dev_free(ctx->d_tmp); ctx->d_tmp_bytes = 0; }
  if (ctx->d_stage2_bytes > 4 * POOL_KEEP_BYTES) { dev_free(ctx->d_stage2); ctx->d_stage2_bytes = 0; }
  if (ctx->d_doc_bytes > 4 * POOL_KEEP_BYTES) { dev_free(ctx->d_doc); ctx->d_doc_bytes = 0; }
  for (sjgpu_ctx::span_slot &sl : ctx->la) { // a parked context keeps no span: streams end, their memory goes away
    if (sl.state == 1 || sl.state == 2) { (void)hipStreamSynchronize(ctx->stream); }
    sl.base = nullptr;
    sl.state = 0;
    if (sl.h_idx) { (void)hipHostFree(sl.h_idx); sl.h_idx = nullptr; sl.h_words = 0; }
    dev_free(sl.d_in);
    sl.d_in_bytes = 0;
    dev_free(sl.d_idx);
    sl.d_idx_bytes = 0;
  }
  if (ctx->h_small_bytes > (size_t(1) << 20)) { // page-locked memory is scarce: a parked context keeps at most 1 MiB of it
    (void)hipHostFree(ctx->h_small);
    ctx->h_small = nullptr;
    ctx->h_small_bytes = 0;
  }
  {
    ctx_pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.m);
    if (pl.parked.size() < POOL_MAX) {
      pl.parked.push_back(ctx);
      return;
    }
  }
  really_destroy(ctx);
}

// Gives back what the parked contexts hold (streams, page-locked blocks, device workspace, copy threads): for callers that have
// destroyed every parser and want the memory, or shut the library down in an orderly way.  Returns the number of contexts freed.
extern "C" int sjgpu_pool_trim(void) {
  std::vector<sjgpu_ctx *> victims;
  {
    ctx_pool &pl = pool();
    std::lock_guard<std::mutex> lk(pl.m);
    victims.swap(pl.parked);
  }
  for (sjgpu_ctx *c : victims) { really_destroy(c); }
  return int(victims.size());
}

extern "C" int sjgpu_set_capacity(sjgpu_ctx *ctx, size_t capacity) {
  if (!ctx || capacity > 0xFFFFFFFFull) { return SJGPU_E_BADARG; }
  ctx->capacity = capacity; // device memory follows the documents actually scanned (ensure_scan_workspace)
  return 0;
}

extern "C" {

size_t sjgpu_capacity(const sjgpu_ctx *ctx) { return ctx ? ctx->capacity : 0; }
const char *sjgpu_last_error(const sjgpu_ctx *ctx) { return ctx ? ctx->err : ""; }

// ---- device-resident entry points ------------------------------------------------------------------
int sjgpu_stage1_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words, void *stream) {
  if (!ctx || !buf_dev || !idx_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(idx_dev) & 15u)) {
    return SJGPU_E_BADARG;
  }
  if (len > ctx->capacity) { return E_CAPACITY; }
  if (len == 0) { return E_EMPTY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  enqueue_stage1(ctx, use_fused(ctx, len, 0), static_cast<const uint8_t *>(buf_dev), len, static_cast<uint32_t *>(idx_dev), idx_words,
                 pick(ctx, stream), next_events(ctx));
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_stage1_tokens_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words, void *tok_dev, size_t tok_bytes, void *stream) {
  if (!ctx || !buf_dev || !idx_dev || !tok_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(idx_dev) & 15u) ||
      (reinterpret_cast<uintptr_t>(tok_dev) & 15u)) { // (the stream leaves as aligned dword stores)
    return SJGPU_E_BADARG;
  }
  if (tok_bytes < idx_words) { return SJGPU_E_BADARG; } // a byte for every word the list may hold
  if (len > ctx->capacity) { return E_CAPACITY; }
  if (len == 0) { return E_EMPTY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  // Which road: the split pipeline stages the structural bytes while it holds them (costs its scan kernel a compaction per chunk), the single-pass kernels
  // gather them out of the document when they emit (costs a second trip of the tile's lines).  Measured (profiles/r06_tokens_fused.txt): see tokens_fused().
  enqueue_stage1(ctx, tokens_fused(ctx, len), static_cast<const uint8_t *>(buf_dev), len, static_cast<uint32_t *>(idx_dev), idx_words, pick(ctx, stream), next_events(ctx),
                 scan_origin{0, 0, 0}, static_cast<uint8_t *>(tok_dev));
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_minify_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *dst_dev, void *stream) {
  if (!ctx || !buf_dev || !dst_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(dst_dev) & 15u)) {
    return SJGPU_E_BADARG;
  }
  if (len > ctx->capacity) { return E_CAPACITY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (const int wrc = ensure_result_only(ctx)) { return wrc; }
  if (len == 0) { // SUCCESS with zero bytes (json_minifier.h:68-97 with an empty reader)
    SJ_TRY(ctx, hipMemsetAsync(ctx->d_result, 0, sizeof(scan_result_dev), pick(ctx, stream)));
    return 0;
  }
  enqueue_minify(ctx, use_fused(ctx, len), static_cast<const uint8_t *>(buf_dev), len, static_cast<uint8_t *>(dst_dev), pick(ctx, stream),
                 next_events(ctx));
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_validate_utf8_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *stream) {
  if (!ctx || (len && (!buf_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u)))) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (const int wrc = ensure_result_only(ctx)) { return wrc; }
  if (len == 0) {
    SJ_TRY(ctx, hipMemsetAsync(ctx->d_result, 0, sizeof(scan_result_dev), pick(ctx, stream)));
    return 0;
  }
  ctx->pending_scan_bytes = 0;
  ctx->last_kernel = "k_validate_utf8";
  launch_validate_utf8(static_cast<const uint8_t *>(buf_dev), len, ctx->d_result, pick(ctx, stream), next_events(ctx));
  SJ_TRY(ctx, hipGetLastError());
  return 0;
}

// ---- shards of ONE large document across GPUs ("general inputs") -------------------------
// Cuts come from sjgpu_clean_cut (host), so escapes, the previous-scalar bit and UTF-8 state are zero at
// every cut; the in-string bit is the only carry and travels as an argument.
int sjgpu_string_parity_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *stream) {
  if (!ctx || (len && (!buf_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u)))) { return SJGPU_E_BADARG; }
  if (len > ctx->capacity) { return E_CAPACITY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (const int wrc = ensure_result_only(ctx)) { return wrc; }
  ctx->pending_scan_bytes = 0;
  launch_string_parity(static_cast<const uint8_t *>(buf_dev), len, ctx->d_result, ctx->esc_tab, pick(ctx, stream));
  SJ_TRY(ctx, hipGetLastError());
  return 0;
}

int sjgpu_stage1_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *idx_dev, size_t idx_words,
                              void *stream) {
  if (!ctx || !buf_dev || !idx_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(idx_dev) & 15u) ||
      len == 0) {
    return SJGPU_E_BADARG;
  }
  if (len > ctx->capacity) { return E_CAPACITY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  enqueue_stage1(ctx, use_fused(ctx, len, 0), static_cast<const uint8_t *>(buf_dev), len, static_cast<uint32_t *>(idx_dev), idx_words,
                 pick(ctx, stream), next_events(ctx), scan_origin{0, 0, CARRY_SHARD | (in_string ? CARRY_IN_STRING : 0u)});
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_minify_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *dst_dev, void *stream) {
  if (!ctx || !buf_dev || !dst_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(dst_dev) & 15u) ||
      len == 0) {
    return SJGPU_E_BADARG;
  }
  if (len > ctx->capacity) { return E_CAPACITY; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  enqueue_minify(ctx, use_fused(ctx, len), static_cast<const uint8_t *>(buf_dev), len, static_cast<uint8_t *>(dst_dev), pick(ctx, stream),
                 next_events(ctx), scan_origin{0, 0, CARRY_SHARD | (in_string ? CARRY_IN_STRING : 0u)});
  SJ_ENQUEUED(ctx);
  return 0;
}

// ---- ranges of ONE resident buffer, scanned one after the other (streaming upload) ---------
static int check_range(const sjgpu_ctx *ctx, const void *buf_dev, size_t begin, size_t end) {
  if (!ctx || !buf_dev || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || begin >= end || (begin % RANGE_ALIGN) != 0 ||
      end > 0xFFFFFFFFull) {
    return SJGPU_E_BADARG;
  }
  return (end - begin > ctx->capacity) ? E_CAPACITY : 0;
}

int sjgpu_stage1_range_device(sjgpu_ctx *ctx, const void *buf_dev, size_t begin, size_t end, int more, int in_string,
                              uint32_t n_before, void *idx_dev, size_t idx_words, void *stream) {
  const int bad = check_range(ctx, buf_dev, begin, end);
  if (bad) { return bad; }
  if (!idx_dev || (reinterpret_cast<uintptr_t>(idx_dev) & 15u)) { return SJGPU_E_BADARG; }
  // in_string is a bit field since round 4 (bit 0: inside a string; SJGPU_F_RANGE_CARRY: the escape carry): a caller of the earlier "any non-zero
  // value" contract that passes 2 or -1 must hear about it, not get offsets for a range that begins outside a string
  if (in_string & ~(1 | int(SJGPU_F_RANGE_CARRY))) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  const scan_origin org{uint64_t(begin), n_before, CARRY_SHARD | ((in_string & 1) ? CARRY_IN_STRING : 0u) | ((in_string & int(SJGPU_F_RANGE_CARRY)) ? CARRY_X : 0u) | (more ? CARRY_MORE : 0u)};
  enqueue_stage1(ctx, use_fused(ctx, end - begin, 0), static_cast<const uint8_t *>(buf_dev), end, static_cast<uint32_t *>(idx_dev), idx_words,
                 pick(ctx, stream), next_events(ctx), org);
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_minify_range_device(sjgpu_ctx *ctx, const void *buf_dev, size_t begin, size_t end, int more, int in_string,
                              uint32_t out_before, void *dst_dev, void *stream) {
  const int bad = check_range(ctx, buf_dev, begin, end);
  if (bad) { return bad; }
  if (!dst_dev || (reinterpret_cast<uintptr_t>(dst_dev) & 15u)) { return SJGPU_E_BADARG; }
  // in_string is a bit field since round 4 (bit 0: inside a string; SJGPU_F_RANGE_CARRY: the escape carry): a caller of the earlier "any non-zero
  // value" contract that passes 2 or -1 must hear about it, not get offsets for a range that begins outside a string
  if (in_string & ~(1 | int(SJGPU_F_RANGE_CARRY))) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  const scan_origin org{uint64_t(begin), out_before, CARRY_SHARD | ((in_string & 1) ? CARRY_IN_STRING : 0u) | ((in_string & int(SJGPU_F_RANGE_CARRY)) ? CARRY_X : 0u) | (more ? CARRY_MORE : 0u)};
  enqueue_minify(ctx, use_fused(ctx, end - begin), static_cast<const uint8_t *>(buf_dev), end, static_cast<uint8_t *>(dst_dev),
                 pick(ctx, stream), next_events(ctx), org);
  SJ_ENQUEUED(ctx);
  return 0;
}

int sjgpu_result(sjgpu_ctx *ctx, void *stream, sjgpu_scan_result *out) {
  if (!ctx || !out) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  return fetch_result(ctx, pick(ctx, stream), out);
}

int sjgpu_debug_trace_stage1(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words,
                             uint64_t *trace_host, uint32_t trace_tiles) {
  if (!ctx || !buf_dev || !idx_dev || !trace_host || len == 0 || len > ctx->capacity || (reinterpret_cast<uintptr_t>(buf_dev) & 15u) ||
      (reinterpret_cast<uintptr_t>(idx_dev) & 15u)) {
    return SJGPU_E_BADARG;
  }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (const int wrc = ensure_scan_workspace(ctx, len, false)) { return wrc; }
  uint64_t *d_trace = nullptr;
  const size_t bytes = size_t(trace_tiles) * 8 * sizeof(uint64_t);
  SJ_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&d_trace), bytes));
  (void)hipMemset(d_trace, 0, bytes);
  launch_stage1_fused_traced(static_cast<const uint8_t *>(buf_dev), len, ctx->desc, static_cast<uint32_t *>(idx_dev), idx_words,
                             ctx->d_result, ctx->max_workgroups, nullptr, d_trace, trace_tiles);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) { e = hipMemcpy(trace_host, d_trace, bytes, hipMemcpyDeviceToHost); }
  (void)hipFree(d_trace);
  ctx->ws_dirty = true; // (a traced kernel cleans up like any other; the next call does not rely on it)
  if (e != hipSuccess) { return fail(ctx, e, "debug_trace"); }
  return 0;
}

int sjgpu_debug_trace_pipelined(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words, uint64_t *trace_host,
                                uint32_t max_records, uint32_t *workgroups_out) {
  if (!ctx || !buf_dev || !idx_dev || !trace_host || !workgroups_out || len == 0 || len > ctx->capacity ||
      (reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(idx_dev) & 15u)) {
    return SJGPU_E_BADARG;
  }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (const int wrc = ensure_scan_workspace(ctx, len, false)) { return wrc; }
  uint64_t *d_trace = nullptr;
  const size_t bytes = size_t(max_records) * 8 * sizeof(uint64_t);
  SJ_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&d_trace), bytes));
  (void)hipMemset(d_trace, 0, bytes);
  *workgroups_out = launch_stage1_pipelined_traced(static_cast<const uint8_t *>(buf_dev), len, ctx->desc, static_cast<uint32_t *>(idx_dev), idx_words,
                                                  ctx->d_result, ctx->max_workgroups, nullptr, d_trace, max_records);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) { e = hipMemcpy(trace_host, d_trace, bytes, hipMemcpyDeviceToHost); }
  (void)hipFree(d_trace);
  ctx->ws_dirty = true;
  if (e != hipSuccess) { return fail(ctx, e, "debug_trace_pipelined"); }
  return *workgroups_out ? 0 : SJGPU_E_BADARG;
}

int sjgpu_set_pipeline(sjgpu_ctx *ctx, int pipeline) {
  if (!ctx || pipeline < 0 || pipeline > 2) { return SJGPU_E_BADARG; }
  ctx->pipeline = pipeline;
  return 0;
}

int sjgpu_last_pipeline(const sjgpu_ctx *ctx) { return ctx ? ctx->last_pipeline : SJGPU_E_BADARG; }
const char *sjgpu_profile_kernel(const sjgpu_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int sjgpu_profile_enable(sjgpu_ctx *ctx, int on) {
  if (!ctx) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  SJ_TRY(ctx, hipDeviceSynchronize());
  drop_events(ctx);
  ctx->profile = on != 0;
  if (ctx->profile) { ctx->events.reserve(MAX_PROFILED_CALLS * PROFILE_EVENTS); } // pointers handed out stay valid
  return 0;
}

int sjgpu_profile_read(sjgpu_ctx *ctx, double *ms_sum, uint32_t *calls) {
  if (!ctx || !ms_sum || !calls) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  SJ_TRY(ctx, hipDeviceSynchronize());
  const size_t ncalls = ctx->events.size() / PROFILE_EVENTS;
  for (int k = 0; k < PROFILE_SLOTS; k++) { ms_sum[k] = 0.0; }
  for (size_t c = 0; c < ncalls; c++) {
    for (int k = 0; k < PROFILE_SLOTS; k++) {
      float ms = 0.f;
      const hipError_t e = hipEventElapsedTime(&ms, ctx->events[c * PROFILE_EVENTS + k], ctx->events[c * PROFILE_EVENTS + k + 1]);
      if (e != hipSuccess) { // a single-kernel call records events 0 and 1 only: its slots 1 and 2 are empty
        (void)hipGetLastError();
        if (k == 0) { SJ_TRY(ctx, e); }
        ms = 0.f;
      }
      ms_sum[k] += double(ms);
    }
  }
  *calls = uint32_t(ncalls);
  drop_events(ctx);
  return 0;
}
} // extern "C"
