// simdjson_amd/csrc/sjgpu_ctx.h -- what the translation units of the C-ABI share (round 6: sjgpu_capi.hip held contexts, pool, stream registry, range scheduler,
// the list passes and stage 2 in 1 950 lines): the context, the copy threads of the overlapped host path, and the helpers every entry point uses (error
// bookkeeping, workspace growth, the pipeline choice, enqueueing a scan).  The helpers live in an unnamed namespace: each unit gets its own copy, none holds state.
//   sjgpu_capi.hip         contexts and the pool, the device-resident entry points
//   sjgpu_capi_host.hip    host buffers: the overlapped path, windows of a registered stream, the stream registry, finish / depth scan, many small documents
//   sjgpu_capi_stage2.hip  strings, key comparison, stage 2 (the tape), sjgpu_parse
#ifndef SJGPU_CTX_H
#define SJGPU_CTX_H
#include "sjgpu.h"
#include "sjgpu_internal.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>

using namespace sjgpu;

// The host-buffer path of a large document moves its bytes on two helper threads, one per direction, each with its own
// stream: a pageable hipMemcpyAsync blocks its caller, so only separate threads keep both directions of the (full
// duplex) PCIe link busy while the calling thread launches scans.  Measured on this box: 56 GB/s either way alone,
// 97 GB/s both ways together (profiles/r01_pcie_overlap.txt).  Works with plain malloc / new[] memory on both sides.
struct copy_worker {
  struct job { void *dst; const void *src; size_t bytes; hipEvent_t record_after; };
  std::thread th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::deque<job> q;
  bool stop = false;
  size_t submitted = 0, finished = 0; // jobs since the last drain
  double busy_s = 0.0;                // time inside copies since the last drain (SJGPU_DEBUG_STREAM)
  size_t busy_bytes = 0;
  hipError_t err = hipSuccess;
  int device = 0;
  hipMemcpyKind kind = hipMemcpyDeviceToHost;
  hipStream_t stream = nullptr;

  void run() {
    (void)hipSetDevice(device);
    if (std::getenv("SJGPU_DEBUG_STREAM")) { std::fprintf(stderr, "[sjgpu] %s thread on cpu %d\n", kind == hipMemcpyHostToDevice ? "upload" : "download", sched_getcpu()); }
    for (;;) {
      std::unique_lock<std::mutex> lk(m);
      cv_job.wait(lk, [&] { return stop || !q.empty(); });
      if (q.empty()) { return; }
      const job j = q.front();
      q.pop_front();
      const bool skip = (err != hipSuccess); // after a failure the remaining jobs are only counted
      lk.unlock();
      hipError_t e = hipSuccess;
      const auto t0 = std::chrono::steady_clock::now();
      if (!skip) {
        e = hipMemcpyAsync(j.dst, j.src, j.bytes, kind, stream);
        if (e == hipSuccess && j.record_after) { e = hipEventRecord(j.record_after, stream); }
        // device-to-host: the caller reads the bytes as soon as we report the job finished
        if (e == hipSuccess && kind == hipMemcpyDeviceToHost) { e = hipStreamSynchronize(stream); }
      }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      lk.lock();
      busy_s += dt;
      busy_bytes += j.bytes;
      if (e != hipSuccess && err == hipSuccess) { err = e; }
      finished++;
      cv_done.notify_all();
    }
  }
  void submit(void *dst, const void *src, size_t bytes, hipEvent_t record_after = nullptr) {
    std::lock_guard<std::mutex> lk(m);
    q.push_back(job{dst, src, bytes, record_after});
    submitted++;
    cv_job.notify_one();
  }
  // blocks until the first `count` jobs since the last drain have been issued (and their events recorded)
  hipError_t wait_finished(size_t count) {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return finished >= count; });
    return err;
  }
  hipError_t drain() { // returns the first error since the last drain
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return finished == submitted; });
    const hipError_t e = err;
    err = hipSuccess;
    submitted = finished = 0;
    const double bs = busy_s;
    const size_t bb = busy_bytes;
    busy_s = 0.0;
    busy_bytes = 0;
    lk.unlock();
    const hipError_t se = hipStreamSynchronize(stream);
    if (bb && std::getenv("SJGPU_DEBUG_STREAM")) {
      std::fprintf(stderr, "[sjgpu]   %s thread: %.1f MB in %.2f ms busy = %.1f GB/s\n", kind == hipMemcpyHostToDevice ? "upload" : "download",
                   bb / 1e6, bs * 1e3, bb / bs / 1e9);
    }
    return e != hipSuccess ? e : se;
  }
  void shutdown() {
    if (th.joinable()) {
      { std::lock_guard<std::mutex> lk(m); stop = true; cv_job.notify_one(); }
      th.join();
    }
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
  }
};

struct sjgpu_ctx {
  int device = 0;
  size_t capacity = 0;    // the caller's limit on document length (CAPACITY beyond it); costs nothing by itself
  size_t ws_capacity = 0; // what the scan workspace below is sized for: grown by the first call that needs more
  hipStream_t stream = nullptr;
  // scan workspace (sized by ws_capacity)
  uint4 *masks = nullptr;
  seg_summary *summ = nullptr;
  seg_prefix *pref = nullptr;
  uint64_t *desc = nullptr; // single-pass pipeline: tile descriptors + ticket
  uint8_t *esc_tab = nullptr; // SEGMENT_BYTES_TABLE bytes: one byte per 16 KiB segment, the scratch of sjgpu_string_parity_device (rounds 1-3: the escape table)
  int pipeline = 2; // 0 split, 1 single pass, 2 auto (use_fused below)
  // AUTO remembers how dense the output of the last large stage-1 scan was (offsets per 1000 input bytes): on sparse
  // output the split pipeline is the faster one, and streams of documents / batches look like their predecessors
  uint32_t density_permille = 1000; // unknown: assume dense
  uint64_t pending_scan_bytes = 0;  // length of the stage-1 scan whose result has not been fetched yet (0: none / a range)
  int last_pipeline = 0;            // pipeline of the last enqueued scan (sjgpu_last_pipeline)
  const char *last_kernel = "";     // its dominant kernel(s), as the launcher reported them (sjgpu_profile_kernel)
  uint32_t max_workgroups = 2048;
  scan_result_dev *d_result = nullptr;
  scan_result_dev *h_result = nullptr; // pinned, 256 bytes: the scan's result; stage 2 reads its two results back at +64 and +128
  // staging for the host-buffer entry points (allocated on first use)
  uint8_t *d_in = nullptr;
  size_t d_in_bytes = 0;
  uint32_t *d_idx = nullptr;
  size_t d_idx_words = 0;
  uint8_t *d_out = nullptr;
  size_t d_out_bytes = 0;
  // sjgpu_stage1_tokens_device: where the segments' structural bytes wait between the two kernels of the split pipeline (one byte per input byte at most)
  uint8_t *d_tokstage = nullptr;
  size_t d_tokstage_bytes = 0;
  // small documents (sjgpu_small.hip): one page-locked block the one-workgroup kernel reads and writes across PCIe
  uint8_t *h_small = nullptr; // [result 64 B][descriptors][input][output]
  size_t h_small_bytes = 0;
  bool small_docs = true;     // env SJGPU_SMALL_DOCS=0 sends small documents through the tile pipelines (A/B, tests)
  int enqueue_rc = 0;         // failure of the workspace allocation inside the last enqueue_* (checked by SJ_ENQUEUED)
  // The single-pass kernels find [result][descriptors][control words] all zero because the kernel before them left them so (leave_and_clean); nothing on
  // the device says whether that kernel ran to its end.  Whatever makes that doubtful -- a HIP error recorded on this context (fail()), a chain that gave
  // up (SJGPU_F_INTERNAL), a traced run -- sets this, and the next single-pass call clears the workspace in front of its kernel instead of trusting it.
  bool ws_dirty = false;
  int device_finish = 1;      // streaming-mode finish: 0 host, 1 device beyond the small-document path, 2 always device
  // scratch of the device-side finish / depth scan (sjgpu_finish.hip), grown on demand
  uint8_t *d_tmp = nullptr;
  size_t d_tmp_bytes = 0;
  // stage 2 (sjgpu_tape.hip): string offsets + the tape builder's arrays; sjgpu_parse's device tape and string buffer
  uint8_t *d_stage2 = nullptr;
  size_t d_stage2_bytes = 0;
  uint8_t *d_doc = nullptr; // [tape words][string buffer] of sjgpu_parse
  size_t d_doc_bytes = 0;
  // look-ahead over a registered stream (sjgpu_stream_register): the raw structurals of ONE span of the stream, in page-locked
  // host memory, from which the windows document_stream asks for are cut without touching the GPU again
  struct span_slot {
    const uint8_t *base = nullptr; // the registered stream the span belongs to (null: empty slot)
    uint64_t stream_id = 0;        // ... and the registration it was made under
    size_t begin = 0, end = 0;     // the span, as offsets into the stream
    int state = 0;                 // 0 empty | 1 scan enqueued | 2 list download enqueued | 3 ready
    uint32_t n = 0;
    bool usable = false;           // false: the span holds an error the windows must find for themselves
    uint8_t *d_in = nullptr;
    size_t d_in_bytes = 0;
    uint32_t *d_idx = nullptr;
    size_t d_idx_bytes = 0;
    uint32_t *h_idx = nullptr;     // page-locked: offsets relative to `begin`
    size_t h_words = 0;
    scan_result_dev *h_res = nullptr; // page-locked copy of the scan's result
    hipEvent_t ev = nullptr;
  } la[2];
  uint32_t last_string_path = 0; // strings_result_dev::path of the last string pass (sjgpu_debug_string_path)
  int la_cur = 0; // the slot windows are being cut from; the other one holds (or awaits) the span behind it
  // overlapped host-buffer path (large documents): one copy thread per direction, one "range uploaded" event per range
  std::vector<copy_worker *> up, down; // range k travels on up[k % up.size()]; output piece k on down[k % down.size()]
  size_t copy_threads = 1;             // per direction (env SJGPU_COPY_THREADS)
  std::vector<hipEvent_t> ev_in;
  // Measured (profiles/r01_host_path_overlap.txt), 1 GiB documents: with page-locked buffers 16 MiB ranges and ONE copy
  // thread per direction give 27.1 ms (large_random) / 20.4 ms (twitter-like) in every context, against 42.1 / 28.3 ms
  // for upload, scan, download one after the other; 8 MiB: 27.6 / 21.6; 4 MiB: 28.8 / 24.5; two threads per direction are
  // slower and erratic (27-35 ms).  With pageable buffers the runtime has to pin every range it has not seen before, which
  // halves the rate of the copy thread (25 instead of 47 GB/s); the overlap then roughly pays for the pinning.
  size_t stream_from = size_t(64) << 20;  // documents at least this long take the overlapped path (env SJGPU_STREAM_FROM_MB, 0 = never)
  size_t stream_chunk = size_t(16) << 20; // range size, a multiple of RANGE_ALIGN (env SJGPU_STREAM_CHUNK_MB)
  // event profiling (sjgpu_profile_*)
  bool profile = false;
  std::vector<hipEvent_t> events; // PROFILE_EVENTS per recorded call
  char err[256] = {0};
};

namespace {

constexpr int E_CAPACITY = 1, E_UTF8 = 11, E_EMPTY = 13, E_UNCLOSED = 15, E_UNEXPECTED = 24;

int fail(sjgpu_ctx *ctx, hipError_t e, const char *what) {
  if (ctx) {
    std::snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, hipGetErrorString(e));
    ctx->ws_dirty = true; // whatever was in flight may not have reached its epilogue
  }
  static const bool trace = std::getenv("SJGPU_TRACE_ERRORS") != nullptr; // diagnostics: the library itself never prints otherwise
  if (trace) { std::fprintf(stderr, "[sjgpu] %s: %s\n", what, hipGetErrorString(e)); }
  return (e == hipErrorOutOfMemory) ? SJGPU_E_NOMEM : SJGPU_E_HIP;
}
#define SJ_TRY(ctx, call)                                  \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) { return fail((ctx), e_, #call); } \
  } while (0)

// behind every enqueue_stage1 / enqueue_minify: workspace allocation failures, then launch failures
#define SJ_ENQUEUED(ctx)                                                    \
  do {                                                                      \
    if ((ctx)->enqueue_rc) { const int r_ = (ctx)->enqueue_rc; (ctx)->enqueue_rc = 0; return r_; } \
    SJ_TRY((ctx), hipGetLastError());                                       \
  } while (0)

template <class T> void dev_free(T *&p) {
  if (p) { (void)hipFree(p); p = nullptr; }
}

int grow(sjgpu_ctx *ctx, void **p, size_t *have, size_t want) {
  if (*have >= want) { return 0; }
  if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
  SJ_TRY(ctx, hipMalloc(p, want));
  *have = want;
  return 0;
}

void release_scan_workspace(sjgpu_ctx *ctx) {
  dev_free(ctx->masks);
  dev_free(ctx->summ);
  dev_free(ctx->pref);
  dev_free(ctx->d_result); // also frees the descriptors behind it
  ctx->desc = nullptr;
  ctx->ws_capacity = 0;
}
void release_staging(sjgpu_ctx *ctx) {
  dev_free(ctx->d_in);
  dev_free(ctx->d_idx);
  dev_free(ctx->d_out);
  dev_free(ctx->d_tokstage);
  ctx->d_tokstage_bytes = 0;
  ctx->d_in_bytes = ctx->d_out_bytes = 0;
  ctx->d_idx_words = 0;
}

// [result][tile descriptors][control words]: one allocation, cleared ONCE, here -- every single-pass kernel puts what it used back to zero when it
// ends (sjgpu_fused.hip: leave_and_clean), so the calls themselves enqueue no clear (rounds 1-4: a hipMemsetAsync in front of every call)
int alloc_result(sjgpu_ctx *ctx, size_t for_len) {
  const size_t tiles = for_len ? num_fused_tiles(for_len) : 0;
  const size_t bytes = sizeof(scan_result_dev) + (tiles + FUSED_WORKSPACE_EXTRA_WORDS) * sizeof(uint64_t);
  SJ_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_result), bytes));
  ctx->desc = reinterpret_cast<uint64_t *>(ctx->d_result + 1);
  SJ_TRY(ctx, hipMemsetAsync(ctx->d_result, 0, bytes, nullptr));
  SJ_TRY(ctx, hipStreamSynchronize(nullptr)); // (the calls run on other streams: the zeros are there before any of them is enqueued)
  ctx->ws_dirty = false;
  return 0;
}

// Device workspace is allocated by the call that first needs it, for what THAT call needs: a context made for
// validate_utf8 or for small documents never pays for the masks of the split pipeline, and sjgpu_set_capacity only
// moves a limit.  Sizes grow geometrically so that a stream of ever larger documents re-allocates O(log) times.
size_t grown(size_t want) {
  size_t g = size_t(1) << 20;
  while (g < want) { g <<= 1; }
  return g > 0xFFFFFFFFull ? 0xFFFFFFFFull : g;
}
// what a scan of `len` bytes needs; split: the masks / summaries of the split pipeline too
int ensure_scan_workspace(sjgpu_ctx *ctx, size_t len, bool split) {
  if (ctx->ws_capacity >= len && ctx->d_result && (!split || (ctx->masks && ctx->summ && ctx->pref))) { return 0; }
  if (ctx->stream) { SJ_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
  const size_t cap = ctx->ws_capacity >= len ? ctx->ws_capacity : grown(len);
  if (cap != ctx->ws_capacity || !ctx->d_result) {
    release_scan_workspace(ctx);
    const int rc = alloc_result(ctx, cap);
    if (rc) { release_scan_workspace(ctx); return rc; }
    ctx->ws_capacity = cap;
  }
  if (split && !(ctx->masks && ctx->summ && ctx->pref)) {
    const size_t nseg = num_segments(cap);
    dev_free(ctx->masks); // all three or none: a later call must never meet half a workspace
    dev_free(ctx->summ);
    dev_free(ctx->pref);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->masks), nseg * (SEG_BYTES / BLOCK_BYTES) * sizeof(uint4));
    if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void **>(&ctx->summ), (nseg + num_groups(cap)) * sizeof(seg_summary)); }
    if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void **>(&ctx->pref), nseg * sizeof(seg_prefix)); }
    if (e != hipSuccess) {
      release_scan_workspace(ctx);
      return fail(ctx, e, "scan workspace");
    }
  }
  return 0;
}
int ensure_result_only(sjgpu_ctx *ctx) { return ctx->d_result ? 0 : alloc_result(ctx, 0); }

int fetch_result(sjgpu_ctx *ctx, hipStream_t s, sjgpu_scan_result *out) {
  SJ_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_result, sizeof(scan_result_dev), hipMemcpyDeviceToHost, s));
  SJ_TRY(ctx, hipStreamSynchronize(s));
  out->n = ctx->h_result->n;
  out->flags = ctx->h_result->flags;
  out->out_len = ctx->h_result->out_len;
  if (out->flags & SJGPU_F_INTERNAL) { ctx->ws_dirty = true; } // a chain that gave up: do not trust what its workgroups left behind
  if (ctx->pending_scan_bytes) {
    ctx->density_permille = uint32_t(uint64_t(out->n) * 1000u / ctx->pending_scan_bytes);
    ctx->pending_scan_bytes = 0;
  }
  return 0;
}

constexpr size_t MAX_PROFILED_CALLS = 4096;

// events for the next call, or nullptr when profiling is off / the ring is full
hipEvent_t *next_events(sjgpu_ctx *ctx) {
  if (!ctx->profile || ctx->events.size() >= MAX_PROFILED_CALLS * PROFILE_EVENTS) { return nullptr; }
  const size_t at = ctx->events.size();
  for (int k = 0; k < PROFILE_EVENTS; k++) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      while (ctx->events.size() > at) { (void)hipEventDestroy(ctx->events.back()); ctx->events.pop_back(); }
      return nullptr;
    }
    ctx->events.push_back(e);
  }
  return ctx->events.data() + at;
}

void drop_events(sjgpu_ctx *ctx) {
  for (hipEvent_t e : ctx->events) { (void)hipEventDestroy(e); }
  ctx->events.clear();
}

// device-resident calls run on the CALLER's stream; NULL is HIP's default (null) stream, which is also
// what torch.cuda.current_stream().cuda_stream reports for torch's default stream.
hipStream_t pick(sjgpu_ctx *, void *stream) { return static_cast<hipStream_t>(stream); }

// Measured on MI355X (profiles/r01_size_sweep.txt, large_random): the single-pass kernel with 16 KiB tiles wins up to
// a few MiB (8-14 us vs 18-20 us per call: one launch instead of four); between ~8 and ~192 MiB the split pipeline
// wins (its kernels fill the chip with 16 KiB work items, the 64 KiB-tile pipelined kernel needs >= 2 tiles per
// workgroup); above that the pipelined single-pass kernel wins on dense output (1 GiB: 0.52 vs 0.58 ms) and is
// loses 3-8 % on sparse output (twitter-like 0.12 offsets per byte: 2 180-2 250 vs 2 260-2 320 GB/s; amazon NDJSON 0.06:
// 2 290-2 440 vs 2 580-2 620), so for stage 1 AUTO goes by the density the previous large scan of this context saw.
// (round 5, profiles/r05_pipeline_sweep.txt: with the split kernels streaming their input the 16 KiB-tile kernel leads up to 4 MiB -- 16 against 20-21 us
// there -- and trails at 8 MiB, 31-37 against 22 us: the limit moved from 8 to 5 MiB)
constexpr size_t AUTO_FUSED_BELOW = size_t(5) << 20;
// (round 4, profiles/r04_pipeline_sweep.txt: with the table launch gone and the emission's shorter chains the split pipeline is the faster one
// on dense output up to 512 MiB -- 287 against 296 us there, 165 against 175 at 256 MiB -- and the single-pass kernel from 768 MiB on: 408
// against 422 us, 521 against 556 at 1 GiB; its fixed cost, one iteration to fill and one to drain, is ~35 us.  Was 192 MiB.)
// (round 4, later: the pipelined kernel with EIGHT waves per workgroup and 128 KiB tiles -- half the per-tile costs per byte -- wins on dense output from
// 256 MiB on: 155 against 162 us there, 268 against 288 at 512 MiB, 479 against 557 at 1 GiB; at 160 MiB the split pipeline still leads, 103 against 110.
// On sparse output the split pipeline stays ahead up to 512 MiB and level at 1 GiB.  The rows in profiles/r04_pipeline_sweep.txt.)
// (round 5: the split kernels request their chunks coalesced and streamed, the masks travel streamed -- profiles/r05_stream_ab.txt -- and lead on dense
// output up to 384 MiB, 197 against 205 us; level at 512 MiB, 265 : 265; the single-pass kernel from there on: 371 against 394 at 768 MiB, 465 against
// 515 at 1 GiB.  profiles/r05_pipeline_sweep.txt.  Was 224 MiB.)
constexpr size_t AUTO_FUSED_FROM = size_t(512) << 20;
// sparse output (twitter-like 0.12 offsets per byte, NDJSON 0.06) stays with the split pipeline at every size (round 5: 289 against 363 us per GiB of
// NDJSON, 363 against 436 on twitter-like text; rounds 1-4 had the two within a few per cent of each other at 1 GiB)
constexpr size_t AUTO_FUSED_FROM_SPARSE = ~size_t(0);
constexpr size_t AUTO_FUSED_FROM_MINIFY = size_t(192) << 20;
constexpr size_t DIRECT_HOST_MAX = size_t(2) << 20; // sjgpu_stage1 on host buffers: up to here the kernels write the offsets into host memory themselves
constexpr uint32_t AUTO_DENSE_PERMILLE = 200;
bool use_fused(const sjgpu_ctx *ctx, size_t len, int op = 1) { // op 0: stage 1, 1: minify
  if (ctx->pipeline != 2) { return ctx->pipeline == 1; }
  if (len <= AUTO_FUSED_BELOW) { return true; }
  if (op != 0) { return len >= AUTO_FUSED_FROM_MINIFY; } // minify: the on-chip kernel reads its input once, the split pipeline twice
  return len >= (ctx->density_permille >= AUTO_DENSE_PERMILLE ? AUTO_FUSED_FROM : AUTO_FUSED_FROM_SPARSE);
}

// sjgpu_stage1_tokens_device: pipeline 0 / 1 as the caller set it; AUTO: the small-input kernel up to its limit (one launch), the split pipeline beyond
bool tokens_fused(const sjgpu_ctx *ctx, size_t len) {
  if (ctx->pipeline != 2) { return ctx->pipeline == 1; }
  return len <= AUTO_FUSED_BELOW;
}

// `len` is the END of the scan (bytes [org.begin, len) are scanned); a whole document has org = {0, 0, 0}
// tok: the token-byte stream beside the offsets (split pipeline: staged by the scan kernel, copied by the emission kernel; single-pass: gathered at emission)
void enqueue_stage1(sjgpu_ctx *ctx, bool fused, const uint8_t *buf, size_t len, uint32_t *idx, size_t idx_words, hipStream_t s,
                    hipEvent_t *ev, scan_origin org = scan_origin{0, 0, 0}, uint8_t *tok = nullptr) {
  ctx->enqueue_rc = ensure_scan_workspace(ctx, len - org.begin, !fused);
  if (!ctx->enqueue_rc && tok && !fused) { ctx->enqueue_rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_tokstage), &ctx->d_tokstage_bytes, size_t(num_segments(grown(len - org.begin))) * SEG_BYTES + 64); }
  if (ctx->enqueue_rc) { return; }
  ctx->last_pipeline = fused ? 1 : 0;
  // the density AUTO decides by is taken from every whole-document scan beyond the small-input kernels' range (round 4 sampled only scans of
  // 224 MiB and more: a context that had once seen sparse output stayed on the split pipeline until another scan of that size measured dense)
  ctx->pending_scan_bytes = (org.begin == 0 && org.base0 == 0 && len > AUTO_FUSED_BELOW) ? len : 0;
  if (fused) {
    ctx->last_kernel = launch_stage1_fused(buf, len, ctx->desc, idx, idx_words, ctx->d_result, org, ctx->max_workgroups, s, ev, !ctx->ws_dirty, tok);
    ctx->ws_dirty = false; // (a dirty workspace was cleared in front of the kernel: clear_fused_workspace)
  }
  else {
    launch_stage1(buf, len, ctx->masks, ctx->summ, ctx->pref, idx, idx_words, ctx->d_result, org, s, ev, tok ? ctx->d_tokstage : nullptr, tok);
    ctx->last_kernel = tok ? "k_stage1_summarize<tokens>+k_resolve_groups+k_resolve_segments+k_stage1_emit<tokens>"
                           : "k_stage1_summarize+k_resolve_groups+k_resolve_segments+k_stage1_emit";
  }
}
void enqueue_minify(sjgpu_ctx *ctx, bool fused, const uint8_t *buf, size_t len, uint8_t *dst, hipStream_t s, hipEvent_t *ev,
                    scan_origin org = scan_origin{0, 0, 0}) {
  ctx->enqueue_rc = ensure_scan_workspace(ctx, len - org.begin, !fused);
  if (ctx->enqueue_rc) { return; }
  ctx->last_pipeline = fused ? 1 : 0;
  ctx->pending_scan_bytes = 0;
  if (fused) {
    ctx->last_kernel = launch_minify_fused(buf, len, ctx->desc, dst, ctx->d_result, org, ctx->max_workgroups, s, ev, !ctx->ws_dirty);
    ctx->ws_dirty = false;
  }
  else {
    launch_minify(buf, len, ctx->summ, ctx->pref, dst, ctx->d_result, org, s, ev);
    ctx->last_kernel = "k_minify_summarize+k_resolve_groups+k_resolve_segments+k_minify_emit";
  }
}

copy_worker *start_worker(sjgpu_ctx *ctx, hipMemcpyKind kind) {
  copy_worker *w = new (std::nothrow) copy_worker();
  if (!w) { return nullptr; }
  w->device = ctx->device;
  w->kind = kind;
  if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
  w->th = std::thread([w] { w->run(); });
  return w;
}

int ensure_streaming(sjgpu_ctx *ctx, size_t nranges) {
  if (ctx->up.empty() && std::getenv("SJGPU_DEBUG_STREAM")) { std::fprintf(stderr, "[sjgpu] caller on cpu %d\n", sched_getcpu()); }
  while (ctx->up.size() < ctx->copy_threads) {
    copy_worker *w = start_worker(ctx, hipMemcpyHostToDevice);
    if (!w) { return SJGPU_E_NOMEM; }
    ctx->up.push_back(w);
  }
  while (ctx->down.size() < ctx->copy_threads) {
    copy_worker *w = start_worker(ctx, hipMemcpyDeviceToHost);
    if (!w) { return SJGPU_E_NOMEM; }
    ctx->down.push_back(w);
  }
  while (ctx->ev_in.size() < nranges) {
    hipEvent_t e;
    SJ_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_in.push_back(e);
  }
  return 0;
}

// The overlapped host-buffer path (the GPU analogue of the reference's stage1_worker,
// dom/document_stream-inl.h:16-85): the document is uploaded and scanned in ranges.  The upload thread streams the
// ranges back to back; as soon as range k is resident this thread scans it (sjgpu_*_range_device's kernels), reads the
// 16-byte result and hands the new output to the download thread.  The only state between ranges is what one call's
// result holds: the output cursor and the in-string bit.
//   op 0: stage 1, out = idx_out (u32 words, room for out_cap words); op 1: minify, out = dst (bytes, room for len);
//   op 2: validate_utf8 (no output: the upload of range k+1 runs under the check of range k, the verdict is fetched once)
//   carry_in: CARRY_IN_STRING if the buffer is a piece of a larger document that begins inside a string (minify)
int run_streamed(sjgpu_ctx *ctx, int op, const uint8_t *buf, size_t len, void *out_host, size_t out_cap, sjgpu_scan_result *res_out,
                 uint32_t carry_in = 0) {
  const size_t chunk = ctx->stream_chunk;
  const size_t nranges = (len + chunk - 1) / chunk;
  int rc = ensure_streaming(ctx, nranges);
  if (rc) { return rc; }
  const size_t unit = (op == 0) ? sizeof(uint32_t) : 1;
  uint8_t *d_out = (op == 0) ? reinterpret_cast<uint8_t *>(ctx->d_idx) : ctx->d_out;
  hipStream_t s = ctx->stream;
  for (size_t k = 0; k < nranges; k++) {
    const size_t b = k * chunk, e = (b + chunk < len) ? b + chunk : len;
    ctx->up[k % ctx->up.size()]->submit(ctx->d_in + b, buf + b, e - b, ctx->ev_in[k]);
  }
  hipError_t he = hipSuccess;
  uint32_t flags = 0, in_string = carry_in & CARRY_IN_STRING, x_carry = 0; // x_carry: SJGPU_F_RANGE_CARRY of the range in front
  uint64_t cursor = 0; // output units produced by the ranges so far
  const bool debug = std::getenv("SJGPU_DEBUG_STREAM") != nullptr;
  double wait_upload_s = 0.0, wait_scan_s = 0.0;
  const auto t_begin = std::chrono::steady_clock::now();
  sjgpu_scan_result res{0, 0, 0};
  for (size_t k = 0; k < nranges && he == hipSuccess && rc == 0; k++) {
    const size_t b = k * chunk, e = (b + chunk < len) ? b + chunk : len;
    const bool last = (k + 1 == nranges);
    const scan_origin org{uint64_t(b), uint32_t(cursor), (in_string ? CARRY_IN_STRING : 0u) | (x_carry ? CARRY_X : 0u) | CARRY_SHARD | (last ? 0u : CARRY_MORE)};
    // the event of range k has been recorded (an unrecorded event would not be waited for): it is job k / T of thread k % T
    const auto tw0 = std::chrono::steady_clock::now();
    he = ctx->up[k % ctx->up.size()]->wait_finished(k / ctx->up.size() + 1);
    const auto tw1 = std::chrono::steady_clock::now();
    wait_upload_s += std::chrono::duration<double>(tw1 - tw0).count();
    if (he == hipSuccess) { he = hipStreamWaitEvent(s, ctx->ev_in[k], 0); }
    if (he != hipSuccess) { break; }
    if (op == 2) { // stateless but for the three bytes in front of the range, which are resident; flags accumulate on the device
      launch_validate_utf8(ctx->d_in, e, ctx->d_result, s, nullptr, b, !last);
      he = hipGetLastError();
      continue;
    }
    for (int attempt = 0; attempt < 2; attempt++) { // a single-pass range that gives up is re-run on the split pipeline
      const bool fused = use_fused(ctx, e - b, op) && attempt == 0;
      if (op == 0) { enqueue_stage1(ctx, fused, ctx->d_in, e, ctx->d_idx, ctx->d_idx_words, s, nullptr, org); }
      else { enqueue_minify(ctx, fused, ctx->d_in, e, ctx->d_out, s, nullptr, org); }
      if (ctx->enqueue_rc) { rc = ctx->enqueue_rc; ctx->enqueue_rc = 0; break; }
      he = hipGetLastError();
      if (he != hipSuccess) { break; }
      rc = fetch_result(ctx, s, &res);
      if (rc || !(res.flags & SJGPU_F_INTERNAL)) { break; }
    }
    wait_scan_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw1).count();
    if (he != hipSuccess || rc) { break; }
    flags |= res.flags & ~uint32_t(SJGPU_F_UNCLOSED_STRING | SJGPU_F_RANGE_CARRY);
    if (res.flags & (SJGPU_F_INTERNAL | SJGPU_F_IDX_OVERFLOW)) { break; }
    const uint64_t now = (op == 0) ? uint64_t(res.n) : res.out_len;
    const uint64_t upto = now + ((op == 0 && last) ? 3 : 0); // the sentinels travel with the last range
    if (now < cursor || upto > out_cap) { rc = SJGPU_E_OVERFLOW; break; }
    if (upto > cursor) {
      ctx->down[k % ctx->down.size()]->submit(static_cast<uint8_t *>(out_host) + cursor * unit, d_out + cursor * unit,
                                              size_t(upto - cursor) * unit);
    }
    cursor = now;
    in_string = res.flags & SJGPU_F_UNCLOSED_STRING;
    x_carry = res.flags & SJGPU_F_RANGE_CARRY;
  }
  // nothing may be left in flight when we return: the caller owns buf and out_host
  const auto t_loop = std::chrono::steady_clock::now();
  hipError_t ue = hipSuccess, de = hipSuccess;
  for (copy_worker *w : ctx->up) {
    const hipError_t e = w->drain();
    if (ue == hipSuccess) { ue = e; }
  }
  for (copy_worker *w : ctx->down) {
    const hipError_t e = w->drain();
    if (de == hipSuccess) { de = e; }
  }
  if (debug) {
    const auto t_end = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[sjgpu] streamed call: %zu ranges, %.2f ms total = %.2f waiting for uploads + %.2f launching/waiting for scans + %.2f draining downloads\n",
                 nranges, std::chrono::duration<double>(t_end - t_begin).count() * 1e3, wait_upload_s * 1e3, wait_scan_s * 1e3,
                 std::chrono::duration<double>(t_end - t_loop).count() * 1e3);
  }
  if (he != hipSuccess) { return fail(ctx, he, "streamed scan"); }
  if (ue != hipSuccess) { return fail(ctx, ue, "streamed scan: upload"); }
  if (de != hipSuccess) { return fail(ctx, de, "streamed scan: download"); }
  if (rc) { return rc; }
  if (op == 2) { return fetch_result(ctx, s, res_out); }
  res_out->n = (op == 0) ? uint32_t(cursor) : 0;
  res_out->out_len = (op == 0) ? 0 : cursor;
  res_out->flags = flags | in_string;
  return 0;
}

bool take_streamed_path(const sjgpu_ctx *ctx, size_t len) {
  return ctx->stream_from != 0 && len >= ctx->stream_from && len > ctx->stream_chunk;
}

int ensure_staging_in(sjgpu_ctx *ctx, size_t len) { return grow(ctx, reinterpret_cast<void **>(&ctx->d_in), &ctx->d_in_bytes, grown(len) + 64); }
int ensure_tmp(sjgpu_ctx *ctx, size_t bytes) { return grow(ctx, reinterpret_cast<void **>(&ctx->d_tmp), &ctx->d_tmp_bytes, bytes); }

} // namespace

#endif
