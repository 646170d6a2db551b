// simdjson_amd/csrc/sjgpu_capi.hip -- the C-ABI of include/sjgpu.h: context/workspace management,
// host<->device staging for the plug-in path, and kernel enqueueing.  No CPU compute path exists
// here: if HIP is unusable every entry point returns a negative code.
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

} // namespace

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

// ---- host-buffer entry points (the plug-in path: H2D, scan, D2H, host finish) ---------------------------
} // extern "C"

namespace {

constexpr size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Streaming-mode documents at least this long are finished on the device (a dozen small launches, ~60 us, against
// downloading and walking a list of millions of offsets); shorter ones keep the host walk, which is O(last document).
constexpr size_t DEVICE_FINISH_FROM = size_t(4) << 20;

int ensure_small(sjgpu_ctx *ctx, size_t bytes) {
  if (ctx->h_small_bytes >= bytes) { return 0; }
  if (ctx->h_small) { (void)hipHostFree(ctx->h_small); ctx->h_small = nullptr; ctx->h_small_bytes = 0; }
  size_t want = size_t(256) << 10;
  while (want < bytes) { want <<= 1; }
  SJ_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_small), want, hipHostMallocDefault));
  ctx->h_small_bytes = want;
  return 0;
}

// ONE small document through the one-workgroup kernel (sjgpu_small.hip): the document is copied into the context's
// page-locked block, the kernel reads it and writes offsets / bytes and the result back across PCIe, the host waits once.
// *out = where the kernel left the output (inside the block, valid until the context's next call).
int small_single(sjgpu_ctx *ctx, int op, const uint8_t *buf, size_t len, sjgpu_scan_result *res, const void **out) {
  const size_t in_at = 64, out_at = in_at + round_up(len, 64) + 64;
  const size_t out_bytes = op == 0 ? (len + 3) * sizeof(uint32_t) : (op == 1 ? len + 16 : 0);
  int rc = ensure_small(ctx, out_at + out_bytes + 64);
  if (rc) { return rc; }
  std::memcpy(ctx->h_small + in_at, buf, len);
  scan_result_dev *r = reinterpret_cast<scan_result_dev *>(ctx->h_small);
  ctx->pending_scan_bytes = 0;
  ctx->last_kernel = op == 0 ? "k_docs<0>" : (op == 1 ? "k_docs<1>" : "k_docs<2>");
  launch_docs(op, ctx->h_small + in_at, nullptr, doc_desc{0, 0, uint32_t(len), 0}, 1, ctx->h_small + out_at, r, ctx->stream);
  SJ_TRY(ctx, hipGetLastError());
  SJ_TRY(ctx, hipStreamSynchronize(ctx->stream));
  res->n = r->n;
  res->flags = r->flags;
  res->out_len = r->out_len;
  if (out) { *out = ctx->h_small + out_at; }
  return 0;
}

int ensure_staging_in(sjgpu_ctx *ctx, size_t len) { return grow(ctx, reinterpret_cast<void **>(&ctx->d_in), &ctx->d_in_bytes, grown(len) + 64); }

// pieces of inputs beyond what one scan addresses (32-bit offsets); env SJGPU_PIECE_MB for tests
size_t piece_bytes() {
  size_t mb = 1024;
  if (const char *v = std::getenv("SJGPU_PIECE_MB")) {
    const size_t x = size_t(std::strtoull(v, nullptr, 10));
    if (x >= 1 && x <= 2048) { mb = x; }
  }
  return mb << 20;
}

// one buffer that is well-formed or not by itself (a whole input, or a piece cut at a character boundary)
int validate_piece(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int *ok) {
  sjgpu_scan_result res{0, 0, 0};
  int rc = 0;
  if (ctx->small_docs && len <= DOCS_SINGLE_MAX) {
    rc = small_single(ctx, 2, buf, len, &res, nullptr);
  } else {
    rc = ensure_result_only(ctx);
    if (!rc) { rc = ensure_staging_in(ctx, len); }
    if (rc) { return rc; }
    ctx->pending_scan_bytes = 0;
    ctx->last_kernel = "k_validate_utf8";
    if (take_streamed_path(ctx, len)) {
      rc = run_streamed(ctx, 2, buf, len, nullptr, 0, &res);
    } else {
      hipStream_t s = ctx->stream;
      SJ_TRY(ctx, hipMemcpyAsync(ctx->d_in, buf, len, hipMemcpyHostToDevice, s));
      launch_validate_utf8(ctx->d_in, len, ctx->d_result, s, nullptr);
      SJ_TRY(ctx, hipGetLastError());
      rc = fetch_result(ctx, s, &res);
    }
  }
  if (rc) { return rc; }
  *ok = (res.flags & SJGPU_F_UTF8_ERROR) ? 0 : 1;
  return 0;
}

// one buffer of at most 4 GiB - 1 bytes; in_string / shard: it is a piece of a larger document (sjgpu_clean_cut)
int minify_piece(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, bool shard, uint32_t in_string, uint8_t *dst, sjgpu_scan_result *res) {
  int rc = 0;
  if (ctx->small_docs && len <= DOCS_SINGLE_MAX && !shard) {
    const void *out = nullptr;
    rc = small_single(ctx, 1, buf, len, res, &out);
    if (rc) { return rc; }
    if (res->out_len > len) { return E_UNEXPECTED; }
    std::memcpy(dst, out, res->out_len);
    return 0;
  }
  rc = ensure_staging_in(ctx, len);
  if (!rc) { rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_out), &ctx->d_out_bytes, grown(len) + 64); }
  if (rc) { return rc; }
  hipStream_t s = ctx->stream;
  const uint32_t carry = (shard ? CARRY_SHARD : 0u) | (in_string ? CARRY_IN_STRING : 0u);
  const bool streamed = take_streamed_path(ctx, len);
  if (streamed) {
    rc = run_streamed(ctx, 1, buf, len, dst, len, res, carry);
    if (rc) { return rc; }
  } else {
    SJ_TRY(ctx, hipMemcpyAsync(ctx->d_in, buf, len, hipMemcpyHostToDevice, s));
    for (int attempt = 0; attempt < 2; attempt++) { // a single-pass call that gives up is re-run on the split pipeline
      enqueue_minify(ctx, use_fused(ctx, len) && attempt == 0, ctx->d_in, len, ctx->d_out, s, nullptr, scan_origin{0, 0, carry});
      SJ_ENQUEUED(ctx);
      rc = fetch_result(ctx, s, res);
      if (rc) { return rc; }
      if (!(res->flags & SJGPU_F_INTERNAL)) { break; }
    }
  }
  if (res->flags & SJGPU_F_INTERNAL) { return E_UNEXPECTED; }
  if (res->out_len > len) { return E_UNEXPECTED; }
  if (!streamed && res->out_len && !((res->flags & SJGPU_F_UNCLOSED_STRING) && !shard)) {
    SJ_TRY(ctx, hipMemcpyAsync(dst, ctx->d_out, res->out_len, hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
  }
  return 0;
}


int ensure_tmp(sjgpu_ctx *ctx, size_t bytes) { return grow(ctx, reinterpret_cast<void **>(&ctx->d_tmp), &ctx->d_tmp_bytes, bytes); }

// What finish() decides for a streaming mode (json_structural_indexer.h:295-394), with the list still on the device:
// the filters and the boundary search run there (sjgpu_finish.hip), the host reads back one small state and applies the
// same scalar edits stage1_finish.cpp applies.  `edit(pos, value)` stores one word of the caller's copy of the list,
// `word(pos)` reads one (device or host copy -- the caller decides where the list lives).
struct finish_decision {
  int error;
  uint32_t n_io;
  bool write_next_start; // idx[n_io] = next_start                     (partial filter modes)
  bool shift_sentinel;   // idx[n_io + 1] = idx[n_io]; idx[n_io] = len (final modes)
  uint32_t next_start;
  bool need_first_word;  // streaming_partial with nothing complete: CAPACITY iff idx[0] == 0, else EMPTY with n_io = 0
};

// Runs the device part for the n_raw structurals of dev_idx and returns the decision; `n_after_unclosed` is filled with
// the list length the reference works on (the dangling opening quote of an unclosed string is dropped first).
int decide_on_device(sjgpu_ctx *ctx, const uint8_t *dev_buf, size_t len, int mode, uint32_t *dev_idx, uint32_t n_raw, uint32_t flags,
                     hipStream_t s, finish_decision *d) {
  const bool partial = mode == SJGPU_STREAMING_PARTIAL || mode == SJGPU_JSON_SEQUENCE_PARTIAL || mode == SJGPU_COMMA_DELIMITED_PARTIAL;
  const bool final_mode = !partial;
  *d = finish_decision{0, n_raw, false, false, uint32_t(len), false};
  if (n_raw == 0) { d->error = E_EMPTY; return 0; }
  uint32_t n = n_raw;
  if (flags & SJGPU_F_UNCLOSED_STRING) { // the last structural is the dangling opening quote
    d->n_io = --n;
    if (partial && n == 0) { d->error = E_CAPACITY; return 0; }
  }
  finish_state st{};
  st.n_report = n;
  if (n > 0) {
    int rc = ensure_tmp(ctx, finish_workspace_bytes(n));
    if (rc) { return rc; }
    launch_finish(mode, dev_buf, len, dev_idx, n, ctx->d_tmp, s);
    SJ_TRY(ctx, hipGetLastError());
    SJ_TRY(ctx, hipMemcpyAsync(&st, ctx->d_tmp, sizeof st, hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
  }
  const uint32_t utf8 = (flags & SJGPU_F_UTF8_ERROR) ? E_UTF8 : 0;
  if (mode == SJGPU_STREAMING_PARTIAL) {
    if (st.keep == 0 && n > 0) { d->need_first_word = true; d->n_io = n; d->error = 0; return 0; } // resolved by the caller
    d->n_io = st.keep;
    d->error = int(utf8);
    return 0;
  }
  if (mode == SJGPU_STREAMING_FINAL) {
    d->n_io = st.keep;
    d->shift_sentinel = true;
    d->error = st.keep == 0 ? E_EMPTY : int(utf8);
    return 0;
  }
  // json_sequence / comma_delimited
  d->next_start = st.next_start;
  if (partial) {
    d->n_io = st.n_report;
    if (st.verdict == FIN_TOO_LARGE) { d->error = E_CAPACITY; return 0; }
    if (st.keep == 0) { d->n_io = 0; d->error = E_EMPTY; return 0; }
    d->n_io = st.keep;
    d->write_next_start = true;
    d->error = int(utf8);
    return 0;
  }
  (void)final_mode;
  d->n_io = st.keep;
  d->shift_sentinel = true;
  d->error = st.keep == 0 ? E_EMPTY : int(utf8);
  return 0;
}

// sjgpu_stage1's tail for the streaming modes when the list is on the device: decide there, fetch only what is kept
int finish_on_device_and_fetch(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words,
                               const sjgpu_scan_result &res, uint32_t *n_io, uint32_t *next_io) {
  (void)buf;
  hipStream_t s = ctx->stream;
  finish_decision d;
  int rc = decide_on_device(ctx, ctx->d_in, len, mode, ctx->d_idx, res.n, res.flags, s, &d);
  if (rc) { return rc; }
  *n_io = d.n_io;
  if (next_io) { *next_io = 0; }
  // the words a caller may look at: idx[0 .. n_io + 2] (never beyond the raw list and its three sentinels)
  size_t words = size_t(d.n_io) + 3;
  if (words > size_t(res.n) + 3) { words = size_t(res.n) + 3; }
  if (words > idx_words) { return SJGPU_E_OVERFLOW; }
  SJ_TRY(ctx, hipMemcpyAsync(idx_out, ctx->d_idx, words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  SJ_TRY(ctx, hipStreamSynchronize(s));
  if (d.need_first_word) {
    if (idx_out[0] == 0) { return E_CAPACITY; } // one document fills the whole window
    *n_io = 0;
    return E_EMPTY; // leading whitespace only; the document may fit the next window
  }
  if (d.write_next_start) { idx_out[d.n_io] = d.next_start; }
  if (d.shift_sentinel) {
    idx_out[d.n_io + 1] = idx_out[d.n_io]; // lets the stream compute truncated_bytes (json_structural_indexer.h:334-337)
    idx_out[d.n_io] = uint32_t(len);
  }
  return d.error;
}

// ---- windows of ONE stream (parse_many): scan a span once, cut the windows out of it --------------------------------------------------
// document_stream calls stage1 on consecutive windows of one buffer (/root/reference/include/simdjson/dom/document_stream-inl.h:285-317:
// &buf[batch_start], batch_size -- 1 MB by default).  One launch and one PCIe round trip per megabyte is what loses to a CPU kernel
// (round 2: 92 us against 68).  The interface hands over a window, not the stream, and nothing beyond a window may be touched on a
// guess -- so the integrator says where the stream lies (sjgpu_stream_register: the in-tree patch calls it from document_stream::start,
// out-of-tree users from simdjson::mi355x::register_stream).  A window inside a registered stream is then answered from a SPAN:
// tens of megabytes uploaded and scanned once, their raw structurals kept in page-locked host memory.  Cutting a window out of them is
// exact because every window of a document stream begins at a structural of the scan in front of it (idx[n] of the previous finish())
// -- outside any string, not escaped, a token start -- or at the start of the span itself; the window's own flags are rebuilt from its
// bytes (does it end inside a string?), and spans that hold an error the windows must report are not used at all.
struct stream_extent {
  const uint8_t *base;
  size_t len;
  bool pinned;
  uint64_t id; // unique per registration: a later stream at the same address must not meet the spans of an earlier one
  uint32_t refs; // registrations alive for this base: two streams over one buffer must not unregister each other
  std::vector<size_t> lens; // the length every live registration named: `len` is the part of the buffer ALL of them vouch for (their minimum)
};
struct stream_registry {
  std::mutex m;
  std::vector<stream_extent> list;
  uint64_t next_id = 1;
};
stream_registry &streams() {
  static stream_registry *r = new stream_registry(); // never destroyed (see ctx_pool)
  return *r;
}
bool find_stream(const uint8_t *buf, size_t len, stream_extent *out) {
  stream_registry &r = streams();
  std::lock_guard<std::mutex> lk(r.m);
  for (const stream_extent &e : r.list) {
    if (buf >= e.base && buf + len <= e.base + e.len) { *out = e; return true; }
  }
  return false;
}

constexpr size_t STREAM_PIN_FROM = size_t(8) << 20;
constexpr size_t LA_WINDOW_MAX = size_t(8) << 20;  // longer windows are worth a scan of their own
constexpr size_t LA_SPAN = size_t(32) << 20;

// index of the first entry >= x
uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint64_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (a[mid] < x) { lo = mid + 1; } else { hi = mid; }
  }
  return lo;
}

// Upload, scan and result read-back of the span [begin, begin + span) of stream e into slot sl: only enqueued.
int span_issue(sjgpu_ctx *ctx, sjgpu_ctx::span_slot &sl, const stream_extent &e, size_t begin, size_t min_len) {
  sl.base = nullptr;
  sl.state = 0;
  size_t span = e.len - begin;
  if (span > LA_SPAN) { span = LA_SPAN > min_len ? LA_SPAN : min_len; }
  if (span > 0xFFFFFFF0ull) { return 0; }
  int rc = grow(ctx, reinterpret_cast<void **>(&sl.d_in), &sl.d_in_bytes, grown(span) + 64);
  if (!rc) { rc = grow(ctx, reinterpret_cast<void **>(&sl.d_idx), &sl.d_idx_bytes, (grown(span) + 16) * sizeof(uint32_t)); }
  if (rc) { return rc; }
  if (!sl.h_res) { SJ_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_res), sizeof(scan_result_dev), hipHostMallocDefault)); }
  if (!sl.ev) { SJ_TRY(ctx, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming)); }
  hipStream_t s = ctx->stream;
  if (hipMemcpyAsync(sl.d_in, e.base + begin, span, hipMemcpyHostToDevice, s) != hipSuccess) {
    (void)hipGetLastError(); // the runtime refuses this host range: no span, the window takes the ordinary path
    return 0;
  }
  const uint32_t carry = (begin + span < e.len) ? CARRY_MORE : 0u; // the stream goes on behind the span: no end-of-input rule
  // the split pipeline: a look-ahead scan must not be able to give up (the single-pass kernels' SJGPU_F_INTERNAL needs a re-run)
  enqueue_stage1(ctx, false, sl.d_in, span, sl.d_idx, sl.d_idx_bytes / sizeof(uint32_t), s, nullptr, scan_origin{0, 0, carry});
  SJ_ENQUEUED(ctx);
  ctx->pending_scan_bytes = 0;
  SJ_TRY(ctx, hipMemcpyAsync(sl.h_res, ctx->d_result, sizeof(scan_result_dev), hipMemcpyDeviceToHost, s));
  SJ_TRY(ctx, hipEventRecord(sl.ev, s));
  sl.base = e.base;
  sl.stream_id = e.id;
  sl.begin = begin;
  sl.end = begin + span;
  sl.state = 1;
  return 0;
}
// Moves a slot towards "ready"; blocking = wait for what is in flight, else only take what has already happened.
int span_advance(sjgpu_ctx *ctx, sjgpu_ctx::span_slot &sl, bool blocking) {
  if (sl.state == 1) {
    if (blocking) { SJ_TRY(ctx, hipEventSynchronize(sl.ev)); }
    else if (hipEventQuery(sl.ev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    sl.n = sl.h_res->n;
    const uint32_t flags = sl.h_res->flags;
    // a control character inside a string or broken UTF-8 SOMEWHERE in the span says nothing about a particular window
    sl.usable = (flags & (SJGPU_F_UNESCAPED_CTRL | SJGPU_F_UTF8_ERROR | SJGPU_F_INTERNAL | SJGPU_F_IDX_OVERFLOW)) == 0;
    if (size_t(sl.n) + 8 > sl.h_words) {
      if (sl.h_idx) { (void)hipHostFree(sl.h_idx); sl.h_idx = nullptr; sl.h_words = 0; }
      size_t want = size_t(1) << 16;
      while (want < size_t(sl.n) + 8) { want <<= 1; }
      SJ_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_idx), want * sizeof(uint32_t), hipHostMallocDefault));
      sl.h_words = want;
    }
    if (sl.n && sl.usable) { SJ_TRY(ctx, hipMemcpyAsync(sl.h_idx, sl.d_idx, size_t(sl.n) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); }
    SJ_TRY(ctx, hipEventRecord(sl.ev, ctx->stream));
    sl.state = 2;
  }
  if (sl.state == 2) {
    if (blocking) { SJ_TRY(ctx, hipEventSynchronize(sl.ev)); }
    else if (hipEventQuery(sl.ev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    sl.state = 3;
  }
  return 0;
}
// does buf[0 .. len) end inside a string, given that its last structural sits at `last`?  Only an OPENING quote is ever a structural.
bool ends_inside_string(const uint8_t *buf, size_t len, uint32_t last) {
  if (buf[last] != '"') { return false; }
  for (size_t j = size_t(last) + 1; j < len; j++) {
    if (buf[j] == '\\') { j++; }
    else if (buf[j] == '"') { return false; }
  }
  return true;
}
// Where the span behind a ready one should begin: the first structural of the document that is still open `window` bytes in front
// of the span's end -- a position the scan has PROVED to lie outside every string and between tokens (what finish() of a partial
// batch computes, json_structural_indexer.h:295-333), chosen so that every window of that size which begins in front of it still
// fits this span.  0 = no such position (no complete document in front of it).
size_t span_successor(const sjgpu_ctx::span_slot &sl, const stream_extent &e, size_t window) {
  if (!sl.usable || sl.n < 2 || sl.end >= e.len) { return 0; }
  const size_t span = sl.end - sl.begin;
  if (window < (size_t(64) << 10)) { window = size_t(64) << 10; }
  if (span < 4 * window) { return 0; }
  const size_t cut = span - window;
  const uint32_t n_cut = lower_bound_u32(sl.h_idx, sl.n, cut);
  if (n_cut < 2) { return 0; }
  const uint8_t *base = e.base + sl.begin;
  const uint32_t flags = ends_inside_string(base, cut, sl.h_idx[n_cut - 1]) ? SJGPU_F_UNCLOSED_STRING : 0u;
  uint32_t n_io = 0, next = 0;
  const uint32_t s0 = sl.h_idx[n_cut], s1 = sl.h_idx[n_cut + 1], s2 = sl.h_idx[n_cut + 2]; // finish() parks its sentinels behind the list it is given
  const int err = sjgpu_stage1_finish_host(base, cut, SJGPU_STREAMING_PARTIAL, sl.h_idx, n_cut, flags, &n_io, &next);
  sl.h_idx[n_cut] = s0; sl.h_idx[n_cut + 1] = s1; sl.h_idx[n_cut + 2] = s2;
  if (err != 0 || n_io == 0 || n_io >= sl.n) { return 0; }
  return sl.begin + sl.h_idx[n_io];
}

// *served = false: take the ordinary path.  len is the window's length after the partial-UTF-8 trim.
int stage1_from_span(sjgpu_ctx *ctx, const stream_extent &e, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words, uint32_t *n_io,
                     uint32_t *next_io, bool *served) {
  *served = false;
  const size_t off = size_t(buf - e.base);
  auto covers = [&](const sjgpu_ctx::span_slot &sl) { return sl.state != 0 && sl.base == e.base && sl.stream_id == e.id && off >= sl.begin && off + len <= sl.end; };
  if (!covers(ctx->la[ctx->la_cur])) {
    if (covers(ctx->la[ctx->la_cur ^ 1])) { ctx->la_cur ^= 1; } // the span that was fetched ahead
    else { // a new span, beginning with this window
      sjgpu_ctx::span_slot &other = ctx->la[ctx->la_cur ^ 1];
      if (other.state == 1 || other.state == 2) { SJ_TRY(ctx, hipStreamSynchronize(ctx->stream)); other.state = 0; other.base = nullptr; } // nothing of ours stays in flight
      const int rc = span_issue(ctx, ctx->la[ctx->la_cur], e, off, len);
      if (rc || ctx->la[ctx->la_cur].state == 0) { return rc; }
    }
  }
  sjgpu_ctx::span_slot &sl = ctx->la[ctx->la_cur];
  const bool fresh = sl.state != 3;
  int rc = span_advance(ctx, sl, true);
  if (rc) { return rc; }
  if (fresh) { // the span has just become readable: fetch the one behind it while the caller works through this one's windows
    sjgpu_ctx::span_slot &next = ctx->la[ctx->la_cur ^ 1];
    const size_t at = span_successor(sl, e, len);
    if (at > sl.begin && !(next.state != 0 && next.base == e.base && next.stream_id == e.id && next.begin == at)) {
      rc = span_issue(ctx, next, e, at, 0);
      if (rc) { return rc; }
    }
  } else {
    rc = span_advance(ctx, ctx->la[ctx->la_cur ^ 1], false); // keep the prefetch moving (its list download waits for its scan)
    if (rc) { return rc; }
  }
  if (!sl.usable) { return 0; }
  const uint64_t rel = off - sl.begin;
  const uint32_t *list = sl.h_idx;
  const uint32_t lo = lower_bound_u32(list, sl.n, rel);
  if (rel != 0 && !(lo < sl.n && list[lo] == rel)) { return 0; } // the window does not begin at a token of the span's scan: not ours to answer
  const uint32_t hi = lower_bound_u32(list, sl.n, rel + len);
  const uint32_t n_raw = hi - lo;
  if (size_t(n_raw) + 3 > idx_words) { return SJGPU_E_OVERFLOW; }
  const uint32_t shift = uint32_t(rel);
  for (uint32_t k = 0; k < n_raw; k++) { idx_out[k] = list[lo + k] - shift; }
  // the window's own flag: does it end inside a string?  Only an opening quote is ever a structural, so that is the case iff the last
  // structural is a quote whose closing quote lies beyond the window.
  const uint32_t flags = (n_raw && ends_inside_string(buf, len, idx_out[n_raw - 1])) ? SJGPU_F_UNCLOSED_STRING : 0u;
  *served = true;
  return sjgpu_stage1_finish_host(buf, len, mode, idx_out, n_raw, flags, n_io, next_io);
}

} // namespace

extern "C" {

int sjgpu_stream_register(const uint8_t *base, size_t len) {
  if (!base || len == 0) { return SJGPU_E_BADARG; }
  stream_extent e{base, len, false, 0, 1, {len}};
  // Page-locking pays for itself on streams of many megabytes (the upload of a span runs at twice the rate and truly asynchronously);
  // small buffers come and go at addresses the allocator hands out again, and registering / unregistering those by the thousand
  // (the reference's document_stream tests) is what the runtime is not made for: they stay pageable.
  static const bool pin = []() { const char *v = std::getenv("SJGPU_STREAM_PIN"); return !v || v[0] != '0'; }();
  if (pin && len >= STREAM_PIN_FROM && sjgpu_device_count() > 0) { e.pinned = hipHostRegister(const_cast<uint8_t *>(base), len, hipHostRegisterDefault) == hipSuccess; }
  (void)hipGetLastError(); // a range that cannot be page-locked (already registered, read-only mapping) still works, only slower
  stream_registry &r = streams();
  std::lock_guard<std::mutex> lk(r.m);
  e.id = r.next_id++;
  for (stream_extent &x : r.list) {
    if (x.base == base) { // registered again (a second stream over the same buffer): spans as good as new, one more unregister to wait for
      // The extent served from spans is what EVERY live registration vouches for.  (Round 4 kept the maximum: when the longer of two streams
      // left first and its owner freed the tail, the survivor still advertised it and a span upload could read freed bytes -- ADVICE r4.)
      x.lens.push_back(len);
      x.len = len < x.len ? len : x.len;
      x.pinned = x.pinned || e.pinned;
      x.id = e.id;
      x.refs++;
      return 0;
    }
  }
  r.list.push_back(e);
  return 0;
}

// len == 0: the caller does not say which registration over `base` leaves
static int stream_unregister_impl(const uint8_t *base, size_t len) {
  if (!base) { return SJGPU_E_BADARG; }
  stream_registry &r = streams();
  bool pinned = false, found = false;
  {
    std::lock_guard<std::mutex> lk(r.m);
    for (size_t i = 0; i < r.list.size(); i++) {
      if (r.list[i].base == base) {
        if (--r.list[i].refs > 0) { // another stream over the same buffer is still at work.  The entry named by `len` leaves; when the caller does not
          // say (or names a length nobody registered) assume the LONGEST did -- the extent never grows beyond what the remaining ones are known to
          // cover (windows beyond it take the ordinary path).  The extent served from spans is the shortest of those that STAY.
          std::vector<size_t> &ls = r.list[i].lens;
          size_t at = ls.size();
          for (size_t k = 0; k < ls.size() && len != 0; k++) { if (ls[k] == len) { at = k; break; } }
          if (at == ls.size()) {
            at = 0;
            for (size_t k = 1; k < ls.size(); k++) { if (ls[k] > ls[at]) { at = k; } }
          }
          if (!ls.empty()) { ls.erase(ls.begin() + long(at)); }
          if (!ls.empty()) {
            size_t m = ls[0];
            for (size_t v : ls) { m = v < m ? v : m; }
            r.list[i].len = m;
          }
          return 0;
        }
        pinned = r.list[i].pinned;
        r.list.erase(r.list.begin() + long(i));
        found = true;
        break;
      }
    }
  }
  if (found) { // a span may still be on its way to some device: the caller is about to let go of the bytes
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess) {
      int before = 0;
      (void)hipGetDevice(&before);
      for (int d = 0; d < ndev; d++) {
        if (hipSetDevice(d) == hipSuccess) { (void)hipDeviceSynchronize(); }
      }
      (void)hipSetDevice(before);
    }
    (void)hipGetLastError();
  }
  if (pinned) { (void)hipHostUnregister(const_cast<uint8_t *>(base)); }
  return found ? 0 : SJGPU_E_BADARG;
}
int sjgpu_stream_unregister(const uint8_t *base) { return stream_unregister_impl(base, 0); }
size_t sjgpu_debug_stream_extent(const uint8_t *base) { // the span-served extent of the registration(s) over `base` (0: none) -- host logic, for the tests
  stream_extent e;
  return (base && find_stream(base, 1, &e) && e.base == base) ? e.len : 0;
}
int sjgpu_stream_unregister_len(const uint8_t *base, size_t len) { return stream_unregister_impl(base, len); }

int sjgpu_stage1(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words, uint32_t *n_io,
                 uint32_t *next_io) {
  if (!ctx || !n_io || mode < SJGPU_REGULAR || mode > SJGPU_COMMA_DELIMITED_FINAL) { return SJGPU_E_BADARG; }
  if (len > ctx->capacity) { return E_CAPACITY; } // json_structural_indexer.h:195
  if (len == 0) { return E_EMPTY; }               // :197
  if (!buf || !idx_out) { return SJGPU_E_BADARG; }
  if (mode != SJGPU_REGULAR) {                    // :198-204
    len = sjgpu_trim_partial_utf8(buf, len);
    if (len == 0) { return E_UTF8; }
  }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  sjgpu_scan_result res;
  int rc = 0;
  if ((mode == SJGPU_STREAMING_PARTIAL || mode == SJGPU_STREAMING_FINAL) && len <= LA_WINDOW_MAX) { // a window of a registered stream?
    stream_extent e;
    if (find_stream(buf, len, &e) && e.len > len) {
      bool served = false;
      rc = stage1_from_span(ctx, e, buf, len, mode, idx_out, idx_words, n_io, next_io, &served);
      if (served || rc) { return rc; }
    }
  }
  if (ctx->small_docs && len <= DOCS_SINGLE_MAX) { // one launch, one wait, no staging copies on the device
    const void *out = nullptr;
    rc = small_single(ctx, 0, buf, len, &res, &out);
    if (rc) { return rc; }
    if (res.flags & SJGPU_F_IDX_OVERFLOW) { return E_UNEXPECTED; }
    if ((res.flags & SJGPU_F_UNCLOSED_STRING) && mode == SJGPU_REGULAR) { return E_UNCLOSED; }
    if (res.flags & SJGPU_F_UNESCAPED_CTRL) { return 14; }
    if (size_t(res.n) + 3 > idx_words) { return SJGPU_E_OVERFLOW; }
    std::memcpy(idx_out, out, (size_t(res.n) + 3) * sizeof(uint32_t));
    return sjgpu_stage1_finish_host(buf, len, mode, idx_out, res.n, res.flags, n_io, next_io);
  }
  rc = ensure_staging_in(ctx, len);
  if (rc) { return rc; }
  size_t idx_bytes = ctx->d_idx_words * sizeof(uint32_t);
  rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_idx), &idx_bytes, (grown(len) + 16) * sizeof(uint32_t));
  ctx->d_idx_words = idx_bytes / sizeof(uint32_t);
  if (rc) { return rc; }
  hipStream_t s = ctx->stream;
  const bool streamed = take_streamed_path(ctx, len);
  // Windows of a document stream (dom::DEFAULT_BATCH_SIZE = 1 MB) and other mid-size documents: the scan kernels write the
  // offsets straight into a page-locked block of the host (posted PCIe writes while they run), so that one wait delivers the
  // result AND the list -- instead of result, wait, list copy, wait.
  const bool direct = !streamed && ctx->small_docs && len <= DIRECT_HOST_MAX && ctx->device_finish != 2;
  if (streamed) { // large document: upload, scan and download overlap range by range; the offsets are on the host afterwards
    rc = run_streamed(ctx, 0, buf, len, idx_out, idx_words, &res);
    if (rc) { return rc; }
  } else if (direct) {
    rc = ensure_small(ctx, (len + 16) * sizeof(uint32_t));
    if (rc) { return rc; }
    uint32_t *h_idx = reinterpret_cast<uint32_t *>(ctx->h_small);
    SJ_TRY(ctx, hipMemcpyAsync(ctx->d_in, buf, len, hipMemcpyHostToDevice, s));
    for (int attempt = 0; attempt < 2; attempt++) { // a single-pass call that gives up is re-run on the split pipeline
      enqueue_stage1(ctx, use_fused(ctx, len, 0) && attempt == 0, ctx->d_in, len, h_idx, len + 3, s, nullptr);
      SJ_ENQUEUED(ctx);
      rc = fetch_result(ctx, s, &res); // the stream is in order: the list is complete when the result has arrived
      if (rc) { return rc; }
      if (!(res.flags & SJGPU_F_INTERNAL)) { break; }
    }
    if (res.flags & SJGPU_F_INTERNAL) { return E_UNEXPECTED; }
    if (res.flags & SJGPU_F_IDX_OVERFLOW) { return E_UNEXPECTED; }
    if ((res.flags & SJGPU_F_UNCLOSED_STRING) && mode == SJGPU_REGULAR) { return E_UNCLOSED; }
    if (res.flags & SJGPU_F_UNESCAPED_CTRL) { return 14; }
    if (size_t(res.n) + 3 > idx_words) { return SJGPU_E_OVERFLOW; }
    std::memcpy(idx_out, h_idx, (size_t(res.n) + 3) * sizeof(uint32_t));
    return sjgpu_stage1_finish_host(buf, len, mode, idx_out, res.n, res.flags, n_io, next_io);
  } else {
    SJ_TRY(ctx, hipMemcpyAsync(ctx->d_in, buf, len, hipMemcpyHostToDevice, s));
    for (int attempt = 0; attempt < 2; attempt++) { // a single-pass call that gives up is re-run on the split pipeline
      enqueue_stage1(ctx, use_fused(ctx, len, 0) && attempt == 0, ctx->d_in, len, ctx->d_idx, ctx->d_idx_words, s, nullptr);
      SJ_ENQUEUED(ctx);
      rc = fetch_result(ctx, s, &res);
      if (rc) { return rc; }
      if (!(res.flags & SJGPU_F_INTERNAL)) { break; }
    }
  }
  if (res.flags & SJGPU_F_INTERNAL) { return E_UNEXPECTED; }
  if (res.flags & SJGPU_F_IDX_OVERFLOW) { return E_UNEXPECTED; }
  // the two early exits of finish() need no index traffic (json_structural_indexer.h:255-263)
  if ((res.flags & SJGPU_F_UNCLOSED_STRING) && mode == SJGPU_REGULAR) { return E_UNCLOSED; }
  if (res.flags & SJGPU_F_UNESCAPED_CTRL) { return 14; }
  if (!streamed) {
    if (size_t(res.n) + 3 > idx_words) { return SJGPU_E_OVERFLOW; }
    if (mode != SJGPU_REGULAR && (ctx->device_finish == 2 || (ctx->device_finish == 1 && len >= DEVICE_FINISH_FROM))) {
      // streaming modes: find the last complete document / filter the list where it lies, then fetch only what is kept
      return finish_on_device_and_fetch(ctx, buf, len, mode, idx_out, idx_words, res, n_io, next_io);
    }
    SJ_TRY(ctx, hipMemcpyAsync(idx_out, ctx->d_idx, (size_t(res.n) + 3) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
  }
  return sjgpu_stage1_finish_host(buf, len, mode, idx_out, res.n, res.flags, n_io, next_io);
}

// ---- the list after the scan, for device-resident callers (sjgpu_finish.hip) --------------------------------------------------
int sjgpu_stage1_finish_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int mode, void *idx_dev, uint32_t n_raw, uint32_t flags,
                               void *stream, uint32_t *n_io, uint32_t *next_start_out) {
  if (!ctx || !buf_dev || !idx_dev || !n_io || mode < SJGPU_STREAMING_PARTIAL || mode > SJGPU_COMMA_DELIMITED_FINAL || len == 0 ||
      len > 0xFFFFFFFFull) {
    return SJGPU_E_BADARG;
  }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if ((flags & SJGPU_F_UNCLOSED_STRING) == 0 && (flags & SJGPU_F_UNESCAPED_CTRL)) { return 14; }
  if (flags & SJGPU_F_UNESCAPED_CTRL) { return 14; }
  hipStream_t s = pick(ctx, stream);
  uint32_t *idx = static_cast<uint32_t *>(idx_dev);
  finish_decision d;
  int rc = decide_on_device(ctx, static_cast<const uint8_t *>(buf_dev), len, mode, idx, n_raw, flags, s, &d);
  if (rc) { return rc; }
  *n_io = d.n_io;
  if (next_start_out) { *next_start_out = d.next_start; }
  if (d.need_first_word) {
    uint32_t first = 0;
    SJ_TRY(ctx, hipMemcpyAsync(&first, idx, sizeof first, hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
    if (first == 0) { return E_CAPACITY; }
    *n_io = 0;
    return E_EMPTY;
  }
  if (d.write_next_start) { SJ_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(idx + d.n_io), int(d.next_start), 1, s)); }
  if (d.shift_sentinel) {
    SJ_TRY(ctx, hipMemcpyAsync(idx + d.n_io + 1, idx + d.n_io, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    SJ_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(idx + d.n_io), int(uint32_t(len)), 1, s));
  }
  return d.error;
}

int sjgpu_depth_scan_device(sjgpu_ctx *ctx, const void *buf_dev, const void *idx_dev, uint32_t n, void *depth_dev, void *stream) {
  if (!ctx || !buf_dev || !idx_dev || !depth_dev) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_tmp(ctx, depth_scan_scratch_bytes(n));
  if (rc) { return rc; }
  launch_depth_scan(static_cast<const uint8_t *>(buf_dev), static_cast<const uint32_t *>(idx_dev), n, static_cast<int32_t *>(depth_dev), ctx->d_tmp,
                    pick(ctx, stream));
  SJ_TRY(ctx, hipGetLastError());
  return 0;
}

int sjgpu_depth_scan_tokens_device(sjgpu_ctx *ctx, const void *tok_dev, uint32_t n, void *depth_dev, void *stream) {
  if (!ctx || !tok_dev || !depth_dev) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_tmp(ctx, depth_scan_scratch_bytes(n));
  if (rc) { return rc; }
  launch_depth_scan(nullptr, nullptr, n, static_cast<int32_t *>(depth_dev), ctx->d_tmp, pick(ctx, stream), static_cast<const uint8_t *>(tok_dev));
  SJ_TRY(ctx, hipGetLastError());
  return 0;
}

// ---- the strings of a document, unescaped (sjgpu_strings.hip) ----------------------------------------------------------------
int sjgpu_parse_strings_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, int allow_replacement,
                               void *string_buf_dev, size_t string_buf_bytes, void *offsets_dev, void *stream, uint64_t *bytes_out,
                               uint32_t *strings_out, uint32_t *first_bad_out) {
  if (!ctx || !buf_dev || !idx_dev || !string_buf_dev) { return SJGPU_E_BADARG; }
  if ((reinterpret_cast<uintptr_t>(buf_dev) & 3u) || (reinterpret_cast<uintptr_t>(offsets_dev) & 3u)) { return SJGPU_E_BADARG; }
  if (len > 2400000000ull || n >= 0xFFFFFFF0u) { return E_CAPACITY; } // record offsets are 32 bits: 5 (len + 1) / 3 bytes of records at most
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  // [result: 32 B, padded to 256][scratch of the string pass][offsets when the caller keeps none]
  const size_t scratch_at = 256, scratch = strings_scratch_bytes(n, len), offs_at = scratch_at + scratch;
  int rc = ensure_tmp(ctx, offs_at + (offsets_dev ? 0 : (size_t(n) + 1) * sizeof(uint32_t)));
  if (rc) { return rc; }
  uint8_t *tmp = static_cast<uint8_t *>(static_cast<void *>(ctx->d_tmp));
  strings_result_dev *res = reinterpret_cast<strings_result_dev *>(tmp);
  uint32_t *offsets = offsets_dev ? static_cast<uint32_t *>(offsets_dev) : reinterpret_cast<uint32_t *>(tmp + offs_at);
  hipStream_t s = pick(ctx, stream);
  // optimistic like stage 2: the stream compaction alone; a document it declines (path 2, nothing written) is run again through the per-string kernels
  strings_result_dev h;
  for (int roads = STRINGS_STREAM_ONLY;; roads = STRINGS_WALK_ONLY) {
    launch_parse_strings(static_cast<const uint8_t *>(buf_dev), len, static_cast<const uint32_t *>(idx_dev), n, allow_replacement != 0,
                         static_cast<uint8_t *>(string_buf_dev), string_buf_bytes, offsets, res, tmp + scratch_at, s, nullptr, roads);
    SJ_TRY(ctx, hipGetLastError());
    SJ_TRY(ctx, hipMemcpyAsync(&h, res, sizeof(h), hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
    if (roads == STRINGS_WALK_ONLY || h.path != 2 || h.overflow) { break; }
  }
  ctx->last_string_path = h.path;
  if (bytes_out) { *bytes_out = h.bytes; }
  if (strings_out) { *strings_out = h.strings; }
  if (first_bad_out) { *first_bad_out = h.first_bad; }
  if (h.overflow) { return SJGPU_E_OVERFLOW; }
  return h.first_bad != 0xFFFFFFFFu ? 5 /* STRING_ERROR */ : 0;
}

int sjgpu_debug_string_path(const sjgpu_ctx *ctx) { return ctx ? int(ctx->last_string_path) : SJGPU_E_BADARG; }

// ---- On-Demand's raw key comparison (sjgpu_strings.hip) ---------------------------------------------------------------------------------
int sjgpu_match_keys_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, const uint8_t *names, const uint32_t *name_lens,
                            uint32_t K, void *match_dev, void *stream, uint32_t *matches_out) {
  if (!ctx || !buf_dev || !idx_dev || !match_dev || !names || !name_lens || K == 0 || K > 256u || (reinterpret_cast<uintptr_t>(match_dev) & 3u)) { return SJGPU_E_BADARG; }
  size_t total = 0;
  for (uint32_t k = 0; k < K; k++) { total += name_lens[k]; }
  if (total > (size_t(64) << 10)) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  const size_t block = size_t(K) * sizeof(uint32_t) + total;
  int rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_stage2), &ctx->d_stage2_bytes, 256 + block);
  if (rc) { return rc; }
  std::vector<uint8_t> host(block);
  std::memcpy(host.data(), name_lens, size_t(K) * sizeof(uint32_t));
  std::memcpy(host.data() + size_t(K) * sizeof(uint32_t), names, total);
  hipStream_t s = pick(ctx, stream);
  uint32_t *d_matches = reinterpret_cast<uint32_t *>(ctx->d_stage2);
  uint8_t *d_block = ctx->d_stage2 + 256;
  SJ_TRY(ctx, hipMemcpyAsync(d_block, host.data(), block, hipMemcpyHostToDevice, s));
  SJ_TRY(ctx, hipStreamSynchronize(s)); // `host` leaves scope with this call
  launch_match_keys(static_cast<const uint8_t *>(buf_dev), len, static_cast<const uint32_t *>(idx_dev), n, d_block, K, static_cast<uint32_t *>(match_dev), d_matches, s);
  SJ_TRY(ctx, hipGetLastError());
  uint32_t m = 0;
  SJ_TRY(ctx, hipMemcpyAsync(&m, d_matches, sizeof m, hipMemcpyDeviceToHost, s));
  SJ_TRY(ctx, hipStreamSynchronize(s));
  if (matches_out) { *matches_out = m; }
  return 0;
}

// ---- stage 2: the tape (sjgpu_tape.hip) -------------------------------------------------------------------------------------------------
int sjgpu_stage2_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, uint32_t max_depth, void *tape_dev,
                        size_t tape_cap_words, void *string_buf_dev, size_t string_buf_bytes, void *stream, uint64_t *tape_words_out,
                        uint64_t *string_bytes_out) {
  return sjgpu_stage2_tokens_device(ctx, buf_dev, len, idx_dev, n, nullptr, max_depth, tape_dev, tape_cap_words, string_buf_dev, string_buf_bytes, stream, tape_words_out,
                                    string_bytes_out);
}

int sjgpu_stage2_tokens_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, const void *tok_dev, uint32_t max_depth,
                               void *tape_dev, size_t tape_cap_words, void *string_buf_dev, size_t string_buf_bytes, void *stream, uint64_t *tape_words_out,
                               uint64_t *string_bytes_out) {
  if (!ctx || !buf_dev || !idx_dev || !tape_dev || !string_buf_dev || max_depth == 0 || max_depth > 4095u) { return SJGPU_E_BADARG; }
  // buf_dev: 16-byte aligned like every device entry point (the string stream's chunk loads are 16-byte loads of an aligned buffer)
  if ((reinterpret_cast<uintptr_t>(buf_dev) & 15u) || (reinterpret_cast<uintptr_t>(tape_dev) & 7u) || (reinterpret_cast<uintptr_t>(idx_dev) & 3u)) { return SJGPU_E_BADARG; }
  if (tape_words_out) { *tape_words_out = 0; }
  if (string_bytes_out) { *string_bytes_out = 0; }
  if (n == 0) { return E_EMPTY; } // walk_document: at_eof() (json_iterator.h:126)
  if (len > 2400000000ull || n >= 0xFFFFFFF0u) { return E_CAPACITY; } // the string pass's 32-bit record offsets
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  // [strings result 32 B, padded to 256][scratch of the string pass][string offsets, n + 1 words][tape workspace]
  const size_t scratch_at = 256, scratch = strings_scratch_bytes(n, len), offs_at = scratch_at + scratch;
  const size_t tape_at = (offs_at + (size_t(n) + 1) * sizeof(uint32_t) + 255) & ~size_t(255);
  int rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_stage2), &ctx->d_stage2_bytes, tape_at + tape_workspace_bytes(n, len));
  if (rc) { return rc; }
  uint8_t *ws = ctx->d_stage2;
  strings_result_dev *sres = reinterpret_cast<strings_result_dev *>(ws);
  uint32_t *offsets = reinterpret_cast<uint32_t *>(ws + offs_at);
  hipStream_t s = pick(ctx, stream);
  // Optimistic: the string buffer by the stream compaction alone, the sort in one pass -- what nearly every document needs.  A document the stream declines
  // (a string the reference rejects, quotes glued to scalars, a look-back that settles nothing) or one nested 64 deep and more says so in its results and is
  // run again with the per-string kernels / the sort's second pass enqueued: ten launches that nearly always did nothing are gone from the common call.
  strings_result_dev hs;
  tape_result_dev ht;
  int roads = STRINGS_STREAM_ONLY;
  bool deep = false;
  for (;;) {
    const int *string_tokens = launch_tape_front(static_cast<const uint8_t *>(buf_dev), len, static_cast<const uint32_t *>(idx_dev), n, max_depth, ws + tape_at, s,
                                                 static_cast<const uint8_t *>(tok_dev));
    const strings_handoff strs = launch_parse_strings(static_cast<const uint8_t *>(buf_dev), len, static_cast<const uint32_t *>(idx_dev), n, false,
                                                      static_cast<uint8_t *>(string_buf_dev), string_buf_bytes, offsets, sres, ws + scratch_at, s, string_tokens, roads);
    launch_tape(static_cast<const uint8_t *>(buf_dev), len, static_cast<const uint32_t *>(idx_dev), n, max_depth, offsets, strs, static_cast<uint8_t *>(string_buf_dev),
                static_cast<uint64_t *>(tape_dev), tape_cap_words, ws + tape_at, s, deep);
    SJ_TRY(ctx, hipGetLastError());
    // (into page-locked memory: a copy into a variable on the stack goes through the runtime's staging buffer and waits for it, twice per call)
    uint8_t *const pinned = reinterpret_cast<uint8_t *>(ctx->h_result);
    static_assert(sizeof(strings_result_dev) <= 64 && sizeof(tape_result_dev) <= 64, "the pinned block's slots");
    SJ_TRY(ctx, hipMemcpyAsync(pinned + 64, sres, sizeof(hs), hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipMemcpyAsync(pinned + 128, ws + tape_at, sizeof(ht), hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
    memcpy(&hs, pinned + 64, sizeof(hs));
    memcpy(&ht, pinned + 128, sizeof(ht));
    bool again = false;
    if (roads == STRINGS_STREAM_ONLY && hs.path == 2 && !hs.overflow) { roads = STRINGS_WALK_ONLY; again = true; }
    if (!deep && ht.max_level >= TAPE_ONE_PASS_LEVELS) { deep = true; again = true; }
    if (!again) { break; }
  }
  ctx->last_string_path = hs.path;
  // the first offender in list order decides; a string's content ranks behind its own position in the grammar (sj_tape_rules.h)
  uint64_t key = ht.error_key;
  if (hs.first_bad != 0xFFFFFFFFu) {
    const uint64_t sk = (uint64_t(hs.first_bad) << 8) | (2u << 4) | 5u; // STRING_ERROR
    if (sk < key) { key = sk; }
  }
  if (key != ~uint64_t(0)) { return int(key & 0xFu); }
  if (hs.overflow || ht.overflow) { return SJGPU_E_OVERFLOW; }
  if (tape_words_out) { *tape_words_out = ht.tape_words; }
  if (string_bytes_out) { *string_bytes_out = hs.bytes; }
  return 0;
}

int sjgpu_parse(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, uint32_t max_depth, uint64_t *tape_out, size_t tape_cap_words, uint8_t *string_buf_out,
                size_t string_buf_bytes, uint64_t *tape_words_out, uint64_t *string_bytes_out) {
  if (!ctx || !tape_out || !string_buf_out) { return SJGPU_E_BADARG; }
  if (tape_words_out) { *tape_words_out = 0; }
  if (string_bytes_out) { *string_bytes_out = 0; }
  if (len > ctx->capacity) { return E_CAPACITY; }
  if (len == 0) { return E_EMPTY; }
  if (!buf) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_staging_in(ctx, len);
  if (rc) { return rc; }
  size_t idx_bytes = ctx->d_idx_words * sizeof(uint32_t);
  rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_idx), &idx_bytes, (grown(len) + 16) * sizeof(uint32_t));
  ctx->d_idx_words = idx_bytes / sizeof(uint32_t);
  if (rc) { return rc; }
  const size_t tape_words_cap = len + 8, str_cap = 5 * (len / 3) + 256, str_at = tape_words_cap * sizeof(uint64_t);
  rc = grow(ctx, reinterpret_cast<void **>(&ctx->d_doc), &ctx->d_doc_bytes, str_at + str_cap);
  if (rc) { return rc; }
  hipStream_t s = ctx->stream;
  SJ_TRY(ctx, hipMemcpyAsync(ctx->d_in, buf, len, hipMemcpyHostToDevice, s));
  sjgpu_scan_result res{0, 0, 0};
  for (int attempt = 0; attempt < 2; attempt++) { // a single-pass scan that gives up is re-run on the split pipeline
    enqueue_stage1(ctx, use_fused(ctx, len, 0) && attempt == 0, ctx->d_in, len, ctx->d_idx, ctx->d_idx_words, s, nullptr);
    SJ_ENQUEUED(ctx);
    rc = fetch_result(ctx, s, &res);
    if (rc) { return rc; }
    if (!(res.flags & SJGPU_F_INTERNAL)) { break; }
  }
  if (res.flags & (SJGPU_F_INTERNAL | SJGPU_F_IDX_OVERFLOW)) { return E_UNEXPECTED; }
  const int e1 = sjgpu_stage1_error_from_flags(res.n, res.flags);
  if (e1) { return e1; }
  uint64_t tw = 0, sb = 0;
  rc = sjgpu_stage2_device(ctx, ctx->d_in, len, ctx->d_idx, res.n, max_depth, ctx->d_doc, tape_words_cap, ctx->d_doc + str_at, str_cap, s, &tw, &sb);
  if (rc) { return rc; }
  if (tw > tape_cap_words || sb > string_buf_bytes) { return SJGPU_E_OVERFLOW; }
  SJ_TRY(ctx, hipMemcpyAsync(tape_out, ctx->d_doc, tw * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  if (sb) { SJ_TRY(ctx, hipMemcpyAsync(string_buf_out, ctx->d_doc + str_at, sb, hipMemcpyDeviceToHost, s)); }
  SJ_TRY(ctx, hipStreamSynchronize(s));
  if (tape_words_out) { *tape_words_out = tw; }
  if (string_bytes_out) { *string_bytes_out = sb; }
  return 0;
}

// ---- many small documents per launch (sjgpu_small.hip) -----------------------------------------------------------------------
int sjgpu_stage1_many(sjgpu_ctx *ctx, sjgpu_doc *docs, size_t count) {
  if (!ctx || (count && !docs) || count > 0xFFFFFFu) { return SJGPU_E_BADARG; }
  if (count == 0) { return 0; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  // block layout: [results: 16 B each][descriptors: 24 B each][inputs, each rounded to 64 B + 64 B of slack][outputs, 16-byte aligned]
  size_t in_bytes = 0, out_words = 0;
  for (size_t i = 0; i < count; i++) {
    docs[i].n = 0;
    docs[i].error = 0;
    if (!docs[i].buf || !docs[i].idx_out || docs[i].len > 0xFFFFFFFFull) { return SJGPU_E_BADARG; }
    if (docs[i].len == 0) { docs[i].error = E_EMPTY; continue; }
    if (docs[i].len > ctx->capacity) { docs[i].error = E_CAPACITY; continue; }
    if (docs[i].idx_words < docs[i].len + 3) { return SJGPU_E_OVERFLOW; }
    in_bytes += round_up(docs[i].len, 64) + 64;
    out_words += round_up(docs[i].len + 3, 4);
  }
  const size_t res_at = 0, desc_at = round_up(count * sizeof(scan_result_dev), 64), in_at = desc_at + round_up(count * sizeof(doc_desc), 64);
  const size_t out_at = in_at + in_bytes, total = out_at + out_words * sizeof(uint32_t) + 64;
  int rc = ensure_small(ctx, total);
  if (rc) { return rc; }
  scan_result_dev *results = reinterpret_cast<scan_result_dev *>(ctx->h_small + res_at);
  doc_desc *descs = reinterpret_cast<doc_desc *>(ctx->h_small + desc_at);
  size_t in_off = 0, out_off = 0;
  uint32_t live = 0;
  for (size_t i = 0; i < count; i++) {
    if (docs[i].error) { continue; }
    std::memcpy(ctx->h_small + in_at + in_off, docs[i].buf, docs[i].len);
    descs[live] = doc_desc{in_off, out_off, uint32_t(docs[i].len), 0};
    in_off += round_up(docs[i].len, 64) + 64;
    out_off += round_up(docs[i].len + 3, 4);
    live++;
  }
  if (live == 0) { return 0; }
  hipStream_t s = ctx->stream;
  // Small batches are read and written by the kernel across PCIe (no copies at all); larger ones are staged through
  // HBM with ONE copy in and ONE copy out, so that the workgroups do not all wait on the link at once.
  const bool zero_copy = total <= (size_t(2) << 20);
  uint8_t *base = ctx->h_small;
  if (!zero_copy) {
    rc = ensure_tmp(ctx, total);
    if (rc) { return rc; }
    SJ_TRY(ctx, hipMemcpyAsync(ctx->d_tmp, ctx->h_small, out_at, hipMemcpyHostToDevice, s));
    base = ctx->d_tmp;
  }
  ctx->pending_scan_bytes = 0;
  ctx->last_kernel = "k_docs<0>";
  launch_docs(0, base + in_at, reinterpret_cast<const doc_desc *>(base + desc_at), doc_desc{0, 0, 0, 0}, live, base + out_at,
              reinterpret_cast<scan_result_dev *>(base + res_at), s);
  SJ_TRY(ctx, hipGetLastError());
  if (!zero_copy) {
    SJ_TRY(ctx, hipMemcpyAsync(ctx->h_small + res_at, ctx->d_tmp + res_at, desc_at, hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipMemcpyAsync(ctx->h_small + out_at, ctx->d_tmp + out_at, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  SJ_TRY(ctx, hipStreamSynchronize(s));
  const uint32_t *out = reinterpret_cast<const uint32_t *>(ctx->h_small + out_at);
  live = 0;
  for (size_t i = 0; i < count; i++) {
    if (docs[i].error) { continue; }
    const scan_result_dev r = results[live];
    const doc_desc d = descs[live];
    live++;
    docs[i].error = sjgpu_stage1_error_from_flags(r.n, r.flags);
    if (r.flags & SJGPU_F_IDX_OVERFLOW) { docs[i].error = E_UNEXPECTED; continue; }
    if (docs[i].error == E_UNCLOSED || docs[i].error == 14) { continue; } // the reference leaves n and the list alone on these two
    docs[i].n = r.n;
    std::memcpy(docs[i].idx_out, out + d.out_off, (size_t(r.n) + 3) * sizeof(uint32_t));
  }
  return 0;
}

// No length limit (include/simdjson/implementation.h:116 has none): inputs beyond 4 GiB - 1 go piece by piece, cut where
// only the in-string bit crosses (sjgpu_clean_cut), exactly like the shards of a document spread over several GPUs.
int sjgpu_minify(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len) {
  if (!ctx || !dst_len) { return SJGPU_E_BADARG; }
  *dst_len = 0;
  if (len == 0) { return 0; }
  if (!buf || !dst) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  const size_t piece = piece_bytes();
  sjgpu_scan_result res{0, 0, 0};
  if (len <= piece) {
    const int rc = minify_piece(ctx, buf, len, false, 0, dst, &res);
    if (rc) { return rc; }
    if (res.flags & SJGPU_F_UNCLOSED_STRING) { return E_UNCLOSED; }
    *dst_len = res.out_len;
    return 0;
  }
  size_t at = 0, out = 0;
  uint32_t in_string = 0;
  while (at < len) {
    size_t cut = (len - at <= piece) ? len : sjgpu_clean_cut(buf, len, at + piece);
    if (cut - at > 0xFFFFFFF0ull) { return E_CAPACITY; } // no clean byte within 4 GiB: not JSON anyone could parse
    const int rc = minify_piece(ctx, buf + at, cut - at, true, in_string, dst + out, &res);
    if (rc) { return rc; }
    out += res.out_len;
    in_string = res.flags & SJGPU_F_UNCLOSED_STRING;
    at = cut;
  }
  if (in_string) { return E_UNCLOSED; } // json_minifier.h:42-47: dst_len stays 0
  *dst_len = out;
  return 0;
}

// No length limit either (include/simdjson/implementation.h:128): pieces are cut in front of a character's first byte, so
// each piece is well-formed or not by itself.
int sjgpu_validate_utf8_pieces(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, size_t piece, int *ok) {
  if (!ctx || !ok) { return SJGPU_E_BADARG; }
  *ok = 1;
  if (len == 0) { return 0; }
  if (!buf) { return SJGPU_E_BADARG; }
  SJ_TRY(ctx, hipSetDevice(ctx->device));
  if (piece == 0) { piece = piece_bytes(); }
  if (piece < 64) { piece = 64; }
  if (piece > (size_t(2048) << 20)) { piece = size_t(2048) << 20; } // what piece_bytes() allows: one scan addresses 32 bits
  size_t at = 0;
  while (at < len) {
    size_t cut = len;
    if (len - at > piece) {
      cut = at + piece;
      int back = 0;
      while (back < 4 && (buf[cut] & 0xC0u) == 0x80u) { cut--; back++; } // continuation bytes belong to the piece in front
      if (back == 4) { *ok = 0; return 0; }                                // four in a row: ill-formed whatever precedes them
    }
    const int rc = validate_piece(ctx, buf + at, cut - at, ok);
    if (rc || !*ok) { return rc; }
    at = cut;
  }
  return 0;
}
int sjgpu_validate_utf8(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int *ok) { return sjgpu_validate_utf8_pieces(ctx, buf, len, 0, ok); }

} // extern "C"
