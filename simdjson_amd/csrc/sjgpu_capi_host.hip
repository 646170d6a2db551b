// simdjson_amd/csrc/sjgpu_capi_host.hip -- the C-ABI of include/sjgpu.h, host buffers: the plug-in path (H2D, scan, D2H, host finish), the overlapped path of large
// documents, windows of a registered stream and the stream registry, the list after the scan for device-resident callers (finish, depth scan), many small
// documents per launch, minify and validate_utf8 of host buffers.  Shared with the other units: sjgpu_ctx.h.
#include "sjgpu_ctx.h"

// ---- host-buffer entry points (the plug-in path: H2D, scan, D2H, host finish) ---------------------------

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
