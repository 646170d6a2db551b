// simdjson_amd/csrc/sjgpu_capi_stage2.hip -- the C-ABI of include/sjgpu.h, what follows the structural list on the device: the strings of a document, On-Demand's raw
// key comparison, stage 2 (the DOM tape) and sjgpu_parse.  Shared with the other units: sjgpu_ctx.h.
#include "sjgpu_ctx.h"

extern "C" {

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
  // the string pass's result lives in the tape workspace's first slot, 64 bytes behind the tape's own result (the slot has 256; k_tape_init clears it, k_strs_init
  // -- later in the stream -- fills it): both results come back in ONE copy (each copy is a 5 us launch of the runtime's copy kernel)
  strings_result_dev *sres = reinterpret_cast<strings_result_dev *>(ws + tape_at + 64);
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
    SJ_TRY(ctx, hipMemcpyAsync(pinned + 64, ws + tape_at, 128, hipMemcpyDeviceToHost, s));
    SJ_TRY(ctx, hipStreamSynchronize(s));
    memcpy(&ht, pinned + 64, sizeof(ht));
    memcpy(&hs, pinned + 128, sizeof(hs));
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

} // extern "C"
