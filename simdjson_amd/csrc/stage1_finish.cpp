// simdjson_amd/csrc/stage1_finish.cpp -- host post-pass of stage 1 (product code, pure C++).
//
// The GPU produces the raw ascending list of structural offsets plus three flags; everything the
// reference's finish() does after that is O(1) for regular mode and O(last document) for the
// streaming modes, touches buf[] at data-dependent positions and walks BACKWARDS over the list, so
// it stays on the host.  Behaviour restated from
// /root/reference/src/generic/stage1/json_structural_indexer.h:249-397 and
// /root/reference/src/generic/stage1/find_next_document_index.h:39-369.
#include "sjgpu.h"

namespace {

enum : int { OK = 0, E_CAPACITY = 1, E_UTF8 = 11, E_EMPTY = 13, E_UNESCAPED = 14, E_UNCLOSED = 15, E_UNEXPECTED = 24 };
constexpr uint32_t TOO_LARGE = 0xFFFFFFFFu; // "a document started but did not fit" (find_next_document_index.h:105)

struct view {
  const uint8_t *buf;
  uint32_t *idx;
  uint8_t at(uint32_t i) const { return buf[idx[i]]; }
};

inline bool opens(uint8_t c) { return c == '{' || c == '['; }
inline bool closes(uint8_t c) { return c == '}' || c == ']'; }
inline bool separates(uint8_t c) { return c == ':' || c == ','; }

// Number of leading structurals that belong to complete documents (find_next_document_index.h:39-98):
// scan backwards for a value that directly follows another value; the tail after that boundary is
// complete iff its brackets balance.
uint32_t complete_prefix(const view &v, uint32_t n) {
  if (n == 0) { return 0; }
  int depth_obj = 0, depth_arr = 0;
  auto count = [&](uint8_t c) {
    depth_obj += (c == '{') - (c == '}');
    depth_arr += (c == '[') - (c == ']');
  };
  for (uint32_t i = n - 1; i > 0; i--) {
    const uint8_t cur = v.at(i);
    if (separates(cur)) { continue; }
    count(cur);
    if (closes(cur)) { continue; }
    const uint8_t before = v.at(i - 1);
    if (opens(before) || separates(before)) { continue; }
    return (depth_obj == 0 && depth_arr == 0) ? n : i;
  }
  count(v.at(0));
  return (depth_obj == 0 && depth_arr == 0) ? n : 0;
}

struct filtered {
  uint32_t keep;       // structurals to keep, 0, or TOO_LARGE
  uint32_t next_start; // where the next batch begins
};

// RFC 7464 record separators (find_next_document_index.h:126-267): drop RS entries, re-insert the
// scalar starts the scanner glued to an RS, then cut at the last RS for partial batches.
filtered filter_record_separators(const view &v, uint32_t &n, size_t len, bool final_batch) {
  filtered r{0, uint32_t(len)};
  if (n == 0) { return r; }
  uint32_t out = 0, rs_seen = 0, last_rs = 0;
  for (uint32_t in = 0; in < n; in++) {
    const uint32_t pos = v.idx[in];
    if (v.buf[pos] != 0x1E) { v.idx[out++] = pos; continue; }
    last_rs = pos;
    rs_seen++;
    uint32_t value = pos + 1;
    for (; value < len; value++) {
      const uint8_t c = v.buf[value];
      if (c == 0x1E) { last_rs = value; rs_seen++; }
      else if (!(c == ' ' || c == '\t' || c == '\n' || c == '\r')) { break; }
    }
    while (in + 1 < n && v.idx[in + 1] < value) { in++; }
    if (value < len) {
      const uint8_t c = v.buf[value];
      const bool is_operator = opens(c) || closes(c) || separates(c);
      const bool scanner_has_it = in + 1 < n && v.idx[in + 1] == value;
      if (!is_operator && !scanner_has_it) { v.idx[out++] = value; }
    }
  }
  n = out;
  if (n == 0) { return r; }
  if (rs_seen == 0) { r.keep = final_batch ? complete_prefix(v, n) : 0; return r; }
  if (final_batch) { r.keep = n; return r; }
  r.next_start = last_rs;
  if (rs_seen < 2) { r.keep = TOO_LARGE; return r; }
  uint32_t k = n;
  while (k > 0 && v.idx[k - 1] >= last_rs) { k--; }
  r.keep = k;
  return r;
}

// Comma-delimited documents (find_next_document_index.h:288-369): drop depth-0 commas, cut after
// the last one for partial batches, then ordinary boundary detection.
filtered filter_root_commas(const view &v, uint32_t &n, size_t len, bool final_batch) {
  filtered r{0, uint32_t(len)};
  if (n == 0) { return r; }
  int depth = 0;
  uint32_t out = 0, commas = 0, last_comma = 0;
  for (uint32_t in = 0; in < n; in++) {
    const uint32_t pos = v.idx[in];
    const uint8_t c = v.buf[pos];
    if (opens(c)) { depth++; }
    else if (closes(c)) { depth--; }
    else if (c == ',' && depth == 0) { last_comma = pos; commas++; continue; }
    v.idx[out++] = pos;
  }
  n = out;
  if (n == 0) { return r; }
  if (final_batch) { r.keep = complete_prefix(v, n); return r; }
  if (commas == 0) { r.keep = TOO_LARGE; return r; }
  r.next_start = last_comma + 1;
  uint32_t k = n;
  while (k > 0 && v.idx[k - 1] >= last_comma) { k--; }
  if (k == 0) { return r; }
  n = k;
  r.keep = complete_prefix(v, k);
  return r;
}

} // namespace

extern "C" size_t sjgpu_trim_partial_utf8(const uint8_t *buf, size_t len) {
  // a trailing lead byte whose sequence cannot be complete within the window (…indexer.h:156-174)
  if (len >= 1 && buf[len - 1] >= 0xC0) { return len - 1; }
  if (len >= 2 && buf[len - 2] >= 0xE0) { return len - 2; }
  if (len >= 3 && buf[len - 3] >= 0xF0) { return len - 3; }
  return len;
}

// A cut is "clean" when the byte in front of it is ASCII whitespace or one of , : [ ] { } : not a backslash
// (nothing is escaped across the cut), neither a scalar byte nor a quote (prev_scalar = 0), ASCII (no UTF-8
// sequence is open).  Only the in-string bit survives such a cut.  Whitespace counts inside strings too.
extern "C" size_t sjgpu_clean_cut(const uint8_t *buf, size_t len, size_t target) {
  if (target == 0) { return 0; }
  for (size_t c = target; c < len; c++) {
    const uint8_t b = buf[c - 1];
    if (b == ' ' || b == '\t' || b == '\n' || b == '\r' || opens(b) || closes(b) || separates(b)) { return c; }
  }
  return len;
}

extern "C" int sjgpu_stage1_error_from_flags(uint32_t n, uint32_t flags) {
  if (flags & SJGPU_F_UNCLOSED_STRING) { return E_UNCLOSED; }
  if (flags & SJGPU_F_UNESCAPED_CTRL) { return E_UNESCAPED; }
  if (n == 0) { return E_EMPTY; }
  return (flags & SJGPU_F_UTF8_ERROR) ? E_UTF8 : OK;
}

extern "C" int sjgpu_stage1_finish_host(const uint8_t *buf, size_t len, int mode, uint32_t *idx, uint32_t n_raw,
                                        uint32_t flags, uint32_t *n_io, uint32_t *next_io) {
  const bool streaming = mode != SJGPU_REGULAR;
  const bool unclosed = (flags & SJGPU_F_UNCLOSED_STRING) != 0;
  if (unclosed && !streaming) { return E_UNCLOSED; } // streaming tolerates it (…indexer.h:255-259)
  if (flags & SJGPU_F_UNESCAPED_CTRL) { return E_UNESCAPED; }
  uint32_t n = n_raw;
  *n_io = n;
  idx[n] = uint32_t(len); // the three sentinels (…indexer.h:284-287)
  idx[n + 1] = uint32_t(len);
  idx[n + 2] = 0;
  if (next_io) { *next_io = 0; } // parser.next_structural_index = 0 (…indexer.h:287)
  if (n == 0) { return E_EMPTY; }
  if (idx[n - 1] > len) { return E_UNEXPECTED; }
  const view v{buf, idx};
  const bool partial = mode == SJGPU_STREAMING_PARTIAL || mode == SJGPU_JSON_SEQUENCE_PARTIAL ||
                       mode == SJGPU_COMMA_DELIMITED_PARTIAL;
  if (streaming && unclosed) { // the last structural is the dangling opening quote
    *n_io = --n;
    if (partial && n == 0) { return E_CAPACITY; }
  }
  switch (mode) {
  case SJGPU_STREAMING_PARTIAL: {
    const uint32_t keep = complete_prefix(v, n);
    if (keep == 0 && n > 0) {
      if (idx[0] == 0) { return E_CAPACITY; } // one document fills the whole window
      *n_io = 0;
      return E_EMPTY; // leading whitespace only; the document may fit the next window
    }
    *n_io = keep;
    break;
  }
  case SJGPU_STREAMING_FINAL: {
    const uint32_t keep = complete_prefix(v, n);
    *n_io = keep;
    idx[keep + 1] = idx[keep]; // lets the stream compute truncated_bytes (…indexer.h:334-337)
    idx[keep] = uint32_t(len);
    if (keep == 0) { return E_EMPTY; }
    break;
  }
  case SJGPU_JSON_SEQUENCE_PARTIAL:
  case SJGPU_COMMA_DELIMITED_PARTIAL: {
    const filtered f = (mode == SJGPU_JSON_SEQUENCE_PARTIAL) ? filter_record_separators(v, n, len, false)
                                                             : filter_root_commas(v, n, len, false);
    *n_io = n;
    if (f.keep == TOO_LARGE) { return E_CAPACITY; }
    if (f.keep == 0) { *n_io = 0; return E_EMPTY; }
    *n_io = f.keep;
    idx[f.keep] = f.next_start;
    break;
  }
  case SJGPU_JSON_SEQUENCE_FINAL:
  case SJGPU_COMMA_DELIMITED_FINAL: {
    const filtered f = (mode == SJGPU_JSON_SEQUENCE_FINAL) ? filter_record_separators(v, n, len, true)
                                                           : filter_root_commas(v, n, len, true);
    *n_io = f.keep;
    idx[f.keep + 1] = idx[f.keep];
    idx[f.keep] = uint32_t(len);
    if (f.keep == 0) { return E_EMPTY; }
    break;
  }
  default:
    break;
  }
  return (flags & SJGPU_F_UTF8_ERROR) ? E_UTF8 : OK;
}
