/* oracle/sj_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sj_oracle.h).
 *
 * Scalar restatement of the reference's stage-1 path.  Everything is position-wise (one byte at a
 * time, three 1-bit carries), which is equivalent to the reference's 64-byte-block formulation
 * because every block-level definition is a prefix function.
 * Citations are to /root/reference.
 */
#include "sj_oracle.h"

#include <string.h>
#include <time.h>

/* ---- character classes: src/haswell.cpp:43-94 (x86 table semantics, incl. the 0x0C / 0x1A quirk) */
static int is_ws(uint8_t b) { return b == 0x20 || b == 0x09 || b == 0x0A || b == 0x0D; }
static int is_op(uint8_t b) {
  if (b >= 0x80) { return 0; } /* pshufb zeroes the lookup for bytes with the top bit set */
  uint8_t c = (uint8_t)(b | 0x20); /* "curlified" compare: src/haswell.cpp:64-72 */
  return c == 0x2C || c == 0x3A || c == 0x7B || c == 0x7D;
}

/* ---- UTF-8 (well-formedness per RFC 3629; what src/generic/stage1/utf8_lookup4_algorithm.h:16-202
 * decides, including the truncated-sequence-at-EOF rule :164-171,198-200) */
int sjo_validate_utf8(const uint8_t *s, size_t len) {
  size_t i = 0;
  while (i < len) {
    uint8_t b = s[i];
    if (b < 0x80) { i++; continue; }
    size_t need;
    uint32_t cp, min;
    if ((b & 0xE0) == 0xC0) { need = 1; cp = b & 0x1Fu; min = 0x80; }
    else if ((b & 0xF0) == 0xE0) { need = 2; cp = b & 0x0Fu; min = 0x800; }
    else if ((b & 0xF8) == 0xF0) { need = 3; cp = b & 0x07u; min = 0x10000; }
    else { return 0; } /* stray continuation byte or 0xF8..0xFF */
    if (len - i <= need) { return 0; } /* sequence runs past the end */
    for (size_t k = 1; k <= need; k++) {
      uint8_t c = s[i + k];
      if ((c & 0xC0) != 0x80) { return 0; }
      cp = (cp << 6) | (c & 0x3Fu);
    }
    if (cp < min) { return 0; }                      /* overlong */
    if (cp >= 0xD800 && cp <= 0xDFFF) { return 0; }  /* surrogate */
    if (cp > 0x10FFFF) { return 0; }                 /* beyond Unicode */
    i += need + 1;
  }
  return 1;
}

/* ---- raw scan ------------------------------------------------------------------------------
 * escaped[i]: src/generic/stage1/json_escape_scanner.h:50-71 (a char is escaped iff preceded by an
 *             unescaped backslash; resolved over the whole buffer, not only inside strings)
 * quote/in_string/string_tail: src/generic/stage1/json_string_scanner.h:62-85, :24-30
 * follows / structural_start: src/generic/stage1/json_scanner.h:44,68-90,128-157
 * unescaped control chars: src/generic/stage1/json_structural_indexer.h:240,246
 */
uint32_t sjo_scan(const uint8_t *buf, size_t len, uint32_t *idx, uint32_t *flags) {
  int next_is_escaped = 0, in_string = 0, prev_nonquote_scalar = 0, ctrl = 0;
  uint32_t n = 0;
  for (size_t i = 0; i < len; i++) {
    uint8_t b = buf[i];
    int escaped = next_is_escaped;
    next_is_escaped = (b == '\\') && !escaped;
    int quote = (b == '"') && !escaped;
    in_string ^= quote; /* includes the opening quote, excludes the closing one */
    int string_tail = in_string ^ quote;
    int ws = is_ws(b), op = is_op(b);
    int scalar = !ws && !op;
    int follows = prev_nonquote_scalar;
    prev_nonquote_scalar = scalar && !quote;
    if ((op || (scalar && !follows)) && !string_tail) { idx[n++] = (uint32_t)i; }
    if (b <= 0x1F && in_string) { ctrl = 1; }
  }
  /* tail padding is 0x20 (src/generic/stage1/buf_block_reader.h:99-104): no structurals, no state change */
  uint32_t f = 0;
  if (in_string) { f |= 1u; }
  if (ctrl) { f |= 2u; }
  if (!sjo_validate_utf8(buf, len)) { f |= 4u; }
  if (flags) { *flags = f; }
  return n;
}

/* ---- src/generic/stage1/json_structural_indexer.h:156-174 */
size_t sjo_trim_partial_utf8(const uint8_t *buf, size_t len) {
  if (len >= 1 && buf[len - 1] >= 0xC0) { return len - 1; }
  if (len >= 2 && buf[len - 2] >= 0xE0) { return len - 2; }
  if (len >= 3 && buf[len - 3] >= 0xF0) { return len - 3; }
  return len;
}

/* ---- src/generic/stage1/find_next_document_index.h:39-98
 * Walk the structurals backwards to the last place where one value directly follows another
 * (no ',' ':' '{' '[' before it, no ',' ':' '}' ']' as itself); the tail after that boundary is
 * complete iff its brackets balance. */
uint32_t sjo_find_next_document_index(const uint8_t *buf, const uint32_t *idx, uint32_t n) {
  if (n == 0) { return 0; }
  int arr = 0, obj = 0;
  for (uint32_t i = n - 1; i > 0; i--) {
    uint8_t cb = buf[idx[i]];
    if (cb == ':' || cb == ',') { continue; }
    if (cb == '}') { obj--; continue; }
    if (cb == ']') { arr--; continue; }
    if (cb == '{') { obj++; }
    else if (cb == '[') { arr++; }
    uint8_t ca = buf[idx[i - 1]];
    if (ca == '{' || ca == '[' || ca == ':' || ca == ',') { continue; }
    return (arr == 0 && obj == 0) ? n : i;
  }
  uint8_t c0 = buf[idx[0]];
  if (c0 == '}') { obj--; }
  else if (c0 == ']') { arr--; }
  else if (c0 == '{') { obj++; }
  else if (c0 == '[') { arr++; }
  return (arr == 0 && obj == 0) ? n : 0;
}

#define SJO_DOCUMENT_TOO_LARGE 0xFFFFFFFFu /* find_next_document_index.h:105 */

/* ---- src/generic/stage1/find_next_document_index.h:126-267 (RFC 7464 record separators) */
static uint32_t seq_filter(const uint8_t *buf, uint32_t *idx, uint32_t *n_io, size_t len, int is_final,
                           uint32_t *next_batch_start) {
  *next_batch_start = (uint32_t)len;
  uint32_t n = *n_io;
  if (n == 0) { return 0; }
  uint32_t w = 0, last_rs = 0, rs_count = 0;
  for (uint32_t r = 0; r < n; r++) {
    uint32_t pos = idx[r];
    if (buf[pos] != 0x1E) { idx[w++] = pos; continue; }
    last_rs = pos;
    rs_count++;
    uint32_t v = pos + 1; /* skip whitespace and further RS bytes to the value start */
    while (v < len) {
      uint8_t c = buf[v];
      if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { v++; }
      else if (c == 0x1E) { last_rs = v; rs_count++; v++; }
      else { break; }
    }
    while (r + 1 < n && idx[r + 1] < v) { r++; } /* structurals swallowed by that run */
    if (v < len) {
      uint8_t c = buf[v];
      int is_operator = (c == '{' || c == '}' || c == '[' || c == ']' || c == ':' || c == ',');
      int already = (r + 1 < n && idx[r + 1] == v);
      if (!is_operator && !already) { idx[w++] = v; } /* scalar start the scanner glued to the RS */
    }
  }
  *n_io = n = w;
  if (n == 0) { return 0; }
  if (rs_count == 0) { return is_final ? sjo_find_next_document_index(buf, idx, n) : 0; }
  if (is_final) { return n; }
  *next_batch_start = last_rs;
  if (rs_count < 2) { return SJO_DOCUMENT_TOO_LARGE; } /* n > 0 here */
  for (uint32_t i = n; i > 0; i--) {
    if (idx[i - 1] < last_rs) { return i; }
  }
  return 0;
}

/* ---- src/generic/stage1/find_next_document_index.h:288-369 (root-level commas) */
static uint32_t comma_filter(const uint8_t *buf, uint32_t *idx, uint32_t *n_io, size_t len, int is_final,
                             uint32_t *next_batch_start) {
  *next_batch_start = (uint32_t)len;
  uint32_t n = *n_io;
  if (n == 0) { return 0; }
  int depth = 0;
  uint32_t w = 0, last_comma = 0, commas = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t p = idx[i];
    uint8_t c = buf[p];
    if (c == '{' || c == '[') { depth++; }
    else if (c == '}' || c == ']') { depth--; }
    else if (c == ',' && depth == 0) { last_comma = p; commas++; continue; }
    idx[w++] = p;
  }
  *n_io = n = w;
  if (n == 0) { return 0; }
  if (is_final) { return sjo_find_next_document_index(buf, idx, n); }
  if (commas == 0) { return SJO_DOCUMENT_TOO_LARGE; }
  *next_batch_start = last_comma + 1;
  uint32_t keep = 0;
  for (uint32_t i = n; i > 0; i--) {
    if (idx[i - 1] < last_comma) { keep = i; break; }
  }
  if (keep == 0) { return 0; }
  *n_io = keep;
  return sjo_find_next_document_index(buf, idx, keep);
}

/* ---- index<STEP>() + finish(): src/generic/stage1/json_structural_indexer.h:193-218,249-397 */
int sjo_stage1(const uint8_t *buf, size_t len, int mode, size_t capacity, uint32_t *idx, uint32_t *n_io) {
  if (len > capacity) { return SJO_CAPACITY; }
  if (len == 0) { return SJO_EMPTY; }
  int streaming = (mode != SJO_REGULAR);
  if (streaming) {
    len = sjo_trim_partial_utf8(buf, len);
    if (len == 0) { return SJO_UTF8_ERROR; }
  }
  uint32_t flags = 0;
  uint32_t n = sjo_scan(buf, len, idx, &flags);
  int unclosed = (flags & 1u) != 0;
  if (unclosed && !streaming) { return SJO_UNCLOSED_STRING; }
  if (flags & 2u) { return SJO_UNESCAPED_CHARS; }
  *n_io = n;
  idx[n] = (uint32_t)len;
  idx[n + 1] = (uint32_t)len;
  idx[n + 2] = 0;
  if (n == 0) { return SJO_EMPTY; }
  if (idx[n - 1] > len) { return SJO_UNEXPECTED_ERROR; }
  uint32_t nbs = (uint32_t)len, r;
  switch (mode) {
  case SJO_STREAMING_PARTIAL:
    if (unclosed) { if (--*n_io == 0) { return SJO_CAPACITY; } }
    r = sjo_find_next_document_index(buf, idx, *n_io);
    if (r == 0 && *n_io > 0) {
      if (idx[0] == 0) { return SJO_CAPACITY; }
      *n_io = 0;
      return SJO_EMPTY;
    }
    *n_io = r;
    break;
  case SJO_STREAMING_FINAL:
    if (unclosed) { --*n_io; }
    *n_io = sjo_find_next_document_index(buf, idx, *n_io);
    idx[*n_io + 1] = idx[*n_io];
    idx[*n_io] = (uint32_t)len;
    if (*n_io == 0) { return SJO_EMPTY; }
    break;
  case SJO_JSON_SEQUENCE_PARTIAL:
  case SJO_COMMA_DELIMITED_PARTIAL:
    if (unclosed) { if (--*n_io == 0) { return SJO_CAPACITY; } }
    r = (mode == SJO_JSON_SEQUENCE_PARTIAL) ? seq_filter(buf, idx, n_io, len, 0, &nbs)
                                            : comma_filter(buf, idx, n_io, len, 0, &nbs);
    if (r == SJO_DOCUMENT_TOO_LARGE) { return SJO_CAPACITY; }
    if (r == 0) { *n_io = 0; return SJO_EMPTY; }
    *n_io = r;
    idx[r] = nbs;
    break;
  case SJO_JSON_SEQUENCE_FINAL:
  case SJO_COMMA_DELIMITED_FINAL:
    if (unclosed) { --*n_io; }
    *n_io = (mode == SJO_JSON_SEQUENCE_FINAL) ? seq_filter(buf, idx, n_io, len, 1, &nbs)
                                              : comma_filter(buf, idx, n_io, len, 1, &nbs);
    idx[*n_io + 1] = idx[*n_io];
    idx[*n_io] = (uint32_t)len;
    if (*n_io == 0) { return SJO_EMPTY; }
    break;
  default:
    break;
  }
  return (flags & 4u) ? SJO_UTF8_ERROR : SJO_SUCCESS;
}

/* ---- src/generic/stage1/json_minifier.h:37-47,68-97: keep every byte that is not whitespace
 * outside a string; an unclosed string voids the output. */
int sjo_minify(const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len) {
  int next_is_escaped = 0, in_string = 0;
  size_t o = 0;
  for (size_t i = 0; i < len; i++) {
    uint8_t b = buf[i];
    int escaped = next_is_escaped;
    next_is_escaped = (b == '\\') && !escaped;
    in_string ^= ((b == '"') && !escaped);
    if (!(is_ws(b) && !in_string)) { dst[o++] = b; }
  }
  if (in_string) { *dst_len = 0; return SJO_UNCLOSED_STRING; }
  *dst_len = o;
  return SJO_SUCCESS;
}

/* ---- strings ------------------------------------------------------------------------------------------ */
static unsigned str_byte(const uint8_t *p, const uint8_t *end) { return p < end ? *p : 0x20u; }

/* jsoncharutils::hex_to_u32_nocheck (/root/reference/include/simdjson/generic/jsoncharutils.h:31-38): four hex digits,
 * or a value with bits above 16 set when any of them is not one */
static uint32_t hex4(const uint8_t *p, const uint8_t *end) {
  uint32_t v = 0;
  for (int k = 0; k < 4; k++) {
    const unsigned c = str_byte(p + k, end);
    uint32_t d;
    if (c >= '0' && c <= '9') { d = c - '0'; }
    else if (c >= 'a' && c <= 'f') { d = c - 'a' + 10; }
    else if (c >= 'A' && c <= 'F') { d = c - 'A' + 10; }
    else { return 0xFFFFFFFFu; }
    v = (v << 4) | d;
  }
  return v;
}

/* jsoncharutils::codepoint_to_utf8 (:52-80): 0 bytes = not a code point */
static size_t put_utf8(uint32_t cp, uint8_t *c) {
  if (cp <= 0x7F) { if (c) { c[0] = (uint8_t)cp; } return 1; }
  if (cp <= 0x7FF) { if (c) { c[0] = (uint8_t)((cp >> 6) + 192); c[1] = (uint8_t)((cp & 63) + 128); } return 2; }
  if (cp <= 0xFFFF) { if (c) { c[0] = (uint8_t)((cp >> 12) + 224); c[1] = (uint8_t)(((cp >> 6) & 63) + 128); c[2] = (uint8_t)((cp & 63) + 128); } return 3; }
  if (cp <= 0x10FFFF) {
    if (c) { c[0] = (uint8_t)((cp >> 18) + 240); c[1] = (uint8_t)(((cp >> 12) & 63) + 128); c[2] = (uint8_t)(((cp >> 6) & 63) + 128); c[3] = (uint8_t)((cp & 63) + 128); }
    return 4;
  }
  return 0;
}

static uint8_t escape_value(unsigned c) { /* escape_map, stringparsing.h:22-43 */
  switch (c) {
  case '"': return 0x22; case '/': return 0x2f; case '\\': return 0x5c;
  case 'b': return 0x08; case 'f': return 0x0c; case 'n': return 0x0a; case 'r': return 0x0d; case 't': return 0x09;
  default: return 0;
  }
}

long sjo_parse_string(const uint8_t *src, const uint8_t *end, uint8_t *dst, int allow_replacement) {
  size_t o = 0;
  for (;;) {
    if (src >= end) { return -1; } /* no closing quote: stage 1 would have said UNCLOSED_STRING */
    const unsigned c = *src;
    if (c == '"') { return (long)o; }
    if (c != '\\') { if (dst) { dst[o] = (uint8_t)c; } o++; src++; continue; }
    const unsigned e = str_byte(src + 1, end);
    if (e != 'u') {
      const uint8_t v = escape_value(e);
      if (!v) { return -1; }
      if (dst) { dst[o] = v; }
      o++;
      src += 2;
      continue;
    }
    /* handle_unicode_codepoint (stringparsing.h:50-96) */
    uint32_t cp = hex4(src + 2, end);
    src += 6;
    if (cp >= 0xd800 && cp < 0xdc00) {
      if (str_byte(src, end) != '\\' || str_byte(src + 1, end) != 'u') {
        if (!allow_replacement) { return -1; }
        cp = 0xfffd;
      } else {
        const uint32_t low = hex4(src + 2, end) - 0xdc00;
        if (low >> 10) {
          if (!allow_replacement) { return -1; }
          cp = 0xfffd; /* the second escape is NOT consumed: it is looked at again on its own */
        } else {
          cp = (((cp - 0xd800) << 10) | low) + 0x10000;
          src += 6;
        }
      }
    } else if (cp >= 0xdc00 && cp <= 0xdfff) {
      if (!allow_replacement) { return -1; }
      cp = 0xfffd;
    }
    const size_t k = put_utf8(cp, dst ? dst + o : NULL);
    if (k == 0) { return -1; }
    o += k;
  }
}

int sjo_string_buffer(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, int allow_replacement, uint8_t *out,
                      size_t out_cap, uint32_t *offsets, uint64_t *bytes, uint32_t *strings, uint32_t *first_bad) {
  uint64_t at = 0;
  uint32_t count = 0, bad = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n; i++) {
    if (offsets) { offsets[i] = 0xFFFFFFFFu; }
    if (idx[i] >= len || buf[idx[i]] != '"') { continue; }
    const long l = sjo_parse_string(buf + idx[i] + 1, buf + len, NULL, allow_replacement);
    if (l < 0) { if (bad == 0xFFFFFFFFu) { bad = i; } continue; }
    if (at + 5 + (uint64_t)l > out_cap) { return SJO_CAPACITY; }
    const uint32_t l32 = (uint32_t)l;
    out[at] = (uint8_t)l32; out[at + 1] = (uint8_t)(l32 >> 8); out[at + 2] = (uint8_t)(l32 >> 16); out[at + 3] = (uint8_t)(l32 >> 24);
    (void)sjo_parse_string(buf + idx[i] + 1, buf + len, out + at + 4, allow_replacement);
    out[at + 4 + l32] = 0;
    if (offsets) { offsets[i] = (uint32_t)at; }
    at += 5 + (uint64_t)l32;
    count++;
  }
  if (bytes) { *bytes = at; }
  if (strings) { *strings = count; }
  if (first_bad) { *first_bad = bad; }
  return bad == 0xFFFFFFFFu ? SJO_SUCCESS : SJO_STRING_ERROR;
}

uint64_t sjo_fnv1a64(const void *data, size_t nbytes) {
  const uint8_t *p = (const uint8_t *)data;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < nbytes; i++) { h = (h ^ p[i]) * 0x100000001b3ull; }
  return h;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double sjo_bench(int which, const uint8_t *buf, size_t len, int iters, void *scratch) {
  double best = 1e300;
  for (int it = 0; it < iters + 1; it++) {
    double t0 = now_s();
    if (which == 0) { uint32_t n = 0; (void)sjo_stage1(buf, len, SJO_REGULAR, len, (uint32_t *)scratch, &n); }
    else if (which == 1) { size_t n = 0; (void)sjo_minify(buf, len, (uint8_t *)scratch, &n); }
    else { (void)sjo_validate_utf8(buf, len); }
    double dt = now_s() - t0;
    if (it > 0 && dt < best) { best = dt; }
  }
  return best;
}
