/* oracle/sj_oracle_stage2.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sj_oracle.h).
 *
 * Scalar restatement of the reference's stage 2 for one document: the serial walk over the structural list
 * (/root/reference/src/generic/stage2/json_iterator.h:121-244, walk_document<STREAMING = false>) with the DOM tape builder as
 * its visitor (/root/reference/src/generic/stage2/tape_builder.h:142-441, tape format /root/reference/doc/tape.md):
 *
 *   tape[0]            'r' | number of tape words                         (visit_document_end, tape_builder.h:160-165)
 *   '{' '['            type | count (saturating at 0xFFFFFF) << 32 | index behind the matching close   (end_container :396-407)
 *   '}' ']'            type | index of the matching open
 *   '"'                type | offset of the string's record in string_buf  (on_start_string :415-419)
 *   numbers            'l' / 'u' / 'd' word followed by the raw 64-bit value (tape_writer.h append_s64 / _u64 / _double)
 *   't' 'f' 'n'        type only
 *   tape[last]         'r' | 0
 *
 * Numbers follow numberparsing::parse_number (/root/reference/include/simdjson/generic/numberparsing.h:859-971) as a grammar;
 * the VALUE of a float is taken from the C library's strtod, which is correctly rounded -- what the reference's Eisel-Lemire
 * path plus its from_chars fallback (:65-330, :535-548, :763-814) compute, by a route that shares nothing with the device code
 * this file checks.  Infinite results are NUMBER_ERROR (:794-797, :806-809).  Bytes at or beyond len read as 0x20, which is what
 * the reference arranges for the only tokens that can touch them (root scalars: tape_builder.h:243-262 and the length-aware atom
 * checks, /root/reference/include/simdjson/generic/atomparsing.h).
 * Pinned against the reference's dom::parser::parse by tests/test_oracle_vs_reference.py (tape and string_buf word for word,
 * error codes on broken documents).
 */
#include "sj_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { E_OK = 0, E_MEMALLOC = 2, E_TAPE = 3, E_DEPTH = 4, E_STRING = 5, E_T_ATOM = 6, E_F_ATOM = 7, E_N_ATOM = 8, E_NUMBER = 9, E_BIGINT = 10, E_EMPTY = 13 };

typedef struct {
  const uint8_t *buf;
  size_t len;
  uint64_t *tape;
  size_t tape_cap, tape_at;
  uint8_t *str;
  size_t str_cap, str_at;
  int overflow;
} builder;

static unsigned at(const builder *b, size_t pos) { return pos < b->len ? b->buf[pos] : 0x20u; }
/* jsoncharutils::is_not_structural_or_whitespace (src/internal/jsoncharutils_tables.cpp:14-29) */
static int not_structural_or_ws(unsigned c) {
  switch (c) {
  case 0x09: case 0x0A: case 0x0D: case 0x20: case ',': case ':': case '[': case ']': case '{': case '}': return 0;
  default: return 1;
  }
}
static void put(builder *b, uint64_t word) {
  if (b->tape_at < b->tape_cap) { b->tape[b->tape_at] = word; } else { b->overflow = 1; }
  b->tape_at++;
}
static uint64_t tagged(char type, uint64_t payload) { return ((uint64_t)(unsigned char)type << 56) | payload; }

/* atomparsing.h: the four letters, then a structural or whitespace byte */
static int atom_ok(const builder *b, size_t pos, const char *word) {
  size_t k = 0;
  for (; word[k]; k++) { if (at(b, pos + k) != (unsigned char)word[k]) { return 0; } }
  return !not_structural_or_ws(at(b, pos + k));
}

/* numberparsing.h:859-971 */
static int number(builder *b, size_t pos) {
  size_t p = pos;
  const int negative = at(b, p) == '-';
  if (negative) { p++; }
  const size_t start_digits = p;
  uint64_t i = 0;
  while (at(b, p) - '0' <= 9u) { i = 10 * i + (at(b, p) - '0'); p++; } /* wraps like the reference's (:617-625) */
  size_t digit_count = p - start_digits;
  if (digit_count == 0 || (at(b, start_digits) == '0' && digit_count > 1)) { return E_NUMBER; }
  int is_float = 0;
  if (at(b, p) == '.') {
    is_float = 1;
    p++;
    const size_t first = p;
    while (at(b, p) - '0' <= 9u) { p++; }
    if (p == first) { return E_NUMBER; } /* "123." (:659-672) */
  }
  if (at(b, p) == 'e' || at(b, p) == 'E') {
    is_float = 1;
    p++;
    if (at(b, p) == '-' || at(b, p) == '+') { p++; }
    const size_t first = p;
    while (at(b, p) - '0' <= 9u) { p++; }
    if (p == first) { return E_NUMBER; } /* (:674-723) */
  }
  if (is_float) {
    /* the token's text, NUL-terminated, through strtod */
    const size_t tl = p - pos;
    char stack[128], *text = tl < sizeof stack ? stack : (char *)malloc(tl + 1);
    if (!text) { return E_MEMALLOC; }
    for (size_t k = 0; k < tl; k++) { text[k] = (char)at(b, pos + k); }
    text[tl] = 0;
    const double d = strtod(text, NULL);
    if (text != stack) { free(text); }
    if (isinf(d)) { return E_NUMBER; }
    uint64_t bits;
    memcpy(&bits, &d, 8);
    put(b, tagged('d', 0));
    put(b, bits);
    return not_structural_or_ws(at(b, p)) ? E_NUMBER : E_OK; /* dirty_end (:916-920) */
  }
  const size_t longest = negative ? 19 : 20;
  if (digit_count > longest) { return E_BIGINT; }
  if (digit_count == longest) {
    if (negative) {
      if (i > (uint64_t)INT64_MAX + 1) { return E_BIGINT; }
      put(b, tagged('l', 0));
      put(b, ~i + 1);
      return not_structural_or_ws(at(b, p)) ? E_NUMBER : E_OK;
    } else if (at(b, pos) != '1' || i <= (uint64_t)INT64_MAX) {
      return E_BIGINT;
    }
  }
  if (i > (uint64_t)INT64_MAX) {
    put(b, tagged('u', 0));
    put(b, i);
  } else {
    put(b, tagged('l', 0));
    put(b, negative ? ~i + 1 : i);
  }
  return not_structural_or_ws(at(b, p)) ? E_NUMBER : E_OK;
}

/* visit_string (tape_builder.h:187-205) */
static int string(builder *b, size_t pos) {
  put(b, tagged('"', b->str_at));
  const long l = sjo_parse_string(b->buf + pos + 1, b->buf + b->len, NULL, 0);
  if (l < 0) { return E_STRING; }
  if (b->str_at + 5 + (size_t)l > b->str_cap) { b->overflow = 1; b->str_at += 5 + (size_t)l; return E_OK; }
  uint8_t *rec = b->str + b->str_at;
  const uint32_t l32 = (uint32_t)l;
  memcpy(rec, &l32, 4);
  (void)sjo_parse_string(b->buf + pos + 1, b->buf + b->len, rec + 4, 0);
  rec[4 + l] = 0;
  b->str_at += 5 + (size_t)l;
  return E_OK;
}

/* visit_primitive (json_iterator.h:342-370) / visit_root_primitive (:313-340).  Inside a container the reference tests
 * `(*value - '0') < 10` in int arithmetic, which holds for EVERY byte below ':' -- a ',' or '!' where a value is expected is
 * handed to parse_number and comes back as NUMBER_ERROR, not TAPE_ERROR; the root value goes through an exact switch. */
static int primitive(builder *b, size_t pos, int root) {
  const unsigned c = at(b, pos);
  if (c == '"') { return string(b, pos); }
  if (root ? (c - '0' <= 9u || c == '-') : ((int)c - '0' < 10 || c == '-')) { return number(b, pos); }
  switch (c) {
  case 't': if (!atom_ok(b, pos, "true")) { return E_T_ATOM; } put(b, tagged('t', 0)); return E_OK;
  case 'f': if (!atom_ok(b, pos, "false")) { return E_F_ATOM; } put(b, tagged('f', 0)); return E_OK;
  case 'n': if (!atom_ok(b, pos, "null")) { return E_N_ATOM; } put(b, tagged('n', 0)); return E_OK;
  default: return E_TAPE;
  }
}

int sjo_stage2(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, uint64_t *tape, size_t tape_cap,
               uint8_t *string_buf, size_t string_cap, uint64_t *tape_words, uint64_t *string_bytes) {
  if (tape_words) { *tape_words = 0; }
  if (string_bytes) { *string_bytes = 0; }
  if (n == 0) { return E_EMPTY; }
  builder B = {buf, len, tape, tape_cap, 0, string_buf, string_cap, 0, 0};
  builder *b = &B;
  /* dom_parser_implementation::open_containers / is_array: max_depth entries (generic/dom_parser_implementation.h:80-89) */
  uint32_t *open_at = (uint32_t *)malloc(((size_t)max_depth + 1) * sizeof(uint32_t));
  uint32_t *count = (uint32_t *)malloc(((size_t)max_depth + 1) * sizeof(uint32_t));
  uint8_t *is_array = (uint8_t *)malloc((size_t)max_depth + 1);
  if (!open_at || !count || !is_array) { free(open_at); free(count); free(is_array); return E_MEMALLOC; }
  uint32_t depth = 0, i = 0;
  int err = E_OK;
  enum { OBJECT_BEGIN, OBJECT_FIELD, OBJECT_CONTINUE, SCOPE_END, ARRAY_BEGIN, ARRAY_VALUE, ARRAY_CONTINUE, DOCUMENT_END } state;
#define TOK(k) ((k) <= n ? at(b, idx[k]) : 0x20u) /* idx[n] = len: the sentinel reads as a space */
#define FAIL(code) do { err = (code); goto done; } while (0)
  put(b, 0); /* visit_document_start: the root word is written at the end */
  {
    const unsigned c = TOK(i);
    const size_t pos = idx[i];
    i++;
    const unsigned last = at(b, idx[n - 1]);
    if (c == '{' && last != '}') { FAIL(E_TAPE); } /* json_iterator.h:138-143 */
    if (c == '[' && last != ']') { FAIL(E_TAPE); }
    if (c == '{') {
      if (TOK(i) == '}') { i++; put(b, tagged('{', b->tape_at + 2)); put(b, tagged('}', b->tape_at - 1)); state = DOCUMENT_END; }
      else { state = OBJECT_BEGIN; }
    } else if (c == '[') {
      if (TOK(i) == ']') { i++; put(b, tagged('[', b->tape_at + 2)); put(b, tagged(']', b->tape_at - 1)); state = DOCUMENT_END; }
      else { state = ARRAY_BEGIN; }
    } else {
      err = primitive(b, pos, 1);
      if (err) { goto done; }
      state = DOCUMENT_END;
    }
  }
  for (;;) {
    switch (state) {
    case OBJECT_BEGIN: {
      depth++;
      if (depth >= max_depth) { FAIL(E_DEPTH); }
      is_array[depth] = 0;
      open_at[depth] = (uint32_t)b->tape_at;
      count[depth] = 0;
      put(b, 0);
      const size_t pos = idx[i < n ? i : n];
      if (TOK(i) != '"') { FAIL(E_TAPE); }
      i++;
      count[depth]++;
      err = string(b, pos);
      if (err) { goto done; }
      state = OBJECT_FIELD;
      break;
    }
    case OBJECT_FIELD: {
      if (TOK(i) != ':') { FAIL(E_TAPE); }
      i++;
      const unsigned c = TOK(i);
      const size_t pos = idx[i < n ? i : n];
      i++;
      if (c == '{') {
        if (TOK(i) == '}') { i++; put(b, tagged('{', b->tape_at + 2)); put(b, tagged('}', b->tape_at - 1)); state = OBJECT_CONTINUE; }
        else { state = OBJECT_BEGIN; }
      } else if (c == '[') {
        if (TOK(i) == ']') { i++; put(b, tagged('[', b->tape_at + 2)); put(b, tagged(']', b->tape_at - 1)); state = OBJECT_CONTINUE; }
        else { state = ARRAY_BEGIN; }
      } else {
        err = primitive(b, pos, 0);
        if (err) { goto done; }
        state = OBJECT_CONTINUE;
      }
      break;
    }
    case OBJECT_CONTINUE: {
      const unsigned c = TOK(i);
      i++;
      if (c == ',') {
        count[depth]++;
        const size_t pos = idx[i < n ? i : n];
        if (TOK(i) != '"') { FAIL(E_TAPE); }
        i++;
        err = string(b, pos);
        if (err) { goto done; }
        state = OBJECT_FIELD;
      } else if (c == '}') {
        put(b, tagged('}', open_at[depth]));
        const uint32_t cnt = count[depth] > 0xFFFFFFu ? 0xFFFFFFu : count[depth];
        if (open_at[depth] < b->tape_cap) { b->tape[open_at[depth]] = tagged('{', (uint64_t)b->tape_at | ((uint64_t)cnt << 32)); }
        state = SCOPE_END;
      } else {
        FAIL(E_TAPE);
      }
      break;
    }
    case SCOPE_END:
      depth--;
      state = depth == 0 ? DOCUMENT_END : (is_array[depth] ? ARRAY_CONTINUE : OBJECT_CONTINUE);
      break;
    case ARRAY_BEGIN:
      depth++;
      if (depth >= max_depth) { FAIL(E_DEPTH); }
      is_array[depth] = 1;
      open_at[depth] = (uint32_t)b->tape_at;
      count[depth] = 1; /* increment_count right behind visit_array_start (json_iterator.h:207-208) */
      put(b, 0);
      state = ARRAY_VALUE;
      break;
    case ARRAY_VALUE: {
      const unsigned c = TOK(i);
      const size_t pos = idx[i < n ? i : n];
      i++;
      if (c == '{') {
        if (TOK(i) == '}') { i++; put(b, tagged('{', b->tape_at + 2)); put(b, tagged('}', b->tape_at - 1)); state = ARRAY_CONTINUE; }
        else { state = OBJECT_BEGIN; }
      } else if (c == '[') {
        if (TOK(i) == ']') { i++; put(b, tagged('[', b->tape_at + 2)); put(b, tagged(']', b->tape_at - 1)); state = ARRAY_CONTINUE; }
        else { state = ARRAY_BEGIN; }
      } else {
        err = primitive(b, pos, 0);
        if (err) { goto done; }
        state = ARRAY_CONTINUE;
      }
      break;
    }
    case ARRAY_CONTINUE: {
      const unsigned c = TOK(i);
      i++;
      if (c == ',') {
        count[depth]++;
        state = ARRAY_VALUE;
      } else if (c == ']') {
        put(b, tagged(']', open_at[depth]));
        const uint32_t cnt = count[depth] > 0xFFFFFFu ? 0xFFFFFFu : count[depth];
        if (open_at[depth] < b->tape_cap) { b->tape[open_at[depth]] = tagged('[', (uint64_t)b->tape_at | ((uint64_t)cnt << 32)); }
        state = SCOPE_END;
      } else {
        FAIL(E_TAPE);
      }
      break;
    }
    case DOCUMENT_END:
      put(b, tagged('r', 0));
      if (b->tape_cap) { b->tape[0] = tagged('r', b->tape_at); }
      if (i != n) { FAIL(E_TAPE); } /* json_iterator.h:236-239 */
      goto done;
    }
  }
done:
  free(open_at);
  free(count);
  free(is_array);
  if (tape_words) { *tape_words = b->tape_at; }
  if (string_bytes) { *string_bytes = b->str_at; }
  if (err == E_OK && b->overflow) { return SJO_CAPACITY; }
  return err;
#undef TOK
#undef FAIL
}

/* ---- On-Demand's raw key comparison, batched --------------------------------------------------------------------------
 * raw_json_string::unsafe_is_equal(length, target) (/root/reference/include/simdjson/generic/ondemand/raw_json_string-inl.h:66-69),
 * as value_iterator::find_field_raw applies it to every key of an object (value_iterator-inl.h:132, :229): the key's RAW bytes
 * (escapes not resolved) against the target's bytes, the byte behind them a quote, and `length` -- the room between this structural
 * and the next one minus the two quotes -- at least the target's length. */
int sjo_raw_key_equal(const uint8_t *raw, size_t length, const uint8_t *target, size_t m) {
  return length >= m && raw[m] == '"' && memcmp(raw, target, m) == 0;
}
/* out[i] = index of the first target equal to structural i when that structural is a KEY (a string whose next structural is ':'),
 * 0xFFFFFFFF otherwise.  targets: K byte strings back to back, lens[k] their lengths.  idx[0..n] incl. the first sentinel.
 * Returns the number of matches. */
uint32_t sjo_match_keys(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, const uint8_t *targets, const uint32_t *lens, uint32_t K,
                        uint32_t *out) {
  uint32_t matches = 0;
  for (uint32_t i = 0; i < n; i++) {
    out[i] = 0xFFFFFFFFu;
    if (buf[idx[i]] != '"' || i + 1 >= n || buf[idx[i + 1]] != ':') { continue; }
    const size_t room = (size_t)idx[i + 1] - idx[i];
    if (room < 2) { continue; }
    const uint8_t *t = targets;
    for (uint32_t k = 0; k < K; t += lens[k], k++) {
      if ((size_t)idx[i] + 1 + lens[k] < len && sjo_raw_key_equal(buf + idx[i] + 1, room - 2, t, lens[k])) { out[i] = k; matches++; break; }
    }
  }
  return matches;
}
