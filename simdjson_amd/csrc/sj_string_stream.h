// simdjson_amd/csrc/sj_string_stream.h -- document::string_buf as a STREAM COMPACTION of the document (per-block math).
//
// The reference fills string_buf one string at a time (stringparsing::parse_string, /root/reference/src/generic/stage2/
// stringparsing.h:150-193, called per token by tape_builder::visit_string, tape_builder.h:187-205, :415-433): a record is
// [u32 length][unescaped bytes][0].  Records follow each other in document order, so the whole buffer is the document with
//   * everything outside strings dropped,
//   * every opening quote replaced by 4 bytes (the length, filled in later) and every closing quote by one 0,
//   * every escape replaced by what it stands for: "\n" -> one byte, "é" -> two, a surrogate pair (12 bytes) -> four.
// That is the shape of minify: a lane owns 64 bytes, works on 64-bit masks, and the position of a byte in the output is a
// prefix sum.  Which bytes are escaped and which quotes are real is stage 1's escape / quote algebra (sj_block.h); new here
// is only the bookkeeping of \u escapes, which may reach over the end of a block (the next block looks back 10 bytes).
// An escape the reference rejects (stringparsing.h:22-43 escape_map, :50-96 handle_unicode_codepoint) is reported as a mask;
// the kernels fall back to the per-string walk of sjgpu_strings.hip for such documents, so this path only ever produces
// buffers of documents whose strings are all valid -- byte for byte the reference's.
// Host + device: sjgpu_string_stream.hip and tests/host/test_string_stream.cpp run the same functions.
#ifndef SJGPU_SJ_STRING_STREAM_H
#define SJGPU_SJ_STRING_STREAM_H

#include "sj_block.h"

namespace sjgpu {

// ---- escapes (the same rules as unescape<> in sjgpu_strings.hip, which stays the per-string fallback) ----------------------------
SJ_HD u32 simple_escape_value(u32 c) { // escape_map, stringparsing.h:22-43 (0 = not an escape)
  switch (c) {
  case '"': return 0x22u;
  case '/': return 0x2fu;
  case '\\': return 0x5cu;
  case 'b': return 0x08u;
  case 'f': return 0x0cu;
  case 'n': return 0x0au;
  case 'r': return 0x0du;
  case 't': return 0x09u;
  default: return 0u;
  }
}
// jsoncharutils::hex_to_u32_nocheck (/root/reference/include/simdjson/generic/jsoncharutils.h:31-38): 0xFFFFFFFF = not hex
template <class SRC> SJ_HD u32 hex4_at(const SRC &src, u32 pos) {
  u32 v = 0;
  for (u32 k = 0; k < 4; k++) {
    const u32 c = src.byte(pos + k);
    u32 d;
    if (c - u32('0') <= 9u) { d = c - u32('0'); }
    else if ((c | 0x20u) - u32('a') <= 5u) { d = (c | 0x20u) - u32('a') + 10u; }
    else { return 0xFFFFFFFFu; }
    v = (v << 4) | d;
  }
  return v;
}

// One \u escape whose 'u' sits at upos (handle_unicode_codepoint, stringparsing.h:50-96; codepoint_to_utf8, jsoncharutils.h:52-80).
struct u_escape {
  u32 len;    // bytes it stands for: 1 ... 4 (0 when rejected)
  u32 span;   // bytes behind the 'u' that belong to it: 4 hex digits, or 10 when a low surrogate's escape was consumed with it
  u32 packed; // those bytes, the first one in the low bits
  bool bad;   // the reference rejects the string
};
template <class SRC> SJ_HD u_escape decode_u_escape(const SRC &src, u32 upos, bool allow_replacement) {
  u_escape e{0u, 4u, 0u, false};
  u32 cp = hex4_at(src, upos + 1);
  if (cp >= 0xd800u && cp < 0xdc00u) {
    if (src.byte(upos + 5) != '\\' || src.byte(upos + 6) != 'u') {
      if (!allow_replacement) { e.bad = true; return e; }
      cp = 0xfffdu;
    } else {
      const u32 low = hex4_at(src, upos + 7) - 0xdc00u;
      if (low >> 10) {
        if (!allow_replacement) { e.bad = true; return e; }
        cp = 0xfffdu; // the second escape is not consumed: it is looked at again on its own
      } else {
        cp = (((cp - 0xd800u) << 10) | low) + 0x10000u;
        e.span = 10;
      }
    }
  } else if (cp >= 0xdc00u && cp <= 0xdfffu) {
    if (!allow_replacement) { e.bad = true; return e; }
    cp = 0xfffdu;
  }
  if (cp <= 0x7Fu) { e.len = 1; e.packed = cp; }
  else if (cp <= 0x7FFu) { e.len = 2; e.packed = ((cp >> 6) + 192u) | (((cp & 63u) + 128u) << 8); }
  else if (cp <= 0xFFFFu) { e.len = 3; e.packed = ((cp >> 12) + 224u) | ((((cp >> 6) & 63u) + 128u) << 8) | (((cp & 63u) + 128u) << 16); }
  else if (cp <= 0x10FFFFu) {
    e.len = 4;
    e.packed = ((cp >> 18) + 240u) | ((((cp >> 12) & 63u) + 128u) << 8) | ((((cp >> 6) & 63u) + 128u) << 16) | (((cp & 63u) + 128u) << 24);
  } else {
    e.bad = true; // not hex
  }
  return e;
}

// ---- byte classes the escapes need, from the bit planes of a block ------------------------------------------------------------------
// bytes equal to V: eight literals in three three-input functions
template <u32 V> SJ_HD u64 plane_eq(const planes &P) {
  constexpr u32 L7 = (V & 0x80u) ? A3 : (~A3 & 0xFFu), L6 = (V & 0x40u) ? B3 : (~B3 & 0xFFu), L5 = (V & 0x20u) ? C3 : (~C3 & 0xFFu);
  constexpr u32 L4 = (V & 0x10u) ? A3 : (~A3 & 0xFFu), L3 = (V & 0x08u) ? B3 : (~B3 & 0xFFu), L2 = (V & 0x04u) ? C3 : (~C3 & 0xFFu);
  constexpr u32 L1 = (V & 0x02u) ? A3 : (~A3 & 0xFFu), L0 = (V & 0x01u) ? B3 : (~B3 & 0xFFu);
  const u64 hi = lut3<SJ_TT3(L7 & L6 & L5)>(P.b[7], P.b[6], P.b[5]);
  const u64 mid = lut3<SJ_TT3(L4 & L3 & L2)>(P.b[4], P.b[3], P.b[2]);
  const u64 lo = lut3<SJ_TT3(L1 & L0 & C3)>(P.b[1], P.b[0], hi);
  return lo & mid;
}
struct escape_classes {
  u64 u;     // 'u'
  u64 same;  // '"' '/' '\\': the escape stands for the byte itself
  u64 remap; // b f n r t: it stands for another byte
};
SJ_HD escape_classes classify_escapes(const planes &P) {
  escape_classes c;
  c.u = plane_eq<0x75u>(P);
  c.same = plane_eq<0x22u>(P) | plane_eq<0x2fu>(P) | plane_eq<0x5cu>(P);
  c.remap = plane_eq<0x62u>(P) | plane_eq<0x66u>(P) | plane_eq<0x6eu>(P) | plane_eq<0x72u>(P) | plane_eq<0x74u>(P);
  return c;
}

// bits lo ... hi of a block (block-relative positions, any sign; clipped to 0 ... 63)
SJ_HD u64 bits_between(int lo, int hi) {
  if (lo < 0) { lo = 0; }
  if (hi > 63) { hi = 63; }
  if (lo > hi) { return 0; }
  const u64 upto_hi = hi == 63 ? ~u64(0) : ((u64(1) << (hi + 1)) - 1);
  return upto_hi & ~((u64(1) << lo) - 1);
}

// ---- one block --------------------------------------------------------------------------------------------------------------------------
// What a block contributes, apart from its quotes: `keep` = the bytes that stand for ONE output byte each if they lie inside a
// string (plain text, the character behind a backslash, the first len bytes of a \u escape); everything else inside a string
// (the backslashes, the rest of a \u escape) is dropped.  Quotes are in neither.
struct string_block {
  u64 keep;
  u64 bad;   // escapes the reference rejects (to be taken seriously inside strings only)
  u64 remap; // kept bytes whose value changes: escaped b f n r t
};
// escapes other than \u: from masks alone.  escaped = stage 1's escaped mask, quote = the real quotes.
SJ_HD string_block simple_escapes(u64 backslash, u64 escaped, u64 quote, const escape_classes &c) {
  string_block b;
  b.keep = ~(andn(backslash, escaped) | quote); // not an escaping backslash, not a real quote
  b.bad = escaped & ~(c.u | c.same | c.remap);
  b.remap = escaped & c.remap;
  return b;
}
// no backslash in sight: everything but the quotes
SJ_HD string_block no_escapes(u64 quote) { return string_block{~quote, 0, 0}; }

// \u escapes: the ones whose 'u' lies in this block (U) and the ones up to 10 bytes in front of it whose bytes may reach into it
// (u_prev: bit k = byte block_pos - 10 + k is an escaped 'u').  The reference consumes a valid low-surrogate escape together with
// the high one in front of it (stringparsing.h:64-81), so the candidates are walked front to back and a consumed one is skipped.
// sink.escape(rel, len, packed): the escape whose 'u' sits at block position rel (-10 ... 63) stands for len bytes (packed, first one low),
// which are the kept bytes at positions rel ... rel + len - 1.
template <class SRC, class SINK>
SJ_HD void unicode_escapes(const SRC &src, u32 block_pos, u64 U, u32 u_prev, bool allow_replacement, string_block &b, SINK &sink) {
  int consumed_at = -100; // block-relative position of the 'u' of a low-surrogate escape that went with its high one
  for (int pass = 0; pass < 2; pass++) {
    u64 todo = pass == 0 ? u64(u_prev & 0x3FFu) : U;
    const int origin = pass == 0 ? -10 : 0;
    while (todo) {
      const int k = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int rel = origin + k;
      if (rel == consumed_at) { continue; }
      if (rel < 0 && block_pos < u32(-rel)) { continue; } // in front of the document (cannot happen: u_prev is zero there)
      const u_escape e = decode_u_escape(src, block_pos + u32(rel), allow_replacement);
      b.keep = (b.keep & ~bits_between(rel - 1, rel + int(e.span))) | bits_between(rel, rel + int(e.len) - 1);
      if (e.bad && rel >= 0) { b.bad |= u64(1) << rel; }
      if (e.len) { sink.escape(rel, e.len, e.packed); }
      if (e.span == 10u) { consumed_at = rel + 6; }
    }
  }
}
struct no_patches {
  SJ_HD void escape(int, u32, u32) {}
};
// the bytes of one escape that fall into the block, one call of f(position 0 ... 63, byte) each
template <class F> SJ_HD void for_each_escape_byte(int rel, u32 len, u32 packed, F &&f) {
  for (u32 j = 0; j < len; j++) {
    const int p = rel + int(j);
    if (p >= 0 && p < 64) { f(u32(p), (packed >> (8u * j)) & 0xFFu); }
  }
}

// output bytes of a block given which of its bytes are inside strings (in_string: stage 1's mask, opening quote included,
// closing quote excluded): data bytes + 4 per opening quote + 1 per closing quote
SJ_HD u32 block_output_bytes(const string_block &b, u64 quote, u64 in_string) {
  return u32(popc64(b.keep & in_string)) + 4u * u32(popc64(quote & in_string)) + u32(popc64(andn(quote, in_string)));
}

} // namespace sjgpu
#endif
