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
// is only the bookkeeping of \u escapes, whose bytes reach over block boundaries: masks, too (unicode_masks below).
// An escape the reference rejects (stringparsing.h:22-43 escape_map, :50-96 handle_unicode_codepoint) is reported as a mask;
// the kernels fall back to the per-string walk of sjgpu_strings.hip for such documents, so this path only ever produces
// buffers of documents whose strings are all valid -- byte for byte the reference's.
// Host + device: sjgpu_string_stream.hip and tests/host/test_string_stream.cpp run the same functions.
#ifndef SJGPU_SJ_STRING_STREAM_H
#define SJGPU_SJ_STRING_STREAM_H

#include "sj_block.h"

namespace sjgpu {

// ---- escapes (the same rules as unescape<> in sjgpu_strings.hip, which stays the per-string fallback) ----------------------------
SJ_HD u32 simple_escape_value(u32 c) { // escape_map, stringparsing.h:22-43 (0 = not an escape); a chain of selects, not a switch (which compiles into divergent branches)
  u32 v = 0u;
  v = c == u32('"') ? 0x22u : v;
  v = c == u32('/') ? 0x2fu : v;
  v = c == u32('\\') ? 0x5cu : v;
  v = c == u32('b') ? 0x08u : v;
  v = c == u32('f') ? 0x0cu : v;
  v = c == u32('n') ? 0x0au : v;
  v = c == u32('r') ? 0x0du : v;
  v = c == u32('t') ? 0x09u : v;
  return v;
}
// ---- \u escapes, without a walk (round 4) -----------------------------------------------------------------------------------------------
// handle_unicode_codepoint (stringparsing.h:50-96) with codepoint_to_utf8 (jsoncharutils.h:52-80) turns "\uXXXX" into 1 - 3 bytes and a
// surrogate pair "\uD8..\uDC.." into 4.  Rounds 2-3 walked a lane's escapes front to back with a decoder (a divergent loop that read the
// document byte by byte: 240 us of k_strs_write's 370 on the synthetic twitter-like text, profiles/r03_strings_escape_cost.txt).  Nothing
// in the result needs the walk:
//   * WHICH bytes of an escape stay is a function of its first three hex digits: the bytes an escape stands for are put on its LAST hex
//     digits -- 1 byte: the 4th; 2 bytes: 3rd, 4th; 3 bytes: 2nd ... 4th; a pair: the high escape's 4th digit and the low escape's 2nd ... 4th --
//     so that every output byte is a function of document bytes AT OR IN FRONT of its own position (UTF-8's first byte needs the leading
//     digits only) and "is the byte at position q kept" is a mask computed from look-back alone: shifts of the escaped-'u' mask ANDed with
//     digit classes.  The stream comes out the same: between the backslash and the last digit nothing else is kept.
//   * WHAT a kept byte becomes is computed by whoever owns the byte, from at most ten bytes in front of it (u_escape_byte).
//   * a rejected escape (a digit that is no hex digit, a lone low surrogate, a high one without a low one behind it) is a mask, too.  It is
//     raised wherever the pattern occurs, inside a string or not: a backslash outside a string makes the document invalid anyway (stage 2
//     rejects the token), and all a raised flag does is send the document down the per-string road, which is always right.
// Masks shift towards higher positions only; what comes in at the bottom is what left the block in front at the top (u_tops), handed
// over in four stages because a shifted mask is ANDed with a class before it is shifted again.
SJ_HD u32 hex_digit_value(u32 c) { return (c & 0xFu) + 9u * (c >> 6); } // of a byte that IS a hex digit
SJ_HD bool byte_is_hex(u32 c) { return c - u32('0') <= 9u || (c | 0x20u) - u32('a') <= 5u; }
SJ_HD bool byte_is_octal(u32 c) { return c - u32('0') <= 7u; }
SJ_HD bool byte_is_d(u32 c) { return (c | 0x20u) == u32('d'); }
SJ_HD bool byte_is_89ab(u32 c) { return c == '8' || c == '9' || (c | 0x20u) == u32('a') || (c | 0x20u) == u32('b'); }
SJ_HD bool byte_is_cdef(u32 c) { return (c | 0x20u) - u32('c') <= 3u; }
struct hex_classes {
  u64 hex, zero, oct, d, s8b, scf; // [0-9a-fA-F], '0', [0-7], [dD], [89abAB], [c-fC-F]
};
SJ_HD hex_classes classify_hex(const planes &P) {
  const u64 b0 = P.b[0], b1 = P.b[1], b2 = P.b[2], b3 = P.b[3], b4 = P.b[4], b5 = P.b[5], b6 = P.b[6], b7 = P.b[7];
  const u64 row3 = ~b7 & ~b6 & b5 & b4;  // 0x30 ... 0x3F
  const u64 al = ~b7 & b6 & ~b4 & ~b3;   // 0x40 ... 0x47, 0x60 ... 0x67
  const u64 low_any = b2 | b1 | b0, low_all = b2 & b1 & b0;
  hex_classes h;
  h.oct = row3 & ~b3;
  h.zero = h.oct & ~low_any;
  h.hex = (row3 & (~b3 | (~b2 & ~b1))) | (al & low_any & ~low_all);
  h.d = al & b2 & ~b1 & ~b0;
  h.s8b = (row3 & b3 & ~b2 & ~b1) | (al & ~b2 & (b1 ^ b0));
  h.scf = al & ((~b2 & b1 & b0) | (b2 & ~(b1 & b0)));
  return h;
}
struct u_tops {
  u32 a, b, c, d; // a: the four highest bits of U; b: the highest bit of z1 | d1 << 1 | h1 << 2; c: of zz2 | h2 << 1, and hi2's six highest << 2; d: of h3
  SJ_HD bool any() const { return (a | b | c | d) != 0u; }
};
struct u_masks {
  u64 drop;       // bytes of \u escapes that leave nothing behind
  u64 bad;        // an escape the reference rejects ends / is recognised here
  u64 k2, k3, k4; // kept bytes on the 2nd / 3rd / 4th hex digit of an escape: their values come from u_escape_byte(position, 2 / 3 / 4)
};
// U = the escaped 'u' of the block; exch(stage, mine) hands `mine` (what leaves this block at the top) on and returns what left the block in front
template <class EXCH> SJ_HD u_masks unicode_masks(u64 U, const hex_classes &h, EXCH &&exch) {
  const u32 pa = exch(0u, u32(U >> 60));
  const u64 A1 = (U << 1) | u64(pa >> 3), U2 = (U << 2) | u64(pa >> 2), U3 = (U << 3) | u64(pa >> 1), U4 = (U << 4) | u64(pa);
  const u64 z1 = A1 & h.zero, d1 = A1 & h.d, h1 = A1 & h.hex; // the first digit: '0', 'd', any
  const u32 pb = exch(1u, u32(z1 >> 63) | (u32(d1 >> 63) << 1) | (u32(h1 >> 63) << 2));
  const u64 z1s = (z1 << 1) | u64(pb & 1u), d1s = (d1 << 1) | u64((pb >> 1) & 1u), h1s = (h1 << 1) | u64((pb >> 2) & 1u);
  const u64 zz2 = z1s & h.zero; // "00": below 0x100
  const u64 zo2 = z1s & h.oct;  // "0[0-7]": below 0x800, at most two bytes
  const u64 hi2 = d1s & h.s8b;  // a high surrogate
  const u64 lo2 = d1s & h.scf;  // a low one
  const u64 h2 = h1s & h.hex;
  const u32 pc = exch(2u, u32(zz2 >> 63) | (u32(h2 >> 63) << 1) | (u32(hi2 >> 58) << 2));
  const u64 one3 = ((zz2 << 1) | u64(pc & 1u)) & h.oct; // "00[0-7]": below 0x80, one byte
  const u64 h3 = ((h2 << 1) | u64((pc >> 1) & 1u)) & h.hex;
  const u64 hi3 = (hi2 << 1) | u64((pc >> 7) & 1u);
  const u64 hi8 = (hi2 << 6) | u64((pc >> 2) & 63u); // where the low escape behind a high one has its second digit
  const u32 pd = exch(3u, u32(h3 >> 63));
  const u64 h4 = ((h3 << 1) | u64(pd & 1u)) & h.hex;
  u_masks m;
  m.drop = U | A1 | zo2 | hi2 | one3 | hi3;
  m.bad = (U4 & ~h4) | (lo2 & ~hi8) | (hi8 & ~lo2);
  m.k2 = U2 & ~(zo2 | hi2);
  m.k3 = U3 & ~(one3 | hi3);
  m.k4 = U4;
  return m;
}
// the byte kept at document position q, the k-th hex digit (2 ... 4) of an escape the reference accepts (codepoint_to_utf8 of the code point
// -- of the surrogate pair's for the high escape's last digit and the low escape's last three); reads src at q - 10 ... q only
template <class SRC> SJ_HD u32 u_escape_byte(const SRC &src, u32 q, u32 k) {
  const u32 p = q - k; // the 'u'
  const u32 d1 = hex_digit_value(src.byte(p + 1)), d2 = hex_digit_value(src.byte(p + 2));
  const bool surrogate = d1 == 0xDu && (d2 & 8u) != 0u, low = surrogate && (d2 & 4u) != 0u;
  if (k == 2u) {
    if (!low) { return 0xE0u | d1; }
    // second byte of the pair: bits 12 ... 17 of 0x10000 + (h << 10 | l) = bits 2 ... 7 of h + 0x40; the high escape's digits sit at p - 5 ... p - 2
    const u32 x = (((hex_digit_value(src.byte(p - 4)) & 3u) << 8) | (hex_digit_value(src.byte(p - 3)) << 4) | hex_digit_value(src.byte(p - 2))) + 0x40u;
    return 0x80u | ((x >> 2) & 63u);
  }
  const u32 d3 = hex_digit_value(src.byte(p + 3));
  if (k == 3u) {
    if (low) { return 0x80u | ((hex_digit_value(src.byte(p - 2)) & 3u) << 4) | ((d2 & 3u) << 2) | (d3 >> 2); } // third byte of the pair
    if (d1 == 0u && d2 < 8u) { return 0xC0u | (d2 << 2) | (d3 >> 2); }
    return 0x80u | (((d2 << 2) | (d3 >> 2)) & 63u);
  }
  const u32 d4 = hex_digit_value(src.byte(p + 4));
  if (surrogate && !low) { return 0xF0u | (((((d2 & 3u) << 8) | (d3 << 4) | d4) + 0x40u) >> 8); } // first byte of the pair
  if (d1 == 0u && d2 == 0u && d3 < 8u) { return (d3 << 4) | d4; }
  return 0x80u | ((d3 & 3u) << 4) | d4;
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

// ---- one block --------------------------------------------------------------------------------------------------------------------------
// What a block contributes, apart from its quotes: `keep` = the bytes that stand for ONE output byte each if they lie inside a
// string (plain text, the character behind a backslash, the last hex digits of a \u escape); everything else inside a string
// (the backslashes, the rest of a \u escape) is dropped.  Quotes are in neither.
struct string_block {
  u64 keep;
  u64 bad;   // escapes the reference rejects (to be taken seriously inside strings only)
  u64 remap; // kept bytes whose value changes: escaped b f n r t
  u64 bad_u; // \u escapes it rejects (unicode_masks: to be taken seriously wherever they are)
};
// escapes other than \u: from masks alone.  escaped = stage 1's escaped mask, quote = the real quotes.
SJ_HD string_block simple_escapes(u64 backslash, u64 escaped, u64 quote, const escape_classes &c) {
  string_block b;
  b.keep = ~(andn(backslash, escaped) | quote); // not an escaping backslash, not a real quote
  b.bad = escaped & ~(c.u | c.same | c.remap);
  b.remap = escaped & c.remap;
  b.bad_u = 0;
  return b;
}
// no backslash in sight: everything but the quotes
SJ_HD string_block no_escapes(u64 quote) { return string_block{~quote, 0, 0, 0}; }

// the \u escapes of a block on top of the simple ones
SJ_HD void apply_unicode(string_block &b, const u_masks &m) {
  b.keep &= ~m.drop;
  b.bad_u = m.bad;
}

// output bytes of a block given which of its bytes are inside strings (in_string: stage 1's mask, opening quote included,
// closing quote excluded): data bytes + 4 per opening quote + 1 per closing quote
SJ_HD u32 block_output_bytes(const string_block &b, u64 quote, u64 in_string) {
  return u32(popc64(b.keep & in_string)) + 4u * u32(popc64(quote & in_string)) + u32(popc64(andn(quote, in_string)));
}

} // namespace sjgpu
#endif
