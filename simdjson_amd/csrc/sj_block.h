// simdjson_amd/csrc/sj_block.h -- per-lane math of the MI355X stage-1 kernels.
//
// Mapping: ONE LANE owns ONE 64-byte block of the input (a wave64 owns 4 KiB).  A lane turns its
// 16 dwords into eight 64-bit BIT PLANES (plane k, bit i = bit k of byte i) with a network of
// byte permutes, rotations and bit-field inserts (v_perm_b32 / v_alignbit_b32 / v_bfi_b32, 7.4 VALU ops per dword), after
// which every character class of the reference is a handful of 64-bit logic ops and the
// string/escape algebra runs on the same 64-bit masks the reference's CPU kernels use.
// This is deliberately NOT the reference's pshufb-nibble-table formulation
// (/root/reference/src/haswell.cpp:43-94): gfx950 has no 16-entry byte shuffle, but it has a
// 2-source byte permute and a full-rate bit-field insert, which make the transposition cheap.
//
// What each function reproduces (behaviour, not code) is cited next to it.  All functions are
// __host__ __device__ so tests/host/test_block_math.cpp can check them on the CPU against
// byte-at-a-time definitions; the kernels in sjgpu_kernels.hip are the only product users.
#ifndef SJGPU_SJ_BLOCK_H
#define SJGPU_SJ_BLOCK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SJ_HD __host__ __device__ __forceinline__
#else
#define SJ_HD inline
#endif

namespace sjgpu {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

// v_perm_b32: result byte j = byte sel[j] of the 8-byte value {hi:lo} (0-3 -> lo, 4-7 -> hi).
SJ_HD u32 byte_perm(u32 hi, u32 lo, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  u64 v = (u64(hi) << 32) | lo;
  u32 r = 0;
  for (int j = 0; j < 4; j++) {
    u32 s = (sel >> (8 * j)) & 0xFF;
    r |= u32((v >> (8 * (s & 7))) & 0xFF) << (8 * j);
  }
  return r;
#endif
}
// v_bfi_b32: bits of a where mask is 1, bits of b elsewhere.
SJ_HD u32 bfi(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }

// Any boolean function of three inputs as ONE instruction: gfx950's v_bitop3_b32 takes the function's truth table as an
// immediate.  TT is the function evaluated on the constants A3, B3, C3 below (operand a, b, c), which is the encoding the
// hardware uses.  The character classes further down are trees of these instead of two-input and / or / and-not chains.
constexpr u32 A3 = 0xF0u, B3 = 0xCCu, C3 = 0xAAu;
#define SJ_TT3(expr) ((expr) & 0xFFu)
template <u32 TT> SJ_HD u32 lut3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
  u32 r = 0;
  if (TT & 0x80u) { r |= a & b & c; }
  if (TT & 0x40u) { r |= a & b & ~c; }
  if (TT & 0x20u) { r |= a & ~b & c; }
  if (TT & 0x10u) { r |= a & ~b & ~c; }
  if (TT & 0x08u) { r |= ~a & b & c; }
  if (TT & 0x04u) { r |= ~a & b & ~c; }
  if (TT & 0x02u) { r |= ~a & ~b & c; }
  if (TT & 0x01u) { r |= ~a & ~b & ~c; }
  return r;
#endif
}
template <u32 TT> SJ_HD u64 lut3(u64 a, u64 b, u64 c) {
  return (u64(lut3<TT>(u32(a >> 32), u32(b >> 32), u32(c >> 32))) << 32) | u64(lut3<TT>(u32(a), u32(b), u32(c)));
}

SJ_HD int popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}

// rotate left by n (0 < n < 32): v_alignbit_b32 with both sources the same register
SJ_HD u32 rotl32(u32 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, u32(32 - n));
#else
  return (x << n) | (x >> (32 - n));
#endif
}
// One butterfly of the serial->parallel bit transposition: exchanges a register-index bit with the bit-position bit of weight
// sh.  a and b hold the same byte positions of the two halves of that index bit; x0 receives the fields of width sh that sit
// LOW in their pair (from a in place, from b moved up), x1 the fields that sit high.  Written with two shifts this is four
// instructions; here b is ROTATED once and x1 comes out rotated left by sh -- a debt that is the same for both inputs of every
// later butterfly (it only depends on the index bits already exchanged), that the field masks do not see (their period divides
// every debt that reaches them: stages run 4, 2, 1) and that is paid once at the end: plane k leaves rotated left by k.
SJ_HD void s2p_butterfly(u32 a, u32 b, u32 himask, int sh, u32 &x0, u32 &x1) {
  const u32 u = rotl32(b, sh);
  x0 = bfi(himask, u, a);
  x1 = bfi(himask, a, u);
}

// 32 bytes (8 dwords, little-endian byte order) -> eight 32-bit plane halves: 16 byte permutes, 12 butterflies of three
// instructions, 7 rotations = 59 instructions (rounds 1-4: 12 steps of two permutes, two shifts and two inserts = 72).
// Two permute stages first bring the two low bits of the byte index into the register index -- r[c] holds bytes c, c + 8,
// c + 16, c + 24 -- so that what remains is an 8 x 8 bit-matrix transposition in every byte slot, done by butterflies alone.
SJ_HD void s2p32(const u32 *s, u32 *p) {
  u32 a[8], r[8], x[8], y[8], z[8];
#pragma unroll
  for (int d2 = 0; d2 < 2; d2++) {
#pragma unroll
    for (int d0 = 0; d0 < 2; d0++) { // dwords j and j + 2: even bytes of both, odd bytes of both
      const int j = 4 * d2 + d0;
      a[4 * d2 + d0] = byte_perm(s[j + 2], s[j], 0x06040200u);     // byte index = 0 mod 2
      a[4 * d2 + 2 + d0] = byte_perm(s[j + 2], s[j], 0x07050301u); // 1 mod 2
    }
  }
#pragma unroll
  for (int y0 = 0; y0 < 2; y0++) {
#pragma unroll
    for (int d0 = 0; d0 < 2; d0++) { // the same pair of the lower and the upper 16 bytes
      const u32 lo = a[2 * y0 + d0], hi = a[4 + 2 * y0 + d0];
      r[4 * d0 + y0] = byte_perm(hi, lo, 0x06040200u);     // byte index = 4 d0 + y0 mod 8
      r[4 * d0 + 2 + y0] = byte_perm(hi, lo, 0x07050301u); // 4 d0 + 2 + y0
    }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) { s2p_butterfly(r[c], r[c + 4], 0xF0F0F0F0u, 4, x[c], x[4 + c]); }           // x[4 k2 + (byte index mod 4)]
#pragma unroll
  for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
    for (int c = 0; c < 2; c++) { s2p_butterfly(x[4 * k2 + c], x[4 * k2 + 2 + c], 0xCCCCCCCCu, 2, y[4 * k2 + c], y[4 * k2 + 2 + c]); } // y[4 k2 + 2 k1 + (mod 2)]
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { s2p_butterfly(y[2 * k], y[2 * k + 1], 0xAAAAAAAAu, 1, z[2 * k], z[2 * k + 1]); } // z[k], rotated left by k
  p[0] = z[0];
#pragma unroll
  for (int k = 1; k < 8; k++) { p[k] = rotl32(z[k], 32 - k); }
}

struct planes { u64 b[8]; };

// 64 bytes (16 dwords) -> eight 64-bit planes.
SJ_HD planes transpose64(const u32 *w) {
  u32 lo[8], hi[8];
  s2p32(w, lo);
  s2p32(w + 8, hi);
  planes P;
#pragma unroll
  for (int k = 0; k < 8; k++) { P.b[k] = (u64(hi[k]) << 32) | lo[k]; }
  return P;
}

// Character classes of the x86 reference kernels (/root/reference/src/haswell.cpp:43-94,
// src/icelake.cpp:48-96): whitespace = {20,09,0A,0D}; "operators" =
// {2C,3A,5B,5D,7B,7D} plus 0C and 1A (the |0x20 artefact), never a byte >= 0x80.
struct classes {
  u64 backslash, quote, ws, op, ctrl;
};
// a & ~b as ONE bit-field insert (v_bfi_b32 with a zero source): gfx950's VALU has no and-not, and spelling
// the complement as a separate v_not doubles the cost of every negative literal below.
SJ_HD u64 andn(u64 a, u64 b) {
  return (u64(bfi(u32(b >> 32), 0u, u32(a >> 32))) << 32) | u64(bfi(u32(b), 0u, u32(a)));
}
SJ_HD classes classify(const planes &P) {
  const u64 b0 = P.b[0], b1 = P.b[1], b2 = P.b[2], b3 = P.b[3], b4 = P.b[4], b5 = P.b[5], b6 = P.b[6], b7 = P.b[7];
  classes c;
  // 17 three-input functions per 32-bit half for the five classes (rounds 3-4: 21; the two-input formulation took ~60)
  c.ctrl = lut3<SJ_TT3(~(A3 | B3 | C3))>(b7, b6, b5);                       // 0x00..0x1F
  const u64 g = lut3<SJ_TT3(~(A3 | B3 | C3))>(b7, b6, b4);                  // high nibble 0x0_ or 0x2_
  const u64 h = lut3<SJ_TT3(A3 & ~B3 & ~C3)>(b5, b0, b3);                   // 0x2_ / 0x3_ / 0x6_ / 0x7_ with a low nibble of 0, 2, 4 or 6
  const u64 sq = lut3<SJ_TT3(A3 & B3 & ~C3)>(g, h, b2);                     // 0x20 or 0x22
  const u64 space = andn(sq, b1);                                           // 0x20
  c.quote = sq & b1;                                                        // 0x22
  // 0x09 0x0A 0x0D: high nibble 0, b3 set, low three bits 001, 010 or 101
  const u64 tlc_low = lut3<SJ_TT3((~A3 & (B3 ^ C3)) | (A3 & ~B3 & C3))>(b2, b1, b0);
  const u64 high0_b3 = lut3<SJ_TT3(A3 & ~B3 & C3)>(g, b5, b3);
  c.ws = lut3<SJ_TT3(A3 | (B3 & C3))>(space, high0_b3, tlc_low);
  const u64 high_010 = lut3<SJ_TT3(~A3 & B3 & ~C3)>(b7, b6, b5);
  const u64 mid_111 = lut3<SJ_TT3(A3 & B3 & C3)>(b4, b3, b2);
  c.backslash = lut3<SJ_TT3(A3 & B3 & ~C3)>(high_010, mid_111, b1 | b0);    // 0x5C
  // operators (b5 is "don't care": the x86 kernels compare b|0x20, which also admits 0x0C and 0x1A):
  //   2C/0C x0x0 1100   3A/1A x0x1 1010   5B/7B x1x1 1011   5D/7D x1x1 1101
  // all four: b7 clear, b3 set, b2 != b1; then b6 clear: b0 clear and b4 == b1; b6 set: b4 and b0 set.  The second half as two functions:
  // y = (b0 ? b4 : b4 == b1) keeps what either case needs of b4 and b1, and the case itself is b6 == b0 (b6 set wants b0 set, b6 clear wants it clear)
  const u64 y = lut3<SJ_TT3((C3 & A3) | (~C3 & ~(A3 ^ B3)))>(b4, b1, b0);
  const u64 by_b6 = lut3<SJ_TT3(A3 & ~(B3 ^ C3))>(y, b6, b0);
  const u64 b3_b2ne1 = lut3<SJ_TT3(A3 & (B3 ^ C3))>(b3, b2, b1);
  c.op = lut3<SJ_TT3(A3 & ~B3 & C3)>(b3_b2ne1, b7, by_b6);
  return c;
}

// prefix-XOR over the 64 bit positions (bit i of the result = XOR of bits 0..i).  The reference
// gets this from one carry-less multiply (/root/reference/include/simdjson/haswell/bitmask.h:18-24);
// on a lane it is six shift-xor steps.
SJ_HD u64 prefix_xor(u64 x) {
  x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
  return x;
}

// Escaped-character mask of a block given its backslash mask and the 1-bit carry "the first byte
// is escaped".  A character is escaped iff preceded by an odd-length backslash run; run parity is
// resolved with the add-carry identity the reference uses
// (/root/reference/src/generic/stage1/json_escape_scanner.h:50-71,96-143).
SJ_HD u64 escaped_mask(u64 backslash, u64 first_is_escaped, u64 &next_is_escaped) {
  const u64 ODD = 0xAAAAAAAAAAAAAAAAull;
  u64 potential = backslash & ~first_is_escaped;
  u64 code = (((potential << 1) | ODD) - potential) ^ ODD; // escaping backslashes + the char each one escapes
  next_is_escaped = (code & backslash) >> 63;
  return code ^ (backslash | first_is_escaped);
}

// ---- UTF-8 ------------------------------------------------------------------------------------
// Position-wise well-formedness test on bit planes (decides exactly what the reference's lookup
// algorithm decides, /root/reference/src/generic/stage1/utf8_lookup4_algorithm.h:16-202: RFC 3629,
// no overlongs, no surrogates, <= U+10FFFF).  `carry` packs what the previous 3 bytes demand of
// this block:
//   bit 0      byte[-1] is a 2/3/4-byte lead
//   bits 1-2   byte[-2], byte[-1] is a 3/4-byte lead
//   bits 3-5   byte[-3], byte[-2], byte[-1] is a 4-byte lead
//   bit 6..9   byte[-1] is E0 / ED / F0 / F4
template <class T> struct utf8_leads_t {
  T cont, l234, l34, l4, e0, ed, f0, f4, bad;
};
typedef utf8_leads_t<u64> utf8_leads;
// b[k] = plane k of a group of 32 (T = u32) or 64 (T = u64) consecutive bytes
template <class T> SJ_HD utf8_leads_t<T> utf8_classify_planes(const T *b) {
  const T b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5], b6 = b[6], b7 = b[7];
  utf8_leads_t<T> L;
  L.cont = b7 & ~b6;
  const T l2 = b7 & b6 & ~b5;                 // C0..DF
  const T l3 = b7 & b6 & b5 & ~b4;            // E0..EF
  L.l4 = b7 & b6 & b5 & b4 & ~b3;             // F0..F7
  L.bad = (b7 & b6 & b5 & b4 & b3)            // F8..FF
          | (l2 & ~b4 & ~b3 & ~b2 & ~b1)      // C0, C1 (overlong 2-byte)
          | (L.l4 & b2 & (b1 | b0));          // F5..F7 (> U+10FFFF)
  const T lowz = ~b2 & ~b1 & ~b0;
  L.e0 = l3 & ~b3 & lowz;
  L.ed = l3 & b3 & b2 & ~b1 & b0;
  L.f0 = L.l4 & lowz;
  L.f4 = L.l4 & b2 & ~b1 & ~b0;
  L.l34 = l3 | L.l4;
  L.l234 = l2 | L.l34;
  return L;
}
SJ_HD utf8_leads utf8_classify(const planes &P) { return utf8_classify_planes<u64>(P.b); }
// What this group demands of the next one (independent of its own carry-in).
template <class T> SJ_HD u32 utf8_carry_out(const utf8_leads_t<T> &L) {
  constexpr int W = int(sizeof(T)) * 8;
  return u32(L.l234 >> (W - 1)) | (u32(L.l34 >> (W - 2)) << 1) | (u32(L.l4 >> (W - 3)) << 3) | (u32(L.e0 >> (W - 1)) << 6) |
         (u32(L.ed >> (W - 1)) << 7) | (u32(L.f0 >> (W - 1)) << 8) | (u32(L.f4 >> (W - 1)) << 9);
}
// Mask of offending positions given the previous group's demands.
template <class T> SJ_HD T utf8_errors_planes(const T *b, const utf8_leads_t<T> &L, u32 carry_in) {
  const T b4 = b[4], b5 = b[5];
  const T expect = ((L.l234 << 1) | T(carry_in & 1u)) | ((L.l34 << 2) | T((carry_in >> 1) & 3u)) |
                   ((L.l4 << 3) | T((carry_in >> 3) & 7u));
  const T after_e0 = (L.e0 << 1) | T((carry_in >> 6) & 1u);
  const T after_ed = (L.ed << 1) | T((carry_in >> 7) & 1u);
  const T after_f0 = (L.f0 << 1) | T((carry_in >> 8) & 1u);
  const T after_f4 = (L.f4 << 1) | T((carry_in >> 9) & 1u);
  const T second = (after_e0 & ~b5)          // E0 80..9F : overlong 3-byte
                   | (after_ed & b5)         // ED A0..BF : surrogate
                   | (after_f0 & ~b5 & ~b4)  // F0 80..8F : overlong 4-byte
                   | (after_f4 & (b5 | b4)); // F4 90..BF : > U+10FFFF
  return (expect ^ L.cont) | L.bad | second;
}
SJ_HD u64 utf8_errors(const planes &P, const utf8_leads &L, u32 carry_in) { return utf8_errors_planes<u64>(P.b, L, carry_in); }

// The same carry word computed from three raw bytes (used once per segment for the look-back).
SJ_HD u32 utf8_carry_from_bytes(u32 p3, u32 p2, u32 p1) {
  auto lead234 = [](u32 x) { return u32(x >= 0xC0 && x <= 0xF7); };
  auto lead34 = [](u32 x) { return u32(x >= 0xE0 && x <= 0xF7); };
  auto lead4 = [](u32 x) { return u32(x >= 0xF0 && x <= 0xF7); };
  return lead234(p1) | (lead34(p2) << 1) | (lead34(p1) << 2) | (lead4(p3) << 3) | (lead4(p2) << 4) | (lead4(p1) << 5) |
         (u32(p1 == 0xE0) << 6) | (u32(p1 == 0xED) << 7) | (u32(p1 == 0xF0) << 8) | (u32(p1 == 0xF4) << 9);
}
// Bits of the carry word that mean "a multi-byte sequence is still open" (EOF rule,
// utf8_lookup4_algorithm.h:164-171).
static const u32 UTF8_CARRY_OPEN = 0x3Fu;

// ---- per-block string/structural algebra -----------------------------------
// Everything a block contributes once its three 1-bit carries are known.  `in_string` here is
// RELATIVE to the state at the segment start; the absolute mask is in_string ^ S with S = all-ones
// iff the segment starts inside a string.
struct block_masks {
  u64 cand;        // op | (scalar & ~follows): structural candidates ignoring strings
  u64 string_tail; // in_string ^ quote (relative)
  u64 in_string;   // relative; includes opening quote, excludes closing quote
};
// Step 1 (needs only the escaped mask): real quotes and "non-quote scalar" bytes.  Their bit 63 /
// parity are what a lane hands to its right-hand neighbour.
struct quote_scalar {
  u64 quote;           // unescaped quotes
  u64 scalar;          // neither whitespace nor operator
  u64 nonquote_scalar; // scalar & ~quote
};
SJ_HD quote_scalar quotes_and_scalars(const classes &c, u64 escaped) {
  quote_scalar q;
  q.quote = andn(c.quote, escaped);
  q.scalar = ~(c.ws | c.op);
  q.nonquote_scalar = lut3<SJ_TT3(~(A3 | B3 | C3))>(c.ws, c.op, q.quote);
  return q;
}
// ({hi, lo} >> n) & 0xFFFFFFFF for 0 < n < 32: v_alignbit_b32
SJ_HD u32 funnel_shift_right(u32 hi, u32 lo, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, u32(n));
#else
  return (hi << (32 - n)) | (lo >> n);
#endif
}
// Step 2 (needs the two 1-bit carries of the lane; in_string_carry is 0 or 1), `follows` = nonquote_scalar one byte position on
SJ_HD block_masks finish_block_follows(const classes &c, const quote_scalar &q, u32 in_string_carry, u64 follows) {
  block_masks m;
  m.in_string = prefix_xor(q.quote ^ u64(in_string_carry)); // (the carry as a quote in front of bit 0: it flips everything, itself included)
  m.cand = lut3<SJ_TT3(A3 | (~B3 & ~C3))>(c.op, c.ws, follows); // op | (scalar & ~follows), scalar = ~(ws | op)
  m.string_tail = m.in_string ^ q.quote;
  return m;
}
SJ_HD block_masks finish_block(const classes &c, const quote_scalar &q, u32 in_string_carry, u32 prev_scalar_carry) {
  return finish_block_follows(c, q, in_string_carry, (q.nonquote_scalar << 1) | prev_scalar_carry);
}

} // namespace sjgpu
#endif
