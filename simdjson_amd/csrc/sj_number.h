// simdjson_amd/csrc/sj_number.h -- one JSON number / atom token -> what the reference's tape holds for it.
//
// Behaviour restated (not code): numberparsing::parse_number, /root/reference/include/simdjson/generic/numberparsing.h:859-971
// (grammar, the integer ranges and their error codes), write_float :763-814 (range handling), compute_float_64 :65-330 (the
// Eisel-Lemire conversion: D. Lemire, "Number Parsing at a Gigabyte per Second", SPE 2021; with the second product, N. Mushtak
// and D. Lemire, "Fast Number Parsing Without Fallback", SPE 2023, it is exact for every 64-bit decimal significand), and the
// atoms of /root/reference/include/simdjson/generic/atomparsing.h.  Where the reference falls back to from_chars
// (src/from_chars.cpp: more than 19 significant digits) this file brackets the value between the conversions of its first 19
// digits w and of w + 1 -- equal results settle it -- and otherwise compares the full decimal text with the midpoint of the two
// candidate doubles in exact integer arithmetic (decide_long_decimal).  Every route yields THE correctly rounded binary64 (or
// "infinite" = NUMBER_ERROR), which is what the reference produces, so tape words agree bit for bit
// (tests/test_number.py: corner cases, exact halfway points, subnormals, the range limits, random decimals of every length).
//
// All functions are host + device: the tape kernels (sjgpu_tape.hip) and tests/host/test_number.cpp (g++) use the same code.
// A token is read through SRC::byte(pos), which returns 0x20 at and beyond the end of the input -- the reference gives root
// scalars a space-padded copy (tape_builder.h:243-262); no other token can reach the end of a document that passes the
// walk's first check (json_iterator.h:138-143).
#ifndef SJGPU_SJ_NUMBER_H
#define SJGPU_SJ_NUMBER_H

#include "sj_block.h"

namespace sjgpu {

// simdjson::error_code values stage 2 reports (/root/reference/include/simdjson/error.h:19-53)
enum : u32 { SJ_SUCCESS = 0, SJ_CAPACITY = 1, SJ_TAPE_ERROR = 3, SJ_DEPTH_ERROR = 4, SJ_STRING_ERROR = 5, SJ_T_ATOM_ERROR = 6, SJ_F_ATOM_ERROR = 7,
                 SJ_N_ATOM_ERROR = 8, SJ_NUMBER_ERROR = 9, SJ_BIGINT_ERROR = 10, SJ_EMPTY = 13 };

constexpr int POW5_SMALLEST = -342, POW5_LARGEST = 308;
#if defined(__HIPCC__)
static __device__ const u64 d_pow5_128[2 * (POW5_LARGEST - POW5_SMALLEST + 1)] = {
#include "sj_pow5_table.inc"
};
#endif
static const u64 h_pow5_128[2 * (POW5_LARGEST - POW5_SMALLEST + 1)] = {
#include "sj_pow5_table.inc"
};
SJ_HD u64 pow5_word(int q, int which) { // which: 0 = high, 1 = low
#if defined(__HIP_DEVICE_COMPILE__)
  return d_pow5_128[2 * (q - POW5_SMALLEST) + which];
#else
  return h_pow5_128[2 * (q - POW5_SMALLEST) + which];
#endif
}

SJ_HD void mul64x64(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  lo = (u64)p;
  hi = (u64)(p >> 64);
#endif
}
SJ_HD int clz64_nonzero(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

// jsoncharutils::is_not_structural_or_whitespace (/root/reference/src/internal/jsoncharutils_tables.cpp:14-29)
SJ_HD bool not_structural_or_whitespace_spelled(u32 c) {
  return !(c == 0x20u || c == 0x09u || c == 0x0Au || c == 0x0Du || c == ',' || c == ':' || c == '[' || c == ']' || c == '{' || c == '}');
}
// the same from two 32-bit masks (tab, LF, CR | space, ',', ':') and one expression for the four brackets ('[' = 0x5B, ']' = 0x5D, '{' = 0x7B, '}' = 0x7D:
// (c | 0x20) ^ 0x7B is 0 for the opening and 6 for the closing ones, and for no other byte): eight instructions where the ten compares above cost twenty
// (round 6: every number token ends with this test, and the kernel that parses them is bound by its instruction count).  tests/host/test_number.cpp
// compares the two on every byte value.
SJ_HD bool not_structural_or_whitespace(u32 c) {
  const u32 low = c < 32u ? (0x00002600u >> c) & 1u : 0u;                          // 0x09, 0x0A, 0x0D
  const u32 mid = (c - 32u) < 32u ? (0x04001001u >> (c - 32u)) & 1u : 0u;          // 0x20, ',' (0x2C), ':' (0x3A)
  const u32 t = (c | 0x20u) ^ 0x7Bu;                                               // '{' '[' -> 0, '}' ']' -> 6
  return (low | mid | ((t == 0u || t == 6u) ? 1u : 0u)) == 0u;
}

// w * 10^q, w != 0, q in [POW5_SMALLEST, POW5_LARGEST] -> the bits of the nearest binary64 (ties to even), sign excluded.
// false: the result is infinite.
SJ_HD bool decimal_to_binary64(u64 w, int q, u64 &bits) {
  const int lz = clz64_nonzero(w);
  w <<= lz;
  u64 hi, lo;
  mul64x64(w, pow5_word(q, 0), hi, lo);
  if ((hi & 0x1FFu) == 0x1FFu) { // the 55 bits we need are not settled by the first product: add the next 64 bits of 5^q
    u64 hi2, lo2;
    mul64x64(w, pow5_word(q, 1), hi2, lo2);
    lo += hi2;
    if (hi2 > lo) { hi++; }
  }
  const u32 upper = u32(hi >> 63);
  u64 m = hi >> (upper + 9); // 54 bits: 53 + one rounding bit
  // binary exponent of 10^q's leading bit: floor(log2(5^q)) + q = (217706 q) >> 16 for every q of the table
  int e = ((217706 * q) >> 16) + 1024 + 63 - lz - int(1u ^ upper); // (32-bit: |217706 q| < 2^27)
  if (e <= 0) { // subnormal (or zero): shift the rounding position up
    if (-e + 1 >= 64) { bits = 0; return true; }
    m >>= u32(-e + 1);
    m += m & 1u;
    m >>= 1;
    bits = m; // m == 2^52 after rounding IS the smallest normal number: the exponent field is the carry
    return true;
  }
  // exactly half way between two doubles: only possible when 5^|q| fits the significand, and then the truncated product is exact
  if (lo <= 1 && q >= -4 && q <= 23 && (m & 3u) == 1u && (m << (upper + 64 - 53 - 2)) == hi) { m &= ~u64(1); }
  m += m & 1u;
  m >>= 1;
  if (m >= (u64(1) << 53)) { m = u64(1) << 52; e++; }
  if (e > 2046) { return false; }
  bits = (m & ~(u64(1) << 52)) | (u64(e) << 52);
  return true;
}

// ---- more than 19 significant digits: exact comparison with the midpoint of the two candidates ----------------------------------------
// A small unsigned big integer: 800 decimal digits, or a 54-bit midpoint times 5^1124 times a power of two, stay below 2 700 bits;
// 4 096 are provided.
struct bigint {
  static constexpr int LIMBS = 128;
  u32 v[LIMBS];
  int n; // limbs in use (no leading zero limbs; 0 = the value zero)
};
SJ_HD void big_set(bigint &a, u64 x) {
  a.n = 0;
  if (x) { a.v[a.n++] = u32(x); }
  if (x >> 32) { a.v[a.n++] = u32(x >> 32); }
}
SJ_HD void big_mul_add(bigint &a, u32 mul, u32 add) { // a = a * mul + add
  u64 carry = add;
  for (int i = 0; i < a.n; i++) {
    const u64 t = u64(a.v[i]) * mul + carry;
    a.v[i] = u32(t);
    carry = t >> 32;
  }
  if (carry && a.n < bigint::LIMBS) { a.v[a.n++] = u32(carry); }
}
SJ_HD void big_mul_pow5(bigint &a, u32 k) {
  for (; k >= 13; k -= 13) { big_mul_add(a, 1220703125u, 0); } // 5^13
  u32 r = 1;
  for (u32 i = 0; i < k; i++) { r *= 5u; }
  if (r != 1) { big_mul_add(a, r, 0); }
}
SJ_HD void big_shl(bigint &a, u32 bits) {
  if (a.n == 0 || bits == 0) { return; }
  const int limbs = int(bits >> 5);
  const u32 b = bits & 31u;
  int top = a.n + limbs + 1;
  if (top > bigint::LIMBS) { top = bigint::LIMBS; }
  for (int i = top - 1; i >= 0; i--) {
    const int s = i - limbs;
    u32 x = 0;
    if (s >= 0 && s < a.n) { x = a.v[s] << b; }
    if (b && s - 1 >= 0 && s - 1 < a.n) { x |= a.v[s - 1] >> (32u - b); }
    a.v[i] = x;
  }
  a.n = top;
  while (a.n > 0 && a.v[a.n - 1] == 0) { a.n--; }
}
SJ_HD int big_cmp(const bigint &a, const bigint &b) {
  if (a.n != b.n) { return a.n < b.n ? -1 : 1; }
  for (int i = a.n - 1; i >= 0; i--) {
    if (a.v[i] != b.v[i]) { return a.v[i] < b.v[i] ? -1 : 1; }
  }
  return 0;
}

// What a number token is made of (filled by scan_number below).
struct number_shape {
  u32 first_sig;    // position of the first significant (non-zero) digit
  u32 digits_end;   // one past the last digit of the mantissa (integer + fraction part)
  u32 dot;          // position of the '.', or digits_end if there is none
  u32 sig_digits;   // significant digits: all mantissa digits from first_sig on
  long long exp10;  // the value is (all mantissa digits as one integer) * 10^exp10
};

// Sign of (the token's exact value) - (mid * 2^e2), mid odd or even, any size up to 2^54.  0 = exactly equal.
template <class SRC>
SJ_HD int compare_with_midpoint(const SRC &src, const number_shape &s, u64 mid, int e2, bigint &lhs, bigint &rhs) {
  // D = the first KEEP significant digits as an integer; `sticky`: a non-zero digit follows them
  constexpr u32 KEEP = 800;
  big_set(lhs, 0);
  u32 taken = 0, chunk = 0, chunk_digits = 0;
  bool sticky = false;
  for (u32 p = s.first_sig; p < s.digits_end; p++) {
    const u32 c = src.byte(p);
    if (c == '.') { continue; }
    if (taken < KEEP) {
      chunk = chunk * 10u + (c - '0');
      chunk_digits++;
      taken++;
      if (chunk_digits == 9) { big_mul_add(lhs, 1000000000u, chunk); chunk = 0; chunk_digits = 0; }
    } else if (c != '0') {
      sticky = true;
    }
  }
  if (chunk_digits) {
    u32 scale = 1;
    for (u32 i = 0; i < chunk_digits; i++) { scale *= 10u; }
    big_mul_add(lhs, scale, chunk);
  }
  // value = D * 10^p10 (+ something below one unit of D if sticky)
  const long long p10 = s.exp10 + (long long)(s.sig_digits - taken);
  big_set(rhs, mid);
  // compare D * 5^p10 * 2^p10 with mid * 2^e2: powers of five to the side where they are positive, then the powers of two
  long long two_l = 0, two_r = e2; // lhs * 2^two_l  vs  rhs * 2^two_r
  if (p10 >= 0) { big_mul_pow5(lhs, u32(p10)); two_l = p10; }
  else { big_mul_pow5(rhs, u32(-p10)); two_r = (long long)e2 - p10; }
  const long long d = two_l - two_r;
  if (d > 0) { big_shl(lhs, u32(d)); } else if (d < 0) { big_shl(rhs, u32(-d)); }
  const int c = big_cmp(lhs, rhs);
  if (c != 0) { return c; }
  return sticky ? 1 : 0;
}

// The token has more than 19 significant digits.  bits = nearest binary64 of its exact value; false = infinite.
// big: two scratch big integers (1 KB together; the caller decides where they live).
template <class SRC>
SJ_HD bool decide_long_decimal(const SRC &src, const number_shape &s, bigint *big, u64 &bits) {
  // w = the first 19 significant digits, value in [w, w + 1) * 10^q
  u64 w = 0;
  u32 taken = 0;
  for (u32 p = s.first_sig; taken < 19; p++) {
    const u32 c = src.byte(p);
    if (c == '.') { continue; }
    w = w * 10u + (c - '0');
    taken++;
  }
  const long long q = s.exp10 + (long long)(s.sig_digits - 19u);
  if (q > POW5_LARGEST) { return false; }         // w >= 10^18: beyond 10^326
  if (q < POW5_SMALLEST) { bits = 0; return true; } // below 10^19 * 10^-343 = 10^-324 < 2^-1075
  u64 lo_bits = 0, hi_bits = 0;
  const bool lo_finite = decimal_to_binary64(w, int(q), lo_bits);
  if (!lo_finite) { return false; } // even the truncated value rounds to infinity
  const bool hi_finite = decimal_to_binary64(w + 1, int(q), hi_bits);
  if (hi_finite && hi_bits == lo_bits) { bits = lo_bits; return true; }
  // lo_bits and its successor are the candidates; the midpoint between them is (2 m + 1) * 2^(e - 1) with lo = m * 2^e
  const u64 frac = lo_bits & ((u64(1) << 52) - 1);
  const u32 ef = u32(lo_bits >> 52);
  const u64 m = ef ? (frac | (u64(1) << 52)) : frac;
  const int e = ef ? int(ef) - 1075 : -1074;
  const int c = compare_with_midpoint(src, s, 2 * m + 1, e - 1, big[0], big[1]);
  const bool up = c > 0 || (c == 0 && (m & 1u)); // above the midpoint, or on it with an odd significand below
  if (!up) { bits = lo_bits; return true; }
  if (lo_bits == 0x7FEFFFFFFFFFFFFFull) { return false; } // the successor of the largest double is infinity
  bits = lo_bits + 1; // the successor: significand + 1, carrying into the exponent field where needed
  return true;
}

// ---- the token -----------------------------------------------------------------------------------------------------------------------
struct number_value {
  u32 error; // SJ_SUCCESS, SJ_NUMBER_ERROR or SJ_BIGINT_ERROR
  u32 type;  // 'l' (int64), 'u' (uint64), 'd' (double)
  u64 bits;  // the second tape word
  bool slow; // type 'd' only: more than 19 significant digits and the bracket did not settle it -- bits is NOT valid yet,
             // the caller runs decide_long_decimal with scratch big integers (error and the terminator check are final)
};

// parse_number (numberparsing.h:859-971).  allow_slow: run the big-integer decision here (big != nullptr) or report `slow`.
template <class SRC>
SJ_HD number_value parse_number_token(const SRC &src, u32 pos, bigint *big, number_shape *shape_out = nullptr) {
  number_value r{SJ_SUCCESS, 'l', 0, false};
  u32 p = pos;
  const bool negative = src.byte(p) == '-';
  if (negative) { p++; }
  const u32 start_digits = p;
  u64 i = 0;
  u32 c = src.byte(p);
  const u32 first_digit = c; // (kept: the leading-zero rule below asked the source for this byte again)
  // i = i * 10 + digit for every digit, modulo 2^64 (may wrap; the digit count decides below) -- accumulated in 32-bit chunks of up to nine digits: chunk and
  // scale take two full-rate instructions per digit each, the 64-bit multiply-add (two quarter-rate instructions) runs once per chunk instead of once per
  // digit (round 6; multiplication modulo 2^64 is a ring homomorphism: i * 10^k + chunk is the same number)
  u32 chunk = 0, scale = 1;
  auto eat_digit = [&]() {
    chunk = 10u * chunk + (c - '0');
    scale *= 10u;
    c = src.byte(++p);
    if (scale == 1000000000u) { i = i * scale + chunk; chunk = 0; scale = 1; }
  };
  auto flush_digits = [&]() { i = i * scale + chunk; chunk = 0; scale = 1; };
  while (c - '0' <= 9u) { eat_digit(); }
  flush_digits();
  u32 digit_count = p - start_digits;
  if (digit_count == 0 || (first_digit == '0' && digit_count > 1)) { r.error = SJ_NUMBER_ERROR; return r; }
  bool is_float = false;
  number_shape s;
  s.dot = 0xFFFFFFFFu;
  long long exp10 = 0;
  if (c == '.') {
    is_float = true;
    s.dot = p;
    const u32 first = ++p;
    c = src.byte(p);
    while (c - '0' <= 9u) { eat_digit(); }
    flush_digits();
    if (p == first) { r.error = SJ_NUMBER_ERROR; return r; } // "1." (:659-672)
    exp10 = -(long long)(p - first);
  }
  s.digits_end = p;
  if (s.dot == 0xFFFFFFFFu) { s.dot = p; }
  if (c == 'e' || c == 'E') {
    is_float = true;
    c = src.byte(++p);
    const bool neg_exp = c == '-';
    if (neg_exp || c == '+') { c = src.byte(++p); }
    u32 first = p;
    if (!(c - '0' <= 9u)) { r.error = SJ_NUMBER_ERROR; return r; } // no exponent digits (:674-723)
    while (c == '0') { c = src.byte(++p); } // leading zeros carry no value
    first = p;
    long long ev = 0;
    while (c - '0' <= 9u) {
      if (p - first < 18u) { ev = 10 * ev + (long long)(c - '0'); }
      c = src.byte(++p);
    }
    if (p - first > 18u) { ev = 999999999999999999ll; } // more than 18 digits: as good as infinite (:713-716)
    exp10 += neg_exp ? -ev : ev;
  }
  if (is_float) {
    r.type = 'd';
    const bool dirty_end = not_structural_or_whitespace(c);
    // significant digits: everything from the first non-zero digit on.  A text of at most 19 digits needs no second look (round 6: three or four more bytes
    // read per number of ordinary text): i holds them exactly, and all that is asked of their count is "none" (i == 0) and "at most 19"
    const u32 all_digits = (s.digits_end - start_digits) - (s.dot < s.digits_end ? 1u : 0u);
    if (all_digits <= 19u && shape_out == nullptr) {
      s.first_sig = start_digits;
      s.sig_digits = i == 0 ? 0u : all_digits;
    } else {
      u32 q = start_digits;
      while (q < s.digits_end && (src.byte(q) == '0' || src.byte(q) == '.')) { q++; }
      s.first_sig = q;
      s.sig_digits = (s.digits_end - q) - ((s.dot >= q && s.dot < s.digits_end) ? 1u : 0u);
    }
    s.exp10 = exp10;
    if (shape_out) { *shape_out = s; }
    const u64 sign = negative ? (u64(1) << 63) : 0;
    if (s.sig_digits == 0) { // the mantissa is zero: +-0.0 whatever the exponent says (:780-789)
      r.bits = sign;
    } else if (s.sig_digits <= 19) { // i holds the digits exactly
      if (exp10 < POW5_SMALLEST) { r.bits = sign; }
      else if (exp10 > POW5_LARGEST) { r.error = SJ_NUMBER_ERROR; return r; }
      else {
        u64 b;
        if (!decimal_to_binary64(i, int(exp10), b)) { r.error = SJ_NUMBER_ERROR; return r; }
        r.bits = b | sign;
      }
    } else if (big) {
      u64 b;
      if (!decide_long_decimal(src, s, big, b)) { r.error = SJ_NUMBER_ERROR; return r; }
      r.bits = b | sign;
    } else {
      // try the bracket without scratch memory: it settles all but a handful of inputs per billion
      u64 w = 0;
      u32 taken = 0;
      for (u32 t = s.first_sig; taken < 19; t++) {
        const u32 d = src.byte(t);
        if (d == '.') { continue; }
        w = w * 10u + (d - '0');
        taken++;
      }
      const long long q10 = exp10 + (long long)(s.sig_digits - 19u);
      u64 lo_bits = 0, hi_bits = 0;
      if (q10 > POW5_LARGEST) { r.error = SJ_NUMBER_ERROR; return r; }
      if (q10 < POW5_SMALLEST) { r.bits = sign; }
      else if (!decimal_to_binary64(w, int(q10), lo_bits)) { r.error = SJ_NUMBER_ERROR; return r; }
      else if (decimal_to_binary64(w + 1, int(q10), hi_bits) && hi_bits == lo_bits) { r.bits = lo_bits | sign; }
      else { r.slow = true; r.bits = sign; }
    }
    if (dirty_end) { r.error = SJ_NUMBER_ERROR; r.slow = false; }
    return r;
  }
  // integers (:922-968)
  const u32 longest = negative ? 19u : 20u;
  if (digit_count > longest) { r.error = SJ_BIGINT_ERROR; return r; }
  if (digit_count == longest) {
    if (negative) {
      if (i > (u64(1) << 63)) { r.error = SJ_BIGINT_ERROR; return r; }
      r.bits = ~i + 1;
      if (not_structural_or_whitespace(c)) { r.error = SJ_NUMBER_ERROR; }
      return r;
    }
    // 20 digits: only those that start with 1 and did not wrap stay below 2^64
    if (src.byte(pos) != '1' || i <= 0x7FFFFFFFFFFFFFFFull) { r.error = SJ_BIGINT_ERROR; return r; }
  }
  if (i > 0x7FFFFFFFFFFFFFFFull) {
    r.type = 'u';
    r.bits = i;
  } else {
    r.bits = negative ? (~i + 1) : i;
  }
  if (not_structural_or_whitespace(c)) { r.error = SJ_NUMBER_ERROR; }
  return r;
}

// true / false / null: the letters, then a structural or whitespace byte (atomparsing.h; the length-aware forms the reference
// uses for root atoms agree with this one when bytes beyond the end read as spaces)
template <class SRC>
SJ_HD bool atom_matches(const SRC &src, u32 pos, u32 l0, u32 l1, u32 l2, u32 l3, u32 l4 /* 0 = four letters */) {
  if (src.byte(pos) != l0 || src.byte(pos + 1) != l1 || src.byte(pos + 2) != l2 || src.byte(pos + 3) != l3) { return false; }
  if (l4) { return src.byte(pos + 4) == l4 && !not_structural_or_whitespace(src.byte(pos + 5)); }
  return !not_structural_or_whitespace(src.byte(pos + 4));
}

} // namespace sjgpu
#endif
