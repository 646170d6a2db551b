// simdjson_amd/csrc/sj_xcarry.h -- the escape carry INSIDE the scan (no escape table, no walk over backslash runs).
//
// The reference carries one bit from block to block, `next_is_escaped` (/root/reference/src/generic/stage1/json_escape_scanner.h:50-71).
// Here a SPAN (the contiguous bytes one wave scans: 4, 8 or 16 KiB) starts from a 64-byte look-back.  Two look-backs do not
// determine the carries:
//   kind B  all 64 bytes are backslashes: "is my first byte escaped?" depends on a run that began further back;
//   kind C  a quote behind 63 backslashes: "is that quote escaped?" (it decides whether my first byte follows a scalar).
// Such a span ASSUMES the answer is no (B: e_in = 0, p_in = 1; C: e_in = 0, p_in = 0) and scans.  One bit per span boundary
// says whether the assumption was wrong: x.  A wrong assumption changes little (tests/host/test_escape_carry_model.cpp):
//   B  the escapedness of byte L, the first byte behind the span's leading backslash run.  It matters only if that byte is a
//      quote: then every in-string bit behind it flips (pick the other of the two hypotheses the span carries anyway, hand on
//      the other parity: F) and byte L + 1, if it is a scalar, starts a token or stops starting one (one candidate bit: P, d);
//   C  the candidate bit of byte 0, if that byte is a scalar (P = 0, d).
// and x itself obeys x(next) = c ^ (dep & x): a span whose last 64 bytes call for kind B / C in its successor knows the answer
// (c) from its own scan unless it is kind B itself and nothing but backslashes (B next) or backslashes and one final quote
// (C next) -- then the answer is its own x, or the opposite (dep).  So a span's summary is what it always was -- the count for
// both in-string hypotheses, the quote parity, the error bits -- plus one word, the XW below, and summaries compose: tiles,
// groups of segments and whole ranges are summarised by the same words (xs_compose), which is what lets the two resolve kernels
// and the look-back of the single-pass kernels carry x along with the in-string bit.
//
// Host + device: tests/host/test_xcarry_model.cpp runs these functions on bytes against a sequential scan.
#ifndef SJGPU_SJ_XCARRY_H
#define SJGPU_SJ_XCARRY_H

#include "sj_block.h"

namespace sjgpu {

// ---- the x word of a summary ------------------------------------------------------------------------------------------------
constexpr u32 XW_C = 1u;     // x of the successor if this summary's own x is 0 ...
constexpr u32 XW_DEP = 2u;   // ... and whether it flips when this summary's own x is 1
constexpr u32 XW_F = 4u;     // x = 1: the in-string state behind this summary's first quote flips (parity handed on, hypothesis picked)
constexpr u32 XW_D_SHIFT = 3; // bits 3-4: d[0], bits 5-6: d[1] (0, 1, 3 = -1): x = 1 adds d[h] to the count of EFFECTIVE hypothesis h
constexpr u32 XW_LOW_BITS = 7; // everything an aggregate needs; spans add:
constexpr u32 XW_P_SHIFT = 8;  // bits 8-22: P, the span-relative position of the candidate bit x = 1 toggles (d says under which hypothesis)
constexpr u32 XW_LOW_MASK = (1u << XW_LOW_BITS) - 1u;

SJ_HD int xw_d(u32 xw, u32 h) {
  const u32 v = (xw >> (XW_D_SHIFT + 2u * h)) & 3u;
  return v == 3u ? -1 : int(v);
}
SJ_HD u32 xw_enc_d(int d0, int d1) { return ((u32(d0) & 3u) << XW_D_SHIFT) | ((u32(d1) & 3u) << (XW_D_SHIFT + 2u)); }
SJ_HD u32 xw_patch_pos(u32 xw) { return xw >> XW_P_SHIFT; }

// what one summary does to the state (s = in-string, x) in front of it
struct xs_step {
  u32 se;    // the effective hypothesis: which of the two counts / error bits / mask selections apply
  int dcount; // added to count[se]
  u32 s_out, x_out;
};
SJ_HD xs_step xs_apply(u32 parity, u32 xw, u32 s, u32 x) {
  xs_step r;
  const u32 f = x & ((xw >> 2) & 1u);
  r.se = s ^ f;
  r.dcount = x ? xw_d(xw, r.se) : 0;
  r.s_out = s ^ parity ^ f;
  r.x_out = (xw & XW_C) ^ (x & ((xw >> 1) & 1u));
  return r;
}

// ---- a summary as a function of the four states in front of it, and back ----------------------------------------------------
// index = s | x << 1
struct xs_fun {
  u32 cnt[4];
  u32 s_out[4], x_out[4];
  u32 err[4]; // caller-defined error bits under that state
};
// summary: counts c[2] (by effective hypothesis), parity, xw, error bits e[2] (by effective hypothesis)
SJ_HD xs_fun xs_expand(u32 c_out, u32 c_in, u32 parity, u32 xw, u32 e_out, u32 e_in) {
  xs_fun f;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (u32 i = 0; i < 4; i++) {
    const xs_step t = xs_apply(parity, xw, i & 1u, i >> 1);
    f.cnt[i] = (t.se ? c_in : c_out) + u32(t.dcount);
    f.s_out[i] = t.s_out;
    f.x_out[i] = t.x_out;
    f.err[i] = t.se ? e_in : e_out;
  }
  return f;
}
// (selects, not indexed loads: on the device these arrays must stay in registers)
SJ_HD u32 xs_pick(const u32 (&a)[4], u32 j) { return j == 0u ? a[0] : (j == 1u ? a[1] : (j == 2u ? a[2] : a[3])); }
// first a, then b
SJ_HD xs_fun xs_then(const xs_fun &a, const xs_fun &b) {
  xs_fun r;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (u32 i = 0; i < 4; i++) {
    const u32 j = a.s_out[i] | (a.x_out[i] << 1);
    r.cnt[i] = a.cnt[i] + xs_pick(b.cnt, j);
    r.s_out[i] = xs_pick(b.s_out, j);
    r.x_out[i] = xs_pick(b.x_out, j);
    r.err[i] = a.err[i] | xs_pick(b.err, j);
  }
  return r;
}
SJ_HD xs_fun xs_identity() {
  xs_fun f;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (u32 i = 0; i < 4; i++) { f.cnt[i] = 0; f.s_out[i] = i & 1u; f.x_out[i] = i >> 1; f.err[i] = 0; }
  return f;
}
struct xs_summary {
  u32 c_out, c_in, parity, xw, e_out, e_in;
  bool exact; // the function has the compact form (it always has when it is a composition of span summaries: the model test asserts it)
};
SJ_HD xs_summary xs_compact(const xs_fun &f) {
  xs_summary r;
  r.c_out = f.cnt[0];
  r.c_in = f.cnt[1];
  r.parity = f.s_out[0]; // s_out under s = 0, x = 0
  r.e_out = f.err[0];
  r.e_in = f.err[1];
  const u32 F = f.s_out[0] ^ f.s_out[2];
  const u32 c = f.x_out[0], dep = f.x_out[0] ^ f.x_out[2];
  // x = 1, s: effective hypothesis s ^ F, count = c[s ^ F] + d[s ^ F]
  const int d0 = int(F ? f.cnt[3] : f.cnt[2]) - int(r.c_out); // effective hypothesis 0 is reached from s = F
  const int d1 = int(F ? f.cnt[2] : f.cnt[3]) - int(r.c_in);
  r.xw = c | (dep << 1) | (F << 2) | xw_enc_d(d0, d1);
  r.exact = d0 >= -1 && d0 <= 1 && d1 >= -1 && d1 <= 1 && f.s_out[1] == (f.s_out[0] ^ 1u) && f.s_out[3] == (f.s_out[2] ^ 1u) &&
            f.x_out[1] == f.x_out[0] && f.x_out[3] == f.x_out[2] && (F ? f.err[3] : f.err[2]) == r.e_out && (F ? f.err[2] : f.err[3]) == r.e_in;
  return r;
}

// ---- summaries in their compact form, composed directly (what the device keeps in registers: four words) -----------------------------
struct xs_sum {
  u32 q, c_out, c_in, xw; // quote parity and counts under x = 0; the x word (low bits)
};
constexpr u32 XW_IDENTITY = XW_DEP; // the summary of nothing: hands x on, counts nothing, flips nothing
SJ_HD u32 xs_sum_count(const xs_sum &a, const xs_step &t) { return (t.se ? a.c_in : a.c_out) + u32(t.dcount); }
// first a, then b: evaluate the pair on the four states, read the compact form off (exact for everything spans can produce:
// tests/host/test_xcarry_model.cpp composes its groups this way, too)
SJ_HD xs_sum xs_compose(const xs_sum &a, const xs_sum &b) {
  u32 cnt[4], so[4], xo[4];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (u32 i = 0; i < 4; i++) {
    const xs_step t1 = xs_apply(a.q, a.xw, i & 1u, i >> 1);
    const xs_step t2 = xs_apply(b.q, b.xw, t1.s_out, t1.x_out);
    cnt[i] = xs_sum_count(a, t1) + xs_sum_count(b, t2);
    so[i] = t2.s_out;
    xo[i] = t2.x_out;
  }
  const u32 F = so[0] ^ so[2];
  xs_sum r;
  r.q = so[0];
  r.c_out = cnt[0];
  r.c_in = cnt[1];
  r.xw = xo[0] | ((xo[0] ^ xo[2]) << 1) | (F << 2) | xw_enc_d(int(F ? cnt[3] : cnt[2]) - int(cnt[0]), int(F ? cnt[2] : cnt[3]) - int(cnt[1]));
  return r;
}

// ---- the x word of ONE span, from the facts its scan collects -----------------------------------------------------------------
constexpr u32 SPAN_EXACT = 0, SPAN_B = 1, SPAN_C = 2;
struct span_facts {
  u32 kind;          // SPAN_*
  u32 bytes;         // N: the span's size (a multiple of 64)
  u32 lead_open;     // kind B: the span is nothing but backslashes
  u32 L;             // kind B, !lead_open: length of the leading backslash run (< N)
  u32 quote_at_L;    // kind B: byte L is a quote
  u32 scalar_behind; // kind B: L + 1 < N and byte L + 1 is neither whitespace nor an operator (bytes beyond the input are spaces)
  u32 scalar_first;  // kind C: byte 0 is neither whitespace nor an operator
  u32 next_b;        // the last 64 bytes are backslashes
  u32 next_c;        // the last byte is a quote behind 63 backslashes
  u32 e_end, p_end;  // the scan's carries behind the last byte (p_end: 0 when the scan does not track scalars: minify)
  u32 resolved;      // stage 1, split pipeline: the span pinned its own in-string state at its first control character ...
  u32 derived;       // ... to this (relative) value: its mask is final and both counts are equal
};
SJ_HD u32 span_xword(const span_facts &f) {
  u32 c = f.next_b ? f.e_end : (f.next_c ? f.p_end : 0u);
  u32 dep = 0, F = 0, P = 0;
  int d0 = 0, d1 = 0;
  if (f.kind == SPAN_B) {
    if (f.lead_open) {
      dep = f.next_b; // all backslashes (an even number): the successor's carry is mine
    } else if (f.quote_at_L) {
      F = 1;
      if (f.L + 1u == f.bytes) {
        dep = f.next_c; // backslashes and one final quote: the quote is escaped iff my first byte is not
      } else if (f.scalar_behind) {
        // under the assumption byte L is escaped iff L is odd: candidate bit of L + 1 = "L even", string_tail of L + 1 = "L even"
        P = f.L + 1u;
        const u32 t = (f.L & 1u) ^ 1u; // tail = cand under the assumption
        const int delta = t ? -1 : 1;
        if (f.resolved) { d0 = d1 = (t == f.derived) ? delta : 0; }
        else if (t) { d1 = delta; }
        else { d0 = delta; }
      }
    }
  } else if (f.kind == SPAN_C && f.scalar_first) {
    // assumed: the quote in front is real, so byte 0 starts a token (cand = 1) outside the string that just closed ... or inside the one
    // that just opened: string_tail of byte 0 is 0 relative to the span
    P = 0;
    if (f.resolved) { d0 = d1 = (f.derived == 0u) ? -1 : 0; }
    else { d0 = -1; }
  }
  const bool patch = d0 != 0 || d1 != 0;
  return c | (dep << 1) | (F << 2) | xw_enc_d(d0, d1) | ((patch ? P : 0u) << XW_P_SHIFT);
}

} // namespace sjgpu
#endif
