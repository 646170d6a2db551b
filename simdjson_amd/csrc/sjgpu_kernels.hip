// simdjson_amd/csrc/sjgpu_kernels.hip -- gfx950 kernels for simdjson's stage 1 / minify / validate_utf8.
//
// Execution shape (MI355X-first, not a translation of the CPU loop):
//   * one LANE owns one 64-byte block, one WAVE64 owns a 4 KiB chunk, and a workgroup IS one wave,
//     so every cross-block carry of the reference (escape parity, in-string parity, previous-scalar,
//     UTF-8 look-back, output cursor) is a wave ballot / popcount / shuffle -- no LDS barriers, no
//     __syncthreads between waves, nothing shaped like a 32-wide warp;
//   * a wave walks SEG_CHUNKS consecutive chunks (a 16 KiB "segment") carrying the 1-bit states in
//     wave-uniform registers, and obtains the segment's escape / scalar / UTF-8 carry-in by looking
//     back at the bytes in front of it (they are position-wise functions of at most the preceding
//     backslash run), so only TWO quantities need a device-wide scan: the in-string parity and the
//     output cursor.  Each segment publishes its count for BOTH in-string hypotheses, one tiny
//     resolve kernel scans them, and the emit kernel selects;
//   * structural offsets are compacted through a per-wave LDS window and leave as coalesced stores.
// All of it is bit manipulation on the HBM-bound path: no MFMA.
//
// Behavioural contract: SURVEY.md App. A (restated from /root/reference/src/generic/stage1/*.h).
#include "sjgpu_internal.h"
#include "sj_block.h"
#include "sjgpu.h"

namespace sjgpu {
namespace {

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 lanemask_lt(u32 lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ u32 readlane(u32 v, int l) { return u32(__builtin_amdgcn_readlane(int(v), l)); }
__device__ __forceinline__ u32 clz64(u64 x) { return u32(__clzll((long long)x)); }      // x != 0
__device__ __forceinline__ u32 ctz64(u64 x) { return u32(__ffsll((long long)x) - 1); } // x != 0

__device__ __forceinline__ u32 wave_incl_scan(u32 v, u32 lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 t = __shfl_up(v, d);
    if (lane >= u32(d)) { v += t; }
  }
  return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { v += __shfl_xor(v, d); }
  return v;
}

// ---- loads ----------------------------------------------------------------------------------------
// A lane's 64 bytes as 16 dwords.  Bytes at or beyond len read as 0x20, exactly the reference's
// space-padded last block (/root/reference/src/generic/stage1/buf_block_reader.h:99-104); nothing
// past len is touched.
__device__ __forceinline__ void load_block(const u8 *__restrict__ buf, u64 pos, u64 len, u32 (&w)[16]) {
  if (pos + BLOCK_BYTES <= len) {
    const uint4 *p = reinterpret_cast<const uint4 *>(buf + pos);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
    w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
  } else {
    const u32 rem = pos < len ? u32(len - pos) : 0u; // 0..63 real bytes
#pragma unroll
    for (int j = 0; j < 16; j++) {
      u32 v = 0x20202020u;
      if (u32(4 * j) < rem) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (u32(4 * j + k) < rem) { v = (v & ~(0xFFu << (8 * k))) | (u32(buf[pos + 4 * j + k]) << (8 * k)); }
        }
      }
      w[j] = v;
    }
  }
}
// bit i set iff byte pos+i is real input
__device__ __forceinline__ u64 valid_mask(u64 pos, u64 len) {
  if (pos + BLOCK_BYTES <= len) { return ~0ull; }
  if (pos >= len) { return 0ull; }
  return (1ull << u32(len - pos)) - 1ull;
}

// ---- carries ----------------------------------------------------------------------------------------
struct wave_carry {
  u32 e;    // first byte of the next block is escaped
  u32 s;    // inside a string (relative or absolute, caller's choice)
  u32 p;    // previous byte is a non-quote scalar
  u32 utf8; // utf8 carry word (sj_block.h)
};

// parity of the maximal backslash run ending at byte end-1 (0 if end == 0 or no run)
__device__ __forceinline__ u32 backslash_run_parity(const u8 *__restrict__ buf, u64 end, u32 lane) {
  for (;;) {
    const u32 byte = (end > lane) ? u32(buf[end - 1 - lane]) : 0u;
    const u64 m = __ballot(byte == 0x5Cu);
    if (~m) { return ctz64(~m) & 1u; }
    end -= 64; // 64 more backslashes: parity unchanged, keep walking
  }
}

// Carry-in of a segment starting at byte `start`, from the bytes in front of it:
//   e = escaped[start]            = parity of the backslash run ending at start-1
//   p = nonquote_scalar[start-1]  = scalar(b) unless b is '"', then "that quote is escaped"
//   utf8 = demands of bytes start-3..start-1
__device__ __forceinline__ wave_carry segment_carry_in(const u8 *__restrict__ buf, u64 start, u32 lane) {
  wave_carry c{0u, 0u, 0u, 0u};
  if (start == 0) { return c; }
  const u32 byte = (start > lane) ? u32(buf[start - 1 - lane]) : 0x20u;
  const u32 b1 = readlane(byte, 0), b2 = readlane(byte, 1), b3 = readlane(byte, 2);
  c.utf8 = utf8_carry_from_bytes(b3, b2, b1);
  c.e = backslash_run_parity(buf, start, lane);
  if (b1 == 0x22u) {
    c.p = backslash_run_parity(buf, start - 1, lane);
  } else {
    const bool ws = b1 == 0x20u || b1 == 0x09u || b1 == 0x0Au || b1 == 0x0Du;
    const u32 cur = b1 | 0x20u;
    const bool op = b1 < 0x80u && (cur == 0x2Cu || cur == 0x3Au || cur == 0x7Bu || cur == 0x7Du);
    c.p = (ws || op) ? 0u : 1u;
  }
  return c;
}

// ---- one chunk (64 blocks) through the scanner -----------------------------------------------------------
struct chunk_masks {
  u64 cand;        // structural candidates (strings ignored)
  u64 string_tail; // in_string ^ quote
  u64 in_string;   // includes opening quotes, excludes closing quotes
  u64 ws;          // whitespace bytes
  u64 ctrl;        // bytes <= 0x1F
  u64 utf8_err;    // offending positions (0 if WANT_UTF8 is false)
};

template <bool WANT_STRUCTURALS, bool WANT_UTF8>
__device__ __forceinline__ chunk_masks scan_chunk(const u32 (&w)[16], wave_carry &wc, u32 lane) {
  const planes P = transpose64(w);
  const classes c = classify(P);
  const u64 lt = lanemask_lt(lane);
  chunk_masks out;

  // escapes: each block is "pass" (64 backslashes) or sets the carry by itself; a lane's carry-in is
  // the setting of the nearest non-pass lane below it (escape monoid).
  u64 escaped = 0;
  if (__ballot(c.backslash != 0) | u64(wc.e)) { // wave-uniform: most chunks have no backslash at all
    const bool all_bs = (c.backslash == ~0ull);
    const u32 own_out = all_bs ? 0u : (clz64(~c.backslash) & 1u); // parity of the trailing run
    const u64 passm = __ballot(all_bs), setm = __ballot(own_out != 0);
    const u64 below = ~passm & lt;
    const u32 e_in = below ? u32((setm >> (63u - clz64(below))) & 1ull) : wc.e;
    u64 unused;
    escaped = escaped_mask(c.backslash, u64(e_in), unused);
    const u64 nonpass = ~passm;
    wc.e = nonpass ? u32((setm >> (63u - clz64(nonpass))) & 1ull) : wc.e;
  }

  const quote_scalar q = quotes_and_scalars(c, escaped);
  // in-string parity: prefix XOR over lanes by ballot + popcount
  const u64 parm = __ballot((popc64(q.quote) & 1) != 0);
  const u32 s_in = (u32(popc64(parm & lt)) & 1u) ^ wc.s;
  wc.s ^= u32(popc64(parm)) & 1u;
  // previous-scalar: bit 63 of the left neighbour
  u32 p_in = 0;
  if (WANT_STRUCTURALS) {
    const u64 msbm = __ballot((q.nonquote_scalar >> 63) != 0);
    p_in = lane ? u32((msbm >> ((lane - 1u) & 63u)) & 1ull) : wc.p;
    wc.p = u32(msbm >> 63);
  }
  const block_masks m = finish_block(c, q, s_in, p_in);
  out.cand = m.cand;
  out.string_tail = m.string_tail;
  out.in_string = m.in_string;
  out.ws = c.ws;
  out.ctrl = c.ctrl;

  out.utf8_err = 0;
  if (WANT_UTF8) {
    if (__ballot(P.b[7] != 0) | u64(wc.utf8)) { // wave-uniform ASCII fast path
      const utf8_leads L = utf8_classify(P);
      const u32 co = utf8_carry_out(L);
      u32 ci = __shfl_up(co, 1);
      if (lane == 0) { ci = wc.utf8; }
      out.utf8_err = utf8_errors(P, L, ci);
      wc.utf8 = readlane(co, 63);
    }
  }
  return out;
}

// =====================================================================================================
// stage 1, kernel 1: scan every segment once, keep the per-block masks, publish the segment summary
// =====================================================================================================
__global__ __launch_bounds__(64) void k_stage1_summarize(const u8 *__restrict__ buf, u64 len, uint4 *__restrict__ masks,
                                                         seg_summary *__restrict__ summ) {
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = u64(seg) * SEG_BYTES;
  wave_carry wc = segment_carry_in(buf, seg_start, lane);
  u32 n_total = 0, n_in_tail = 0;
  u64 ctrl_in = 0, ctrl_out = 0, uerr = 0;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    load_block(buf, pos, len, w);
    const chunk_masks m = scan_chunk<true, true>(w, wc, lane);
    n_total += u32(popc64(m.cand));
    n_in_tail += u32(popc64(m.cand & m.string_tail));
    ctrl_in |= m.ctrl & m.in_string;   // offends if the relative view is the true one
    ctrl_out |= m.ctrl & ~m.in_string; // offends if the segment really starts inside a string
    uerr |= m.utf8_err;
    masks[pos / BLOCK_BYTES] = make_uint4(u32(m.cand), u32(m.cand >> 32), u32(m.string_tail), u32(m.string_tail >> 32));
  }
  // a multi-byte sequence still open at the very end of the input (utf8_lookup4_algorithm.h:164-171);
  // when len is not a multiple of the chunk the space padding has already flagged it.
  if (seg_start + SEG_BYTES >= len && (wc.utf8 & UTF8_CARRY_OPEN)) { uerr |= 1; }
  const u32 tot = wave_sum(n_total), tail = wave_sum(n_in_tail);
  u32 flags = wc.s ? SF_PARITY : 0u;
  if (__ballot(ctrl_in != 0)) { flags |= SF_CTRL_IF_OUT; }
  if (__ballot(ctrl_out != 0)) { flags |= SF_CTRL_IF_IN; }
  if (__ballot(uerr != 0)) { flags |= SF_UTF8; }
  if (lane == 0) {
    seg_summary s;
    s.count_if_out = tot - tail; // structural = cand & ~string_tail
    s.count_if_in = tail;        // starting inside a string flips string_tail
    s.flags = flags;
    s.pad = 0;
    summ[seg] = s;
  }
}

// =====================================================================================================
// resolve: device-wide scan of (parity, count) over segments.  One workgroup; the summaries are
// 16 B per 16 KiB of input, i.e. 0.1 % of the traffic.
// =====================================================================================================
constexpr u32 RESOLVE_THREADS = 1024;

__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *sh, bool use_xor, u32 &total) {
  const u32 tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (u32 d = 1; d < RESOLVE_THREADS; d <<= 1) {
    const u32 t = (tid >= d) ? sh[tid - d] : 0u;
    __syncthreads();
    sh[tid] = use_xor ? (sh[tid] ^ t) : (sh[tid] + t);
    __syncthreads();
  }
  const u32 incl = sh[tid];
  total = sh[RESOLVE_THREADS - 1];
  __syncthreads();
  return use_xor ? (incl ^ v) : (incl - v);
}

// what: 0 = stage1 (writes n, flags and the three sentinels), 1 = minify (writes out_len, flags)
__global__ __launch_bounds__(RESOLVE_THREADS) void k_resolve_segments(const seg_summary *__restrict__ summ,
                                                                     seg_prefix *__restrict__ pref, u32 nseg, u64 len,
                                                                     u32 *__restrict__ idx, u64 idx_words,
                                                                     scan_result_dev *__restrict__ result, int what) {
  __shared__ u32 sh[RESOLVE_THREADS];
  __shared__ u32 sh_flags;
  const u32 tid = threadIdx.x;
  if (tid == 0) { sh_flags = 0; }
  const u32 per = (nseg + RESOLVE_THREADS - 1) / RESOLVE_THREADS;
  const u32 lo = min(tid * per, nseg), hi = min(lo + per, nseg);
  // pass 1: quote parity
  u32 par = 0;
  for (u32 i = lo; i < hi; i++) { par ^= summ[i].flags & SF_PARITY; }
  u32 final_parity;
  u32 s = block_excl_scan(par, sh, true, final_parity);
  // pass 2: counts under the now-known in-string state
  u32 cnt = 0, st = s, flags = 0;
  for (u32 i = lo; i < hi; i++) {
    const seg_summary x = summ[i];
    cnt += st ? x.count_if_in : x.count_if_out;
    if (x.flags & (st ? SF_CTRL_IF_IN : SF_CTRL_IF_OUT)) { flags |= SJGPU_F_UNESCAPED_CTRL; }
    if (x.flags & SF_UTF8) { flags |= SJGPU_F_UTF8_ERROR; }
    st ^= x.flags & SF_PARITY;
  }
  u32 total;
  u32 base = block_excl_scan(cnt, sh, false, total);
  if (flags) { atomicOr(&sh_flags, flags); }
  // pass 3: per-segment carry-in
  st = s;
  for (u32 i = lo; i < hi; i++) {
    const seg_summary x = summ[i];
    seg_prefix p;
    p.base = base;
    p.in_string = st;
    pref[i] = p;
    base += st ? x.count_if_in : x.count_if_out;
    st ^= x.flags & SF_PARITY;
  }
  __syncthreads();
  if (tid == 0) {
    u32 f = sh_flags | (final_parity ? SJGPU_F_UNCLOSED_STRING : 0u);
    if (what == 0) {
      if (u64(total) + 3 <= idx_words) { // sentinels (json_structural_indexer.h:284-286)
        idx[total] = u32(len);
        idx[total + 1] = u32(len);
        idx[total + 2] = 0;
      } else {
        f |= SJGPU_F_IDX_OVERFLOW;
      }
      result->n = total;
      result->out_len = 0;
    } else {
      result->n = 0;
      result->out_len = final_parity ? 0ull : u64(total); // unclosed string voids the output (json_minifier.h:42-47)
    }
    atomicOr(&result->flags, f);
  }
}

// =====================================================================================================
// stage 1, kernel 2: select the right hypothesis per segment, flatten bitmaps to ascending offsets
// =====================================================================================================
constexpr u32 EMIT_WINDOW = 2048; // offsets staged per wave in LDS (8 KiB); dense chunks take two rounds

__global__ __launch_bounds__(64) void k_stage1_emit(const uint4 *__restrict__ masks, const seg_prefix *__restrict__ pref,
                                                    u64 len, u32 *__restrict__ idx, u64 idx_words,
                                                    scan_result_dev *__restrict__ result) {
  __shared__ u32 stage[EMIT_WINDOW];
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = u64(seg) * SEG_BYTES;
  const seg_prefix pf = pref[seg];
  u32 base = pf.base;
  const u64 flip = pf.in_string ? ~0ull : 0ull;
  bool overflow = false;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    const uint4 mk = masks[pos / BLOCK_BYTES];
    const u64 cand = (u64(mk.y) << 32) | mk.x, tail = (u64(mk.w) << 32) | mk.z;
    const u64 structural = cand & ~(tail ^ flip);
    u32 lo = u32(structural), hi = u32(structural >> 32);
    const u32 cnt = u32(__popc(lo) + __popc(hi));
    const u32 incl = wave_incl_scan(cnt, lane);
    const u32 total = readlane(incl, 63);
    u32 off = incl - cnt;
    const u32 p32 = u32(pos);
    for (u32 w0 = 0; w0 < total; w0 += EMIT_WINDOW) { // wave-uniform
      const u32 lim = w0 + EMIT_WINDOW;
      while (off < lim) {
        u32 v;
        if (lo) { v = p32 + u32(__ffs(int(lo)) - 1); lo &= lo - 1; }
        else if (hi) { v = p32 + 32u + u32(__ffs(int(hi)) - 1); hi &= hi - 1; }
        else { break; }
        stage[off - w0] = v;
        off++;
      }
      __syncthreads();
      const u32 here = min(EMIT_WINDOW, total - w0);
      for (u32 i = lane; i < here; i += 64) {
        const u64 slot = u64(base) + w0 + i;
        if (slot < idx_words) { idx[slot] = stage[i]; } else { overflow = true; }
      }
      __syncthreads();
    }
    base += total;
  }
  if (__ballot(overflow) && lane == 0) { atomicOr(&result->flags, SJGPU_F_IDX_OVERFLOW); }
}

// =====================================================================================================
// minify: summarize (kept-byte counts for both hypotheses) -> resolve -> re-scan + byte compaction
// =====================================================================================================
__global__ __launch_bounds__(64) void k_minify_summarize(const u8 *__restrict__ buf, u64 len, seg_summary *__restrict__ summ) {
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = u64(seg) * SEG_BYTES;
  wave_carry wc = segment_carry_in(buf, seg_start, lane);
  u32 kept_out = 0, kept_in = 0;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    load_block(buf, pos, len, w);
    const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
    const u64 valid = valid_mask(pos, len);
    kept_out += u32(popc64(valid & ~(m.ws & ~m.in_string))); // dropped: whitespace outside strings (json_scanner.h:46)
    kept_in += u32(popc64(valid & ~(m.ws & m.in_string)));
  }
  const u32 a = wave_sum(kept_out), b = wave_sum(kept_in);
  if (lane == 0) {
    seg_summary s;
    s.count_if_out = a;
    s.count_if_in = b;
    s.flags = wc.s ? SF_PARITY : 0u;
    s.pad = 0;
    summ[seg] = s;
  }
}

constexpr u32 MINIFY_STAGE_BYTES = CHUNK_BYTES + 16;

__global__ __launch_bounds__(64) void k_minify_emit(const u8 *__restrict__ buf, u64 len, const seg_prefix *__restrict__ pref,
                                                    u8 *__restrict__ dst) {
  __shared__ __attribute__((aligned(16))) u8 stage[MINIFY_STAGE_BYTES];
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = u64(seg) * SEG_BYTES;
  const seg_prefix pf = pref[seg];
  wave_carry wc = segment_carry_in(buf, seg_start, lane);
  wc.s = pf.in_string; // absolute from here on
  u32 base = pf.base;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    load_block(buf, pos, len, w);
    const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
    const u64 keep = valid_mask(pos, len) & ~(m.ws & ~m.in_string);
    const u32 cnt = u32(popc64(keep));
    const u32 incl = wave_incl_scan(cnt, lane);
    const u32 total = readlane(incl, 63);
    // stage the kept bytes so that LDS offset and destination address agree modulo 16
    const u32 skew = base & 15u;
    u32 o = skew + (incl - cnt);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const u32 k4 = u32(keep >> (4 * j)) & 0xFu;
      const u32 x = w[j];
      if (k4 & 1u) { stage[o] = u8(x); o++; }
      if (k4 & 2u) { stage[o] = u8(x >> 8); o++; }
      if (k4 & 4u) { stage[o] = u8(x >> 16); o++; }
      if (k4 & 8u) { stage[o] = u8(x >> 24); o++; }
    }
    __syncthreads();
    // bytes [skew, skew+total) of the stage go to dst[base ...]; whole 16-byte vectors in the middle
    u8 *const gbase = dst + (u64(base) - skew);
    const u32 end = skew + total;
    const u32 v_first = (skew + 15u) / 16u, v_last = end / 16u; // full vectors [v_first, v_last)
    if (v_last > v_first) {
      for (u32 v = v_first + lane; v < v_last; v += 64) {
        *reinterpret_cast<uint4 *>(gbase + 16u * v) = *reinterpret_cast<const uint4 *>(stage + 16u * v);
      }
      const u32 head_end = 16u * v_first; // bytes [skew, head_end)
      if (skew + lane < head_end) { gbase[skew + lane] = stage[skew + lane]; }
      const u32 tail_start = 16u * v_last; // bytes [tail_start, end)
      if (tail_start + lane < end) { gbase[tail_start + lane] = stage[tail_start + lane]; }
    } else { // fewer than one full vector: plain byte copy
      for (u32 i = skew + lane; i < end; i += 64) { gbase[i] = stage[i]; }
    }
    __syncthreads();
    base += total;
  }
}

// =====================================================================================================
// validate_utf8: stateless (3-byte look-back), pure HBM read
// =====================================================================================================
__global__ __launch_bounds__(64) void k_validate_utf8(const u8 *__restrict__ buf, u64 len, scan_result_dev *__restrict__ result) {
  const u32 lane = lane_id();
  const u64 nchunks = (len + CHUNK_BYTES - 1) / CHUNK_BYTES;
  bool bad = false;
  for (u64 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const u64 cstart = ch * CHUNK_BYTES;
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    load_block(buf, pos, len, w);
    u32 hi = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { hi |= w[j]; }
    u32 c0 = 0;
    if (ch > 0) {
      const u32 pw = *reinterpret_cast<const u32 *>(buf + cstart - 4);
      c0 = utf8_carry_from_bytes((pw >> 8) & 0xFFu, (pw >> 16) & 0xFFu, pw >> 24);
    }
    if (__ballot((hi & 0x80808080u) != 0) == 0) { // all-ASCII chunk
      if (c0 & UTF8_CARRY_OPEN) { bad = true; }
      continue;
    }
    const planes P = transpose64(w);
    const utf8_leads L = utf8_classify(P);
    const u32 co = utf8_carry_out(L);
    u32 ci = __shfl_up(co, 1);
    if (lane == 0) { ci = c0; }
    if (utf8_errors(P, L, ci) != 0) { bad = true; }
    if (ch == nchunks - 1 && (readlane(co, 63) & UTF8_CARRY_OPEN)) { bad = true; } // sequence open at EOF
  }
  if (__ballot(bad) && lane == 0) { atomicOr(&result->flags, SJGPU_F_UTF8_ERROR); }
}

} // namespace

// ---- launchers ----------------------------------------------------------------------------------------
static inline void mark(hipEvent_t *ev, int k, hipStream_t stream) {
  if (ev) { (void)hipEventRecord(ev[k], stream); }
}

void launch_stage1(const uint8_t *buf, uint64_t len, uint4 *masks, seg_summary *summ, seg_prefix *pref, uint32_t *idx,
                   uint64_t idx_words, scan_result_dev *result, hipStream_t stream, hipEvent_t *ev) {
  const u32 nseg = num_segments(len);
  (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
  mark(ev, 0, stream);
  hipLaunchKernelGGL(k_stage1_summarize, dim3(nseg), dim3(64), 0, stream, buf, len, masks, summ);
  mark(ev, 1, stream);
  hipLaunchKernelGGL(k_resolve_segments, dim3(1), dim3(RESOLVE_THREADS), 0, stream, summ, pref, nseg, len, idx, idx_words,
                     result, 0);
  mark(ev, 2, stream);
  hipLaunchKernelGGL(k_stage1_emit, dim3(nseg), dim3(64), 0, stream, masks, pref, len, idx, idx_words, result);
  mark(ev, 3, stream);
}

void launch_minify(const uint8_t *buf, uint64_t len, seg_summary *summ, seg_prefix *pref, uint8_t *dst,
                   scan_result_dev *result, hipStream_t stream, hipEvent_t *ev) {
  const u32 nseg = num_segments(len);
  (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
  mark(ev, 0, stream);
  hipLaunchKernelGGL(k_minify_summarize, dim3(nseg), dim3(64), 0, stream, buf, len, summ);
  mark(ev, 1, stream);
  hipLaunchKernelGGL(k_resolve_segments, dim3(1), dim3(RESOLVE_THREADS), 0, stream, summ, pref, nseg, len,
                     static_cast<u32 *>(nullptr), u64(0), result, 1);
  mark(ev, 2, stream);
  hipLaunchKernelGGL(k_minify_emit, dim3(nseg), dim3(64), 0, stream, buf, len, pref, dst);
  mark(ev, 3, stream);
}

void launch_validate_utf8(const uint8_t *buf, uint64_t len, scan_result_dev *result, hipStream_t stream, hipEvent_t *ev) {
  (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
  const u64 nchunks = (len + CHUNK_BYTES - 1) / CHUNK_BYTES;
  const u32 grid = u32(nchunks < 8192 ? nchunks : 8192); // 256 CUs x 32 single-wave workgroups, grid-stride beyond
  mark(ev, 0, stream);
  hipLaunchKernelGGL(k_validate_utf8, dim3(grid), dim3(64), 0, stream, buf, len, result);
  mark(ev, 1, stream);
  mark(ev, 2, stream);
  mark(ev, 3, stream);
}

} // namespace sjgpu
