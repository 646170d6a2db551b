// simdjson_amd/csrc/sjgpu_kernels.hip -- gfx950 kernels: the SPLIT pipeline (summarize -> resolve -> emit) for
// stage 1 and minify, and validate_utf8.  The single-pass pipeline lives in sjgpu_fused.hip; this one is
// its fallback and A/B partner (SJGPU_PIPELINE=split).
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
#include "sjgpu_device.h"

#include <cstdlib>

namespace sjgpu {
namespace {

// =====================================================================================================
// stage 1, kernel 1: scan every segment once, keep the per-block masks, publish the segment summary
// =====================================================================================================
// Local resolution of the in-string state: a raw control character (<= 0x1F) inside a string is an error
// (UNESCAPED_CHARS, json_structural_indexer.h:240,261-263) after which callers never look at the indexes, so
// at the first control character of a segment the string state MUST be "outside" for any observable output.
// That pins the segment's carry-in:  s_in = (relative in-string bit at that character).  When the first chunk
// of a segment contains such a character (pretty-printed JSON, NDJSON, large_random: virtually always) the
// segment is RESOLVED: it writes ONE final 64-bit mask per block instead of two and its count has a single
// value.  The true parity chain is still scanned: a resolved segment whose true carry-in differs from the
// derived one contains a control character inside a string and raises the same error the reference raises.
// Four independent waves (four consecutive segments) share a workgroup for ONE reason: the UTF-8 blocks they have
// noted but not yet validated when their segments end.  A wave of its own would validate a handful of blocks with a
// handful of lanes (one or two non-ASCII blocks per chunk is what NDJSON / pretty-printed text hold) -- ~75 of its
// ~415 VALU instructions per chunk (profiles/r02_pmc_ndjson.txt: this kernel issues 81 % of the time).  The four
// leave their left-over rows where they are and go; wave 0 validates all of them together.
constexpr u32 SUMM_WAVES = 4;
// TOKENS (round 5, sjgpu_stage1_tokens_device): the segment's structural BYTES leave with its masks.  A lane holds the 64 bytes of its block and, in a
// resolved segment, their final structural mask: the kept bytes of every chunk are compacted through a per-wave LDS window (emit_bytes, the minifier's
// compaction) into the segment's staging area tokstage[seg * SEG_BYTES ...], in list order; k_stage1_emit, which learns where the segment's offsets go, copies
// them behind the same cursor.  Every later pass over the LIST then reads one byte per structural instead of fetching the whole document through 128-byte
// lines to pick that byte.  Segments that could not resolve themselves (no control character in their first chunk: minified text) stage
// nothing (no SF_TOKENS in their summary): emit gathers their bytes from the document.  Costs this kernel the compaction (~170 VALU per chunk on top of
// ~410) and 16.5 KiB more LDS per workgroup (47 KiB: THREE workgroups per CU instead of five): the token stream is opt-in.
// Occupancy request = what the LDS allows (160 KiB per CU): 31 KiB -> 5 workgroups of four waves = 5 waves per SIMD (96 VGPRs each), 47 KiB -> 3 (128 VGPRs).
// (Rounds 4-5 asked for 6 and 4, which the LDS forbids: the compiler warned and fell back to these budgets by itself.)
template <bool TOKENS>
__global__ __launch_bounds__(64 * SUMM_WAVES) SJ_WAVES_PER_EU(TOKENS ? 3 : 5, TOKENS ? 3 : 5) void k_stage1_summarize(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0,
                                                                     u64 *__restrict__ mask1, seg_summary *__restrict__ summ,
                                                                     scan_origin org, u32 nseg, u8 *__restrict__ tokstage) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  const u32 seg = blockIdx.x * SUMM_WAVES + wave; // relative to the scan's origin: workspace index
  __shared__ __attribute__((aligned(16))) u32 park[SUMM_WAVES][UTF8P_ROWS * UTF8P_ROW_WORDS]; // the blocks the UTF-8 check still has to look at (utf8_park)
  __shared__ u32 sh_left[SUMM_WAVES]; // rows every wave has left when its segment ends
  __shared__ __attribute__((aligned(16))) uint4 xbuf[SUMM_WAVES][CHUNK_BYTES / 16]; // load_chunk_stream's exchange buffer of each wave
  __shared__ __attribute__((aligned(16))) u8 tok_window[TOKENS ? SUMM_WAVES : 1][TOKENS ? MINIFY_STAGE_BYTES : 16]; // TOKENS: the compaction window of each wave
  __shared__ u32 tok_lut[MINIFY_LUT_WORDS];
  if (TOKENS) { // (every wave fills the table with the same words: no barrier needed before its own use)
    init_compaction_lut(tok_lut, lane);
    clear_minify_stage(tok_window[wave], lane);
  }
  u32 tok_run = 0; // TOKENS: structural bytes of this segment staged so far (wave-uniform)
  const bool more = (org.carry & CARRY_MORE) != 0;
  if (seg >= nseg) { // past the last segment (wave-uniform; wave 0 always has one): only the rendezvous below
    if (lane == 0) { sh_left[wave] = 0; }
    lds_writes_done();
    __syncthreads();
    return;
  }
  const u64 seg_start = org.begin + u64(seg) * SEG_BYTES;
  const u64 lane_off = u64(lane) * BLOCK_BYTES;
  const u32 lookback = lookback_issue(buf, seg_start, lane); // consumed after chunk 0 has been requested
  wave_carry wc{0u, 0u, 0u};
  span_x sx;
  // (the in-line check of dense chunks is part of the design here: it bounds the rows a chunk can park)
  utf8_park uq{park[wave], 0u, 0u, 0u, 0x20202020u, buf, len, more ? 1u : 0u, UTF8P_DENSE_FROM};
  if (((org.carry >> 16) & 0xFFu) != 0u && ((org.carry >> 16) & 0xFFu) < UTF8P_DENSE_FROM) { uq.dense_from = (org.carry >> 16) & 0xFFu; } // A/B: env SJGPU_UTF8_DENSE_FROM (downwards only)
  u32 n_a = 0, n_b = 0; // resolved: n_a = final count; else n_a = candidates, n_b = candidates in a string tail
  bool any_a = false, any_b = false; // wave-uniform: a control character offends under hypothesis a / b (folded per chunk: two
                                     // compares instead of four VGPRs of masks carried through the segment)
  bool resolved = false;
  u32 derived = 0;
  u64 flip = 0;
  // The segment's masks leave in ONE burst behind the last chunk (32 bytes per lane and plane, lane-major), not chunk
  // by chunk: an 8-byte store in front of the next chunk's loads sits in the same in-order counter those loads are
  // waited on, so every chunk paid the store's round trip (loads + OR + such a store: 4.6 TB/s, without: 5.9).
  u64 keep0[SEG_CHUNKS] = {0, 0, 0, 0}, keep1[SEG_CHUNKS] = {0, 0, 0, 0};
#pragma unroll
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + lane_off;
    u32 w[16];
    if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf[wave], w); } // interior chunk: coalesced streaming loads, block order through LDS
    else { load_block(buf, pos, len, w); }
    if (c == 0) { // the look-back's latency hid behind the loads above
      wc = span_carry_assume(seg_start, lane, lookback, sx);
      utf8_park_begin(uq, lookback, lane);
    }
    span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
    const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, u32(cstart / BLOCK_BYTES));
    if (c == SEG_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
    if (c == 0) {
      const u64 cm = __ballot(m.ctrl != 0);
      if (cm) { // wave-uniform
        const u32 lc = ctz64(cm);
        u32 v = 0;
        if (lane == lc) { v = u32(m.in_string >> ctz64(m.ctrl)) & 1u; }
        derived = readlane_dyn(v, lc);
        resolved = true;
        flip = derived ? ~0ull : 0ull;
      }
    }
    if (resolved) {
      const u64 structural = m.cand & ~(m.string_tail ^ flip);
      n_a += u32(popc64(structural));
      any_a |= __ballot((m.ctrl & (m.in_string ^ flip)) != 0) != 0;
      keep0[c] = structural;
      if (TOKENS) { emit_bytes(w, structural, lane, tokstage + size_t(seg) * SEG_BYTES, tok_run, tok_window[wave], tok_lut); }
    } else {
      n_a += u32(popc64(m.cand));
      n_b += u32(popc64(m.cand & m.string_tail));
      any_a |= __ballot((m.ctrl & m.in_string) != 0) != 0;  // offends if the relative view is the true one
      any_b |= __ballot((m.ctrl & ~m.in_string) != 0) != 0; // offends if the segment really starts inside a string
      keep0[c] = m.cand;
      keep1[c] = m.string_tail;
    }
  }
  // left-overs (fewer than UTF8P_DRAIN_AT per wave) are validated by wave 0 with full lanes, straight from the four waves' rows
  if (lane == 0) { sh_left[wave] = uq.count; }
  lds_writes_done();
  __syncthreads();
  if (wave == 0) {
    const u32 n0 = sh_left[0], n1 = n0 + sh_left[1], n2 = n1 + sh_left[2], total = n2 + sh_left[3];
    for (u32 done = 0; done < total; done += 64) {
      const u32 g = done + lane;
      bool bad = false;
      if (g < total) {
        const u32 v = (g >= n0) + (g >= n1) + (g >= n2);
        const u32 r = g - (v == 0 ? 0u : (v == 1 ? n0 : (v == 2 ? n1 : n2)));
        bad = utf8_check_row(park[v] + r * UTF8P_ROW_WORDS, len, more); // incl. a sequence open at the end of the input
      }
      if (__ballot(bad)) { uq.error = 1u; }
    }
  }
  const u32 ta = wave_sum(n_a), tb = wave_sum(n_b);
  u32 flags = wc.s ? SF_PARITY : 0u;
  if (uq.error) { flags |= SF_UTF8; }
  seg_summary s;
  if (resolved) {
    flags |= SF_RESOLVED;
    if (TOKENS) { flags |= SF_TOKENS; }
    s.count_if_out = ta;
    s.count_if_in = ta;
    // true carry-in == derived: error iff a control character sits inside a string in the resolved view;
    // true carry-in != derived: the resolving control character itself is inside a string
    const bool ok_view = !any_a;
    if (derived ? true : !ok_view) { flags |= SF_CTRL_IF_OUT; }
    if (derived ? !ok_view : true) { flags |= SF_CTRL_IF_IN; }
  } else {
    s.count_if_out = ta - tb; // structural = cand & ~string_tail
    s.count_if_in = tb;       // starting inside a string flips string_tail
    if (any_a) { flags |= SF_CTRL_IF_OUT; }
    if (any_b) { flags |= SF_CTRL_IF_IN; }
  }
  s.flags = flags;
  s.xw = span_finish(sx, buf, seg_start, SEG_BYTES, len, wc, true, resolved, derived);
  // round 6: a handful of candidates leave as a list, not as planes (below); not for segments whose x word patches a candidate bit (the patch is defined on planes)
  const bool sparse = !TOKENS && ta != 0u && ta <= SPARSE_MAX && ((s.xw >> XW_D_SHIFT) & 0xFu) == 0u; // wave-uniform (ta: resolved: the final count; else every candidate)
  if (sparse) { s.flags |= SF_SPARSE; }
  if (lane == 0) { summ[seg] = s; } // (in front of the masks: whatever this store has to wait for -- span_finish may have read bytes -- is waited for before they are issued)
  // The masks leave LAST, behind the rendezvous and behind the summary: nothing waits behind them, the wave ends with its stores in flight (in front of the
  // barrier its fence made every wave wait for their acknowledgement -- this ISA counts loads and stores in one in-order counter -- which measured no
  // different once both orders ran in one process; what the masks cost is their traffic: a lab build that writes none runs 189 against 258 us per GiB of
  // NDJSON, and a pure mover pays the same for them -- scripts/micro/split_lab.hip).  They are written with the non-temporal hint: read once, by k_stage1_emit.
  // A segment without a single candidate (the inside of a long string, of a backslash run, of whitespace) publishes two zero counts and
  // writes NO masks: k_stage1_emit takes the zeros from the summary (escape_heavy: 0.27 GB of masks written and 0.28 GB read back for 300 000
  // structurals in a GiB -- profiles/r04_pmc_summary.txt -- are gone; ordinary input has no such segments and pays one ballot)
  // Round 6: a segment with a HANDFUL of candidates (a string boundary between two long backslash runs, the one comma in 16 KiB of text: escape_heavy holds
  // ~2 structurals per segment and shipped 2 x 2 KiB of planes for them -- 0.14 GB written and 0.15 GB read back per GiB, profiles/r05_pmc_summary.txt) ships
  // them as a list: one word per candidate in ascending order -- its offset in the segment, bit 31 its string_tail bit -- in the first line of the segment's
  // plane-0 area.  k_stage1_emit selects by hypothesis and copies.
  if (sparse) {
    // per-chunk counts packed into one word (each <= 32: no carry between the 8-bit fields), one scan for the four chunks' exclusive prefixes
    const u32 packed = u32(popc64(keep0[0])) | (u32(popc64(keep0[1])) << 8) | (u32(popc64(keep0[2])) << 16) | (u32(popc64(keep0[3])) << 24);
    const u32 incl = wave_incl_scan(packed), tot = readlane(incl, 63), excl = incl - packed;
    u32 *list = reinterpret_cast<u32 *>(mask0) + size_t(seg) * (SEG_BYTES / BLOCK_BYTES * 2u); // the segment's 2 KiB of plane 0, as words
    u32 chunk_base = 0;
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      u32 slot = chunk_base + ((excl >> (8u * c)) & 0xFFu);
      u64 bits = keep0[c];
      while (bits) { // (a few lanes, a few bits)
        const u32 b = ctz64(bits);
        bits &= bits - 1;
        list[slot++] = (c * CHUNK_BYTES + lane * BLOCK_BYTES + b) | (u32((keep1[c] >> b) & 1ull) << 31);
      }
      chunk_base += (tot >> (8u * c)) & 0xFFu;
    }
    return;
  }
  if (__ballot(n_a != 0u)) {
    // in 16-byte units: [segment][chunk pair][lane] -- an instruction of the wave writes (and k_stage1_emit's reads) one contiguous KiB: whole lines
    // (rounds 1-5a: [segment][lane][chunk], 16 bytes of every 32 per instruction: half of every line, twice)
    const size_t at = size_t(seg) * 128 + lane;
    uint4 *p0 = reinterpret_cast<uint4 *>(mask0) + at;
    store16_stream(p0, make_uint4(u32(keep0[0]), u32(keep0[0] >> 32), u32(keep0[1]), u32(keep0[1] >> 32)));
    store16_stream(p0 + 64, make_uint4(u32(keep0[2]), u32(keep0[2] >> 32), u32(keep0[3]), u32(keep0[3] >> 32)));
    if (!resolved) {
      uint4 *p1 = reinterpret_cast<uint4 *>(mask1) + at;
      store16_stream(p1, make_uint4(u32(keep1[0]), u32(keep1[0] >> 32), u32(keep1[1]), u32(keep1[1] >> 32)));
      store16_stream(p1 + 64, make_uint4(u32(keep1[2]), u32(keep1[2] >> 32), u32(keep1[3]), u32(keep1[3] >> 32)));
    }
  }
}

// =====================================================================================================
// resolve, level 2: scan of (parity, count) over at most a few thousand GROUP summaries in one workgroup
// (level 1 = k_resolve_groups below folds 64 segment summaries into one group summary).
// =====================================================================================================
constexpr u32 RESOLVE_THREADS = 1024;

// Exclusive scans over the workgroup's threads, one value per thread: a DPP scan inside every wave, the sixteen wave totals through LDS, two barriers
// (rounds 1-4: ten Hillis-Steele steps through LDS with two barriers each -- sixty barriers for the three scans of k_resolve_segments, most of its 7.7 us)
constexpr u32 RESOLVE_WAVES = RESOLVE_THREADS / 64;
__device__ __forceinline__ u32 wave_incl_scan_xor(u32 v) {
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));
  v ^= u32(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *sh, bool use_xor, u32 &total) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 incl = use_xor ? wave_incl_scan_xor(v) : wave_incl_scan(v);
  if (lane == 63) { sh[wave] = incl; }
  __syncthreads();
  u32 base = 0, sum = 0;
  for (u32 w = 0; w < RESOLVE_WAVES; w++) {
    const u32 t = sh[w];
    if (w < wave) { base = use_xor ? (base ^ t) : (base + t); }
    sum = use_xor ? (sum ^ t) : (sum + t);
  }
  total = sum;
  __syncthreads();
  return use_xor ? (base ^ incl ^ v) : (base + incl - v);
}

// the successor function of a summary, x -> c ^ (dep & x), as two bits: bit 0 = f(0), bit 1 = f(1)
__device__ __forceinline__ u32 xfun_of(u32 xw) { return (xw & 1u) | (((xw ^ (xw >> 1)) & 1u) << 1); }
__device__ __forceinline__ u32 xfun_then(u32 a, u32 b) { // first a, then b
  const u32 r0 = (b >> (a & 1u)) & 1u, r1 = (b >> ((a >> 1) & 1u)) & 1u;
  return r0 | (r1 << 1);
}
constexpr u32 XFUN_ID = 2u;
// exclusive scan of function composition over the workgroup's threads (lanes without a source compose with the identity)
__device__ __forceinline__ u32 wave_incl_scan_xfun(u32 v) {
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x111, 0xf, 0xf, false)), v);
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x112, 0xf, 0xf, false)), v);
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x114, 0xf, 0xf, false)), v);
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x118, 0xf, 0xf, false)), v);
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x142, 0xa, 0xf, false)), v);
  v = xfun_then(u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(v), 0x143, 0xc, 0xf, false)), v);
  return v;
}
__device__ __forceinline__ u32 block_excl_scan_xfun(u32 v, u32 *sh) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 incl = wave_incl_scan_xfun(v);
  if (lane == 63) { sh[wave] = incl; }
  __syncthreads();
  u32 base = XFUN_ID;
  for (u32 w = 0; w < RESOLVE_WAVES; w++) {
    if (w < wave) { base = xfun_then(base, sh[w]); }
  }
  // exclusive: everything in front of my wave, then the lanes in front of me
  u32 prev = u32(__builtin_amdgcn_update_dpp(int(XFUN_ID), int(incl), 0x138, 0xf, 0xf, false)); // wave_shr:1 (lane 0 keeps the identity)
  __syncthreads();
  return xfun_then(base, prev);
}

// what: 0 = stage1 (writes n, flags and the three sentinels), 1 = minify (writes out_len, flags)
__global__ __launch_bounds__(RESOLVE_THREADS) void k_resolve_segments(const seg_summary *__restrict__ summ,
                                                                     seg_prefix *__restrict__ pref, u32 nseg, u64 len,
                                                                     u32 *__restrict__ idx, u64 idx_words,
                                                                     scan_result_dev *__restrict__ result, int what,
                                                                     scan_origin org) {
  const u32 carry = org.carry;
  __shared__ u32 sh[RESOLVE_THREADS];
  __shared__ u32 sh_flags, sh_x_end;
  const u32 tid = threadIdx.x;
  if (tid == 0) { sh_flags = 0; }
  const u32 per = (nseg + RESOLVE_THREADS - 1) / RESOLVE_THREADS;
  const u32 lo = min(tid * per, nseg), hi = min(lo + per, nseg);
  // pass 0: x in front of my first summary (sj_xcarry.h: x of the successor = c ^ (dep & x), a scan over functions of one bit)
  u32 fun = XFUN_ID;
  for (u32 i = lo; i < hi; i++) { fun = xfun_then(fun, xfun_of(summ[i].xw)); }
  const u32 x_call = (carry & CARRY_X) ? 1u : 0u; // a later range of a buffer: what the range in front said about its successor
  const u32 x0 = (block_excl_scan_xfun(fun, sh) >> x_call) & 1u;
  // pass 1: quote parity (a wrong assumption in front of a quote flips what the summary hands on)
  u32 par = 0, xx = x0;
  for (u32 i = lo; i < hi; i++) {
    const seg_summary v = summ[i];
    par ^= (v.flags & SF_PARITY) ^ (xx & (v.xw >> 2) & 1u);
    xx = (v.xw & 1u) ^ (xx & (v.xw >> 1) & 1u);
  }
  if (tid == RESOLVE_THREADS - 1) { sh_x_end = xx; }
  u32 final_parity;
  u32 s = block_excl_scan(par, sh, true, final_parity);
  s ^= carry & CARRY_IN_STRING; // a shard of a larger document may begin inside a string
  final_parity ^= carry & CARRY_IN_STRING;
  // pass 2: counts under the now-known state
  u32 cnt = 0, st = s, flags = 0;
  xx = x0;
  for (u32 i = lo; i < hi; i++) {
    const seg_summary v = summ[i];
    const xs_step t = xs_apply(v.flags & SF_PARITY, v.xw, st, xx);
    cnt += xs_count(v.count_if_out, v.count_if_in, t);
    if (v.flags & (t.se ? SF_CTRL_IF_IN : SF_CTRL_IF_OUT)) { flags |= SJGPU_F_UNESCAPED_CTRL; }
    if (v.flags & SF_UTF8) { flags |= SJGPU_F_UTF8_ERROR; }
    st = t.s_out;
    xx = t.x_out;
  }
  u32 total;
  u32 base = block_excl_scan(cnt, sh, false, total) + org.base0; // output continues where the previous range stopped
  total += org.base0;
  if (flags) { atomicOr(&sh_flags, flags); }
  // pass 3: per-segment carry-in
  st = s;
  xx = x0;
  for (u32 i = lo; i < hi; i++) {
    const seg_summary v = summ[i];
    seg_prefix p;
    p.base = base;
    p.in_string = st | (xx << 1);
    pref[i] = p;
    const xs_step t = xs_apply(v.flags & SF_PARITY, v.xw, st, xx);
    base += xs_count(v.count_if_out, v.count_if_in, t);
    st = t.s_out;
    xx = t.x_out;
  }
  __syncthreads();
  if (tid == 0) {
    u32 f = sh_flags | (final_parity ? SJGPU_F_UNCLOSED_STRING : 0u);
    if ((carry & CARRY_MORE) && sh_x_end) { f |= SJGPU_F_RANGE_CARRY; } // the next range of the buffer starts from it
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
      // unclosed string voids the output (json_minifier.h:42-47) -- of a whole document, not of a shard
      result->out_len = (final_parity && !(carry & CARRY_SHARD)) ? 0ull : u64(total);
    }
    result->flags = f; // this kernel is the first writer of the call's result (the emit kernel only ORs into it)
  }
}

// =====================================================================================================
// resolve, level 1: one wave per GROUP of 64 segments (1 MiB of input) folds the 64 summaries into one
// summary of the same shape; level 2 (k_resolve_segments above, run over the group summaries) scans at
// most 4096 of those in one workgroup; the emit kernels redo the in-group part themselves with one
// coalesced 1 KiB load (segment_prefix below).  Replaces a 180 us single-workgroup scan by ~10 us.
// =====================================================================================================
__global__ __launch_bounds__(64) void k_resolve_groups(const seg_summary *__restrict__ summ, seg_summary *__restrict__ gsum,
                                                       u32 nseg) {
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x * RESOLVE_GROUP + lane;
  seg_summary x{0u, 0u, 0u, XW_IDENTITY};
  if (seg < nseg) { x = summ[seg]; }
  u32 err_out = 0, err_in = 0;
  const xs_sum g = wave_fold<false>(xs_sum{x.flags & SF_PARITY, x.count_if_out, x.count_if_in, x.xw}, lane, x.flags & SF_CTRL_IF_OUT, x.flags & SF_CTRL_IF_IN,
                                    &err_out, &err_in);
  u32 flags = g.q ? SF_PARITY : 0u;
  if (err_out) { flags |= SF_CTRL_IF_OUT; }
  if (err_in) { flags |= SF_CTRL_IF_IN; }
  if (__ballot((x.flags & SF_UTF8) != 0)) { flags |= SF_UTF8; }
  if (lane == 0) {
    seg_summary s;
    s.count_if_out = g.c_out;
    s.count_if_in = g.c_in;
    s.flags = flags;
    s.xw = g.xw;
    gsum[blockIdx.x] = s;
  }
}
// (Round 6, measured and not kept: both levels in ONE launch -- every wave folds a group, the workgroup that finds all others done scans the group
// summaries.  The hand-over needs a release fence behind the group summaries and an acquire fence in front of the scan, and an agent-scope fence on this
// part writes back and invalidates the XCD's L2, which at that moment holds what k_stage1_summarize has just written: the resolve step went from 12 to
// 30 us per call, every workload -- profiles/r06_split_ab.txt.)

// carry-in (in-string bit, x, output cursor) of segment `seg`: its group's prefix + the segments in front of
// it inside the group
__device__ __forceinline__ seg_prefix segment_prefix(const seg_summary *__restrict__ summ, const seg_prefix *__restrict__ gpref,
                                                     u32 seg, u32 lane) {
  const u32 group = seg / RESOLVE_GROUP, r = seg % RESOLVE_GROUP;
  const seg_prefix gp = gpref[group];
  seg_summary x{0u, 0u, 0u, XW_IDENTITY};
  if (lane < r) { x = summ[group * RESOLVE_GROUP + lane]; }
  // Ordinary input: no segment in front of this one inside the group carries an x word and the group starts with x = 0 -- then x is 0 in front of every
  // one of them and nothing of the x algebra below applies: parity by one ballot, the cursor by one sum (round 5: 71 -> ~30 VALU instructions per segment
  // of a kernel that, on sparse output, is bound by its instructions)
  if (__ballot(lane < r && x.xw != 0u) == 0 && ((gp.in_string >> 1) & 1u) == 0u) { // wave-uniform
    const u64 qm = __ballot((x.flags & SF_PARITY) != 0u); // (lanes from r on hold the summary of nothing: no parity, no counts)
    const u32 s = (__builtin_amdgcn_mbcnt_hi(u32(qm >> 32), __builtin_amdgcn_mbcnt_lo(u32(qm), gp.in_string & 1u))) & 1u;
    seg_prefix p;
    p.base = gp.base + wave_sum(s ? x.count_if_in : x.count_if_out);
    p.in_string = readlane_dyn(s, r);
    return p;
  }
  const u64 lt = lanemask_lt(lane);
  const lane_x mine = wave_x_chain<false>(x.xw, lt);
  const u32 xin = mine.base ^ (mine.free & (gp.in_string >> 1) & 1u);
  const u32 f = xin & (x.xw >> 2) & 1u;
  const u64 qm = __ballot(((((x.flags & SF_PARITY) ? 1u : 0u) ^ f)) != 0u);
  const u32 s = (gp.in_string & 1u) ^ (u32(popc64(qm & lt)) & 1u);
  const u32 se = s ^ f;
  seg_prefix p;
  p.base = gp.base + wave_sum((se ? x.count_if_in : x.count_if_out) + u32(xin ? xw_d(x.xw, se) : 0));
  p.in_string = readlane_dyn(s | (xin << 1), r); // lane r holds the summary of nothing: the state in front of it is the segment's
  return p;
}

// =====================================================================================================
// stage 1, kernel 2: select the right hypothesis per segment, flatten bitmaps to ascending offsets
// =====================================================================================================
// TOKENS: tok[i] = buf[idx[i]] for the segment's offsets, behind the same cursor: copied from the staging area k_stage1_summarize<true> filled, or -- a
// segment that staged nothing, or whose one patched candidate bit (sj_xcarry.h) changes the list the staging was made for -- gathered from the document
// (measured in round 5 and not kept: four waves -- four segments -- per workgroup: the dispatcher has a quarter of the workgroups to place, and the kernel
// is SLOWER, 75 -> 85 us per GiB of NDJSON and 32 -> 46 us on escape_heavy, profiles/r05_scan_steps_ab.txt: a workgroup then waits for four free wave
// slots and 25 KiB of LDS at once)
template <bool TOKENS>
__global__ __launch_bounds__(64) void k_stage1_emit(const u64 *__restrict__ mask0, const u64 *__restrict__ mask1,
                                                    const seg_summary *__restrict__ summ, const seg_prefix *__restrict__ gpref,
                                                    u64 len, u32 *__restrict__ idx, u64 idx_words,
                                                    scan_result_dev *__restrict__ result, scan_origin org, const u8 *__restrict__ buf,
                                                    const u8 *__restrict__ tokstage, u8 *__restrict__ tok) {
  __shared__ __attribute__((aligned(16))) u32 stage[EMIT_STAGE_WORDS];
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = org.begin + u64(seg) * SEG_BYTES;
  // Independent loads first, consumers later: the four chunks' masks (8 bytes per lane each) are requested before
  // the group-prefix fold, so the segment pays ONE round trip to HBM, not one per chunk plus one for the prefix.
  const seg_summary own = summ[seg];
  const bool resolved = (own.flags & SF_RESOLVED) != 0; // masks are already final
  const bool empty = own.count_if_out == 0u && own.count_if_in == 0u; // no candidate at all: k_stage1_summarize wrote no masks
  if (empty && ((own.xw >> XW_D_SHIFT) & 0xFu) == 0u) { return; }  // ... and no bit a wrong assumption could add: nothing to emit
  if (own.flags & SF_SPARSE) { // a list of at most SPARSE_MAX candidates instead of planes (k_stage1_summarize): select by hypothesis, copy
    const u32 ncand = resolved ? own.count_if_out : own.count_if_out + own.count_if_in;
    const u32 *list = reinterpret_cast<const u32 *>(mask0) + size_t(seg) * (SEG_BYTES / BLOCK_BYTES * 2u);
    const u32 e = lane < ncand ? list[lane] : 0u; // requested before the prefix fold: one round trip for both
    const seg_prefix pf = segment_prefix(summ, gpref, seg, lane);
    const xs_step t = xs_apply(own.flags & SF_PARITY, own.xw, pf.in_string & 1u, pf.in_string >> 1); // (no patched bit: sparse segments have d = 0)
#ifndef SJGPU_SELFTEST_SPARSE_IGNORES_HYPOTHESIS // tests/test_kernels_emu.py: with the selection dropped the sparse-segment documents must FAIL
    const bool keep = lane < ncand && (resolved || (e >> 31) == t.se); // structural = candidate & ~(string_tail ^ hypothesis)
#else
    const bool keep = lane < ncand && (resolved || (e >> 31) == 0u);
#endif
    const u64 km = __ballot(keep);
    const u64 at = u64(pf.base) + u32(popc64(km & lanemask_lt(lane)));
    if (u64(pf.base) + u32(popc64(km)) > idx_words) {
      if (lane == 0) { atomicOr(&result->flags, SJGPU_F_IDX_OVERFLOW); }
    } else if (keep) {
      idx[at] = u32(seg_start) + (e & 0x7FFFFFFFu);
    }
    return;
  }
  u64 m0[SEG_CHUNKS] = {0, 0, 0, 0}, m1[SEG_CHUNKS] = {0, 0, 0, 0};
  if (!empty) {
    const size_t at = size_t(seg) * 128 + lane; // [segment][chunk pair][lane]; chunks beyond len hold zero masks; read once: streamed
    const uint4 *p0 = reinterpret_cast<const uint4 *>(mask0) + at;
    const uint4 x = load16_stream(p0), y = load16_stream(p0 + 64);
    m0[0] = (u64(x.y) << 32) | x.x; m0[1] = (u64(x.w) << 32) | x.z;
    m0[2] = (u64(y.y) << 32) | y.x; m0[3] = (u64(y.w) << 32) | y.z;
    if (!resolved) {
      const uint4 *p1 = reinterpret_cast<const uint4 *>(mask1) + at;
      const uint4 u = load16_stream(p1), v = load16_stream(p1 + 64);
      m1[0] = (u64(u.y) << 32) | u.x; m1[1] = (u64(u.w) << 32) | u.z;
      m1[2] = (u64(v.y) << 32) | v.x; m1[3] = (u64(v.w) << 32) | v.z;
    }
  }
  const seg_prefix pf = segment_prefix(summ, gpref, seg, lane);
  const u32 x = pf.in_string >> 1;
  const xs_step t = xs_apply(own.flags & SF_PARITY, own.xw, pf.in_string & 1u, x); // which hypothesis, and the bit a wrong assumption toggles
  u32 base = pf.base;
  const u64 flip = t.se ? ~0ull : 0ull;
  bool overflow = false;
  u64 st[SEG_CHUNKS];
#pragma unroll
  for (u32 c = 0; c < SEG_CHUNKS; c++) { st[c] = m0[c]; } // chunks beyond len hold zero masks
  if (!resolved) { // (wave-uniform: a scalar branch, not eight selects)
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) { st[c] = andn(m0[c], m1[c] ^ flip); }
  }
  span_patch(st, own.xw, x, t.se, lane);
  const u32 span_count = (org.carry & CARRY_DEBUG_NO_SPAN_HINT) ? 0u : xs_count(own.count_if_out, own.count_if_in, t);
  const u32 base_before = base;
  emit_span4_adaptive<EMIT_WINDOW>(st, u32(seg_start), lane, idx, idx_words, base, stage, overflow, span_count); // sparse segments go out in one piece
  if (__ballot(overflow) && lane == 0) { atomicOr(&result->flags, SJGPU_F_IDX_OVERFLOW); }
  if (TOKENS && !__ballot(overflow)) {
    const u32 cnt = base - base_before; // wave-uniform
    if ((own.flags & SF_TOKENS) != 0u && t.dcount == 0) {
      // the staged bytes (16-byte aligned source) behind the cursor (any alignment): bytes up to the destination's next dword, whole dwords -- read with
      // unaligned loads -- and the last few bytes (the first version copied byte by byte: 16 round trips per lane for a segment of NDJSON, +55 us per GiB)
      const u8 *src = tokstage + size_t(seg) * SEG_BYTES;
      u8 *dst = tok + size_t(base_before);
      const u32 head_want = (4u - (base_before & 3u)) & 3u, head = head_want < cnt ? head_want : cnt;
      if (lane < head) { dst[lane] = src[lane]; }
      const u32 body = (cnt - head) >> 2;
      typedef u32 __attribute__((aligned(1))) u32_any;
#pragma unroll 1
      for (u32 i = lane; i < body; i += 64) { *reinterpret_cast<u32 *>(dst + head + 4u * i) = *reinterpret_cast<const u32_any *>(src + head + 4u * i); }
      const u32 done = head + 4u * body;
      if (lane < cnt - done) { dst[done + lane] = src[done + lane]; }
    } else { // the offsets this WAVE has just written, read back by itself once its stores have been acknowledged (no fence: an agent-scope fence writes
             // back and invalidates the XCD's L2 -- see leave_and_clean in sjgpu_fused.hip); the road of segments that resolved nothing and of patched bits
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const volatile u32 *back = idx + size_t(base_before);
#pragma unroll 1
      for (u32 i = lane; i < cnt; i += 64) {
        const u32 p = back[i];
        tok[size_t(base_before) + i] = u64(p) < len ? buf[p] : u8(0x20);
      }
    }
  }
}

// =====================================================================================================
// minify: summarize (kept-byte counts for both hypotheses) -> resolve -> re-scan + byte compaction
// =====================================================================================================
__global__ __launch_bounds__(64) void k_minify_summarize(const u8 *__restrict__ buf, u64 len, seg_summary *__restrict__ summ,
                                                         scan_origin org) {
  __shared__ __attribute__((aligned(16))) uint4 xbuf[CHUNK_BYTES / 16]; // load_chunk_stream's exchange buffer
  const u32 lane = lane_id();
  const u32 seg = blockIdx.x;
  const u64 seg_start = org.begin + u64(seg) * SEG_BYTES;
  const u32 lookback = lookback_issue(buf, seg_start, lane);
  wave_carry wc{0u, 0u, 0u};
  span_x sx;
  u32 kept_out = 0, kept_in = 0;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf, w); }
    else { load_block(buf, pos, len, w); }
    if (c == 0) { wc = span_carry_assume(seg_start, lane, lookback, sx); }
    span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
    const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
    if (c == SEG_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
    const u64 valid = valid_mask(pos, len);
    kept_out += u32(popc64(valid & ~(m.ws & ~m.in_string))); // dropped: whitespace outside strings (json_scanner.h:46)
    kept_in += u32(popc64(valid & ~(m.ws & m.in_string)));
  }
  const u32 a = wave_sum(kept_out), b = wave_sum(kept_in);
  const u32 xw = span_finish(sx, buf, seg_start, SEG_BYTES, len, wc, false);
  if (lane == 0) {
    seg_summary s;
    s.count_if_out = a;
    s.count_if_in = b;
    s.flags = wc.s ? SF_PARITY : 0u;
    s.xw = xw;
    summ[seg] = s;
  }
}

__global__ __launch_bounds__(64) void k_minify_emit(const u8 *__restrict__ buf, u64 len, const seg_summary *__restrict__ summ,
                                                    const seg_prefix *__restrict__ gpref, u8 *__restrict__ dst, scan_origin org) {
  __shared__ __attribute__((aligned(16))) uint4 xbuf[CHUNK_BYTES / 16]; // load_chunk_stream's exchange buffer
  __shared__ __attribute__((aligned(16))) u8 stage[MINIFY_STAGE_BYTES];
  __shared__ u32 lut[MINIFY_LUT_WORDS];
  const u32 lane = lane_id();
  init_compaction_lut(lut, lane);
  clear_minify_stage(stage, lane);
  const u32 seg = blockIdx.x;
  const u64 seg_start = org.begin + u64(seg) * SEG_BYTES;
  const u32 lookback = lookback_issue(buf, seg_start, lane);
  const seg_prefix pf = segment_prefix(summ, gpref, seg, lane);
  wave_carry wc{0u, 0u, 0u};
  u32 base = pf.base;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf, w); }
    else { load_block(buf, pos, len, w); }
    if (c == 0) { // the same scan as k_minify_summarize's, assumption included; the state it starts from is the EFFECTIVE one: behind the
                  // quote a wrong assumption misjudged both agree with the truth, in front of it there is nothing but backslashes
      span_x sx;
      wc = span_carry_assume(seg_start, lane, lookback, sx);
      const seg_summary own = summ[seg];
      wc.s = xs_apply(own.flags & SF_PARITY, own.xw, pf.in_string & 1u, pf.in_string >> 1).se; // absolute from here on
    }
    const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
    emit_bytes(w, valid_mask(pos, len) & ~(m.ws & ~m.in_string), lane, dst, base, stage, lut);
  }
}

// =====================================================================================================
// validate_utf8: stateless (3-byte look-back), pure HBM read
// =====================================================================================================
// Bytes [begin, len) of buf (begin a multiple of the chunk size; the bytes in front of it are resident and were checked
// by an earlier launch); `more`: the input continues behind len, no end-of-input rule.
__global__ __launch_bounds__(64) void k_validate_utf8(const u8 *__restrict__ buf, u64 len, scan_result_dev *__restrict__ result, u64 begin,
                                                      u32 more) {
  __shared__ __attribute__((aligned(16))) uint4 xbuf[CHUNK_BYTES / 16]; // load_chunk_stream's exchange buffer
  const u32 lane = lane_id();
  const u64 nchunks = (len + CHUNK_BYTES - 1) / CHUNK_BYTES;
  bool bad = false;
  for (u64 ch = begin / CHUNK_BYTES + blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const u64 cstart = ch * CHUNK_BYTES;
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    // (measured and not kept, round 5: requesting the wave's NEXT chunk before this one is looked at -- 173 -> 193 us per GiB, and the same in
    // k_stage1_summarize, 16 registers more: 288 -> 306 us per GiB of NDJSON; profiles/r05_stream_ab.txt, session Q)
    if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf, w); }
    else { load_block(buf, pos, len, w); }
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
    if (ch == nchunks - 1 && !more && (readlane(co, 63) & UTF8_CARRY_OPEN)) { bad = true; } // sequence open at EOF
  }
  if (__ballot(bad) && lane == 0) { atomicOr(&result->flags, SJGPU_F_UTF8_ERROR); }
}


// =====================================================================================================
// string parity of a shard: 1 iff it holds an odd number of unescaped quotes (read-only pre-pass of the
// general multi-GPU sharding, SURVEY 8(e): the ranks all-gather these bits, then scan with the right carry-in)
// =====================================================================================================
__global__ __launch_bounds__(64) void k_string_parity(const u8 *__restrict__ buf, u64 len, u32 nseg, u8 *__restrict__ seg_bits) {
  __shared__ __attribute__((aligned(16))) uint4 xbuf[CHUNK_BYTES / 16]; // load_chunk_stream's exchange buffer
  const u32 lane = lane_id();
  for (u32 seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const u64 seg_start = u64(seg) * SEG_BYTES;
    const u32 lookback = lookback_issue(buf, seg_start, lane);
    wave_carry wc{0u, 0u, 0u};
    span_x sx;
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
      if (cstart >= len) { break; }
      const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
      u32 w[16];
      if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf, w); }
      else { load_block(buf, pos, len, w); }
      if (c == 0) { wc = span_carry_assume(seg_start, lane, lookback, sx); }
      span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
      const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
      if (c == SEG_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
    }
    const u32 xw = span_finish(sx, buf, seg_start, SEG_BYTES, len, wc, false);
    if (lane == 0) { seg_bits[seg] = u8(wc.s | ((xw & 7u) << 1)); } // parity under the assumption; c, dep, F
  }
}
// the fold of those bytes: x along the segments, then the parity with the flips x asks for
__global__ __launch_bounds__(RESOLVE_THREADS) void k_string_parity_fold(const u8 *__restrict__ seg_bits, u32 nseg, scan_result_dev *__restrict__ result) {
  __shared__ u32 sh[RESOLVE_THREADS];
  const u32 tid = threadIdx.x;
  const u32 per = (nseg + RESOLVE_THREADS - 1) / RESOLVE_THREADS;
  const u32 lo = min(tid * per, nseg), hi = min(lo + per, nseg);
  u32 fun = XFUN_ID;
  for (u32 i = lo; i < hi; i++) { fun = xfun_then(fun, xfun_of(u32(seg_bits[i]) >> 1)); }
  u32 xx = block_excl_scan_xfun(fun, sh) & 1u; // a shard begins behind a clean byte: x = 0 in front of it
  u32 par = 0;
  for (u32 i = lo; i < hi; i++) {
    const u32 v = seg_bits[i], xw = v >> 1;
    par ^= (v & 1u) ^ (xx & (xw >> 2) & 1u);
    xx = (xw & 1u) ^ (xx & (xw >> 1) & 1u);
  }
  u32 total;
  (void)block_excl_scan(par, sh, true, total);
  if (tid == 0) { result->n = total & 1u; }
}


} // namespace

// ---- launchers ----------------------------------------------------------------------------------------
static inline void mark(hipEvent_t *ev, int k, hipStream_t stream) {
  if (ev) { (void)hipEventRecord(ev[k], stream); }
}

void launch_stage1(const uint8_t *buf, uint64_t len, uint4 *masks, seg_summary *summ, seg_prefix *pref, uint32_t *idx,
                   uint64_t idx_words, scan_result_dev *result, scan_origin org, hipStream_t stream, hipEvent_t *ev, uint8_t *tokstage, uint8_t *tok) {
  const u32 nseg = num_segments(len - org.begin);
  static const bool no_hint = std::getenv("SJGPU_NO_SPAN_HINT") != nullptr; // A/B switch
  if (no_hint) { org.carry |= CARRY_DEBUG_NO_SPAN_HINT; }
  static const bool queue_only = std::getenv("SJGPU_UTF8_QUEUE_ONLY") != nullptr; // A/B switch
  if (queue_only) { org.carry |= CARRY_DEBUG_QUEUE_UTF8; }
  static const unsigned dense_from = []() { const char *v = std::getenv("SJGPU_UTF8_DENSE_FROM"); return v ? unsigned(std::atoi(v)) & 0xFFu : 0u; }();
  org.carry |= dense_from << 16;
  mark(ev, 0, stream);
  u64 *mask0 = reinterpret_cast<u64 *>(masks);
  u64 *mask1 = mask0 + size_t(nseg) * (SEG_BYTES / BLOCK_BYTES); // second plane, only written by unresolved segments
  const bool tokens = tokstage != nullptr && tok != nullptr;
  if (tokens) {
    hipLaunchKernelGGL(k_stage1_summarize<true>, dim3((nseg + SUMM_WAVES - 1) / SUMM_WAVES), dim3(64 * SUMM_WAVES), 0, stream, buf, len, mask0, mask1, summ,
                       org, nseg, tokstage);
  } else {
    hipLaunchKernelGGL(k_stage1_summarize<false>, dim3((nseg + SUMM_WAVES - 1) / SUMM_WAVES), dim3(64 * SUMM_WAVES), 0, stream, buf, len, mask0, mask1, summ,
                       org, nseg, static_cast<u8 *>(nullptr));
  }
  mark(ev, 1, stream);
  const u32 ngroups = (nseg + RESOLVE_GROUP - 1) / RESOLVE_GROUP;
  seg_summary *gsum = summ + nseg; // the group summaries live behind the segment summaries
  hipLaunchKernelGGL(k_resolve_groups, dim3(ngroups), dim3(64), 0, stream, summ, gsum, nseg);
  hipLaunchKernelGGL(k_resolve_segments, dim3(1), dim3(RESOLVE_THREADS), 0, stream, gsum, pref, ngroups, len, idx, idx_words,
                     result, 0, org);
  mark(ev, 2, stream);
  if (tokens) {
    hipLaunchKernelGGL(k_stage1_emit<true>, dim3(nseg), dim3(64), 0, stream, mask0, mask1, summ, pref, len, idx, idx_words, result, org, buf,
                       static_cast<const u8 *>(tokstage), tok);
  } else {
    hipLaunchKernelGGL(k_stage1_emit<false>, dim3(nseg), dim3(64), 0, stream, mask0, mask1, summ, pref, len, idx, idx_words, result, org, buf,
                       static_cast<const u8 *>(nullptr), static_cast<u8 *>(nullptr));
  }
  mark(ev, 3, stream);
}

void launch_minify(const uint8_t *buf, uint64_t len, seg_summary *summ, seg_prefix *pref, uint8_t *dst,
                   scan_result_dev *result, scan_origin org, hipStream_t stream, hipEvent_t *ev) {
  const u32 nseg = num_segments(len - org.begin);
  mark(ev, 0, stream);
  hipLaunchKernelGGL(k_minify_summarize, dim3(nseg), dim3(64), 0, stream, buf, len, summ, org);
  mark(ev, 1, stream);
  const u32 ngroups = (nseg + RESOLVE_GROUP - 1) / RESOLVE_GROUP;
  seg_summary *gsum = summ + nseg;
  hipLaunchKernelGGL(k_resolve_groups, dim3(ngroups), dim3(64), 0, stream, summ, gsum, nseg);
  hipLaunchKernelGGL(k_resolve_segments, dim3(1), dim3(RESOLVE_THREADS), 0, stream, gsum, pref, ngroups, len,
                     static_cast<u32 *>(nullptr), u64(0), result, 1, org);
  mark(ev, 2, stream);
  hipLaunchKernelGGL(k_minify_emit, dim3(nseg), dim3(64), 0, stream, buf, len, summ, pref, dst, org);
  mark(ev, 3, stream);
}

void launch_validate_utf8(const uint8_t *buf, uint64_t len, scan_result_dev *result, hipStream_t stream, hipEvent_t *ev, uint64_t begin,
                          bool more) {
  mark(ev, 0, stream);
  if (begin == 0) { (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream); } // later ranges add to the flags
  const u64 nchunks = (len - begin + CHUNK_BYTES - 1) / CHUNK_BYTES;
  const u32 grid = u32(nchunks < 8192 ? nchunks : 8192); // 256 CUs x 32 single-wave workgroups, grid-stride beyond
  hipLaunchKernelGGL(k_validate_utf8, dim3(grid), dim3(64), 0, stream, buf, len, result, begin, more ? 1u : 0u);
  mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded; sjgpu_profile_read reports them as 0)
}

void launch_string_parity(const uint8_t *buf, uint64_t len, scan_result_dev *result, uint8_t *workspace, hipStream_t stream) {
  (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
  const u32 nseg = num_segments(len);
  if (nseg == 0) { return; }
  const u32 grid = nseg < 8192u ? nseg : 8192u;
  hipLaunchKernelGGL(k_string_parity, dim3(grid), dim3(64), 0, stream, buf, len, nseg, workspace);
  hipLaunchKernelGGL(k_string_parity_fold, dim3(1), dim3(RESOLVE_THREADS), 0, stream, workspace, nseg, result);
}

} // namespace sjgpu
