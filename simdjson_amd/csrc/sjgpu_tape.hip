// simdjson_amd/csrc/sjgpu_tape.hip -- SURVEY 8(f).3: stage 2 of a resident document on the device -- the reference's DOM tape.
//
// The reference builds the tape with ONE core walking the structural list (/root/reference/src/generic/stage2/json_iterator.h:121-244,
// visitor /root/reference/src/generic/stage2/tape_builder.h:142-441, format /root/reference/doc/tape.md): a state machine with a
// stack of open containers.  Nothing in the result needs that order.  Per token (sj_tape_rules.h):
//   * its tape position is a prefix sum of word counts (0 for ':' ',', 2 for numbers, 1 for the rest);
//   * its nesting depth is a prefix sum of +1 / -1 over the brackets;
//   * whether the walk would accept it depends on the token in front and on the KIND of the innermost open container;
//   * its content (number, atom; strings: sjgpu_strings.hip) is a function of its own bytes.
// The stack is replaced by a stable SORT of the brackets and commas by nesting level: inside one level the tokens of one container
// are contiguous and in document order -- open, its commas, close -- so a count of the opening brackets in front names the
// container, the distance between an open and its close counts the commas, and a close finds its partner without any walk.
// The first error of the serial walk is the smallest (list index, rank) over all tokens that break their rule: one atomicMin.
//
// Passes (all asynchronous on the caller's stream; one 32-byte result is read back by the caller):
//   k_tok_stage / k_tok_scan_sums      the byte of every token, the block totals of the six per-token counters -- and, from the same staged bytes of the
//                     document, the spelling of true / false / null and the VALUES of the number tokens (round 6: the document is fetched once for all of them)
//   (the string buffer: sjgpu_strings.hip / sjgpu_string_stream.hip, which take the number of string tokens from here)
//   k_tok_apply       tape position and nesting depth of every token in one sweep over the token bytes (the depth stays in registers: what it
//                     decides -- the root value has ended, the nesting limit -- is said here); the tape words of the strings (the
//                     k-th string token's record is the k-th of the buffer), of the atoms and of the numbers (values parked by k_tok_stage);
//                     (level, kind, token) of every bracket and comma into the sort's input
//   radix passes      stable LSD radix sort on the level, 6 bits per pass: histogram, scan, scatter   1 wave / 2048 elements;
//                     the second pass only runs for documents nested 64 deep and more
//   (container ordinal per sorted element and sorted position of every open: written by the sort's last scatter from a second histogram)
//   k_tape_match      commas judge the two tokens behind them by their container's kind; closes write BOTH bracket words (count, partner
//                     index), kinds checked
//   k_tape_slow_numbers  the handful of number tokens whose rounding needs exact big-integer arithmetic (sj_number.h)
// Parity: tests/test_gpu_parity.py::test_tape_* against the live reference's dom::parser::parse (tape and string_buf word for word,
// error codes of broken documents); the same steps run on the CPU in tests/host/test_tape_model.cpp.
#include "sjgpu_device.h"
#include "sj_tape_rules.h"
#ifdef SJGPU_LAB_STAGE_PHASES
#include <vector>
#include <algorithm>
#endif

namespace sjgpu {
namespace {

constexpr u32 TP_THREADS = 256;
constexpr u32 RADIX_BITS = 6, RADIX_BINS = 1u << RADIX_BITS, RADIX_TILE = 2048; // one wave sorts one tile
constexpr u32 RADIX_STEPS = RADIX_TILE / 64, RADIX_DEAD = 0xFFFFFFFFu; // (keys have 16 bits)
static_assert(RADIX_BINS == TAPE_ONE_PASS_LEVELS, "sjgpu_stage2_device decides from max_level whether the second pass is needed");

struct dev_bytes {
  const u8 *buf;
  u32 len;
  __device__ __forceinline__ u32 byte(u32 pos) const { return pos < len ? u32(buf[pos]) : 0x20u; }
};
// The same bytes through an 8-byte window in registers: a token is read front to back, so one (unaligned) 8-byte load serves eight
// byte() calls -- the per-byte loads of the first version were what made the per-token kernel (then k_tape_write) the longest kernel of the tape
// (profiles/r03_tape_kernel_stats.txt: 1.9 ms of 6.5 per 256 MiB document, numbers parsed with a round trip to L2 per digit).
struct windowed_bytes {
  const u8 *buf;
  u32 len;
  mutable u64 window = 0;
  mutable u32 at = 0xFFFFFFF0u; // position of the window's first byte (nothing loaded yet)
  typedef u64 __attribute__((aligned(1))) u64_unaligned;
  __device__ __forceinline__ u32 byte(u32 pos) const {
    const u32 d = pos - at;
    if (d >= 8u) { // also true for pos < at (wraps)
      at = pos;
      if (u64(pos) + 8u <= len) { window = *reinterpret_cast<const u64_unaligned *>(buf + pos); }
      else {
        window = 0;
        for (u32 k = 0; k < 8; k++) { window |= u64(pos + k < len ? u32(buf[pos + k]) : 0x20u) << (8u * k); }
      }
      return u32(window) & 0xFFu;
    }
    return u32(window >> (8u * d)) & 0xFFu;
  }
};

// The same through a 32-byte window, all of it requested at once: a number token of the twitter-like text is 19 bytes long (an 18-digit id and the
// byte that ends it) -- three dependent round trips through the 8-byte window, one here.  k_tape_numbers: one token per lane, waiting for memory
// 93 % of its cycles (profiles/r04_pmc_summary.txt).
struct wide_window_bytes {
  const u8 *buf;
  u32 len;
  mutable u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  mutable u32 at = 0xFFFFFF00u;
  typedef u64 __attribute__((aligned(1))) u64_unaligned;
  __device__ __forceinline__ void fill(u32 pos) const { // request the 32 bytes from pos on (nothing waits for them here)
    at = pos;
    if (u64(pos) + 32u <= len) {
      const u64_unaligned *p = reinterpret_cast<const u64_unaligned *>(buf + pos);
      w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
    } else {
      u64 w[4] = {0, 0, 0, 0};
      for (u32 k = 0; k < 32; k++) { w[k >> 3] |= u64(pos + k < len ? u32(buf[pos + k]) : 0x20u) << (8u * (k & 7u)); }
      w0 = w[0]; w1 = w[1]; w2 = w[2]; w3 = w[3];
    }
  }
  __device__ __forceinline__ u32 byte(u32 pos) const {
    u32 d = pos - at;
    if (d >= 32u) {
      fill(pos);
      d = 0;
    }
    const u64 lo = d < 8u ? w0 : w1, hi = d < 24u ? w2 : w3;
    return u32((d < 16u ? lo : hi) >> (8u * (d & 7u))) & 0xFFu;
  }
};

__device__ __forceinline__ void report_error(tape_result_dev *res, u64 key) { atomicMin(reinterpret_cast<unsigned long long *>(&res->error_key), (unsigned long long)key); }

// the result and the control words: `words` dwords from res on are cleared, then the three that are not zero
// ... and the two tables of "what a token's byte is" (round 6): rounds 3-5 had every workgroup of the token kernels build its LDS table from the predicates of
// sj_tape_rules.h -- one entry per thread, ~400 instructions with their branches for an entry of k_tok_apply's table, a tenth of all that kernel issued
// (a workgroup only handles 16 tokens per thread).  Built once per call here, loaded with one instruction there.
__global__ void k_tape_init(tape_result_dev *__restrict__ res, u32 *__restrict__ n_words, u32 words, u32 n1, u32 scan_len, u32 *__restrict__ entry_tab,
                            uint4 *__restrict__ stage_tab) {
  u32 *w = reinterpret_cast<u32 *>(res);
  for (u32 k = threadIdx.x; k < words; k += 64u) { w[k] = 0u; } // (launched with 64 threads)
  for (u32 v = threadIdx.x; v < 256u; v += 64u) {
    const u32 x = token_props_of(v);
    const tok_packed pk = tok_contribution_of_props(x);
    entry_tab[v] = token_entry_of(v);               // k_tok_apply
    stage_tab[v] = make_uint4(pk.a, pk.b, pk.c, x); // k_tok_stage
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    res->error_key = NO_ERROR_KEY;
    n_words[0] = n1;
    n_words[1] = scan_len;
  }
}

// ---- the token front: every prefix sum the tape needs, in one sweep over the token bytes ----------------------------------------------
// Per token (sj_tape_rules.h): tape words (0 / 1 / 2), "goes into the sort" (brackets and commas), "is a string", opening, closing.
// Their exclusive prefix sums are the token's tape position, its slot in the sort's input, its ordinal among the strings and
// (opens - closes) its nesting depth.  The first version wrote three int arrays and ran the generic three-kernel scan over each
// (profiles/r03_pmc_summary.txt: 2.6 GB of the 9.2 GB a twitter-like call moved); all five sums are functions of ONE byte per token,
// so here a block of 4096 tokens adds them up while it fetches its token bytes (k_tok_stage) and reads the 4 KiB of bytes once more
// for the prefixes (k_tok_apply) -- nothing else.  Inside a block the five counters travel packed in three dwords (each field < 2^16).
constexpr u32 TS_THREADS = 256, TS_ROW = TS_THREADS * 4, TS_ROWS = 4, TS_BLOCK = TS_ROW * TS_ROWS, TS_SUMS = 6;
// (tok_packed / tok_contribution, the kinds of the sort's elements and the placement of the value lists live in sj_tape_rules.h: the CPU
// model of tests/host/test_tape_model.cpp runs the same functions)
typedef u32 __attribute__((aligned(1))) u32_unaligned_t;
// the four token bytes i0 ... i0 + 3 (tokc is two bytes off the dword grid: one unaligned load)
__device__ __forceinline__ u32 four_tokens(const u8 *__restrict__ tokc, u64 i0) { return *reinterpret_cast<const u32_unaligned_t *>(tokc + 2 + i0); }

// ---- the staged token front (round 6) ------------------------------------------------------------------------------------------------------------
// Rounds 3-5 fetched the document once per kernel that looks at a token's bytes: k_tok_classify (one byte per token: a 128-byte line each, the whole
// document on dense lists), k_tape_numbers (32 bytes per listed number behind two dependent round trips -- list entry, idx[i], the bytes --: 93 % of its
// cycles waiting for memory), k_tape_atoms (the same for true / false / null): 0.95 GB of the 3.4 GB a twitter-like call moved and 1.35 of the 5.8 of
// large_random, 345 / 680 us of 1300 / 2230 (profiles/r06_pmc_summary.txt, r06_final_kernel_stats.txt).  The list is SORTED, so the bytes a run of
// consecutive tokens needs are one contiguous span of the document.  Here a wave owns SG_WAVE_TOKENS consecutive tokens (one row of k_tok_apply), parks
// their offsets in LDS, and works through them in GROUPS of whole rows of 64 tokens: as many rows as have their span (first token ... last token + SG_OVER
// bytes) inside SG_WINDOW bytes are staged with coalesced 16-byte loads into the wave's LDS window, and everything that asks for a token's bytes asks the
// window: the token byte (-> tokc and the six counters, as before), the spelling of the atoms (eight bytes at the token, compared as words), and the
// NUMBERS -- the group's number tokens are gathered in an LDS list (ballot + mbcnt: list order) and parsed one per lane with every lane busy (parse_number,
// sj_number.h, through a 32-byte register window filled from LDS).  Nothing waits for a second round trip and the document comes in once, line by line.
// A row whose 64 tokens span more than the window (sparse text: more than ~60 bytes per token) is read in place like before -- there a gather touches few
// lines per token anyway; a number that runs over the staged bytes (more than SG_OVER digits) continues from memory.
// Where the values go: a number's tape position is not known yet (k_tok_scan_sums comes next), so the parsed value waits in numbits / numtype at
// [first token of the wave + ordinal of the number inside the wave's 1024 tokens]; k_tok_apply, whose rows are the same 1024 tokens, knows that ordinal
// from its own scan and writes both tape words.  The lists of rounds 3-5 (number_list, value_list: 8 bytes per value token written and read back) are gone.
constexpr u32 SG_WAVE_TOKENS = TS_ROW, SG_ROWS = SG_WAVE_TOKENS / 64u;
constexpr u32 SG_OVER = 64; // bytes staged behind a group's last token: the longest number that is parsed without leaving the window
#ifndef SJGPU_STAGE_WINDOW
#define SJGPU_STAGE_WINDOW 4096
#endif
constexpr u32 SG_WINDOW = SJGPU_STAGE_WINDOW; // bytes of LDS per wave (a multiple of 1024: every lane stores 16 bytes per staging step)
constexpr u32 SG_STEPS = SG_WINDOW / 1024u;
static_assert(SG_WINDOW % 1024u == 0 && SG_WINDOW >= 1024u, "a staging step is 64 lanes x 16 bytes");

// The document through the wave's window, in the window's own coordinates: bytes [0, span) are staged (bytes at or beyond len as the spaces of a
// padded_string), byte `span` is a zero.  A byte is ONE LDS read (ds_read_u8) and two VALU instructions, without a branch and without a clamp: parse_number
// reads a token front to back and stops at the first byte that is no digit -- the zero at the latest -- and everything else it looks at lies in front of
// that byte, so no read goes beyond `span`.  The accessor remembers how far it was asked (`reach`): a token that reached the zero is parsed AGAIN, from
// memory, by the caller (wide_window_bytes; more than SG_OVER bytes of number: a handful per document at most).  The first version kept 32 bytes of the
// token in registers (the window of k_tape_numbers, filled from LDS) and paid ~15 VALU instructions per byte for the selects and the 64-bit shift that
// pick a byte out of them, the second branched per byte between LDS and memory (600 global loads and 950 branches in the kernel's code), the third clamped:
// the kernel is bound by its instruction count (profiles/r06_tape_stage.txt: 251 M / 214 M / 180 M VALU instructions per large_random call at four cycles
// each; the phase clocks of scripts/lab/stage_phases.py: a wave spends its time issuing, not waiting).
struct staged_bytes {
  const u8 *win; // the window (LDS)
  mutable u32 reach = 0; // the largest offset asked for
  __device__ __forceinline__ u32 byte(u32 off) const {
    reach = off > reach ? off : reach;
    return u32(win[off]);
  }
};
// bytes [a0, a0 + bytes) of the document into the window (a0 and bytes: multiples of 16, bytes <= SG_WINDOW): all loads issued, then all stores
__device__ __forceinline__ void stage_window(const u8 *__restrict__ buf, u32 len, u32 a0, u32 bytes, u32 lane, uint4 *__restrict__ win) {
  uint4 v[SG_STEPS];
#pragma unroll
  for (u32 j = 0; j < SG_STEPS; j++) {
    const u32 off = 1024u * j + 16u * lane;
    v[j] = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
    if (off < bytes) {
      const u64 p = u64(a0) + off;
      if (p + 16u <= len) { v[j] = *reinterpret_cast<const uint4 *>(buf + p); }
      else if (p < len) { // the input ends inside this piece: what lies behind it reads as spaces
        u32 w[4] = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
        for (u32 k = 0; k < 16u && p + k < len; k++) { w[k >> 2] = (w[k >> 2] & ~(0xFFu << (8u * (k & 3u)))) | (u32(buf[p + k]) << (8u * (k & 3u))); }
        v[j] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }
#pragma unroll
  for (u32 j = 0; j < SG_STEPS; j++) {
    const u32 off = 1024u * j + 16u * lane;
    if (off < bytes) { win[off >> 4] = v[j]; }
  }
}
// the eight bytes of the document at p (bytes at or beyond len: spaces), read in place
__device__ __forceinline__ uint2 eight_bytes_in_place(const u8 *__restrict__ buf, u32 len, u32 p) {
  typedef u64 __attribute__((aligned(1))) u64_unaligned;
  if (u64(p) + 8u <= len) {
    const u64 x = *reinterpret_cast<const u64_unaligned *>(buf + p);
    return make_uint2(u32(x), u32(x >> 32));
  }
  u64 x = 0;
  for (u32 k = 0; k < 8; k++) { x |= u64(u64(p) + k < len ? u32(buf[p + k]) : 0x20u) << (8u * k); }
  return make_uint2(u32(x), u32(x >> 32));
}
// true / false / null at a token whose first eight bytes are {lo, hi}: 0, or the error the reference's visitor raises (atomparsing.h: the letters, then a
// structural or whitespace byte; atom_matches of sj_number.h says the same byte by byte -- tests/host/test_tape_rules.cpp compares the two)
__device__ __forceinline__ u32 atom_error_of(u32 lo, u32 hi) {
  const u32 c = lo & 0xFFu;
  if (c == 't') { return (lo == 0x65757274u && !not_structural_or_whitespace(hi & 0xFFu)) ? 0u : u32(SJ_T_ATOM_ERROR); }
  if (c == 'f') { return (lo == 0x736C6166u && (hi & 0xFFu) == 'e' && !not_structural_or_whitespace((hi >> 8) & 0xFFu)) ? 0u : u32(SJ_F_ATOM_ERROR); }
  if (c == 'n') { return (lo == 0x6C6C756Eu && !not_structural_or_whitespace(hi & 0xFFu)) ? 0u : u32(SJ_N_ATOM_ERROR); }
  return 0u;
}

#ifdef SJGPU_LAB_STAGE_PHASES
constexpr u32 LAB_WAVES = 131072;
__device__ u32 d_stage_trace[LAB_WAVES][8]; // per wave: ticks per phase, no atomics
#define PHASE(k) do { const u64 now_ = wall_clock64(); acc_phase[k] += u32(now_ - t_phase); t_phase = now_; } while (0)
#else
#define PHASE(k) do { } while (0)
#endif
// a parsed number token i: its error, or its value where k_tok_apply will look for it
__device__ __forceinline__ void park_number(const number_value &v, u32 i, u64 at, u64 *__restrict__ numbits, u8 *__restrict__ numtype, u32 *__restrict__ slow_list,
                                            u32 slow_cap, tape_result_dev *__restrict__ res) {
  if (v.error) { report_error(res, error_key(i, 2, v.error)); return; }
  numbits[at] = v.bits; // sign only when v.slow: k_tape_slow_numbers completes the word on the tape
  numtype[at] = u8(v.type);
  if (v.slow) {
    const u32 s = atomicAdd(&res->slow_numbers, 1u);
    if (s < slow_cap) { slow_list[s] = i; }
  }
}
// tokc holds the token bytes with TWO zero bytes in front and behind: tokc[i + 2] = byte of token i (k_tape_rules looks two tokens back
// and one ahead).  The same sweep leaves the block totals: sums[k * nblocks + block], k = tape words, sort flags, strings, opens,
// closes, numbers.
// LDS per wave: the tokens' offsets as 16-bit distances from the first token of their row of 64 (a row that spans 64 KiB is read in place and asks the list
// again), the starts of the rows, the number list, the window: 8.3 KiB with a 4 KiB window -- four workgroups per CU.
#ifndef SJGPU_STAGE_OCC
#define SJGPU_STAGE_OCC 4
#endif
__global__ __launch_bounds__(TS_THREADS, SJGPU_STAGE_OCC) void k_tok_stage(const u8 *__restrict__ buf, u32 len, const u32 *__restrict__ idx, u32 n, u8 *__restrict__ tokc,
                                                            int *__restrict__ sums, u32 nblocks, u64 *__restrict__ numbits, u8 *__restrict__ numtype,
                                                            u32 *__restrict__ slow_list, u32 slow_cap, tape_result_dev *__restrict__ res,
                                                            const uint4 *__restrict__ stage_tab) {
  __shared__ u32 sh[3][TS_THREADS / 64];
  __shared__ uint4 sh_tab[256]; // what a token's byte IS: its packed contribution to the six counters (tok_packed) and its properties (token_props_of) -- one 16-byte
                                // LDS read per token instead of ~20 compares, or of the nine VALU instructions that unpack the contribution from the properties
  __shared__ unsigned short sh_rel[TS_THREADS / 64][SG_WAVE_TOKENS];  // the wave's offsets, relative to the first token of their row (0xFFFF: 64 KiB and more)
  __shared__ unsigned short sh_list[TS_THREADS / 64][SG_WAVE_TOKENS]; // its number tokens (which of its tokens), in list order
  __shared__ u32 sh_row[TS_THREADS / 64][32];                        // where its rows begin
  __shared__ uint4 sh_win[TS_THREADS / 64][SG_WINDOW / 16u + 1u];     // its window
  static_assert(TS_THREADS == 256 && TS_BLOCK == 4u * SG_WAVE_TOKENS && SG_ROWS <= 31u, "one table entry per thread; a wave per row of k_tok_apply");
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  sh_tab[tid] = stage_tab[tid]; // (k_tape_init built it)
  lds_writes_done();
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) { tokc[0] = 0; tokc[1] = 0; tokc[n + 2] = 0; tokc[n + 3] = 0; }
  const u64 t0_64 = u64(blockIdx.x) * TS_BLOCK + u64(wave) * SG_WAVE_TOKENS;
  const u32 cnt = t0_64 < n ? (n - u32(t0_64) < SG_WAVE_TOKENS ? n - u32(t0_64) : SG_WAVE_TOKENS) : 0u; // wave-uniform
  const u32 t0 = u32(t0_64);
  unsigned short *const rel = sh_rel[wave];
  unsigned short *const list = sh_list[wave];
  u32 *const row_start = sh_row[wave];
  uint4 *const win = sh_win[wave];
  const u32 *const win32 = reinterpret_cast<const u32 *>(win);
  const u8 *const win8 = reinterpret_cast<const u8 *>(win);
  u8 *const tokc_w = tokc + 2u + t0_64;
  const u32 *const idx_w = idx + t0_64;
  u32 a = 0, b = 0, c = 0;
#ifdef SJGPU_LAB_STAGE_PHASES
  u64 t_phase = wall_clock64();
  u32 acc_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  if (cnt) { // (wave-uniform; a wave without tokens only takes part in the block's sums)
    const u32 rows = (cnt + 63u) / 64u;
    u32 E = 0; // lane l < rows: where row l begins; lane rows: where the wave's span ends (behind its last token)
    {
      u32 v[SG_ROWS];
#pragma unroll
      for (u32 r = 0; r < SG_ROWS; r++) {
        const u32 tl = 64u * r + lane;
        v[r] = tl < cnt ? idx_w[tl] : 0xFFFFFFFFu;
      }
      const u32 last_row = (cnt - 1u) >> 6, last_lane = (cnt - 1u) & 63u;
#pragma unroll
      for (u32 r = 0; r < SG_ROWS; r++) {
        const u32 first = readlane(v[r], 0);
        if (lane == r && r <= last_row) { E = first; }
        if (r == last_row) { // wave-uniform
          const u32 end = readlane_dyn(v[r], last_lane) + 1u;
          if (lane == r + 1u) { E = end; }
        }
        const u32 d = v[r] - first;
        rel[64u * r + lane] = (unsigned short)(d < 0xFFFFu ? d : 0xFFFFu);
      }
      if (lane <= rows) { row_start[lane] = E; }
    }
    wave_lds_fence();
    PHASE(0); // the list rows have arrived and are parked
    const bool first_wave = t0 == 0u; // holds the root token
    u32 ncount = 0; // number tokens listed so far (wave-uniform)
    u32 ra = 0;
#pragma unroll 1
    while (ra < rows) {
      const u32 a0 = readlane_dyn(E, ra) & ~15u;
      // rows ra ... l - 1 fit the window: their last token lies in front of E[l], and SG_OVER bytes from it on are wanted (positions ascend: the lanes that fit
      // are ra + 1 ... rb)
      const bool fits = lane > ra && lane <= rows && (E - a0) + (SG_OVER - 1u) + 4u <= SG_WINDOW;
      const u32 nfit = u32(popc64(__ballot(fits)));
      const bool staged = nfit != 0u; // wave-uniform
      const u32 rb = staged ? ra + nfit : ra + 1u;
      u32 span = 0;
      if (staged) {
        const u32 want = (readlane_dyn(E, rb) - a0) + (SG_OVER - 1u) + 4u; // <= SG_WINDOW
        const u32 bytes = (want + 15u) & ~15u;
        stage_window(buf, len, a0, bytes, lane, win);
        span = bytes;
        if (lane == 0) { reinterpret_cast<u32 *>(win)[bytes >> 2] = 0u; } // what an offset beyond the staged bytes reads (staged_bytes): no digit
        wave_lds_fence();
      }
      PHASE(1); // staged
      const u32 g0 = ncount;
      // Dense text (large_random: 3.4 bytes per token) stages ALL 1024 tokens of a full wave at once.  Then a lane takes FOUR consecutive tokens per step (four
      // steps instead of sixteen rows): their offsets are one 8-byte LDS read, their bytes one dword store, the numbers among them are placed by one wave scan
      // of the lanes' counts instead of a ballot and a v_mbcnt per row -- the row loop below issues ~64 instructions per 64 tokens whatever they hold, and
      // the kernel is bound by what it issues (profiles/r06_tape_stage.txt).
      const bool dense = staged && ra == 0u && rb == SG_ROWS && cnt == SG_WAVE_TOKENS; // wave-uniform
      if (dense) {
#pragma unroll 1
        for (u32 sr = 0; sr < SG_ROWS / 4u; sr++) {
          const u32 tl0 = 256u * sr + 4u * lane; // the lane's four tokens lie in one row of 64: row 4 sr + lane / 16
          const uint2 r4 = *reinterpret_cast<const uint2 *>(rel + tl0);
          const u32 rs = row_start[4u * sr + (lane >> 4)] - a0;
          const u32 offs[4] = {rs + (r4.x & 0xFFFFu), rs + (r4.x >> 16), rs + (r4.y & 0xFFFFu), rs + (r4.y >> 16)};
          u32 lo[4], hi[4];
#pragma unroll
          for (u32 j = 0; j < 4; j++) {
            const u32 sft = offs[j] & 3u;
            const u32 *q = win32 + (offs[j] >> 2);
            const u32 d0 = q[0], d1 = q[1], d2 = q[2];
            lo[j] = __builtin_amdgcn_alignbyte(d1, d0, sft);
            hi[j] = __builtin_amdgcn_alignbyte(d2, d1, sft);
          }
          uint4 tab[4];
#pragma unroll
          for (u32 j = 0; j < 4; j++) { tab[j] = sh_tab[lo[j] & 0xFFu]; }
          if (first_wave && sr == 0u) { // wave-uniform: the root token's number path differs (takes_number_path) -- one token per document
            if (lane == 0) {
              const tok_packed pk = tok_contribution(lo[0] & 0xFFu, true);
              tab[0].x = pk.a; tab[0].y = pk.b; tab[0].z = pk.c;
            }
          }
          *reinterpret_cast<u32_unaligned_t *>(tokc_w + tl0) = (lo[0] & 0xFFu) | ((lo[1] & 0xFFu) << 8) | ((lo[2] & 0xFFu) << 16) | (lo[3] << 24);
          u32 mine = 0, atoms = 0; // number tokens of the lane, atoms of the lane
#pragma unroll
          for (u32 j = 0; j < 4; j++) {
            a += tab[j].x; b += tab[j].y; c += tab[j].z;
            mine += tab[j].z >> 16;
            atoms |= tab[j].w & TP_ATOM;
          }
          if (__ballot(atoms != 0u)) { // wave-uniform: steps without true / false / null skip the spelling
#pragma unroll
            for (u32 j = 0; j < 4; j++) {
              const u32 g = (tab[j].w & TP_ATOM) ? atom_error_of(lo[j], hi[j]) : 0u;
              if (g) { report_error(res, error_key(t0 + tl0 + j, 2, g)); }
            }
          }
          const u32 incl = wave_incl_scan(mine);
          u32 at = ncount + incl - mine;
#pragma unroll
          for (u32 j = 0; j < 4; j++) {
            if (tab[j].z >> 16) { list[at] = (unsigned short)(tl0 + j); }
            at += tab[j].z >> 16;
          }
          ncount += readlane(incl, 63);
        }
      }
#pragma unroll 1
      for (u32 r = dense ? rb : ra; r < rb; r++) {
        const u32 tl = 64u * r + lane;
        const bool live = tl < cnt;
        u32 lo, hi;
        if (staged) {
          const u32 rs = readlane_dyn(E, r) - a0; // the row's first token inside the window (wave-uniform)
          const u32 off = live ? rs + u32(rel[tl]) : 0u, sft = off & 3u;
          const u32 *q = win32 + (off >> 2);
          const u32 d0 = q[0], d1 = q[1], d2 = q[2];
          lo = __builtin_amdgcn_alignbyte(d1, d0, sft);
          hi = __builtin_amdgcn_alignbyte(d2, d1, sft);
        } else {
          const uint2 x = eight_bytes_in_place(buf, len, live ? idx_w[tl] : 0u);
          lo = x.x; hi = x.y;
        }
        const u32 ch = lo & 0xFFu;
        uint4 tab = sh_tab[ch];
        if (first_wave && r == 0u) { // wave-uniform: the root token's number path differs (takes_number_path) -- one token per document
          if (lane == 0) {
            const tok_packed pk = tok_contribution(ch, true);
            tab.x = pk.a; tab.y = pk.b; tab.z = pk.c;
          }
        }
        if (live) {
          tokc_w[tl] = u8(ch);
          a += tab.x; b += tab.y; c += tab.z;
        }
        const bool is_number = live && (tab.z >> 16) != 0u;
        const bool is_atom = live && (tab.w & TP_ATOM) != 0u;
        if (__ballot(is_atom)) { // wave-uniform: rows without true / false / null skip the spelling
          const u32 g = is_atom ? atom_error_of(lo, hi) : 0u; // visit_true_atom ..., tape_builder.h:278-329 (the words themselves: k_tok_apply)
          if (g) { report_error(res, error_key(t0 + tl, 2, g)); }
        }
        const u64 nm = __ballot(is_number);
        const u32 k = __builtin_amdgcn_mbcnt_hi(u32(nm >> 32), __builtin_amdgcn_mbcnt_lo(u32(nm), 0u));
        if (is_number) { list[ncount + k] = (unsigned short)tl; }
        ncount += u32(popc64(nm));
      }
      wave_lds_fence();
      PHASE(2); // rows
      // the group's numbers, one per lane: visit_number (tape_builder.h:213-275) = parse_number (numberparsing.h:859-971, sj_number.h).  First from the window;
      // a token that reaches beyond it -- and every token of a row read in place -- is flagged in the list and parsed from memory in a second sweep.
      u32 again = staged ? 0u : 1u; // wave-uniform
#ifdef SJGPU_LAB_SKIP_NUMBERS
      again = 0;
#endif
      if (staged
#ifdef SJGPU_LAB_SKIP_NUMBERS
          && false
#endif
      ) {
#pragma unroll 1
        for (u32 k0 = g0; k0 < ncount; k0 += 64u) {
          const u32 k = k0 + lane;
          bool redo = false;
          if (k < ncount) {
            const u32 tl = list[k];
            const u32 p = (row_start[tl >> 6] - a0) + u32(rel[tl]); // inside the window: p + SG_OVER <= span
            const staged_bytes src{win8};
            const number_value v = parse_number_token(src, p, static_cast<bigint *>(nullptr));
            redo = src.reach >= span;
            if (redo) { list[k] = (unsigned short)(tl | 0x8000u); }
            else { park_number(v, t0 + tl, t0_64 + k, numbits, numtype, slow_list, slow_cap, res); }
          }
          if (__ballot(redo)) { again = 1u; }
        }
      }
      if (again) { // (wave-uniform, rare in ordinary text)
        wave_lds_fence();
#pragma unroll 1
        for (u32 k0 = g0; k0 < ncount; k0 += 64u) {
          const u32 k = k0 + lane;
          if (k < ncount) {
            const u32 e = list[k], tl = e & 0x7FFFu;
            if (!staged || (e & 0x8000u)) { // through the 8-byte register window (k_tape_numbers' 32-byte one costs this kernel, whose registers are its occupancy, 40 more)
              const u32 p = idx_w[tl];
              const windowed_bytes src{buf, len};
              park_number(parse_number_token(src, p, static_cast<bigint *>(nullptr)), t0 + tl, t0_64 + k, numbits, numtype, slow_list, slow_cap, res);
            }
          }
        }
      }
      wave_lds_fence(); // (the next group's staging overwrites the window: LDS traffic of one wave stays in order)
      PHASE(3); // numbers
#ifdef SJGPU_LAB_STAGE_PHASES
      acc_phase[6]++;
#endif
      ra = rb;
    }
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); // a wave: at most 1024 tokens x 2 words: the fields do not overflow
  if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; sh[2][wave] = c; }
  __syncthreads();
  PHASE(4); // waiting for the workgroup's other waves
#ifdef SJGPU_LAB_STAGE_PHASES
  { const u32 wid = blockIdx.x * 4u + wave; if (lane < 8 && wid < LAB_WAVES) { u32 vv = 0;
#pragma unroll
      for (u32 q = 0; q < 8; q++) { if (lane == q) { vv = acc_phase[q]; } }
      if (lane == 7) { vv = 1u; }
      d_stage_trace[wid][lane] = vv; } }
#endif
  if (tid == 0) {
    u32 slots = 0, sel = 0, strs = 0, opens = 0, closes = 0, numbers = 0;
    for (u32 w = 0; w < TS_THREADS / 64; w++) {
      slots += sh[0][w] & 0xFFFFu; sel += sh[0][w] >> 16;
      strs += sh[1][w] & 0xFFFFu; opens += sh[1][w] >> 16;
      closes += sh[2][w] & 0xFFFFu; numbers += sh[2][w] >> 16;
    }
    sums[0 * nblocks + blockIdx.x] = int(slots);
    sums[1 * nblocks + blockIdx.x] = int(sel);
    sums[2 * nblocks + blockIdx.x] = int(strs);
    sums[3 * nblocks + blockIdx.x] = int(opens);
    sums[4 * nblocks + blockIdx.x] = int(closes);
    sums[5 * nblocks + blockIdx.x] = int(numbers);
  }
}
// one workgroup per row: the block totals become exclusive prefixes, in place; totals[k] = the sum of row k (row 2: the string tokens of
// the list, which the string pass -- it runs between this kernel and k_tok_apply -- takes instead of counting them itself)
__global__ __launch_bounds__(1024) void k_tok_scan_sums(int *__restrict__ sums, u32 nblocks, int *__restrict__ totals) {
  __shared__ int sh[1024];
  const u32 per = (nblocks + 1023) / 1024;
  const u32 lo = min(threadIdx.x * per, nblocks), hi = min(lo + per, nblocks);
  {
    const u32 k = blockIdx.x;
    int *row = sums + size_t(k) * nblocks;
    int sum = 0;
    { // (round 6: eight entries requested at a time -- one dependent load per entry made this kernel 26 us for large_random's 20 000 blocks)
      u32 i = lo;
      for (; i + 8 <= hi; i += 8) {
        int v[8];
#pragma unroll
        for (u32 q = 0; q < 8; q++) { v[q] = row[i + q]; }
#pragma unroll
        for (u32 q = 0; q < 8; q++) { sum += v[q]; }
      }
      for (; i < hi; i++) { sum += row[i]; }
    }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
      const int t = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    int run = sh[threadIdx.x] - sum;
    if (threadIdx.x == 1023) { totals[k] = sh[1023]; }
    {
      u32 i = lo;
      for (; i + 8 <= hi; i += 8) {
        int v[8];
#pragma unroll
        for (u32 q = 0; q < 8; q++) { v[q] = row[i + q]; }
#pragma unroll
        for (u32 q = 0; q < 8; q++) { row[i + q] = run; run += v[q]; }
      }
      for (; i < hi; i++) {
        const int x = row[i];
        row[i] = run;
        run += x;
      }
    }
    __syncthreads();
  }
}
// tpos[i] for i in [0, n] (entry n = the total).  Every token that writes a VALUE word writes it here, where its tape position sits in a register:
// strings (the k-th string token's record is the k-th of the string buffer), atoms (the spelling was checked by k_tok_stage) and -- round 6 -- numbers,
// whose values k_tok_stage parked at [first token of this row + ordinal of the number inside the row] (rounds 3-5 LISTED numbers and atoms for dense
// kernels of their own, 8 bytes per value token written here and read there).  The brackets
// and commas go straight into the sort's input with
// their level: the depth in front of an opening bracket, the depth behind a closing one, and that of the container a comma separates
// the members of -- clamped to [0, kmax] (beyond the nesting limit an error is already certain); *m_out = how many went in
__global__ __launch_bounds__(TS_THREADS, 4) void k_tok_apply(const u8 *__restrict__ tokc, u32 n, u32 kmax, u32 max_depth, const int *__restrict__ sums, u32 nblocks,
                                                         int *__restrict__ tpos, unsigned short *__restrict__ key, u32 *__restrict__ tok,
                                                         int *__restrict__ m_out, int *__restrict__ max_level, const u64 *__restrict__ numbits,
                                                         const u8 *__restrict__ numtype, const u32 *__restrict__ str_offsets, strings_handoff strs,
                                                         u8 *__restrict__ string_buf, u64 *__restrict__ tape, u64 tape_cap, tape_result_dev *__restrict__ res,
                                                         const u32 *__restrict__ entry_tab) {
  // Round 4, second half: the nesting depth is not written anywhere.  What it decides -- "the root value has ended" and the nesting limit
  // (depth_rule, sj_tape_rules.h) and "the list ends inside a container" -- is said HERE, where it sits in a register; the levels of the brackets and
  // commas go into the sort's keys as before.  (k_tape_rules read 4 B per token for it, this kernel wrote them.)
  // Round 4: the string buffer exists when this kernel runs, and the tape words of the string tokens and of the atoms are written HERE,
  // where the token's tape position and its ordinal among the strings sit in registers -- on_start_string (tape_builder.h:415-419): the payload
  // of the word is where the record begins; when the stream compaction wrote the buffer the k-th string token's record begins at outq[k] and
  // its length word is still missing (a record ends where the next begins: [u32 length][bytes][0], on_end_string :428-433), otherwise the
  // per-string kernels left the offsets per token.  k_tape_strings, the list it read and the tape stores of k_tape_atoms (which still checks
  // the spelling of true / false / null) are gone.
  const bool stream_strings = strs.go_stream != nullptr && *strs.go_stream != 0; // uniform
  const u32 cap32 = tape_cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : u32(tape_cap); // (tape positions are ints)
  __shared__ u32 sh[3][TS_THREADS / 64];
  __shared__ u32 sh_props[256]; // what a token's byte IS, as bits (token_entry_of, sj_tape_rules.h: its properties, the walk's state behind it, the states that
                                // accept it): one LDS read instead of ~20 compares per token -- and, round 6, all that the token's own rule needs
  static_assert(TS_THREADS == 256, "one table entry per thread");
  sh_props[threadIdx.x] = entry_tab[threadIdx.x]; // (k_tape_init built it)
  __syncthreads();
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 block0 = blockIdx.x * TS_BLOCK; // (32-bit list indexes throughout: n < 2^32 - 16, sjgpu_stage2_device, and the last block ends below n + 4096)
  const int slots0 = sums[0 * nblocks + blockIdx.x], sel0 = sums[1 * nblocks + blockIdx.x], strs0 = sums[2 * nblocks + blockIdx.x];
  const int opens0 = sums[3 * nblocks + blockIdx.x], closes0 = sums[4 * nblocks + blockIdx.x];
  const int depth0 = opens0 - closes0;
  u32 ra = 0, rb = 0, rc = 0; // what the rows in front of this one hold (packed)
  int top = 0;                // highest level this thread sent into the sort
  u32 err_index = 0, err_low = 0; // this thread's first error: list index, rank << 4 | code (0 = none)
#pragma unroll 1
  for (u32 row = 0; row < TS_ROWS; row++) {
    const u32 i0 = block0 + row * TS_ROW + 4u * tid;
    // the bytes of tokens i0 - 2 ... i0 + 5 in one load (tokc[2 + i] = token i: two zero bytes lead the array, and it has room for n + 9 bytes; what lies behind
    // token n - 1 is only looked at where it is zeros): the thread's four, the two behind them (depth_rule looks at the token behind an opening bracket, a comma at
    // its two followers) and the two in front (the token's own rule, token_rule_self_entries: k_tape_rules of rounds 3-5)
    typedef u64 __attribute__((aligned(1))) u64_unaligned_t;
    u64 around8 = 0;
    if (i0 < n) { around8 = *reinterpret_cast<const u64_unaligned_t *>(tokc + i0); }
    const u32 four = u32(around8 >> 16), behind2 = u32(around8 >> 48);
    const u32 xm2 = sh_props[u32(around8) & 0xFFu], xm1 = sh_props[(u32(around8) >> 8) & 0xFFu]; // the entries of tokens i0 - 2, i0 - 1 (in front of the list: of byte 0)
    u32 x[6]; // the table entries of tokens i0 ... i0 + 5 (what lies at or behind n: zero -- no token)
#pragma unroll
    for (u32 q = 0; q < 6; q++) {
      const u32 c = q < 4 ? (four >> (8u * q)) & 0xFFu : (behind2 >> (8u * (q - 4))) & 0xFFu;
      x[q] = i0 + q < n ? sh_props[c] : 0u;
    }
    // (a token's contribution is unpacked from its entry here and again where it is applied: twelve registers across the scan cost more than nine instructions)
    auto contribution = [&](u32 j) {
      tok_packed pj = tok_contribution_of_props(x[j]);
      if (j == 0 && i0 == 0 && n != 0) { pj = tok_contribution(four & 0xFFu, true); } // the root token: its number path differs (one token per document)
      return pj;
    };
    u32 ta = 0, tb = 0, tc = 0;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      const tok_packed pj = contribution(j);
      ta += pj.a; tb += pj.b; tc += pj.c;
    }
    const u32 ia = wave_incl_scan(ta), ib = wave_incl_scan(tb), ic = wave_incl_scan(tc);
    if (lane == 63) { sh[0][wave] = ia; sh[1][wave] = ib; sh[2][wave] = ic; }
    __syncthreads();
    const u32 rc_row = rc; // what the rows in front hold: (ec - rc_row) >> 16 = the number tokens of THIS row in front of the thread's four
    u32 ea = ra + ia - ta, eb = rb + ib - tb, ec = rc + ic - tc;
    for (u32 w = 0; w < TS_THREADS / 64; w++) {
      if (w < wave) { ea += sh[0][w]; eb += sh[1][w]; ec += sh[2][w]; }
      ra += sh[0][w]; rb += sh[1][w]; rc += sh[2][w];
    }
    __syncthreads();
    if (i0 <= n) {
      int tp[4];
      // where the records of this thread's string tokens begin: its (at most four) strings have consecutive ordinals, so the five words are
      // requested HERE, all at once, not one by one inside the loop below (as dependent loads they made this kernel 120 us longer than the
      // separate pass over a list of string tokens had been: profiles/r04_tape_kernel_stats.txt)
      const u32 k0 = u32(strs0) + (eb & 0xFFFFu);
      u32 oq[5] = {0u, 0u, 0u, 0u, 0u}; // stream: the record starts of strings k0 ... k0 + 4; else (the same registers): the offsets the per-string kernels left for tokens i0 ... i0 + 3
      if (((tb & 0xFFFFu) != 0u)) { // this thread holds string tokens
        if (stream_strings) {
#pragma unroll
          for (u32 q = 0; q < 5; q++) { if (k0 + q <= n + 1u) { oq[q] = strs.outq[k0 + q]; } } // (outq has n + 2 entries)
        } else {
#pragma unroll
          for (u32 q = 0; q < 4; q++) { if (i0 + q < n) { oq[q] = str_offsets[i0 + q]; } }
        }
      }
      // the values of this thread's number tokens: consecutive entries of what k_tok_stage parked for this row (requested at once, like the records above)
      u64 nbits[4] = {0, 0, 0, 0};
      u32 ntypes = 0; // one byte each
      if ((tc >> 16) != 0u) {
        const u32 at0 = (block0 + row * TS_ROW) + ((ec - rc_row) >> 16);
#pragma unroll
        for (u32 q = 0; q < 4; q++) { if (at0 + q <= n && q < (tc >> 16)) { nbits[q] = numbits[at0 + q]; ntypes |= u32(numtype[at0 + q]) << (8u * q); } }
      }
      u32 sk = 0, nk = 0; // strings / numbers of this thread so far
#pragma unroll
      for (u32 j = 0; j < 4; j++) {
        // (written for few branches: a wave of 64 x 4 consecutive tokens holds every kind of token, so every branch is taken by somebody and costs its
        // mask bookkeeping for all -- 1 280 scalar and 940 vector instructions per row before, profiles/r04_tape_kernel_stats.txt; stores of the same
        // width share one site with a selected address, errors are kept in two registers and reported once)
        const u32 i = i0 + j;
        const bool live = i < n;
        const tok_packed pj = contribution(j);
        const u32 ch = (four >> (8u * j)) & 0xFFu, c1 = j < 3 ? (four >> (8u * (j + 1))) & 0xFFu : behind2 & 0xFFu;
        const int d = depth0 + int(eb >> 16) - int(ec & 0xFFFFu);
        tp[j] = slots0 + int(ea & 0xFFFFu);
        {
          u32 rank = 0, rank_self = 0;
          u32 g = live ? depth_rule(i == 0, ch, i + 1 < n ? c1 : 0u, d, max_depth, &rank) : 0u;
          if (i == n && d != 0) { g = SJ_TAPE_ERROR; rank = 0; } // the walk meets the sentinel inside a container
          // what the token says about itself from the two tokens in front of it (the followers of a comma are judged by the comma: k_tape_match)
          const u32 gs = live ? token_rule_self_entries(i == 0, x[j], j == 0 ? xm1 : x[j > 0 ? j - 1 : 0], j == 0 ? xm2 : (j == 1 ? xm1 : x[j > 1 ? j - 2 : 0]), &rank_self) : 0u;
          const u32 low_d = g ? (rank << 4) | g : 0xFFFFu, low_s = gs ? (rank_self << 4) | gs : 0xFFFFu, low = low_d < low_s ? low_d : low_s; // the smaller key of one token
          if (low != 0xFFFFu && err_low == 0u) { err_index = i; err_low = low; } // (a thread meets its tokens in list order: its first error is its smallest key)
        }
        const int strings_before = strs0 + int(eb & 0xFFFFu);
        const int slot = sel0 + int(ea >> 16);
        if (i == n) { *m_out = slot; m_out[3] = slot + 1; m_out[4] = strings_before; }
        const u32 list = value_list_of(pj); // (of the packed contribution, not of the table entry: the root token's differs)
        const bool is_number = list == LIST_NUMBERS, is_string = list == LIST_STRINGS, is_rest = list == LIST_REST;
        const u32 at = u32(tp[j]) + 1u; // (tape positions are ints: below 2^31)
        // (sk is a compile-time-bounded counter: selects, not indexed registers)
        const u32 begin = sk == 0 ? oq[0] : (sk == 1 ? oq[1] : (sk == 2 ? oq[2] : oq[3]));
        const u32 next = sk == 0 ? oq[1] : (sk == 1 ? oq[2] : (sk == 2 ? oq[3] : oq[4]));
        if (is_string && stream_strings) { *reinterpret_cast<u32_unaligned_t *>(string_buf + begin) = next - begin - 5u; }
        const bool is_atom = is_rest && (x[j] & TP_ATOM) != 0u; // visit_true_atom ..., tape_builder.h:278-329
        // the token's (first) value word -- ONE store site for strings, atoms and numbers (visit_number, tape_builder.h:213-275: the type word, then the value; a
        // number k_tok_stage rejected left nothing: the document is in error and its tape is nobody's)
        if ((is_string || is_atom || is_number) && at < cap32) {
          const u32 hi = is_number ? ((ntypes >> (8u * nk)) & 0xFFu) << 24 : (is_string ? u32('"') << 24 : ch << 24);
          const u32 lo = is_string ? (stream_strings ? begin : oq[j]) : 0u;
          tape[at] = (u64(hi) << 32) | lo;
        }
        if (is_number && at + 1u < cap32) { tape[at + 1] = nk == 0 ? nbits[0] : (nk == 1 ? nbits[1] : (nk == 2 ? nbits[2] : nbits[3])); }
        nk += is_number ? 1u : 0u;
        sk += is_string ? 1u : 0u;
        if (live && (pj.a >> 16)) {
          int k = (pj.b >> 16) ? d : d - 1;
          k = k < 0 ? 0 : (k > int(kmax) ? int(kmax) : k);
          // a bracket travels with its tape position, a comma with its list index and with what it knows about its two followers (sj_tape_rules.h)
          const bool comma = (x[j] & TP_COMMA) != 0u;
          const u32 fine = comma_fine_bits_of_props(x[j + 1], i + 1 < n, x[j + 2], i + 2 < n);
          key[slot] = (unsigned short)sort_key_of_props(u32(k), x[j], fine);
          tok[slot] = comma ? i : u32(tp[j]);
          top = k > top ? k : top;
        }
        ea += pj.a; eb += pj.b; ec += pj.c;
      }
      if (i0 + 3 <= n) {
        *reinterpret_cast<int4 *>(tpos + i0) = make_int4(tp[0], tp[1], tp[2], tp[3]);
      } else {
        for (u32 j = 0; j < 4 && i0 + j <= n; j++) { tpos[i0 + j] = tp[j]; }
      }
    }
  }
  if (err_low != 0u) { report_error(res, (u64(err_index) << 8) | err_low); }
  // one atomic per wave, and only from waves that raise the mark (it decides whether the sort needs its second pass)
  const u32 wave_top = wave_max(u32(top));
  if (lane == 0 && int(wave_top) > __hip_atomic_load(max_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicMax(max_level, int(wave_top)); }
}

// ---- stable radix sort on the level, one digit of RADIX_BITS per pass ----------------------------------------------------------------
// m = number of elements = selpos[n] (device memory).  hist is digit-major: hist[d * live + t] (live = the tiles that hold elements), so that ONE exclusive scan of the
// whole table yields, for every (digit, tile), where that tile's elements with that digit begin in the output.
// second pass (shift > 0): nothing to do when every level fits into the first digit; the first pass tells the second's scan how long it is
// Round 4: a SECOND table behind the first counts the opening brackets per (digit, tile): scanned with the first (one scan over both), it
// tells the scatter how many opening brackets lie in front of every sorted element -- the container ordinals -- so that the pass over the
// sorted keys, the scan over m + 1 flags and the pass that listed the opens' positions (k_tape_opens / k_tape_openpos, 110 us per
// twitter-like call) are gone: the last scatter writes both arrays on its way.
__global__ __launch_bounds__(64) void k_radix_hist(const unsigned short *__restrict__ key, const int *__restrict__ m_ptr, u32 shift, u32 tiles, int *__restrict__ hist,
                                                   const int *__restrict__ max_level, u32 *__restrict__ scan_len) {
  const u32 lane = threadIdx.x, tile = blockIdx.x;
  const bool one_pass = u32(*max_level) < RADIX_BINS;
  const u32 m = u32(*m_ptr);
  // Round 6: the tables are as wide as the tiles that HOLD elements (live = ceil(m / 2048), known on the device only; the grid and the room are sized for
  // n + 1 tokens): the scans run over 128 x live entries and the tiles behind write nothing -- large_random: two tiles in three, each 128 scattered words of
  // zeros per call (0.18 GB of partial lines) for the scan to read.  The first pass tells both scans their lengths.
  const u32 live = (m + RADIX_TILE - 1u) / RADIX_TILE;
  (void)tiles;
  if (shift == 0 && tile == 0 && lane == 0) {
    scan_len[0] = 2u * live * RADIX_BINS;                   // this pass's scan (n_words[1])
    scan_len[1] = one_pass ? 0u : 2u * live * RADIX_BINS;   // the second pass's (n_words[2])
  }
  if (shift != 0 && one_pass) { return; }
  if (tile >= live) { return; }
  const u32 base = tile * RADIX_TILE;
  // the tile's keys are requested at once (a wave per tile and a load per step made 32 dependent round trips: there are fewer tiles than the chip holds waves)
  u32 kk[RADIX_STEPS];
#pragma unroll
  for (u32 q = 0; q < RADIX_STEPS; q++) {
    const u32 j = base + q * 64u + lane;
    kk[q] = j < m ? u32(key[j]) : RADIX_DEAD;
  }
  // Round 6: lane l counts bin l in a register -- the elements of a step whose digit is l are an AND of the six ballots of the digit's bits (each taken or
  // complemented by the bit of l), no LDS and no atomic.  Rounds 3-5 added every element to an LDS word with an atomic: the 64 elements of a step hold two
  // or three different levels in ordinary text, and 64 atomics on two or three words are served one after the other (large_random: 27 M bank-conflict cycles
  // per call, 133 us for a kernel of 9 M VALU instructions, profiles/r06_pmc_summary.txt); a loop over the DISTINCT digits of a step took that to 83 us and
  // twitter-like, with six to eight levels per step, from 39 to 53 (profiles/r06_tape_stage.txt).
  u32 c0 = 0, c1 = 0;
  {
#pragma unroll
    for (u32 q = 0; q < RADIX_STEPS; q++) {
      const u32 k = kk[q];
      const bool live = k != RADIX_DEAD;
      const u32 d = (k >> shift) & (RADIX_BINS - 1);
      u64 mine = __ballot(live);
#pragma unroll
      for (u32 bit = 0; bit < RADIX_BITS; bit++) {
        const u64 ones = __ballot(live && ((d >> bit) & 1u));
        mine &= ((lane >> bit) & 1u) ? ones : ~ones;
      }
      c0 += u32(popc64(mine));
      c1 += u32(popc64(mine & __ballot(live && kind_is_open(k >> KIND_SHIFT))));
    }
  }
  hist[lane * live + tile] = int(c0);
  hist[(RADIX_BINS + lane) * live + tile] = int(c1);
}
__global__ __launch_bounds__(64) void k_radix_scatter(const unsigned short *__restrict__ key_in, const u32 *__restrict__ tok_in, const int *__restrict__ m_ptr, u32 shift,
                                                      u32 tiles, const int *__restrict__ hist, unsigned short *__restrict__ key_out, u32 *__restrict__ tok_out,
                                                      const int *__restrict__ max_level, int *__restrict__ opens_before, u32 *__restrict__ openpos) {
  __shared__ u32 next[RADIX_BINS];  // where the next element of each digit goes
  __shared__ u32 onext[RADIX_BINS]; // opening brackets in front of it (the scan ran over both tables: the first one's total, m, is in every entry of the second)
  const u32 lane = threadIdx.x, tile = blockIdx.x;
  if (shift != 0 && u32(*max_level) < RADIX_BINS) { return; }
  const u32 m = u32(*m_ptr);
  // the grid is sized for n + 1 elements (m is only known on the device): a tile behind the m that exist has nothing to move.  Rounds 3-5 let it run its 32
  // steps of ballots over dead lanes -- two tiles in three of large_random's grid, 226 us for the kernel (profiles/r06_tape_stage.txt)
  if (tile * RADIX_TILE >= m) { return; }
  const u32 live = (m + RADIX_TILE - 1u) / RADIX_TILE; // the tables' width: k_radix_hist
  (void)tiles;
  next[lane] = u32(hist[lane * live + tile]);
  onext[lane] = u32(hist[(RADIX_BINS + lane) * live + tile]) - m;
  wave_lds_fence();
  const u32 base = tile * RADIX_TILE;
  // keys and payloads of the whole tile are requested at once (see k_radix_hist), then the 32 steps run out of registers
  u32 kk[RADIX_STEPS], tt[RADIX_STEPS];
#pragma unroll
  for (u32 q = 0; q < RADIX_STEPS; q++) {
    const u32 j = base + q * 64u + lane;
    kk[q] = j < m ? u32(key_in[j]) : RADIX_DEAD;
    tt[q] = j < m ? tok_in[j] : 0u;
  }
#pragma unroll
  for (u32 q = 0; q < RADIX_STEPS; q++) { // 64 consecutive elements per step, in order: the sort is stable
    const bool live = kk[q] != RADIX_DEAD;
    const u32 k = live ? kk[q] : 0u;
    const u32 d = (k >> shift) & (RADIX_BINS - 1);
    u64 peers = __ballot(live); // lanes with my digit
#pragma unroll
    for (u32 b = 0; b < RADIX_BITS; b++) {
      const u64 ones = __ballot(live && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? ones : ~ones;
    }
    const u32 rank = u32(popc64(peers & lanemask_lt(lane)));
    const bool open = live && kind_is_open(k >> KIND_SHIFT);
    const u64 open_peers = peers & __ballot(open);
    if (live) {
      const u32 at = next[d] + rank;
      key_out[at] = (unsigned short)k;
      tok_out[at] = tt[q];
      // (valid in the order of the LAST pass that runs; an earlier pass's values are overwritten by it)
      const u32 ob = onext[d] + u32(popc64(open_peers & lanemask_lt(lane))); // opening brackets in front of the element
      opens_before[at] = int(ob);
      if (open) { openpos[ob] = at; }
    }
    wave_lds_fence();
    if (live && rank == 0) { // one lane per digit present
      next[d] += u32(popc64(peers));
      onext[d] += u32(popc64(open_peers));
    }
    wave_lds_fence();
  }
}

// Where the sorted elements are: two passes leave them in the first buffer pair; a document whose levels all fit into one digit
// (nesting below 64 -- nearly every document) is sorted after the first pass, the second is skipped and they sit in the other pair.
struct sorted_pairs {
  const unsigned short *key_two, *key_one;
  const u32 *tok_two, *tok_one;
  const int *max_level;
  __device__ __forceinline__ bool one_pass() const { return u32(*max_level) < RADIX_BINS; }
  __device__ __forceinline__ const unsigned short *key() const { return one_pass() ? key_one : key_two; }
  __device__ __forceinline__ const u32 *tok() const { return one_pass() ? tok_one : tok_two; }
};

// ---- containers -------------------------------------------------------------------------------------------------------------------------
// (opens_before[j] = opening brackets in front of sorted element j and openpos[k] = sorted position of the k-th opening bracket come from the
// sort's last scatter)
// commas: judge the two tokens behind them by their container's kind (rounds 3-4a: wrote that kind to a byte array for k_tape_rules to read --
// a one-byte scatter per comma).  Closing brackets: the two bracket words of the tape
// (end_container, tape_builder.h:396-407; an empty container is the same formula with count 0, :386-391).
// Four consecutive sorted elements per thread, the loads of each step of the chain (element -> its container's open -> that open's key and
// payload) issued for all four before any is used: with one element per thread this kernel ran at the latency of its chain.  Rounds 3-4a had two more
// steps -- the tape positions of the close and of its open, gathered from the per-token array, and a byte scattered (then: two gathered) per comma.
constexpr u32 TM_PER = 4;
__global__ __launch_bounds__(TP_THREADS) void k_tape_match(sorted_pairs sorted, const int *__restrict__ m_ptr, const int *__restrict__ opens_before,
                                                          const u32 *__restrict__ openpos, const int *__restrict__ tpos, const u8 *__restrict__ tokc, u32 n,
                                                          u64 *__restrict__ tape, u64 tape_cap, tape_result_dev *__restrict__ res) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { // what belongs to no token (k_tape_rules' last thread, rounds 3-5): the root words, the sizes, the root container's end
    const u64 words = u64(u32(tpos[n])) + 2;
    res->tape_words = words;
    res->max_level = u32(*sorted.max_level);
    if (words <= tape_cap) {
      tape[0] = tape_word('r', words);           // visit_document_end, tape_builder.h:160-165
      tape[words - 1] = tape_word('r', 0);
    } else {
      res->overflow = 1;
    }
    const u32 c0 = tokc[2], last = tokc[n + 1];
    if ((c0 == '{' && last != '}') || (c0 == '[' && last != ']')) { report_error(res, error_key(0, 0, SJ_TAPE_ERROR)); } // json_iterator.h:138-143
  }
  const unsigned short *__restrict__ key = sorted.key();
  const u32 *__restrict__ tok = sorted.tok();
  const u64 j0 = (u64(blockIdx.x) * TP_THREADS + threadIdx.x) * TM_PER;
  const u32 m = u32(*m_ptr);
  if (j0 >= m) { return; }
  if (key == nullptr) { return; } // a document that needs the sort's second pass, which was not enqueued: the caller runs the tape again (launch_tape)
  u32 kj[TM_PER], cid[TM_PER], ti[TM_PER];
  bool live[TM_PER];
#pragma unroll
  for (u32 e = 0; e < TM_PER; e++) {
    const u64 j = j0 + e;
    const bool in = j < m;
    const u64 at = in ? j : j0; // (j0 < m: a valid element to read instead)
    kj[e] = key[at];
    cid[e] = u32(opens_before[at]); // an element that is not an open: opens in front of it = opens at positions <= j
    ti[e] = tok[at];
    live[e] = in && !kind_is_open(kj[e] >> KIND_SHIFT) && cid[e] != 0;
  }
  u32 jo[TM_PER];
#pragma unroll
  for (u32 e = 0; e < TM_PER; e++) { jo[e] = live[e] ? openpos[cid[e] - 1] : 0u; } // (openpos[0] does not exist in a list without brackets)
  u32 ko[TM_PER], io[TM_PER];
#pragma unroll
  for (u32 e = 0; e < TM_PER; e++) {
    ko[e] = key[jo[e]];
    io[e] = tok[jo[e]]; // the opening bracket's tape position
    live[e] = live[e] && ((ko[e] ^ kj[e]) & ((1u << KIND_SHIFT) - 1)) == 0; // else: no container of this level in front, the token's own rule reports it
  }
#pragma unroll
  for (u32 e = 0; e < TM_PER; e++) {
    if (!live[e]) { continue; }
    const u32 kind = kj[e] >> KIND_SHIFT;
    const bool object = (ko[e] >> KIND_SHIFT) == KIND_OPEN_OBJECT;
    if (kind_is_comma(kind)) {
      // the comma judges its two followers: nobody else knows what its container wants there (sj_tape_rules.h).  It said what it could when it was sent
      // into the sort -- fine in an object, fine in an array -- and has nothing to fetch unless the document is broken here
      if ((kind - KIND_COMMA) & (object ? COMMA_FINE_IN_OBJECT : COMMA_FINE_IN_ARRAY)) { continue; }
      const u32 i = ti[e];
      typedef unsigned short __attribute__((aligned(1))) u16_unaligned_t;
      const u32 behind = u32(*reinterpret_cast<const u16_unaligned_t *>(tokc + i + 3)); // (tokc[i + 2] = token i; two zero bytes end the array)
      const follower_keys f = comma_followers_rule(i, object ? CTX_OBJECT : CTX_ARRAY, behind & 0xFFu, i + 1 < n, behind >> 8, i + 2 < n);
      const u64 k01 = f.k[0] < f.k[1] ? f.k[0] : f.k[1], k = k01 < f.k[2] ? k01 : f.k[2];
      if (k != NO_ERROR_KEY) { report_error(res, k); }
      continue;
    }
    const u32 open_tp = io[e], close_tp = ti[e]; // tape positions travel with the brackets: nothing to gather
    if ((kind == KIND_CLOSE_OBJECT) != object) { report_error(res, error_key(token_at_tape_position(tpos, n, close_tp), 0, SJ_TAPE_ERROR)); }
    const u64 open_at = 1 + u64(open_tp), close_at = 1 + u64(close_tp);
    const u64 between = (j0 + e) - jo[e]; // commas + 1
    const u64 count = (close_tp == open_tp + 1) ? 0 : (between > 0xFFFFFFull ? 0xFFFFFFull : between);
    if (close_at < tape_cap) {
      tape[close_at] = tape_word(kind == KIND_CLOSE_OBJECT ? u32('}') : u32(']'), open_at);
      tape[open_at] = tape_word(object ? u32('{') : u32('['), (count << 32) | (close_at + 1));
    } else {
      res->overflow = 1;
    }
  }
}

// (Rounds 3-5 had a kernel per token here, k_tape_rules: the part of the walk's rule a token can check from the two bytes in front of it, from three LDS
// tables -- 144 us per large_random call and the token bytes read once more.  Round 6: k_tok_apply holds every token's table entry anyway and says it there
// (token_rule_self_entries); the root words and the check that belongs to no token moved to k_tape_match's first thread.)
// (Rounds 3-5 had two dense kernels here, k_tape_atoms and k_tape_numbers, fed from lists k_tok_apply wrote: one listed token per lane, the token's offset
// and then its bytes fetched through two dependent round trips.  Round 4 tried the spelling inside the gathering classifier -- 120 -> 244 us, a second
// dependent fetch in a kernel bound by the latency of its gather; round 6 staged the document's bytes in LDS (k_tok_stage) and both kernels are gone.)
// Number tokens with more than 19 significant digits whose two bracketing conversions disagree: the exact decision needs two
// big integers of 516 bytes each per thread (private memory) -- kept out of k_tape_numbers, which then needs no scratch at all.
__global__ __launch_bounds__(64) void k_tape_slow_numbers(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, const int *__restrict__ tpos, const u32 *__restrict__ slow_list,
                                                          u32 slow_cap, u64 *__restrict__ tape, u64 tape_cap, tape_result_dev *__restrict__ res) {
  const u32 count = res->slow_numbers < slow_cap ? res->slow_numbers : slow_cap;
  const dev_bytes src{buf, u32(len)};
  bigint big[2];
  for (u32 k = blockIdx.x * 64 + threadIdx.x; k < count; k += gridDim.x * 64) {
    const u32 i = slow_list[k];
    number_shape shape;
    const number_value v = parse_number_token(src, idx[i], static_cast<bigint *>(nullptr), &shape);
    u64 bits = 0;
    if (!decide_long_decimal(src, shape, big, bits)) { report_error(res, error_key(i, 2, SJ_NUMBER_ERROR)); continue; }
    const u64 at = 1 + u64(u32(tpos[i]));
    if (at + 1 < tape_cap) { tape[at + 1] = v.bits | bits; }
  }
}

} // namespace

#ifdef SJGPU_LAB_STAGE_PHASES
extern "C" int sjgpu_lab_stage_phases(unsigned long long *out) { // sums the per-wave records (ticks of 10 ns; [6] groups, [7] waves) and clears them
  static std::vector<u32> host;
  host.resize(size_t(LAB_WAVES) * 8);
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(d_stage_trace), host.size() * 4) != hipSuccess) { return -1; }
  for (u32 k = 0; k < 8; k++) { out[k] = 0; }
  for (size_t w = 0; w < LAB_WAVES; w++) { for (u32 k = 0; k < 8; k++) { out[k] += host[w * 8 + k]; } }
  std::fill(host.begin(), host.end(), 0u);
  return hipMemcpyToSymbol(HIP_SYMBOL(d_stage_trace), host.data(), host.size() * 4) == hipSuccess ? 0 : -1;
}
#endif
static inline u32 blocks_of(u64 n, u32 per) { return u32((n + per - 1) / per); }

// Workspace layout for n structurals (every array 256-byte aligned): see tape_workspace below.
struct tape_workspace {
  tape_result_dev *res;
  u32 *n_words;       // [0] = n + 1 (scan lengths), [1] = n (upper bound of the sorted elements + 1 for the opens scan), [2] = hist length per pass
  u8 *tokc;
  int *slots;                // tape position of every token (entry n: the total)
  int *m;                    // brackets and commas = elements of the sort
  int *sums;                 // k_tok_stage's block totals (6 rows)
  int *totals;               // ... and the sums of the rows (k_tok_scan_sums): [2] = the string tokens of the list
  u64 *numbits;              // the value words of the number tokens as k_tok_stage parked them: [first token of the wave's row + ordinal inside the row]
  u8 *numtype;               // ... and their types ('l', 'u', 'd')
  u32 *entry_tab;            // token_entry_of of every byte value (k_tok_apply's table), built by k_tape_init
  uint4 *stage_tab;          // k_tok_stage's table
  u32 tok_blocks;
  unsigned short *key_a, *key_b;
  u32 *tok_a, *tok_b, *openpos, *slow_list;
  int *hist, *opens, *partial;
  u32 tiles, slow_cap;
  size_t bytes;
};
static tape_workspace carve(uint8_t *base, uint32_t n, uint64_t len) {
  tape_workspace w{};
  size_t at = 0;
  auto take = [&](size_t bytes) { uint8_t *p = base ? base + at : nullptr; at += (bytes + 255) & ~size_t(255); return p; };
  const size_t n1 = size_t(n) + 1;
  w.tiles = blocks_of(n1, RADIX_TILE);
  w.slow_cap = u32(len / 20 + 64 < n1 ? len / 20 + 64 : n1);
  w.res = reinterpret_cast<tape_result_dev *>(take(sizeof(tape_result_dev)));
  w.n_words = reinterpret_cast<u32 *>(take(128)); // (behind the result: launch_tape_front clears both with one memset)
  w.totals = reinterpret_cast<int *>(w.n_words) + 16; // [16 .. 21]
  w.entry_tab = reinterpret_cast<u32 *>(take(256 * 4));
  w.stage_tab = reinterpret_cast<uint4 *>(take(256 * 16));
  w.m = reinterpret_cast<int *>(w.n_words) + 8; // [8] = m, [9] = highest level in the sort, [11] = m + 1 (length of the opens scan), [12] = string tokens;
  // n_words[2] = length of the second pass's scan
  w.tokc = take(n1 + 8);
  w.slots = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.numbits = reinterpret_cast<u64 *>(take(n1 * 8 + 64));
  w.numtype = take(n1 + 64);
  w.tok_blocks = blocks_of(n1, TS_BLOCK);
  w.sums = reinterpret_cast<int *>(take(size_t(w.tok_blocks) * TS_SUMS * 4 + 64));
  w.key_a = reinterpret_cast<unsigned short *>(take(n1 * 2 + 64));
  w.key_b = reinterpret_cast<unsigned short *>(take(n1 * 2 + 64));
  w.tok_a = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.tok_b = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.openpos = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.opens = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.slow_list = reinterpret_cast<u32 *>(take(size_t(w.slow_cap) * 4 + 64));
  w.hist = reinterpret_cast<int *>(take(size_t(w.tiles) * RADIX_BINS * 2 * 4 + 64)); // two tables: elements, opening brackets
  const size_t longest = n1 > size_t(w.tiles) * RADIX_BINS * 2 ? n1 : size_t(w.tiles) * RADIX_BINS * 2;
  w.partial = reinterpret_cast<int *>(take((longest / 4096 + 80) * 4));
  w.bytes = at;
  return w;
}
size_t tape_workspace_bytes(uint32_t n, uint64_t len) { return carve(nullptr, n, len).bytes; }

// Stage 2 in two halves, the string buffer in between (sjgpu_capi_stage2.hip: sjgpu_stage2_device).  idx[0 .. n] (n >= 1; idx[n] = len, stage 1's first
// sentinel); workspace: tape_workspace_bytes(n, len), its first bytes are the tape_result_dev afterwards.
// launch_tape_front: the byte of every token and the block totals of the six per-token counters.  Returns (a device pointer to) the number of
// string tokens, which the string pass takes instead of counting them itself.  launch_tape (behind the string pass): tape position and depth of
// every token, the string and atom words, the lists of the other value tokens, the sort, the brackets, the rules, the numbers.
const int *launch_tape_front(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, void *workspace, hipStream_t s, const uint8_t *tok) {
  (void)max_depth;
  const tape_workspace w = carve(static_cast<uint8_t *>(workspace), n, len);
  const u32 n1 = n + 1;
  // one small kernel for the result and the control words behind it (they share the first 512 bytes of the workspace): all zero but three words
  // (rounds 3-4a: four memsets -- four launches of the runtime's fill kernel, 4 us each)
  hipLaunchKernelGGL(k_tape_init, dim3(1), dim3(64), 0, s, w.res, w.n_words, u32(reinterpret_cast<uint8_t *>(w.n_words + 16) - reinterpret_cast<uint8_t *>(w.res)) / 4u, n1,
                     2u * w.tiles * RADIX_BINS, w.entry_tab, w.stage_tab);
  // (tok: since round 6 the front stages the document's bytes for the numbers and the atoms anyway and takes the token bytes from the same window; the
  // stream is accepted and not read)
  (void)tok;
  hipLaunchKernelGGL(k_tok_stage, dim3(w.tok_blocks), dim3(TS_THREADS), 0, s, buf, u32(len), idx, n, w.tokc, w.sums, w.tok_blocks, w.numbits, w.numtype, w.slow_list, w.slow_cap,
                     w.res, w.stage_tab);
  hipLaunchKernelGGL(k_tok_scan_sums, dim3(TS_SUMS), dim3(1024), 0, s, w.sums, w.tok_blocks, w.totals);
  return w.totals + 2; // the number of string tokens (device)
}

// launch_tape: the rest, behind the string pass.  str_offsets: what the per-string kernels left (n + 1 words), read when they wrote the buffer;
// strs: where the records of the stream compaction begin, read when IT wrote the buffer (the flag decides on the device); string_buf: the
// buffer, for the length words.
void launch_tape(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, const uint32_t *str_offsets, strings_handoff strs,
                 uint8_t *string_buf, uint64_t *tape, uint64_t tape_cap, void *workspace, hipStream_t s, bool deep) {
  const tape_workspace w = carve(static_cast<uint8_t *>(workspace), n, len);
  const u32 n1 = n + 1;
  const int *m_ptr = w.m;
  const u32 kmax = max_depth < 4095u ? max_depth : 4095u;
  hipLaunchKernelGGL(k_tok_apply, dim3(w.tok_blocks), dim3(TS_THREADS), 0, s, w.tokc, n, kmax, max_depth, w.sums, w.tok_blocks, w.slots, w.key_a, w.tok_a, w.m, w.m + 1,
                     w.numbits, w.numtype, str_offsets, strs, string_buf, tape, tape_cap, w.res, w.entry_tab);
  // two passes of six bits cover levels up to 4095; the second one only runs for documents nested 64 deep and more
  const int *max_level = w.m + 1;
  hipLaunchKernelGGL(k_radix_hist, dim3(w.tiles), dim3(64), 0, s, w.key_a, m_ptr, 0u, w.tiles, w.hist, max_level, w.n_words + 1);
  enqueue_scan(w.hist, 2 * w.tiles * RADIX_BINS, w.n_words + 1, w.partial, s);
  hipLaunchKernelGGL(k_radix_scatter, dim3(w.tiles), dim3(64), 0, s, w.key_a, w.tok_a, m_ptr, 0u, w.tiles, w.hist, w.key_b, w.tok_b, max_level, w.opens, w.openpos);
  if (deep) {
    hipLaunchKernelGGL(k_radix_hist, dim3(w.tiles), dim3(64), 0, s, w.key_b, m_ptr, RADIX_BITS, w.tiles, w.hist, max_level, w.n_words + 1);
    enqueue_scan(w.hist, 2 * w.tiles * RADIX_BINS, w.n_words + 2, w.partial, s);
    hipLaunchKernelGGL(k_radix_scatter, dim3(w.tiles), dim3(64), 0, s, w.key_b, w.tok_b, m_ptr, RADIX_BITS, w.tiles, w.hist, w.key_a, w.tok_a, max_level, w.opens, w.openpos);
  }
  // (without the second pass a document nested 64 deep and more is NOT sorted: k_tape_match then leaves at once -- key_two == nullptr says so -- and the
  // caller, who finds max_level in the result, comes back with deep = true)
  const sorted_pairs sorted{deep ? w.key_a : nullptr, w.key_b, deep ? w.tok_a : nullptr, w.tok_b, max_level};
  // containers: the ordinals came with the last scatter
  hipLaunchKernelGGL(k_tape_match, dim3(blocks_of(n1, TP_THREADS * TM_PER)), dim3(TP_THREADS), 0, s, sorted, m_ptr, w.opens, w.openpos, w.slots, w.tokc, n, tape, tape_cap, w.res);
  hipLaunchKernelGGL(k_tape_slow_numbers, dim3(64), dim3(64), 0, s, buf, len, idx, w.slots, w.slow_list, w.slow_cap, tape, tape_cap, w.res);
}

} // namespace sjgpu
