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
//   k_tape_classify   token byte, word count, bracket delta, "goes into the sort" flag            1 thread / token
//   3 x scan          tape positions, depths, sort slots (sjgpu_finish.hip's scan kernels)
//   k_tape_select     (level, token) of every bracket and comma into the sort's input
//   2 x radix pass    stable LSD radix sort on the level, 6 bits per pass: histogram, scan, scatter   1 wave / 2048 elements
//   k_tape_opens + scan + k_tape_openpos   container ordinal per sorted element, sorted position of every open
//   k_tape_match      commas learn their container's kind; closes write BOTH bracket words (count, partner index), kinds checked
//   k_tape_write      per token: the walk's rule, nesting limit, numbers / atoms / string words, root words
//   k_tape_slow_numbers  the handful of number tokens whose rounding needs exact big-integer arithmetic (sj_number.h)
// Parity: tests/test_gpu_parity.py::test_tape_* against the live reference's dom::parser::parse (tape and string_buf word for word,
// error codes of broken documents); the same steps run on the CPU in tests/host/test_tape_model.cpp.
#include "sjgpu_device.h"
#include "sj_tape_rules.h"

namespace sjgpu {
namespace {

constexpr u32 TP_THREADS = 256;
constexpr u32 RADIX_BITS = 6, RADIX_BINS = 1u << RADIX_BITS, RADIX_TILE = 2048; // one wave sorts one tile

struct dev_bytes {
  const u8 *buf;
  u32 len;
  __device__ __forceinline__ u32 byte(u32 pos) const { return pos < len ? u32(buf[pos]) : 0x20u; }
};
// The same bytes through an 8-byte window in registers: a token is read front to back, so one (unaligned) 8-byte load serves eight
// byte() calls -- the per-byte loads of the first version were what made k_tape_write the longest kernel of the tape
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

__device__ __forceinline__ void report_error(tape_result_dev *res, u64 key) { atomicMin(reinterpret_cast<unsigned long long *>(&res->error_key), (unsigned long long)key); }

// tokc holds the token bytes with TWO zero bytes in front and behind: tokc[i + 2] = byte of token i
__global__ __launch_bounds__(TP_THREADS) void k_tape_classify(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, u32 n, u8 *__restrict__ tokc,
                                                             int *__restrict__ slots, int *__restrict__ delta, int *__restrict__ sel) {
  const u64 i = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  if (i > n) { return; }
  if (i == n) { // the slot behind the list: the scans turn it into the totals
    slots[n] = 0; delta[n] = 0; sel[n] = 0;
    tokc[0] = 0; tokc[1] = 0; tokc[n + 2] = 0; tokc[n + 3] = 0;
    return;
  }
  const u32 pos = idx[i];
  const u32 c = pos < len ? u32(buf[pos]) : 0x20u;
  tokc[i + 2] = u8(c);
  slots[i] = int(tape_slots(c, i == 0));
  delta[i] = is_open_char(c) ? 1 : (is_close_char(c) ? -1 : 0);
  sel[i] = (is_open_char(c) || is_close_char(c) || c == ',') ? 1 : 0;
}

// level of a sorted element: the depth in front of an opening bracket, the depth behind a closing one, and that of the
// container a comma separates the members of -- clamped to [0, kmax] (beyond the nesting limit an error is already certain)
__global__ __launch_bounds__(TP_THREADS) void k_tape_select(const u8 *__restrict__ tokc, const int *__restrict__ depth, const int *__restrict__ selpos, u32 n, u32 kmax,
                                                           unsigned short *__restrict__ key, u32 *__restrict__ tok) {
  const u64 i = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  if (i >= n) { return; }
  const u32 c = tokc[i + 2];
  if (!(is_open_char(c) || is_close_char(c) || c == ',')) { return; }
  int k = is_open_char(c) ? depth[i] : depth[i] - 1;
  k = k < 0 ? 0 : (k > int(kmax) ? int(kmax) : k);
  const u32 at = u32(selpos[i]);
  key[at] = (unsigned short)k;
  tok[at] = u32(i);
}

// ---- stable radix sort on the level, one digit of RADIX_BITS per pass ----------------------------------------------------------------
// m = number of elements = selpos[n] (device memory).  hist is digit-major: hist[d * tiles + t], so that ONE exclusive scan of the
// whole table yields, for every (digit, tile), where that tile's elements with that digit begin in the output.
__global__ __launch_bounds__(64) void k_radix_hist(const unsigned short *__restrict__ key, const int *__restrict__ m_ptr, u32 shift, u32 tiles, int *__restrict__ hist) {
  __shared__ u32 cnt[RADIX_BINS];
  const u32 lane = threadIdx.x, tile = blockIdx.x;
  const u32 m = u32(*m_ptr);
  cnt[lane] = 0;
  wave_lds_fence();
  const u32 base = tile * RADIX_TILE;
  for (u32 r = 0; r < RADIX_TILE; r += 64) {
    const u32 j = base + r + lane;
    if (j < m) { atomicAdd(&cnt[(key[j] >> shift) & (RADIX_BINS - 1)], 1u); }
  }
  wave_lds_fence();
  hist[lane * tiles + tile] = int(cnt[lane]);
}
__global__ __launch_bounds__(64) void k_radix_scatter(const unsigned short *__restrict__ key_in, const u32 *__restrict__ tok_in, const int *__restrict__ m_ptr, u32 shift,
                                                      u32 tiles, const int *__restrict__ hist, unsigned short *__restrict__ key_out, u32 *__restrict__ tok_out) {
  __shared__ u32 next[RADIX_BINS]; // where the next element of each digit goes
  const u32 lane = threadIdx.x, tile = blockIdx.x;
  const u32 m = u32(*m_ptr);
  next[lane] = u32(hist[lane * tiles + tile]);
  wave_lds_fence();
  const u32 base = tile * RADIX_TILE;
  for (u32 r = 0; r < RADIX_TILE; r += 64) { // 64 consecutive elements per step, in order: the sort is stable
    const u32 j = base + r + lane;
    const bool live = j < m;
    const u32 k = live ? u32(key_in[j]) : 0u;
    const u32 d = (k >> shift) & (RADIX_BINS - 1);
    u64 peers = __ballot(live); // lanes with my digit
#pragma unroll
    for (u32 b = 0; b < RADIX_BITS; b++) {
      const u64 ones = __ballot(live && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? ones : ~ones;
    }
    const u32 rank = u32(popc64(peers & lanemask_lt(lane)));
    if (live) {
      const u32 at = next[d] + rank;
      key_out[at] = (unsigned short)k;
      tok_out[at] = tok_in[j];
    }
    wave_lds_fence();
    if (live && rank == 0) { next[d] += u32(popc64(peers)); } // one lane per digit present
    wave_lds_fence();
  }
}

// ---- containers -------------------------------------------------------------------------------------------------------------------------
// opens[j] = 1 where the sorted element j is an opening bracket (the scan turns it into "opens in front of j"); opens[m] = 0
__global__ __launch_bounds__(TP_THREADS) void k_tape_opens(const u8 *__restrict__ tokc, const u32 *__restrict__ tok, const int *__restrict__ m_ptr, u32 n, int *__restrict__ opens) {
  const u64 j = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  const u32 m = u32(*m_ptr);
  if (j > n) { return; }
  opens[j] = (j < m && is_open_char(tokc[tok[j] + 2])) ? 1 : 0; // zeros behind the m sorted elements: the scan runs over n + 1 entries
}
// openpos[k] = sorted position of the k-th opening bracket
__global__ __launch_bounds__(TP_THREADS) void k_tape_openpos(const u8 *__restrict__ tokc, const u32 *__restrict__ tok, const int *__restrict__ m_ptr, const int *__restrict__ opens_before,
                                                            u32 *__restrict__ openpos) {
  const u64 j = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  const u32 m = u32(*m_ptr);
  if (j >= m) { return; }
  if (is_open_char(tokc[tok[j] + 2])) { openpos[opens_before[j]] = u32(j); }
}
// commas: ctx[token] = kind of their container.  Closing brackets: the two bracket words of the tape
// (end_container, tape_builder.h:396-407; an empty container is the same formula with count 0, :386-391).
__global__ __launch_bounds__(TP_THREADS) void k_tape_match(const u8 *__restrict__ tokc, const unsigned short *__restrict__ key, const u32 *__restrict__ tok, const int *__restrict__ m_ptr,
                                                          const int *__restrict__ opens_before, const u32 *__restrict__ openpos, const int *__restrict__ tpos, u8 *__restrict__ ctx,
                                                          u64 *__restrict__ tape, u64 tape_cap, tape_result_dev *__restrict__ res) {
  const u64 j = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  const u32 m = u32(*m_ptr);
  if (j >= m) { return; }
  const u32 i = tok[j], c = tokc[i + 2];
  if (is_open_char(c)) { return; }
  const u32 cid = u32(opens_before[j]); // an element that is not an open: opens in front of it = opens at positions <= j
  if (cid == 0) { return; }
  const u32 jo = openpos[cid - 1];
  if (key[jo] != key[j]) { return; } // no container of this level in front: the token's own rule reports it
  const u32 io = tok[jo], co = tokc[io + 2];
  if (c == ',') {
    ctx[i] = u8(co == '{' ? CTX_OBJECT : CTX_ARRAY);
    return;
  }
  if ((c == '}') != (co == '{')) { report_error(res, error_key(i, 0, SJ_TAPE_ERROR)); }
  const u64 open_at = 1 + u64(u32(tpos[io])), close_at = 1 + u64(u32(tpos[i]));
  const u64 between = j - jo; // commas + 1
  const u64 count = (i == io + 1) ? 0 : (between > 0xFFFFFFull ? 0xFFFFFFull : between);
  if (close_at < tape_cap) {
    tape[close_at] = tape_word(c, open_at);
    tape[open_at] = tape_word(co, (count << 32) | (close_at + 1));
  } else {
    res->overflow = 1;
  }
}

// ---- per token: rule, limit, content ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TP_THREADS) void k_tape_write(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, u32 n, u32 max_depth, const u8 *__restrict__ tokc,
                                                          const int *__restrict__ tpos, const int *__restrict__ depth, const u8 *__restrict__ ctx, const u32 *__restrict__ str_offsets,
                                                          u64 *__restrict__ tape, u64 tape_cap, u32 *__restrict__ slow_list, u32 slow_cap, tape_result_dev *__restrict__ res) {
  const u64 i = u64(blockIdx.x) * TP_THREADS + threadIdx.x;
  if (i > n) { return; }
  if (i == n) { // behind the last token: the root words and the checks that belong to no token
    const u64 words = u64(u32(tpos[n])) + 2;
    res->tape_words = words;
    if (words <= tape_cap) {
      tape[0] = tape_word('r', words);           // visit_document_end, tape_builder.h:160-165
      tape[words - 1] = tape_word('r', 0);
    } else {
      res->overflow = 1;
    }
    const u32 c0 = tokc[2], last = tokc[n + 1];
    if ((c0 == '{' && last != '}') || (c0 == '[' && last != ']')) { report_error(res, error_key(0, 0, SJ_TAPE_ERROR)); } // json_iterator.h:138-143
    if (depth[n] != 0) { report_error(res, error_key(n, 0, SJ_TAPE_ERROR)); } // the walk meets the sentinel inside a container
    return;
  }
  const u32 c = tokc[i + 2], prev = tokc[i + 1], prev2 = tokc[i], next = tokc[i + 3];
  const u32 ctx_prev = i >= 1 ? ctx[i - 1] : 0u, ctx_prev2 = i >= 2 ? ctx[i - 2] : 0u;
  u32 rank = 0;
  const u32 g = token_grammar_error(i, c, prev, prev2, next, ctx_prev, ctx_prev2, (long long)depth[i], max_depth, &rank);
  if (g) { report_error(res, error_key(i, rank, g)); }
  const u64 at = 1 + u64(u32(tpos[i]));
  const bool root = i == 0;
  const windowed_bytes src{buf, u32(len)};
  if (c == '"') {
    if (at < tape_cap) { tape[at] = tape_word('"', str_offsets[i]); } // on_start_string, tape_builder.h:415-419
  } else if (c == ',') {
    if (comma_in_value_position(i, prev, ctx_prev)) { report_error(res, error_key(i, 2, SJ_NUMBER_ERROR)); }
  } else if (is_open_char(c) || is_close_char(c) || c == ':') {
    // bracket words come from k_tape_match
  } else if (takes_number_path(c, root)) {
    const number_value v = parse_number_token(src, idx[i], static_cast<bigint *>(nullptr));
    if (v.error) { report_error(res, error_key(i, 2, v.error)); }
    else if (at + 1 < tape_cap) {
      tape[at] = tape_word(v.type, 0);
      tape[at + 1] = v.bits; // sign only when v.slow: k_tape_slow_numbers completes it
      if (v.slow) {
        const u32 k = atomicAdd(&res->slow_numbers, 1u);
        if (k < slow_cap) { slow_list[k] = u32(i); }
      }
    }
  } else if (c == 't' || c == 'f' || c == 'n') {
    const bool ok = c == 't' ? atom_matches(src, idx[i], 't', 'r', 'u', 'e', 0)
                             : (c == 'f' ? atom_matches(src, idx[i], 'f', 'a', 'l', 's', 'e') : atom_matches(src, idx[i], 'n', 'u', 'l', 'l', 0));
    if (!ok) { report_error(res, error_key(i, 2, c == 't' ? SJ_T_ATOM_ERROR : (c == 'f' ? SJ_F_ATOM_ERROR : SJ_N_ATOM_ERROR))); }
    if (at < tape_cap) { tape[at] = tape_word(c, 0); }
  }
}

// Number tokens with more than 19 significant digits whose two bracketing conversions disagree: the exact decision needs two
// big integers of 516 bytes each per thread (private memory) -- kept out of k_tape_write, which then needs no scratch at all.
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

static inline u32 blocks_of(u64 n, u32 per) { return u32((n + per - 1) / per); }

// Workspace layout for n structurals (every array 256-byte aligned): see tape_workspace below.
struct tape_workspace {
  tape_result_dev *res;
  u32 *n_words;       // [0] = n + 1 (scan lengths), [1] = n (upper bound of the sorted elements + 1 for the opens scan), [2] = hist length per pass
  u8 *tokc, *ctx;
  int *slots, *depth, *sel; // in place: tape positions, depths, sort slots
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
  w.n_words = reinterpret_cast<u32 *>(take(64));
  w.tokc = take(n1 + 8);
  w.ctx = take(n1 + 8);
  w.slots = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.depth = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.sel = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.key_a = reinterpret_cast<unsigned short *>(take(n1 * 2 + 64));
  w.key_b = reinterpret_cast<unsigned short *>(take(n1 * 2 + 64));
  w.tok_a = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.tok_b = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.openpos = reinterpret_cast<u32 *>(take(n1 * 4 + 64));
  w.opens = reinterpret_cast<int *>(take(n1 * 4 + 64));
  w.slow_list = reinterpret_cast<u32 *>(take(size_t(w.slow_cap) * 4 + 64));
  w.hist = reinterpret_cast<int *>(take(size_t(w.tiles) * RADIX_BINS * 4 + 64));
  const size_t longest = n1 > size_t(w.tiles) * RADIX_BINS ? n1 : size_t(w.tiles) * RADIX_BINS;
  w.partial = reinterpret_cast<int *>(take((longest / 4096 + 80) * 4));
  w.bytes = at;
  return w;
}
size_t tape_workspace_bytes(uint32_t n, uint64_t len) { return carve(nullptr, n, len).bytes; }

// idx[0 .. n] (n >= 1; idx[n] = len, stage 1's first sentinel); str_offsets: what launch_parse_strings left (n + 1 words);
// workspace: tape_workspace_bytes(n, len).  Leaves tape_result_dev at the start of the workspace.
void launch_tape(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, const uint32_t *str_offsets, uint64_t *tape, uint64_t tape_cap,
                 void *workspace, hipStream_t s) {
  const tape_workspace w = carve(static_cast<uint8_t *>(workspace), n, len);
  const u32 n1 = n + 1;
  (void)hipMemsetAsync(w.res, 0, sizeof(tape_result_dev), s);
  (void)hipMemsetAsync(&w.res->error_key, 0xFF, sizeof(u64), s); // NO_ERROR_KEY
  (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.n_words), int(n1), 1, s);
  (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.n_words + 1), int(w.tiles * RADIX_BINS), 1, s);
  (void)hipMemsetAsync(w.ctx, 0, size_t(n1) + 8, s);
  const u32 grid = blocks_of(n1, TP_THREADS);
  hipLaunchKernelGGL(k_tape_classify, dim3(grid), dim3(TP_THREADS), 0, s, buf, len, idx, n, w.tokc, w.slots, w.depth, w.sel);
  enqueue_scan(w.slots, n1, w.n_words, w.partial, s); // slots -> tape positions (the root word in front not counted)
  enqueue_scan(w.depth, n1, w.n_words, w.partial, s); // deltas -> depth in front of every token
  enqueue_scan(w.sel, n1, w.n_words, w.partial, s);   // flags -> slot in the sort's input; sel[n] = m
  const int *m_ptr = w.sel + n;
  const u32 kmax = max_depth < 4095u ? max_depth : 4095u;
  hipLaunchKernelGGL(k_tape_select, dim3(grid), dim3(TP_THREADS), 0, s, w.tokc, w.depth, w.sel, n, kmax, w.key_a, w.tok_a);
  // two passes of six bits cover levels up to 4095
  hipLaunchKernelGGL(k_radix_hist, dim3(w.tiles), dim3(64), 0, s, w.key_a, m_ptr, 0u, w.tiles, w.hist);
  enqueue_scan(w.hist, w.tiles * RADIX_BINS, w.n_words + 1, w.partial, s);
  hipLaunchKernelGGL(k_radix_scatter, dim3(w.tiles), dim3(64), 0, s, w.key_a, w.tok_a, m_ptr, 0u, w.tiles, w.hist, w.key_b, w.tok_b);
  hipLaunchKernelGGL(k_radix_hist, dim3(w.tiles), dim3(64), 0, s, w.key_b, m_ptr, RADIX_BITS, w.tiles, w.hist);
  enqueue_scan(w.hist, w.tiles * RADIX_BINS, w.n_words + 1, w.partial, s);
  hipLaunchKernelGGL(k_radix_scatter, dim3(w.tiles), dim3(64), 0, s, w.key_b, w.tok_b, m_ptr, RADIX_BITS, w.tiles, w.hist, w.key_a, w.tok_a);
  // containers
  hipLaunchKernelGGL(k_tape_opens, dim3(grid), dim3(TP_THREADS), 0, s, w.tokc, w.tok_a, m_ptr, n, w.opens);
  enqueue_scan(w.opens, n1, w.n_words, w.partial, s);
  hipLaunchKernelGGL(k_tape_openpos, dim3(grid), dim3(TP_THREADS), 0, s, w.tokc, w.tok_a, m_ptr, w.opens, w.openpos);
  hipLaunchKernelGGL(k_tape_match, dim3(grid), dim3(TP_THREADS), 0, s, w.tokc, w.key_a, w.tok_a, m_ptr, w.opens, w.openpos, w.slots, w.ctx, tape, tape_cap, w.res);
  hipLaunchKernelGGL(k_tape_write, dim3(grid), dim3(TP_THREADS), 0, s, buf, len, idx, n, max_depth, w.tokc, w.slots, w.depth, w.ctx, str_offsets, tape, tape_cap, w.slow_list,
                     w.slow_cap, w.res);
  hipLaunchKernelGGL(k_tape_slow_numbers, dim3(64), dim3(64), 0, s, buf, len, idx, w.slots, w.slow_list, w.slow_cap, tape, tape_cap, w.res);
}

} // namespace sjgpu
