// simdjson_amd/csrc/sjgpu_fused.hip -- the SINGLE-PASS pipeline: stage 1 (and minify) in one kernel that reads
// every input byte from HBM once and writes every output byte once (algorithmic traffic).
//
// The two true device-wide dependencies of the path -- in-string parity and output cursor -- are resolved
// with a chained scan ("decoupled look-back"): a tile publishes its AGGREGATE (quote parity + the output
// count for BOTH in-string hypotheses) as soon as it has scanned its bytes, then walks back over its
// predecessors' descriptors until it meets one that already knows its INCLUSIVE prefix, composes, and
// publishes its own inclusive prefix.  Everything else the reference carries from block to block
// (escape parity, previous-scalar, UTF-8 look-back) is position-wise and is re-derived from the 64 bytes in
// front of each wave's span; where those do not settle it (64 backslashes, a quote behind 63) the span assumes,
// and one more bit -- x, "the assumption of the next tile's first span is wrong" -- travels with the in-string bit
// through the same descriptors (sj_xcarry.h; sjgpu_device.h: span_carry_assume).
//
// MI355X specifics:
//   * tile = 128 KiB = one 512-thread workgroup = 8 waves x 4 chunks (k_fused_pipelined, the large-input kernel since round 4; rounds 1-3:
//     64 KiB, 4 waves -- SJGPU_PIPE_WAVES=4 brings that shape back; k_fused, the small-input kernel: 16 KiB, 4 waves x 1 chunk;
//     k_minify_onchip: 64 KiB, 8 waves x 2 chunks).  Tiles must be big: descriptors live in other XCDs' L2s, a hand-off costs 1-3 us under load
//     (MI355X_MICROARCH.md, handoff rows), a ticket and a look-back cost ~17 ns per tile device-wide whatever the tile does
//     (profiles/r04_gather_lab.txt), and at ~5 TB/s a new 128 KiB tile becomes ready every ~26 ns, so the look-back window (256
//     descriptors per round trip, 4 coalesced 512-byte loads per wave) must cover "tiles per round trip" (~75).
//   * descriptors are single naturally-aligned 8-byte granules written with ONE relaxed agent-scope
//     atomic store (sc1 write-through) and read with relaxed agent-scope loads: the data is the flag, no
//     fences (cdna_hip_programming.md G16, form R2).  Dispatch order is not assumed: tile ids come from an
//     atomic ticket, so every predecessor of a running tile has already started.
//   * every spin is bounded (wall-clock timeout); a timed-out tile poisons its descriptor, raises
//     SJGPU_F_INTERNAL and the host re-runs the call on the split pipeline.
//   * the per-chunk masks wait for the look-back in a 4-deep register FIFO (rolled loops, no dynamic
//     register indexing, no LDS), offsets leave through the per-wave LDS window as 16-byte stores.
//   * a call is ONE dispatch (round 5): nothing is cleared in front of the kernel, its last workgroup to leave puts descriptors, ticket and
//     flag word back to zero for the next call (leave_and_clean).
#include "sjgpu_device.h"

#include <cstdlib>

namespace sjgpu {
namespace {

constexpr u32 FUSED_WAVES = 4; // waves per workgroup of k_fused (and of the four-wave A/B shapes of the pipelined kernels, whose default is NW = 8: launch_fused);
                               // each wave owns WC consecutive chunks of the tile
constexpr u32 LOOKBACK_LOADS = 4;                                                // x64 descriptors per round trip
constexpr u64 LOOKBACK_TIMEOUT_TICKS = 100ull * 1000 * 1000;                     // wall_clock64 is 100 MHz: 1 s

// ---- descriptor encoding (one u64 per tile) -------------------------------------------------------------
//   [63:62] status   0 invalid | 1 aggregate | 2 inclusive | 3 poison
//   aggregate: [49:43] the tile's x word (sj_xcarry.h, XW_LOW_BITS), [42] quote parity, [41:21] count if the tile starts inside a
//              string, [20:0] count if outside
//   inclusive: [33] x in front of the next tile, [32] in-string after the tile, [31:0] output cursor after the tile
constexpr u64 ST_AGG = 1, ST_INCL = 2, ST_POISON = 3;
constexpr u32 DESC_XW_SHIFT = 43;
__device__ __forceinline__ u64 make_agg(u32 q, u32 c_out, u32 c_in, u32 xw) {
  return (ST_AGG << 62) | (u64(xw & XW_LOW_MASK) << DESC_XW_SHIFT) | (u64(q & 1u) << 42) | (u64(c_in) << 21) | u64(c_out);
}
__device__ __forceinline__ u64 make_incl(u32 s, u32 x, u32 base) { return (ST_INCL << 62) | (u64(x & 1u) << 33) | (u64(s & 1u) << 32) | u64(base); }

__device__ __forceinline__ u64 desc_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void desc_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the workspace cleans itself (round 5) ----------------------------------------------------------------
// Rounds 1-4 cleared result, descriptors and ticket with a hipMemsetAsync in front of every call: a fill kernel and a dispatch gap, 10-18 us of
// a 0.49 ms call and most of a 16 us one.  Now the LAST workgroup to leave a kernel puts everything back to zero for the next call, and a call
// is ONE dispatch.  Behind the ntiles descriptors lie the control words (fused_ctl_words): [ticket][done][flags][pad].  A workgroup that has
// finished its last tile -- it will read no descriptor, draw no ticket and raise no flag any more -- adds itself to `done`; the one that
// finds everybody else gone moves the accumulated flags into the result and zeroes descriptors and control words.  Flags are ORed into the
// control word, not into the result: the result needs no clearing either (its other two fields are written by the last TILE's owner).
// The host clears once, when the workspace is allocated, and in front of the first call after anything that makes the state doubtful -- a HIP error on the
// context, a chain that gave up, a traced run (sjgpu_ctx::ws_dirty, sjgpu_ctx.h).
constexpr u32 FUSED_CTL_WORDS = FUSED_WORKSPACE_EXTRA_WORDS; // u64 words behind the descriptors: [ticket, done][flags, -] ... and, a 128-byte line further, the second ticket counter
__device__ __forceinline__ u32 *ctl_done(u32 *ticket) { return ticket + 1; }
__device__ __forceinline__ u32 *ctl_flags(u32 *ticket) { return ticket + 2; }
// Tickets from TWO counters (round 6, A/B: env SJGPU_TWO_TICKETS): workgroups with an even blockIdx draw the even tiles, the others the odd ones.  Atomics on ONE
// address are served at ~30 ns each on this part (profiles/r06_direct_ab.txt), k_minify_onchip draws 16 384 tickets per GiB in ~450 us: two counters in two lines
// halve what each has to serve.  Progress: a tile still waits only for smaller tiles; those of the other stream are drawn by workgroups that wait for nobody
// larger, so the smallest unfinished tile of either stream always moves.
__device__ __forceinline__ u32 draw_ticket(u32 *ticket, bool two) {
  if (!two) { return atomicAdd(ticket, 1u); }
  const u32 odd = blockIdx.x & 1u;
  return 2u * atomicAdd(ticket + 32u * odd, 1u) + odd;
}
// all threads of the workgroup, behind its last tile
template <u32 THREADS>
__device__ __forceinline__ void leave_and_clean(u64 *desc, u32 ntiles, u32 *ticket, scan_result_dev *result) {
  __shared__ u32 sh_last;
  // What this wave ORed into the flag word and stored into descriptors has been PERFORMED before the workgroup counts itself out: all of it went out as
  // agent-scope atomics (performed at the device's coherence point, past the XCD's L2), so waiting for their acknowledgements is enough.  NOT
  // __threadfence(): an agent-scope fence is buffer_wbl2 + buffer_inv on this part -- a write-back and an invalidation of the XCD's whole L2 per WAVE that
  // leaves: the first version of this function cost the headline kernel 0.22 ms of 0.70 (profiles/r05_selfclean_ab.txt).
  __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = atomicAdd(ctl_done(ticket), 1u) == gridDim.x - 1u;
    if (last) { // the flags are read BEFORE the barrier that lets the other waves of this workgroup start zeroing (the flag word among the rest)
      result->flags = __hip_atomic_load(ctl_flags(ticket), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh_last = last ? 1u : 0u;
  }
  lds_writes_done();
  __syncthreads();
  if (sh_last == 0u) { return; }
  for (u32 i = threadIdx.x; i < ntiles + FUSED_CTL_WORDS; i += THREADS) { desc_store(desc + i, 0ull); }
}

// Wave-wide look-back for tile `tile` (all 64 lanes of ONE wave call this).  On success S = in-string at
// the tile start, X = x in front of the tile, B = output cursor at the tile start.
// State of a walk: the summary of tiles [end, tile) in its compact form (sj_xcarry.h: parity, the two counts, x word), which maps
// the state at `end` to the state in front of `tile`.  Windows whose aggregates all have x word zero -- every window of ordinary
// input -- are folded as they always were (ballot, two sums); the first window that holds an x word sends the walk to
// lookback_general, which folds whole windows with wave_fold (documents with backslash runs of 64 bytes and more across spans).
struct lookback_state {
  xs_sum acc;
  int end; // tile numbers fit 32 bits (4 GiB / 32 KiB tiles)
};
__device__ __forceinline__ u64 lookback_front(scan_origin org) { // in front of tile 0: the call's carry-in and cursor
  return make_incl(org.carry & CARRY_IN_STRING, (org.carry & CARRY_X) ? 1u : 0u, org.base0);
}
__device__ __forceinline__ void lookback_load(const u64 *desc, int end, u32 lane, u64 (&d)[LOOKBACK_LOADS], scan_origin org) {
#pragma unroll
  for (u32 w = 0; w < LOOKBACK_LOADS; w++) {
    const int t = end - 1 - int(w * 64 + lane);
    d[w] = (t >= 0) ? desc_load(desc + t) : lookback_front(org);
  }
}
// the walk has met the inclusive prefix of tile k: apply what lies between
__device__ __forceinline__ void lookback_finish(const lookback_state &st, u32 lo, u32 hi, u32 &S, u32 &X, u32 &B) {
  const xs_step t = xs_apply(st.acc.q, st.acc.xw, hi & 1u, (hi >> 1) & 1u);
  S = t.s_out;
  X = t.x_out;
  B = lo + xs_sum_count(st.acc, t);
}
// 1 = finished (S, X, B valid), 0 = a needed aggregate is not published yet (reload from st.end), -1 = poisoned chain,
// 2 = the window at st.end holds an x word: lookback_general
__device__ __forceinline__ int lookback_consume(const u64 (&d)[LOOKBACK_LOADS], lookback_state &st, u32 lane, u32 &S, u32 &X, u32 &B) {
#pragma unroll
  for (u32 w = 0; w < LOOKBACK_LOADS; w++) {
    const u32 status = u32(d[w] >> 62);
    const u64 incl = __ballot(status == ST_INCL), valid = __ballot(status != 0), poison = __ballot(status == ST_POISON);
    const u32 k = incl ? ctz64(incl) : 64u;                       // nearest inclusive predecessor in this window
    const u64 below = (k == 64u) ? ~0ull : ((1ull << k) - 1ull);  // lanes nearer than it: must be aggregates
    const u64 upto = (k == 64u) ? ~0ull : (below | (1ull << k));
    if (poison & upto) { return -1; }
    if (~valid & below) { return 0; }
    // G = composition of the aggregates on lanes k-1 ... 0 (ascending tile order)
    const bool mine = lane < k;
    if (__ballot(mine && (u32(d[w] >> DESC_XW_SHIFT) & XW_LOW_MASK) != 0u)) { return 2; }
    const u32 q = mine ? u32(d[w] >> 42) & 1u : 0u;
    const u32 c_out = mine ? u32(d[w]) & 0x1FFFFFu : 0u;
    const u32 c_in = mine ? u32(d[w] >> 21) & 0x1FFFFFu : 0u;
    const u64 qm = __ballot(q != 0);
    // parity already flipped, relative to the window start, when the walk reaches my tile: quotes of the
    // farther lanes lane+1 .. k-1
    const u32 flipped = u32(popc64(qm & ~((2ull << lane) - 1ull))) & 1u;
    const u32 g_out = wave_sum(flipped ? c_in : c_out);
    const u32 g_in = wave_sum(flipped ? c_out : c_in);
    const u32 gq = u32(popc64(qm)) & 1u;
    if (k) { // acc := G then acc.  G answers "no" for its successor, whatever x it met: acc is met with x = 0
      const u32 nf_out = g_out + (gq ? st.acc.c_in : st.acc.c_out), nf_in = g_in + (gq ? st.acc.c_out : st.acc.c_in);
      st.acc.c_out = nf_out;
      st.acc.c_in = nf_in;
      st.acc.q ^= gq;
      st.acc.xw &= XW_C;
    }
    if (k < 64u) {
      lookback_finish(st, readlane_dyn(u32(d[w]), k), readlane_dyn(u32(d[w] >> 32), k), S, X, B);
      return 1;
    }
    st.end -= 64;
  }
  return 0; // whole batch consumed, no inclusive prefix yet: keep walking from st.end
}
// The same walk for windows with x words, one window of 64 per round trip, every window folded by the whole wave (wave_fold) and
// composed in front of what the walk holds (xs_compose).  Only documents that need it come here.
__device__ __forceinline__ bool lookback_general(const u64 *desc, lookback_state &st, u32 lane, u32 &S, u32 &X, u32 &B, scan_origin org, u64 t_start) {
  for (;;) {
    const int t = st.end - 1 - int(lane);
    const u64 d = (t >= 0) ? desc_load(desc + t) : lookback_front(org);
    const u32 status = u32(d >> 62);
    const u64 incl = __ballot(status == ST_INCL), valid = __ballot(status != 0), poison = __ballot(status == ST_POISON);
    const u32 k = incl ? ctz64(incl) : 64u;
    const u64 below = (k == 64u) ? ~0ull : ((1ull << k) - 1ull);
    const u64 upto = (k == 64u) ? ~0ull : (below | (1ull << k));
    if (poison & upto) { return false; }
    if (~valid & below) {
      if (wall_clock64() - t_start > LOOKBACK_TIMEOUT_TICKS) { return false; }
      __builtin_amdgcn_s_sleep(4);
      continue;
    }
    const bool mine = lane < k;
    xs_sum v{0u, 0u, 0u, XW_IDENTITY};
    if (mine) { v = xs_sum{u32(d >> 42) & 1u, u32(d) & 0x1FFFFFu, u32(d >> 21) & 0x1FFFFFu, u32(d >> DESC_XW_SHIFT) & XW_LOW_MASK}; }
    const xs_sum g = wave_fold<true>(v, lane); // lane 0 holds the last tile of the window
    st.acc = xs_compose(g, st.acc);
    if (k < 64u) {
      lookback_finish(st, readlane_dyn(u32(d), k), readlane_dyn(u32(d >> 32), k), S, X, B);
      return true;
    }
    st.end -= 64;
  }
}
__device__ __forceinline__ bool lookback(const u64 *desc, u32 tile, u32 lane, u32 &S, u32 &X, u32 &B, scan_origin org) {
  lookback_state st{xs_sum{0u, 0u, 0u, XW_IDENTITY}, int(tile)};
  const u64 t_start = wall_clock64();
  for (;;) {
    u64 d[LOOKBACK_LOADS];
    lookback_load(desc, st.end, lane, d, org);
    const int r = lookback_consume(d, st, lane, S, X, B);
    if (r == 2) { return lookback_general(desc, st, lane, S, X, B, org, t_start); }
    if (r != 0) { return r > 0; }
    if (wall_clock64() - t_start > LOOKBACK_TIMEOUT_TICKS) { return false; }
    __builtin_amdgcn_s_sleep(4);
  }
}

// the aggregate of a tile from its waves' summaries ([parity, count if out, count if in, flags, x word] each, in LDS)
struct tile_agg {
  u32 q, c_out, c_in, xw;
};
template <u32 NW>
__device__ __forceinline__ tile_agg tile_aggregate(const u32 (*wv)[5]) {
  tile_agg r{0u, 0u, 0u, 0u};
  u32 any = 0;
#pragma unroll
  for (u32 v = 0; v < NW; v++) { any |= wv[v][4]; }
  any = u32(__builtin_amdgcn_readfirstlane(int(any)));
  if (any == 0u) { // ordinary input
#pragma unroll
    for (u32 v = 0; v < NW; v++) {
      const u32 q = wv[v][0], o = wv[v][1], i = wv[v][2];
      const u32 no = r.c_out + (r.q ? i : o), ni = r.c_in + (r.q ? o : i);
      r.c_out = no;
      r.c_in = ni;
      r.q ^= q;
    }
    return r;
  }
  // some span of the tile assumed, or tells its successor something: compose the waves' summaries in their compact form (a rolled
  // loop: this is the rare road, it must not cost the kernel registers)
  xs_sum acc{0u, 0u, 0u, XW_IDENTITY};
#pragma unroll 1
  for (u32 v = 0; v < NW; v++) { acc = xs_compose(acc, xs_sum{wv[v][0], wv[v][1], wv[v][2], wv[v][4] & XW_LOW_MASK}); }
  r.q = acc.q;
  r.c_out = acc.c_out;
  r.c_in = acc.c_in;
  r.xw = acc.xw;
  return r;
}
// the state in front of wave `wave`'s span from the state in front of the tile
__device__ __forceinline__ void wave_state(const u32 (*wv)[5], u32 wave, u32 &s, u32 &x, u32 &base) {
  for (u32 v = 0; v < wave; v++) {
    const xs_step t = xs_apply(wv[v][0], wv[v][4], s, x);
    base += xs_count(wv[v][1], wv[v][2], t);
    s = t.s_out;
    x = t.x_out;
  }
}

// TOKENS (round 6, sjgpu_stage1_tokens_device on the single-pass road): tok[i] = buf[idx[i]] for the offsets a wave has just written, idx[from .. to).  The
// wave reads its own stores back once they have been acknowledged (no fence: an agent-scope fence writes back and invalidates the XCD's L2 -- leave_and_clean)
// and gathers the bytes out of the document -- the road k_stage1_emit<true> takes for segments that staged nothing.  The tile was read an iteration ago; its
// lines come back from wherever they still are (the Infinity Cache, mostly): what that costs is measured, not assumed (profiles/r06_tokens_fused.txt).
__device__ __forceinline__ void gather_tokens(const u8 *__restrict__ buf, u64 len, const u32 *idx, u8 *__restrict__ tok, u32 from, u32 to, u32 lane) {
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const volatile u32 *back = idx;
#pragma unroll 1
  for (u32 i = from + lane; i < to; i += 64) {
    const u32 p = back[i];
    tok[i] = u64(p) < len ? buf[p] : u8(0x20);
  }
}

// what a wave tells its workgroup about its 16 KiB span (UTF-8 verdicts go straight to the result: utf8_queue)
constexpr u32 WF_CTRL_IF_OUT = 1u, WF_CTRL_IF_IN = 2u;

// OP 0: stage 1 (out = u32 structural offsets); OP 1: minify (out = bytes)
// TRACE: wave 0 / lane 0 of the first `trace_tiles` tiles records wall_clock64() at the phase boundaries
// (8 stamps per tile) so the per-phase latency budget can be read off a real run (sjgpu_debug_trace_stage1).
constexpr u32 TRACE_STAMPS = 8;
// WC = chunks per wave: 4 (64 KiB tiles) for throughput, 1 (16 KiB tiles) for small inputs, where the serial
// latency of one tile (WC loads + scans, then WC emits) is the whole call.
template <int OP, bool TRACE, u32 WC, bool TOKENS = false>
__global__ __launch_bounds__(256) void k_fused(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ desc,
                                               u32 *__restrict__ ticket, u32 ntiles, void *__restrict__ out, u64 out_words,
                                               scan_result_dev *__restrict__ result, u64 *__restrict__ trace, u32 trace_tiles,
                                               scan_origin org, u8 *__restrict__ tok = nullptr) {
  const u32 carry = org.carry;
#define SJ_STAMP(k) do { if (TRACE && threadIdx.x == 0 && tile < trace_tiles) { trace[u64(tile) * TRACE_STAMPS + (k)] = wall_clock64(); } } while (0)
  constexpr u32 STAGE_WORDS = (OP == 0) ? EMIT_STAGE_WORDS : (MINIFY_STAGE_BYTES / 4);
  static_assert(WC == 1 || WC == 4, "the mask FIFO is written for 1 or 4 chunks per wave");
  constexpr u32 FUSED_WAVE_CHUNKS = WC;
  constexpr u32 FUSED_WAVE_BYTES = WC * CHUNK_BYTES;
  constexpr u32 FUSED_TILE_BYTES = FUSED_WAVES * FUSED_WAVE_BYTES;
  __shared__ u32 sh_tile;
  __shared__ u32 sh_wave[FUSED_WAVES][5]; // parity, count_if_out, count_if_in, flags, x word (sj_xcarry.h)
  __shared__ u32 sh_prefix[4];            // S, B, ok, X
  __shared__ __attribute__((aligned(16))) u32 sh_stage[FUSED_WAVES][STAGE_WORDS];
  __shared__ u32 sh_lut[MINIFY_LUT_WORDS];
  __shared__ u32 sh_uq[(OP == 0) ? FUSED_WAVES : 1][(OP == 0) ? 64 : 1]; // left-over UTF-8 list entries between tiles

  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (OP == 1) {
    if (wave == 0) { init_compaction_lut(sh_lut, lane); }
    clear_minify_stage(reinterpret_cast<u8 *>(sh_stage[wave]), lane);
  }
  const bool more = (carry & CARRY_MORE) != 0;
  // small inputs (16 KiB tiles) check dense non-ASCII chunks in line; the 64 KiB-tile variant has no registers to spare for that
  utf8_queue uq{sh_uq[(OP == 0) ? wave : 0], 0u, 0u, 0u, (WC == 1 && !(carry & CARRY_DEBUG_QUEUE_UTF8)) ? buf : nullptr, len, (carry & CARRY_MORE) ? 1u : 0u}; // lives across tiles: blocks are validated 64 at a time

  for (;;) {
    // Take the ticket only when we are ready to start the tile: a ticket claimed early would make every
    // successor's look-back wait for a tile nobody is scanning yet.
    const u64 t_loop = TRACE ? wall_clock64() : 0;
    if (threadIdx.x == 0) { sh_tile = atomicAdd(ticket, 1u); }
    __syncthreads();
    const u32 tile = sh_tile;
    if (tile >= ntiles) { break; }
    if (TRACE && threadIdx.x == 0 && tile < trace_tiles) { trace[u64(tile) * TRACE_STAMPS + 0] = t_loop; }
    SJ_STAMP(1); // ticket known

    // ---- phase 1: scan my 4 chunks with a relative in-string state; masks go into the register FIFO ----
    const u64 wave_start = org.begin + u64(tile) * FUSED_TILE_BYTES + u64(wave) * FUSED_WAVE_BYTES;
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0; // slot 3 = oldest chunk
    u32 n_out = 0, n_in = 0;
    bool f_ci = false, f_co = false; // wave-uniform error facts
    u32 parity = 0, xw = 0;
    if (wave_start < len) { // wave-uniform
      const u32 lookback = lookback_issue(buf, wave_start, lane); // consumed after chunk 0 has been requested
      wave_carry wc{0u, 0u, 0u};
      span_x sx;
      if (OP == 0) { utf8_resume(uq, sh_stage[wave], sh_uq[wave], lane); } // the output window is idle while we scan
#pragma unroll 1
      for (u32 c = 0; c < FUSED_WAVE_CHUNKS; c++) {
        const u64 cstart = wave_start + u64(c) * CHUNK_BYTES;
        u64 a = 0, b = 0;
        if (cstart < len) {
          const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
          u32 w[16];
          if (cstart + CHUNK_BYTES <= len) { load_block_full(buf, pos, w); }
          else { load_block(buf, pos, len, w); }
          if (c == 0) {
            wc = span_carry_assume(wave_start, lane, lookback, sx);
            uq.pending = utf8_pending_from(lookback, lane);
          }
          span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
          if (OP == 0) {
            const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, u32(cstart / BLOCK_BYTES));
            if (c == FUSED_WAVE_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
            a = m.cand;
            b = m.string_tail;
            n_out += u32(popc64(a & ~b));
            n_in += u32(popc64(a & b));
            f_ci |= __ballot((m.ctrl & m.in_string) != 0) != 0;
            f_co |= __ballot((m.ctrl & ~m.in_string) != 0) != 0;
          } else {
            const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
            if (c == FUSED_WAVE_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
            const u64 valid = valid_mask(pos, len);
            a = valid & m.ws; // drop candidates
            b = m.in_string;
            n_out += u32(popc64(valid & ~(a & ~b)));
            n_in += u32(popc64(valid & ~(a & b)));
          }
        }
        a3 = a2; a2 = a1; a1 = a0; a0 = a;
        b3 = b2; b2 = b1; b1 = b0; b0 = b;
      }
      parity = wc.s;
      xw = span_finish(sx, buf, wave_start, FUSED_WAVE_BYTES, len, wc, OP == 0);
      if (OP == 0) { utf8_settle(uq, sh_uq[wave], buf, len, more, lane); }
    }
    SJ_STAMP(2); // wave 0 finished scanning
    {
      const u32 t_out = wave_sum(n_out), t_in = wave_sum(n_in);
      u32 f = 0;
      if (f_ci) { f |= WF_CTRL_IF_OUT; }
      if (f_co) { f |= WF_CTRL_IF_IN; }
      if (lane == 0) {
        sh_wave[wave][0] = parity;
        sh_wave[wave][1] = t_out;
        sh_wave[wave][2] = t_in;
        sh_wave[wave][3] = f;
        sh_wave[wave][4] = xw;
      }
    }
    __syncthreads();
    SJ_STAMP(3); // all waves finished scanning

    // ---- phase 2 (wave 0): publish the tile aggregate, look back, publish the inclusive prefix ----------
    if (wave == 0) {
      const tile_agg ta = tile_aggregate<FUSED_WAVES>(sh_wave);
      if (lane == 0) { desc_store(desc + tile, make_agg(ta.q, ta.c_out, ta.c_in, ta.xw)); }
      u32 S = 0, X = 0, B = 0;
      const bool ok = lookback(desc, tile, lane, S, X, B, org);
      SJ_STAMP(4); // look-back done
      if (lane == 0) {
        if (ok) {
          const xs_step te = xs_apply(ta.q, ta.xw, S, X);
          const u32 total = B + xs_count(ta.c_out, ta.c_in, te), s_end = te.s_out;
          desc_store(desc + tile, make_incl(s_end, te.x_out, total));
          if (tile == ntiles - 1) { // the last tile knows the totals: n / out_len, unclosed string, sentinels
            u32 f = s_end ? SJGPU_F_UNCLOSED_STRING : 0u;
            if (OP == 0) {
              u32 *idx = static_cast<u32 *>(out);
              if (u64(total) + 3 <= out_words) { // json_structural_indexer.h:284-286
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
              result->out_len = (s_end && !(carry & CARRY_SHARD)) ? 0ull : u64(total); // json_minifier.h:42-47
            }
            if ((carry & CARRY_MORE) && te.x_out) { f |= SJGPU_F_RANGE_CARRY; }
            if (f) { atomicOr(ctl_flags(ticket), f); }
          }
        } else {
          desc_store(desc + tile, ST_POISON << 62);
          atomicOr(ctl_flags(ticket), SJGPU_F_INTERNAL);
        }
        sh_prefix[0] = S;
        sh_prefix[1] = B;
        sh_prefix[2] = ok ? 1u : 0u;
        sh_prefix[3] = X;
      }
    }
    __syncthreads();
    SJ_STAMP(5);

    // ---- phase 3: every wave derives its own carry-in from the tile prefix and emits ------------------------
    if (sh_prefix[2] == 0u) { continue; } // poisoned chain: nothing to emit (workgroup-uniform)
    u32 s = sh_prefix[0], x = sh_prefix[3], base = sh_prefix[1];
    wave_state(sh_wave, wave, s, x, base);
    if (wave_start >= len) { continue; }
    const xs_step own = xs_apply(parity, xw, s, x); // which hypothesis holds for my span, and the one bit a wrong assumption toggles
    {
      const u32 f = sh_wave[wave][3];
      u32 g = 0;
      if (f & (own.se ? WF_CTRL_IF_IN : WF_CTRL_IF_OUT)) { g |= SJGPU_F_UNESCAPED_CTRL; }
      if (g && lane == 0) { atomicOr(ctl_flags(ticket), g); }
    }
    const u64 flip = own.se ? ~0ull : 0ull;
    const bool patch = OP == 0 && x != 0u && own.dcount != 0; // wave-uniform
    const u32 patch_at = xw_patch_pos(xw);
    bool overflow = false;
    const u32 base_before = base;
#pragma unroll 1
    for (u32 c = 0; c < FUSED_WAVE_CHUNKS; c++) {
      // oldest chunk first: after WC pushes chunk c sits in slot WC-1-c
      const u64 a = (WC == 1) ? a0 : a3, b = (WC == 1) ? b0 : b3;
      if (WC != 1) {
        a3 = a2; a2 = a1; a1 = a0;
        b3 = b2; b2 = b1; b1 = b0;
      }
      const u64 cstart = wave_start + u64(c) * CHUNK_BYTES;
      if (cstart >= len) { break; }
      const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
      if (OP == 0) {
        u64 st = a & ~(b ^ flip);
        if (patch && c == patch_at / CHUNK_BYTES && lane == ((patch_at / BLOCK_BYTES) & 63u)) { st ^= 1ull << (patch_at & 63u); }
        emit_indices(st, u32(pos), lane, static_cast<u32 *>(out), out_words, base, sh_stage[wave], overflow);
      } else {
        u32 w[16];
        load_block(buf, pos, len, w); // second touch of the same 4 KiB: served by L2 / Infinity Cache
        emit_bytes(w, valid_mask(pos, len) & ~(a & ~(b ^ flip)), lane, static_cast<u8 *>(out), base,
                   reinterpret_cast<u8 *>(sh_stage[wave]), sh_lut);
      }
    }
    if (OP == 0 && __ballot(overflow) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_IDX_OVERFLOW); }
    if (OP == 0 && TOKENS && !__ballot(overflow)) { gather_tokens(buf, len, static_cast<const u32 *>(out), tok, base_before, base, lane); }
    SJ_STAMP(6); // wave 0 finished emitting
  }
  if (OP == 0) { // what is left of the wave's UTF-8 list, then its verdict
    utf8_drain_rest(uq, buf, len, more, lane);
    if (uq.error && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_UTF8_ERROR); }
  }
  leave_and_clean<256>(desc, ntiles, ticket, result);
#undef SJ_STAMP
}

// =====================================================================================================
// Pipelined variant (large inputs): a workgroup defers the look-back + emission of tile T(i-1) until it has
// scanned and published tile T(i).  The in-kernel trace of the plain kernel showed 29 % of a tile's residency
// spent waiting for the slowest of its ~150 predecessors to publish an aggregate; with one whole tile scan
// (~15 us) between "publish my aggregate" and "ask for my prefix" that wait is over before it starts.
// The pending tile's masks wait in LDS (16 KiB per workgroup), the tile being scanned uses the register FIFO.
// Progress: a workgroup never waits before it has published the aggregate of every tile it holds, so every
// wait is on an aggregate that some running workgroup will publish without waiting on anyone.
// =====================================================================================================
constexpr u32 PIPE_WINDOW = 1280; // 4 x 5 KiB of emission windows + 16 KiB of pending masks -> 4 workgroups per CU
// (measured in round 5 and not kept: the chunks of this kernel requested coalesced and streamed like the split kernels' (load_chunk_stream, the exchange buffer
// in the idle emission window): configs[1] 474.9 -> 472.6 us, NDJSON 335.7 -> 356.0, deep nesting 1101 -> 1131 -- this kernel is not waiting for its loads;
// nor the wave's NEXT chunk requested into that buffer with global_load_lds_dwordx4 (no registers) while it scans this one: 460.9 -> 458.7 us, NDJSON
// 335.5 -> 351.1 -- session V; the window grown to 1336 words for the buffer alone cost deep nesting 5 %: its emission rounds are sized by it)
constexpr u32 NO_TILE = 0xFFFFFFFFu;

// (Round 2 carried a variant that requested chunk c+1 while chunk c was classified: two register sets, 128 VGPRs with 20 B of
// scratch in the hot loop.  Same-box A/B in round 3, profiles/r03_overlap_experiment.txt: 0.5504 vs 0.5501 ms on configs[1],
// 0.4241 vs 0.4267 ms on NDJSON -- nothing, so the kernel without it, which spills nothing, is the only one left.)
// TRACE: thread 0 of every workgroup stamps wall_clock64() at 8 phase boundaries of its first PIPE_TRACE_ITERS iterations
// (sjgpu_debug_trace_pipelined): 0 loop top, 1 ticket known, 2 wave 0 scanned, 3 all scanned, 4 wave 0 published +
// looked back, 5 prefix broadcast, 6 wave 0 emitted, 7 masks parked.
constexpr u32 PIPE_TRACE_ITERS = 32;
// wave priority by phase (s_setprio 0..3): which of the four workgroups of a CU gets the issue slots when a scanning wave (VALU) and an
// emitting or look-back wave (LDS, stores, descriptor loads) compete.  Policy = bits 24-26 of scan_origin::carry (env SJGPU_PRIO), 0 = leave alone.
__device__ __forceinline__ void set_prio(u32 p) {
  switch (p) {
  case 1: __builtin_amdgcn_s_setprio(1); break;
  case 2: __builtin_amdgcn_s_setprio(2); break;
  case 3: __builtin_amdgcn_s_setprio(3); break;
  default: __builtin_amdgcn_s_setprio(0); break;
  }
}
__device__ __forceinline__ void phase_prio(u32 policy, u32 phase /* 0 scan, 1 look-back, 2 emit */) {
  if (policy == 0) { return; }
  // policy: 1 = (0,3,3)  2 = (3,0,0)  3 = (0,3,0)  4 = (0,0,3)  5 = (1,3,2)
  const u32 table[6][3] = {{0, 0, 0}, {0, 3, 3}, {3, 0, 0}, {0, 3, 0}, {0, 0, 3}, {1, 3, 2}};
  set_prio(table[policy < 6 ? policy : 0][phase]);
}
// PWC: chunks per wave and tile: 4 (64 KiB tiles) or 2 (32 KiB tiles: half the fixed cost of filling and draining the pipeline -- one iteration
// each --, twice the descriptors, look-backs and barriers per byte; A/B: env SJGPU_PIPE_WC)
// NW: waves per workgroup: 4, or 8 (128 KiB tiles at the same wave count per CU: half the tickets, look-backs and descriptors per byte, two workgroups per
// CU instead of four to hide them behind; A/B: env SJGPU_PIPE_WAVES)
template <int OP, bool TRACE = false, u32 PWC = FUSED_WAVE_CHUNKS, u32 NW = FUSED_WAVES, bool TOKENS = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fused_pipelined(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ desc,
                                                         u32 *__restrict__ ticket, u32 ntiles, void *__restrict__ out,
                                                         u64 out_words, scan_result_dev *__restrict__ result, scan_origin org,
                                                         u64 *__restrict__ trace = nullptr, u8 *__restrict__ tok = nullptr) {
#define SJ_PSTAMP(k) do { if (TRACE && threadIdx.x == 0 && iter < PIPE_TRACE_ITERS) { trace[(u64(blockIdx.x) * PIPE_TRACE_ITERS + iter) * TRACE_STAMPS + (k)] = wall_clock64(); } } while (0)
  const u32 carry = org.carry;
  constexpr u32 WC = PWC;
  static_assert(WC == 2 || WC == 4, "the mask FIFO is written for 2 or 4 chunks per wave");
  constexpr u32 TILE_BYTES = NW * WC * CHUNK_BYTES;
  constexpr u32 STAGE_WORDS = (OP == 0) ? emit_stage_words(PIPE_WINDOW) : (MINIFY_STAGE_BYTES / 4);
  constexpr u32 WAVE_BYTES = WC * CHUNK_BYTES;
  __shared__ u32 sh_tile[2];                 // [iteration parity]: the ticket of an iteration is drawn one iteration ahead
  __shared__ u32 sh_wave[3][NW][5]; // [iteration mod 3][wave]: parity, count_if_out, count_if_in, flags, x word (sj_xcarry.h); three, so that a wave that is an
                                    // iteration ahead (there is no barrier at the loop top since round 6) does not write the rows a slower one still reads
  __shared__ u32 sh_agg[2][4];               // tile aggregate of the same two tiles (tq, tout, tin, x word)
  __shared__ u32 sh_prefix[4];               // S, B, ok, X of the tile being emitted
  __shared__ u64 sh_mask_a[NW][WC][64], sh_mask_b[NW][WC][64]; // the pending tile's masks
  __shared__ __attribute__((aligned(16))) u32 sh_stage[NW][STAGE_WORDS];
  __shared__ u32 sh_lut[MINIFY_LUT_WORDS];
  __shared__ u32 sh_uq[(OP == 0) ? NW : 1][(OP == 0) ? 64 : 1]; // left-over UTF-8 list entries between tiles

  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (OP == 1) {
    if (wave == 0) { init_compaction_lut(sh_lut, lane); }
    clear_minify_stage(reinterpret_cast<u8 *>(sh_stage[wave]), lane);
  }
  const bool more = (carry & CARRY_MORE) != 0;
  // always the queue here: the in-line check of dense non-ASCII chunks (utf8_dense_chunk) costs this kernel 100 B of scratch in
  // its hot loop and 10-15 % on EVERY workload (profiles/r02_utf8_dense_ab.txt); text-heavy documents have sparse output, for
  // which AUTO picks the split pipeline, whose summarize kernel has the in-line path
  utf8_queue uq{sh_uq[(OP == 0) ? wave : 0], 0u, 0u, 0u, nullptr, len, (carry & CARRY_MORE) ? 1u : 0u}; // lives across tiles: blocks are validated 64 at a time
  const u32 prio_policy = (carry >> 24) & 7u;
  u32 pend_tile = NO_TILE; // workgroup-uniform
  // Tickets are drawn one iteration AHEAD (the atomic's round trip, 1.4-2 us of an iteration of 21-27, hides behind
  // the scan): thread 0 keeps the next ticket in a register and hands it over through LDS at the end of the iteration.
  // Progress: a workgroup still publishes the aggregate of every tile it holds before it waits for anyone, and all its
  // waits are on smaller tile numbers; the owner of the smallest unpublished tile is therefore never blocked.
  u32 next_ticket = 0;
  const bool early = (org.carry & CARRY_DEBUG_LATE_TICKET) == 0;
  // Round 6: TWO barriers per iteration, not three.  The one at the loop top handed over the next ticket and kept a fast wave's next summary row from
  // landing in the array a slow wave was still reading; the ticket now crosses with the barrier behind wave 0's publication (it is known since the top of
  // the iteration), the summary rows are triple-buffered, and everything else in LDS is either owned by one wave or written between the two remaining
  // barriers and read behind the second.  A wave that has emitted starts loading its next span at once -- which pays for minify and costs stage 1 on dense
  // output 2 % (its waves' emission phases fall out of step): the launcher keeps the third barrier for stage 1 (CARRY_DEBUG_TOP_BARRIER, launch_fused).
  const bool top_barrier = (org.carry & CARRY_DEBUG_TOP_BARRIER) != 0;
  const bool two_tickets = (org.carry & CARRY_DEBUG_TWO_TICKETS) != 0 && gridDim.x >= 2u; // (both parities need a workgroup)
  if (threadIdx.x == 0 && early) { sh_tile[0] = draw_ticket(ticket, two_tickets); }
  lds_writes_done();
  __syncthreads();
  for (u32 iter = 0;; iter++) {
    const u32 cur = iter & 1u, cw = iter % 3u, pw = (iter + 2u) % 3u; // ticket / aggregate slot; summary rows of this iteration's tile and of the pending one
    SJ_PSTAMP(0);
    if (!early) {
      if (threadIdx.x == 0) { sh_tile[cur] = draw_ticket(ticket, two_tickets); }
      lds_writes_done();
      __syncthreads();
    } else if (top_barrier) {
      __syncthreads();
    }
    const u32 tile = sh_tile[cur];
    const bool have = tile < ntiles;
    const bool pend = pend_tile != NO_TILE;
    if (!have && !pend) { break; }
    if (threadIdx.x == 0 && have && early) { next_ticket = draw_ticket(ticket, two_tickets); } // consumed at the end of this iteration
    SJ_PSTAMP(1);

    // ---- scan the new tile into the register FIFO ------------------------------------------------------------
    phase_prio(prio_policy, 0);
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0; // slot 3 = oldest chunk
    if (have) {
      const u64 wave_start = org.begin + u64(tile) * TILE_BYTES + u64(wave) * WAVE_BYTES;
      u32 n_out = 0, n_in = 0, parity = 0, xw = 0;
      bool f_ci = false, f_co = false;
      if (wave_start < len) {
        const u32 lookback = lookback_issue(buf, wave_start, lane);
        wave_carry wc{0u, 0u, 0u};
        span_x sx;
        if (OP == 0) { utf8_resume(uq, sh_stage[wave], sh_uq[wave], lane); } // the output window is idle while we scan
        // one chunk through the scanner; its masks enter the FIFO
        auto scan_one = [&](const u32 (&w)[16], u64 cstart, bool last) {
          const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
          u64 a, b;
          if (OP == 0) {
            const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, u32(cstart / BLOCK_BYTES));
            if (last) { span_note_tail(sx, m.backslash, m.quote_raw); }
            a = m.cand;
            b = m.string_tail;
            n_out += u32(popc64(a & ~b));
            n_in += u32(popc64(a & b));
            f_ci |= __ballot((m.ctrl & m.in_string) != 0) != 0;
            f_co |= __ballot((m.ctrl & ~m.in_string) != 0) != 0;
          } else {
            const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
            if (last) { span_note_tail(sx, m.backslash, m.quote_raw); }
            const u64 valid = valid_mask(pos, len);
            a = valid & m.ws;
            b = m.in_string;
            n_out += u32(popc64(valid & ~(a & ~b)));
            n_in += u32(popc64(valid & ~(a & b)));
          }
          a3 = a2; a2 = a1; a1 = a0; a0 = a;
          b3 = b2; b2 = b1; b1 = b0; b0 = b;
        };
        {
#pragma unroll 1
          for (u32 c = 0; c < WC; c++) {
            const u64 cstart = wave_start + u64(c) * CHUNK_BYTES;
            if (cstart < len) {
              const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
              u32 w[16];
              if (cstart + CHUNK_BYTES <= len) { load_block_full(buf, pos, w); }
              else { load_block(buf, pos, len, w); }
              if (c == 0) {
                wc = span_carry_assume(wave_start, lane, lookback, sx);
                uq.pending = utf8_pending_from(lookback, lane);
              }
              span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
              scan_one(w, cstart, c == WC - 1);
            } else {
              a3 = a2; a2 = a1; a1 = a0; a0 = 0;
              b3 = b2; b2 = b1; b1 = b0; b0 = 0;
            }
          }
        }
        parity = wc.s;
        xw = span_finish(sx, buf, wave_start, WAVE_BYTES, len, wc, OP == 0);
        if (OP == 0) { utf8_settle(uq, sh_uq[wave], buf, len, more, lane); }
      }
      SJ_PSTAMP(2);
      const u32 t_out = wave_sum(n_out), t_in = wave_sum(n_in);
      u32 f = 0;
      if (f_ci) { f |= WF_CTRL_IF_OUT; }
      if (f_co) { f |= WF_CTRL_IF_IN; }
      if (lane == 0) {
        sh_wave[cw][wave][0] = parity;
        sh_wave[cw][wave][1] = t_out;
        sh_wave[cw][wave][2] = t_in;
        sh_wave[cw][wave][3] = f;
        sh_wave[cw][wave][4] = xw;
      }
    }
    // Round 6: the pending tile's look-back needs nothing of the tile just scanned -- wave 0 walks the descriptors BEFORE the barrier, while the other
    // seven waves finish their spans (the phase trace: they arrive 0.9-1.6 us behind wave 0, the walk takes 1.9-2.9), and behind the barrier only
    // publishes: the aggregate of the new tile, the inclusive prefix of the pending one.  Progress holds: the owner of the SMALLEST tile whose
    // aggregate is unpublished waits, in this walk, only for tiles smaller than its pending one -- all published -- so it reaches the barrier and publishes.
    u32 lb_S = 0, lb_X = 0, lb_B = 0;
    bool lb_ok = false;
    const bool early_lb = (carry & CARRY_DEBUG_LATE_LOOKBACK) == 0;
    if (wave == 0 && pend && early_lb) {
      phase_prio(prio_policy, 1);
      lb_ok = lookback(desc, pend_tile, lane, lb_S, lb_X, lb_B, org);
    }
    __syncthreads();
    SJ_PSTAMP(3);

    // ---- wave 0: publish the new tile's aggregate and the PENDING tile's inclusive prefix -------------------------
    if (wave == 0) {
      phase_prio(prio_policy, 1);
      if (have) {
        const tile_agg ta = tile_aggregate<NW>(sh_wave[cw]);
        if (lane == 0) {
          desc_store(desc + tile, make_agg(ta.q, ta.c_out, ta.c_in, ta.xw));
          sh_agg[cur][0] = ta.q;
          sh_agg[cur][1] = ta.c_out;
          sh_agg[cur][2] = ta.c_in;
          sh_agg[cur][3] = ta.xw;
        }
      }
      if (pend) {
        const u32 tq = sh_agg[cur ^ 1u][0], tout = sh_agg[cur ^ 1u][1], tin = sh_agg[cur ^ 1u][2], txw = sh_agg[cur ^ 1u][3];
        u32 S = lb_S, X = lb_X, B = lb_B;
        const bool ok = early_lb ? lb_ok : lookback(desc, pend_tile, lane, S, X, B, org);
        if (lane == 0) {
          if (ok) {
            const xs_step te = xs_apply(tq, txw, S, X);
            const u32 total = B + xs_count(tout, tin, te), s_end = te.s_out;
            desc_store(desc + pend_tile, make_incl(s_end, te.x_out, total));
            if (pend_tile == ntiles - 1) { // the last tile knows the totals
              u32 f = s_end ? SJGPU_F_UNCLOSED_STRING : 0u;
              if (OP == 0) {
                u32 *idx = static_cast<u32 *>(out);
                if (u64(total) + 3 <= out_words) { // json_structural_indexer.h:284-286
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
                result->out_len = (s_end && !(carry & CARRY_SHARD)) ? 0ull : u64(total);
              }
              if ((carry & CARRY_MORE) && te.x_out) { f |= SJGPU_F_RANGE_CARRY; }
              if (f) { atomicOr(ctl_flags(ticket), f); }
            }
          } else {
            desc_store(desc + pend_tile, ST_POISON << 62);
            atomicOr(ctl_flags(ticket), SJGPU_F_INTERNAL);
          }
          sh_prefix[0] = S;
          sh_prefix[1] = B;
          sh_prefix[2] = ok ? 1u : 0u;
          sh_prefix[3] = X;
        }
      }
      if (lane == 0 && early) { sh_tile[cur ^ 1u] = have ? next_ticket : NO_TILE; } // the next iteration's ticket crosses with this barrier
      SJ_PSTAMP(4);
    }
    lds_writes_done();
    __syncthreads();
    SJ_PSTAMP(5);

    // ---- every wave: emit its share of the pending tile from the LDS masks -----------------------------------------
    phase_prio(prio_policy, 2);
    if (pend && sh_prefix[2] != 0u) {
      const u64 wave_start = org.begin + u64(pend_tile) * TILE_BYTES + u64(wave) * WAVE_BYTES;
      u32 s = sh_prefix[0], x = sh_prefix[3], base = sh_prefix[1];
      wave_state(sh_wave[pw], wave, s, x, base);
      if (wave_start < len) {
        const u32 pxw = sh_wave[pw][wave][4];
        const xs_step own = xs_apply(sh_wave[pw][wave][0], pxw, s, x); // which hypothesis holds for my span, and the bit a wrong assumption toggles
        const u32 f = sh_wave[pw][wave][3];
        u32 g = 0;
        if (f & (own.se ? WF_CTRL_IF_IN : WF_CTRL_IF_OUT)) { g |= SJGPU_F_UNESCAPED_CTRL; }
        if (g && lane == 0) { atomicOr(ctl_flags(ticket), g); }
        const u64 flip = own.se ? ~0ull : 0ull;
        bool overflow = false;
        const u32 base_before = base;
        if (OP == 0) { // sparse spans leave in one piece, medium ones as two pairs of chunks, dense ones chunk by chunk
          u64 st[WC];
#pragma unroll
          for (u32 c = 0; c < WC; c++) { st[c] = sh_mask_a[wave][c][lane] & ~(sh_mask_b[wave][c][lane] ^ flip); } // zero beyond len
          span_patch(st, pxw, x, own.se, lane);
          const u32 span_count = (carry & CARRY_DEBUG_NO_SPAN_HINT) ? 0u : xs_count(sh_wave[pw][wave][1], sh_wave[pw][wave][2], own);
          if constexpr (WC == 4) {
            emit_span4_adaptive<PIPE_WINDOW>(st, u32(wave_start), lane, static_cast<u32 *>(out), out_words, base, sh_stage[wave], overflow, span_count);
          } else { // two chunks: in one piece when they fit the window, else chunk by chunk
            if (!(span_count <= PIPE_WINDOW && emit_span<PIPE_WINDOW, 2>(st, u32(wave_start), lane, static_cast<u32 *>(out), out_words, base, sh_stage[wave], overflow))) {
              emit_indices<PIPE_WINDOW>(st[0], u32(wave_start) + lane * BLOCK_BYTES, lane, static_cast<u32 *>(out), out_words, base, sh_stage[wave], overflow);
              emit_indices<PIPE_WINDOW>(st[1], u32(wave_start) + CHUNK_BYTES + lane * BLOCK_BYTES, lane, static_cast<u32 *>(out), out_words, base, sh_stage[wave], overflow);
            }
          }
        } else {
#pragma unroll 1
          for (u32 c = 0; c < WC; c++) {
            const u64 cstart = wave_start + u64(c) * CHUNK_BYTES;
            if (cstart >= len) { break; }
            const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
            const u64 a = sh_mask_a[wave][c][lane], b = sh_mask_b[wave][c][lane];
            u32 w[16];
            load_block(buf, pos, len, w);
            emit_bytes(w, valid_mask(pos, len) & ~(a & ~(b ^ flip)), lane, static_cast<u8 *>(out), base,
                       reinterpret_cast<u8 *>(sh_stage[wave]), sh_lut);
          }
        }
        if (OP == 0 && __ballot(overflow) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_IDX_OVERFLOW); }
        if (OP == 0 && TOKENS && !__ballot(overflow)) { gather_tokens(buf, len, static_cast<const u32 *>(out), tok, base_before, base, lane); }
      }
    }
    SJ_PSTAMP(6);
    // ---- the tile scanned in this iteration becomes the pending one: park its masks in (this wave's rows of) LDS ----
    if (have) { // chunk c sits in FIFO slot WC - 1 - c
      if constexpr (WC == 4) {
        sh_mask_a[wave][0][lane] = a3; sh_mask_b[wave][0][lane] = b3;
        sh_mask_a[wave][1][lane] = a2; sh_mask_b[wave][1][lane] = b2;
        sh_mask_a[wave][2][lane] = a1; sh_mask_b[wave][2][lane] = b1;
        sh_mask_a[wave][3][lane] = a0; sh_mask_b[wave][3][lane] = b0;
      } else {
        sh_mask_a[wave][0][lane] = a1; sh_mask_b[wave][0][lane] = b1;
        sh_mask_a[wave][1][lane] = a0; sh_mask_b[wave][1][lane] = b0;
      }
    }
    pend_tile = have ? tile : NO_TILE;
    lds_writes_done();
    SJ_PSTAMP(7);
  }
  if (OP == 0) { // what is left of the wave's UTF-8 list, then its verdict
    utf8_drain_rest(uq, buf, len, more, lane);
    if (uq.error && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_UTF8_ERROR); }
  }
  leave_and_clean<NW * 64>(desc, ntiles, ticket, result);
#undef SJ_PSTAMP
}


// =====================================================================================================
// k_minify_onchip: the pipelined minifier with the pending tile's BYTES kept on the chip.
// k_fused_pipelined<1> defers the compaction of a tile by one iteration and then loads the tile a second time; with
// ~130 MiB of tiles between the two reads (1024 workgroups x 2 x 64 KiB) the second read misses the 4 MiB L2s, so the
// kernel moves 2 x len + dst_len through the fabric (profiles/traffic.json: FETCH_SIZE = 2.0 x len) and sits at the
// rate a copy of THAT many bytes reaches.  Here a wave scans TWO chunks per tile, keeps their 2 x 16 input dwords in
// registers until the pending tile has left, and parks them in LDS ([16-byte piece][lane]: conflict-free b128 accesses)
// where the next iteration's compaction finds them.  The 4 KiB + 32 B of a parked chunk double as that chunk's
// compaction window (emit_bytes wants a zeroed window: the chunk is read into registers, its rows are zeroed, the
// kept bytes are OR-merged back in and leave from there), so a workgroup of NW waves needs NW x 2 x 4128 B of LDS:
// 33 KiB for 4 waves (4 workgroups per CU, 32 KiB tiles) or 66 KiB for 8 waves (2 per CU, 64 KiB tiles).
// The masks of the pending tile (2 x 2 x u64 per lane) stay in registers.  Protocol = k_fused_pipelined's.
// =====================================================================================================
constexpr u32 ONCHIP_REGION_BYTES = MINIFY_STAGE_BYTES; // 4 KiB of input + the 32 bytes of skew/slack the window needs
static_assert(ONCHIP_REGION_BYTES % 16 == 0, "regions are accessed with b128 operations");

template <u32 NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_minify_onchip(
    const u8 *__restrict__ buf, u64 len, u64 *__restrict__ desc, u32 *__restrict__ ticket, u32 ntiles, u8 *__restrict__ out,
    scan_result_dev *__restrict__ result, scan_origin org) {
  constexpr u32 WC = ONCHIP_WAVE_CHUNKS;
  static_assert(WC == 2, "two register sets, two parked chunks");
  constexpr u32 WAVE_BYTES = WC * CHUNK_BYTES;
  constexpr u32 TILE_BYTES = NW * WAVE_BYTES;
  const u32 carry = org.carry;
  __shared__ u32 sh_tile[2];
  __shared__ u32 sh_wave[3][NW][5]; // [iteration mod 3][wave]: quote parity, kept if out, kept if in, (unused), x word (sj_xcarry.h); three: no barrier at the loop top (k_fused_pipelined)
  __shared__ u32 sh_agg[2][4];
  __shared__ u32 sh_prefix[4];
  __shared__ __attribute__((aligned(16))) u8 sh_bytes[NW][WC][ONCHIP_REGION_BYTES];
  __shared__ u32 sh_lut[MINIFY_LUT_WORDS];

  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (wave == 0) { init_compaction_lut(sh_lut, lane); }
  const u32 prio_policy = (carry >> 24) & 7u;
  u32 pend_tile = NO_TILE;
  u64 pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0; // the pending tile's masks: droppable-if-outside-a-string, in-string (relative)
  u32 next_ticket = 0;
  const bool top_barrier = (org.carry & CARRY_DEBUG_TOP_BARRIER) != 0; // (round 6: two barriers per iteration -- see k_fused_pipelined)
  const bool two_tickets = (org.carry & CARRY_DEBUG_TWO_TICKETS) != 0 && gridDim.x >= 2u; // (both parities need a workgroup)
  if (threadIdx.x == 0) { sh_tile[0] = draw_ticket(ticket, two_tickets); }
  lds_writes_done();
  __syncthreads();
  for (u32 iter = 0;; iter++) {
    const u32 cur = iter & 1u, cw = iter % 3u, pw = (iter + 2u) % 3u;
    if (top_barrier) { __syncthreads(); }
    const u32 tile = sh_tile[cur];
    const bool have = tile < ntiles;
    const bool pend = pend_tile != NO_TILE;
    if (!have && !pend) { break; }
    if (threadIdx.x == 0 && have) { next_ticket = draw_ticket(ticket, two_tickets); }

    // ---- scan the new tile; its bytes stay in wa / wb until the pending tile has left the LDS ------------------------
    phase_prio(prio_policy, 0);
    u32 wa[16], wb[16];
    u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    bool park = false;
    if (have) {
      const u64 wave_start = org.begin + u64(tile) * TILE_BYTES + u64(wave) * WAVE_BYTES;
      u32 n_out = 0, n_in = 0, parity = 0, xw = 0;
      if (wave_start < len) {
        park = true;
        const u32 lookback = lookback_issue(buf, wave_start, lane);
        const u64 pos0 = wave_start + u64(lane) * BLOCK_BYTES, pos1 = pos0 + CHUNK_BYTES;
        if (wave_start + WAVE_BYTES <= len) { // wave-uniform: the whole span is input
          load_block_full(buf, pos0, wa);
          load_block_full(buf, pos1, wb);
        } else {
          load_block(buf, pos0, len, wa); // zero beyond the end
          load_block(buf, pos1, len, wb);
        }
        span_x sx;
        wave_carry wc = span_carry_assume(wave_start, lane, lookback, sx);
        span_note_chunk(sx, wa, 0u, lane);
        {
          const chunk_masks m = scan_chunk<false, false>(wa, wc, lane);
          const u64 valid = valid_mask(pos0, len);
          a0 = valid & m.ws;
          b0 = m.in_string;
          n_out += u32(popc64(valid & ~(a0 & ~b0))); // dropped: whitespace outside strings (json_scanner.h:46)
          n_in += u32(popc64(valid & ~(a0 & b0)));
        }
        span_note_chunk(sx, wb, CHUNK_BYTES, lane);
        {
          const chunk_masks m = scan_chunk<false, false>(wb, wc, lane);
          span_note_tail(sx, m.backslash, m.quote_raw);
          const u64 valid = valid_mask(pos1, len);
          a1 = valid & m.ws;
          b1 = m.in_string;
          n_out += u32(popc64(valid & ~(a1 & ~b1)));
          n_in += u32(popc64(valid & ~(a1 & b1)));
        }
        parity = wc.s;
        xw = span_finish(sx, buf, wave_start, WAVE_BYTES, len, wc, false);
      }
      const u32 t_out = wave_sum(n_out), t_in = wave_sum(n_in);
      if (lane == 0) {
        sh_wave[cw][wave][0] = parity;
        sh_wave[cw][wave][1] = t_out;
        sh_wave[cw][wave][2] = t_in;
        sh_wave[cw][wave][3] = 0;
        sh_wave[cw][wave][4] = xw;
      }
    }
    // (round 6: the pending tile's look-back in front of the barrier, while the other waves finish scanning: k_fused_pipelined)
    u32 lb_S = 0, lb_X = 0, lb_B = 0;
    bool lb_ok = false;
    const bool early_lb = (carry & CARRY_DEBUG_LATE_LOOKBACK) == 0;
    if (wave == 0 && pend && early_lb) { lb_ok = lookback(desc, pend_tile, lane, lb_S, lb_X, lb_B, org); }
    __syncthreads();

    // ---- wave 0: publish the new tile's aggregate and the PENDING tile's inclusive prefix -------------------------
    if (wave == 0) {
      if (have) {
        const tile_agg ta = tile_aggregate<NW>(sh_wave[cw]);
        if (lane == 0) {
          desc_store(desc + tile, make_agg(ta.q, ta.c_out, ta.c_in, ta.xw));
          sh_agg[cur][0] = ta.q;
          sh_agg[cur][1] = ta.c_out;
          sh_agg[cur][2] = ta.c_in;
          sh_agg[cur][3] = ta.xw;
        }
      }
      if (pend) {
        const u32 tq = sh_agg[cur ^ 1u][0], tout = sh_agg[cur ^ 1u][1], tin = sh_agg[cur ^ 1u][2], txw = sh_agg[cur ^ 1u][3];
        u32 S = lb_S, X = lb_X, B = lb_B;
        const bool ok = early_lb ? lb_ok : lookback(desc, pend_tile, lane, S, X, B, org);
        if (lane == 0) {
          if (ok) {
            const xs_step te = xs_apply(tq, txw, S, X);
            const u32 total = B + xs_count(tout, tin, te), s_end = te.s_out;
            desc_store(desc + pend_tile, make_incl(s_end, te.x_out, total));
            if (pend_tile == ntiles - 1) { // the last tile knows the totals
              if (s_end) { atomicOr(ctl_flags(ticket), SJGPU_F_UNCLOSED_STRING); }
              if ((carry & CARRY_MORE) && te.x_out) { atomicOr(ctl_flags(ticket), SJGPU_F_RANGE_CARRY); }
              result->n = 0;
              result->out_len = (s_end && !(carry & CARRY_SHARD)) ? 0ull : u64(total); // json_minifier.h:42-47
            }
          } else {
            desc_store(desc + pend_tile, ST_POISON << 62);
            atomicOr(ctl_flags(ticket), SJGPU_F_INTERNAL);
          }
          sh_prefix[0] = S;
          sh_prefix[1] = B;
          sh_prefix[2] = ok ? 1u : 0u;
          sh_prefix[3] = X;
        }
      }
      if (lane == 0) { sh_tile[cur ^ 1u] = have ? next_ticket : NO_TILE; } // the next iteration's ticket crosses with this barrier
    }
    lds_writes_done();
    __syncthreads();

    // ---- every wave: compact its two parked chunks of the pending tile ----------------------------------------------
    phase_prio(prio_policy, 2);
    if (pend && sh_prefix[2] != 0u) {
      const u64 wave_start = org.begin + u64(pend_tile) * TILE_BYTES + u64(wave) * WAVE_BYTES;
      u32 s = sh_prefix[0], x = sh_prefix[3], base = sh_prefix[1];
      wave_state(sh_wave[pw], wave, s, x, base);
      const u64 flip = xs_apply(sh_wave[pw][wave][0], sh_wave[pw][wave][4], s, x).se ? ~0ull : 0ull; // the effective hypothesis of my span
#pragma unroll
      for (u32 c = 0; c < WC; c++) {
        const u64 cstart = wave_start + u64(c) * CHUNK_BYTES;
        if (cstart < len) { // wave-uniform
          const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
          const u64 a = c ? pa1 : pa0, b = c ? pb1 : pb0;
          uint4 *const region = reinterpret_cast<uint4 *>(sh_bytes[wave][c]);
          u32 w[16];
#pragma unroll
          for (u32 j = 0; j < 4; j++) {
            const uint4 q = region[j * 64 + lane];
            w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
          }
          wave_lds_fence();
#pragma unroll
          for (u32 j = 0; j < 4; j++) { region[j * 64 + lane] = make_uint4(0, 0, 0, 0); }
          if (lane < (ONCHIP_REGION_BYTES - CHUNK_BYTES) / 16) { region[256 + lane] = make_uint4(0, 0, 0, 0); }
          wave_lds_fence();
          emit_bytes<false>(w, valid_mask(pos, len) & ~(a & ~(b ^ flip)), lane, out, base, sh_bytes[wave][c], sh_lut);
        }
      }
    }
    // ---- the tile scanned in this iteration becomes the pending one: park its bytes, keep its masks ----------------
    if (park) {
      uint4 *const r0 = reinterpret_cast<uint4 *>(sh_bytes[wave][0]), *const r1 = reinterpret_cast<uint4 *>(sh_bytes[wave][1]);
#pragma unroll
      for (u32 j = 0; j < 4; j++) {
        r0[j * 64 + lane] = make_uint4(wa[4 * j], wa[4 * j + 1], wa[4 * j + 2], wa[4 * j + 3]);
        r1[j * 64 + lane] = make_uint4(wb[4 * j], wb[4 * j + 1], wb[4 * j + 2], wb[4 * j + 3]);
      }
    }
    pa0 = a0; pa1 = a1; pb0 = b0; pb1 = b1;
    pend_tile = have ? tile : NO_TILE;
    lds_writes_done();
  }
  leave_and_clean<NW * 64>(desc, ntiles, ticket, result);
}

// =====================================================================================================
// k_stage1_direct (round 6): the split pipeline's scan with the emission INSIDE it -- one pass, no masks in HBM -- for PLAIN input.
//
// k_stage1_summarize + k_stage1_emit move the masks out and in again (0.135 + 0.151 GB per GiB of NDJSON, 18 % of everything the pipeline moves; both kernels
// run at ~0.9 of the copy ceiling on the bytes they move: only fewer bytes help -- VERDICT r05).  A segment whose first chunk holds a control character pins
// its own string state there (k_stage1_summarize: "resolved"): one final mask plane, ONE count whatever the state in front of it.  When EVERY segment of an
// input is like that and no span had to assume its escape carry (x word 0) -- pretty-printed text, NDJSON, large_random: "plain" input -- the two quantities
// that cross segments are ADDITIVE: the output cursor is a sum of counts, the string state a sum of quote parities mod 2.  Sums need no chain of
// compositions: every tile adds (1 arrival, parity, count) to the word of its group of 64 tiles with one atomicAdd, and a tile's prefix is two kinds of
// coalesced loads -- the tiles in front inside its group, the groups in front -- complete when the arrival fields say so.  No look-back walk, no
// aggregate / inclusive protocol.
//   * persistent workgroups of EIGHT waves (a wave = one 16 KiB segment, k_stage1_summarize's body: streamed coalesced chunk loads, UTF-8 rows parked in
//     LDS), 128 KiB tiles by atomic ticket drawn one iteration ahead.  (Four waves and 64 KiB tiles -- 16 384 tickets and 2 x 16 384 adds per GiB -- sat at
//     67 ns per TILE whatever the tile did, 1.1 ms per GiB with ideal traffic: atomics on one address are served at ~30 ns each on this part.)
//   * a tile's emission is DEFERRED by one iteration (the first version of this kernel emitted at once: every workgroup waited for the slowest of its
//     ~1280 in-flight predecessors and the kernel lost 20-45 % to both pipelines, profiles/r06_direct_ab.txt): the pending segment's final masks wait in
//     eight VGPRs per lane while the wave scans its next segment; by then the tiles in front have long published;
//   * the offsets leave through the LDS the chunk loads do not need between two scans (load_chunk_stream's exchange buffer = the emission window);
//   * a segment that is NOT plain (no control character in its first chunk: minified text, brackets only; or an x word) cannot be summed: the kernel
//     raises SJGPU_F_INTERNAL and the call is re-run on the split pipeline like every single-pass call that gives up; AUTO only comes here for contexts
//     whose previous scan found the input plain (sjgpu_capi.hip).
// Workspace (the single-pass kernels' descriptor array): [ntiles tile words][group words][control words: ticket, done, flags].
//   tile word:   [63] published, [32] quote parity of the tile, [31:0] its count        (one relaxed agent-scope store)
//   group word:  [63:48] arrivals, [47:32] sum of parities, [31:0] sum of counts        (agent-scope atomicAdd, one per tile)
// MEASURED (profiles/r06_direct_ab.txt), and why AUTO does not select it: correct and with IDEAL traffic (1.081 GB fetched + 0.271 written for 1.336 algorithmic
// per GiB of NDJSON: the masks' round trip is gone) it takes 447 us per GiB of NDJSON against 287 for the split pipeline and 335 for k_fused_pipelined, 588
// against 472 on large_random.  Thread 0's wall-clock per phase: a wave scans 56 % of the time (prefix 5 us and publication 2 us per 27 us iteration, emission
// 4-17), and a scan that is bound by what it has IN FLIGHT -- one 4 KiB chunk per scanning wave -- runs at the fraction of its waves that are scanning:
// k_stage1_summarize keeps 20 waves per CU loading all the time, this kernel ~9 of 16.  The split pipeline wins on this part BECAUSE its kernels have one phase.
// =====================================================================================================
constexpr u32 DIRECT_WAVES = 8;  // (four waves and 64 KiB tiles, 16 384 tickets and 2 x 16 384 adds per GiB: the atomics on the ticket and on the sum words are served at
                                 // ~30 ns each per ADDRESS and the kernel sat at 67 ns per tile whatever it did -- 1.1 ms per GiB with ideal traffic, session J)
constexpr u32 DIRECT_TILE_BYTES = DIRECT_WAVES * SEG_BYTES;
constexpr u32 DIRECT_GROUP = 64; // tiles per group (8 MiB): one sum word per group, 128 of them per GiB
constexpr u32 DIRECT_WINDOW = CHUNK_BYTES / 4 - 8; // the exchange buffer's 1024 words minus skew and dump slots (emit_stage_words)
static_assert(emit_stage_words(DIRECT_WINDOW) * 4 == CHUNK_BYTES && DIRECT_WINDOW % 4 == 0, "the emission window is the exchange buffer");
constexpr u64 DIRECT_VALID = 1ull << 63;
__host__ __device__ inline u32 direct_groups(u32 ntiles) { return (ntiles + DIRECT_GROUP - 1) / DIRECT_GROUP; }
__host__ __device__ inline u32 direct_words(u32 ntiles) { return ntiles + direct_groups(ntiles); } // in front of the control words

// prefix (count, parity) of tile `tile` = everything in front of it: wave-wide, all 64 lanes.  false: timed out
__device__ __forceinline__ bool direct_prefix(const u64 *__restrict__ desc, u32 ntiles, u32 tile, u32 lane, u32 &B, u32 &S, scan_origin org) {
  const u64 *A = desc, *G = desc + ntiles;
  const u32 g = tile / DIRECT_GROUP, na = tile % DIRECT_GROUP; // groups in front of mine; tiles in front of me inside my group
  const u64 t_start = wall_clock64();
  for (;;) {
    const u64 a = lane < na ? desc_load(A + g * DIRECT_GROUP + lane) : DIRECT_VALID;
    u32 cnt = lane < na ? u32(a) : 0u, par = lane < na ? u32(a >> 32) & 1u : 0u;
    bool ready = (a >> 63) != 0;
#pragma unroll 1
    for (u32 j = lane; j < g; j += 64u) { // 128 groups per GiB: two rounds (wave-uniform trip count up to the last round)
      const u64 gg = desc_load(G + j);
      ready = ready && (gg >> 48) == DIRECT_GROUP;
      cnt += u32(gg);
      par += u32(gg >> 32) & 0xFFFFu;
    }
    if (!__ballot(!ready)) {
      B = org.base0 + wave_sum(cnt);
      S = (wave_sum(par) & 1u) ^ (org.carry & CARRY_IN_STRING);
      return true;
    }
    if (wall_clock64() - t_start > LOOKBACK_TIMEOUT_TICKS) { return false; }
    __builtin_amdgcn_s_sleep(4);
  }
}

#ifndef SJGPU_DIRECT_EU
#define SJGPU_DIRECT_EU 4 // waves per SIMD the kernel is compiled for: 4 = 128 VGPRs (24 B of cold scratch), 5 = 96 VGPRs (136 B); lab builds override it
#endif
__global__ __launch_bounds__(64 * DIRECT_WAVES) SJ_WAVES_PER_EU(SJGPU_DIRECT_EU, SJGPU_DIRECT_EU) void k_stage1_direct(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ desc, u32 *__restrict__ ticket,
                                                                                       u32 ntiles, u32 *__restrict__ idx, u64 idx_words,
                                                                                       scan_result_dev *__restrict__ result, scan_origin org) {
  const u32 lane = lane_id();
  const u32 wave = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) u32 park[DIRECT_WAVES][UTF8P_ROWS * UTF8P_ROW_WORDS];
  __shared__ __attribute__((aligned(16))) uint4 xbuf[DIRECT_WAVES][CHUNK_BYTES / 16]; // chunk loads while a wave scans, its emission window in between
  __shared__ u32 sh_wave[2][DIRECT_WAVES][2]; // [iteration parity][wave]: count, quote parity
  __shared__ u32 sh_prefix[3];                // B, S, ok of the pending tile
  __shared__ u32 sh_tile[2];
  const bool more = (org.carry & CARRY_MORE) != 0;
  u64 *const G = desc + ntiles;
  // the pending segment (what this wave scanned an iteration ago and has not emitted yet)
  u64 keep_prev[SEG_CHUNKS] = {0, 0, 0, 0};
  u32 pend_tile = NO_TILE, pend_count = 0, pend_wflags = 0, pend_base_rel = 0, pend_par_rel = 0;
  u32 pend_tile_count = 0, pend_tile_par = 0; // (wave 0: the whole pending tile)
  u32 utf8_error = 0;
  u32 next_ticket = 0;
  if (threadIdx.x == 0) { sh_tile[0] = atomicAdd(ticket, 1u); }
  lds_writes_done();
  __syncthreads();
  for (u32 iter = 0;; iter++) {
    const u32 cur = iter & 1u;
    const u32 tile = sh_tile[cur];
    const bool have = tile < ntiles, pend = pend_tile != NO_TILE; // workgroup-uniform
    if (!have && !pend) { break; }
    if (threadIdx.x == 0 && have) { next_ticket = atomicAdd(ticket, 1u); }
    // ---- scan my segment of the new tile (k_stage1_summarize's body); its final masks stay in registers ----
    const u64 seg_start = org.begin + (u64(tile) * DIRECT_WAVES + wave) * SEG_BYTES;
    const bool mine = have && seg_start < len; // wave-uniform
    u64 keep[SEG_CHUNKS] = {0, 0, 0, 0};
    u32 count = 0, parity = 0, wflags = 0;
    utf8_park uq{park[wave], 0u, 0u, 0u, 0x20202020u, buf, len, more ? 1u : 0u, UTF8P_DENSE_FROM};
    if (mine) {
      const u64 lane_off = u64(lane) * BLOCK_BYTES;
      const u32 lookback = lookback_issue(buf, seg_start, lane);
      wave_carry wc{0u, 0u, 0u};
      span_x sx;
      u32 n_a = 0;
      bool any_a = false, resolved = false;
      u32 derived = 0;
      u64 flip = 0;
      // (a rolled loop over the four chunks, their masks in a register FIFO -- no dynamic register indexing: unrolled like k_stage1_summarize's, the kernel is
      // 57 KB of code with the emission inlined behind it; rolled, 31 KB)
      u64 k0 = 0, k1 = 0, k2 = 0, k3 = 0; // slot 3 = oldest chunk
#pragma unroll 1
      for (u32 c = 0; c < SEG_CHUNKS; c++) {
        const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
        u64 structural = 0;
        if (cstart < len) { // wave-uniform
          const u64 pos = cstart + lane_off;
          u32 w[16];
          if (cstart + CHUNK_BYTES <= len) { load_chunk_stream(buf, cstart, lane, xbuf[wave], w); }
          else { load_block(buf, pos, len, w); }
          if (c == 0) {
            wc = span_carry_assume(seg_start, lane, lookback, sx);
            utf8_park_begin(uq, lookback, lane);
          }
          span_note_chunk(sx, w, c * CHUNK_BYTES, lane);
          const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, u32(cstart / BLOCK_BYTES));
          if (c == SEG_CHUNKS - 1) { span_note_tail(sx, m.backslash, m.quote_raw); }
          if (c == 0) { // the first control character pins the segment's string state (k_stage1_summarize)
            const u64 cm = __ballot(m.ctrl != 0);
            if (cm) {
              const u32 lc = ctz64(cm);
              u32 v = 0;
              if (lane == lc) { v = u32(m.in_string >> ctz64(m.ctrl)) & 1u; }
              derived = readlane_dyn(v, lc);
              resolved = true;
              flip = derived ? ~0ull : 0ull;
            }
          }
          // (an unresolved segment makes the call give up below: what is kept of it does not matter)
          structural = m.cand & ~(m.string_tail ^ flip);
          n_a += u32(popc64(structural));
          any_a |= __ballot((m.ctrl & (m.in_string ^ flip)) != 0) != 0;
        }
        k3 = k2; k2 = k1; k1 = k0; k0 = structural;
      }
      keep[0] = k3; keep[1] = k2; keep[2] = k1; keep[3] = k0;
      // the rows this wave has left are validated by itself, now (k_stage1_summarize hands them to wave 0 behind its one barrier: here that would be eight
      // waves standing by in every iteration, and the row check's registers in a kernel that has none to spare)
      if (uq.count) { utf8_park_drain(uq, lane, uq.count); }
      count = wave_sum(n_a);
      parity = wc.s;
      const bool ok_view = !any_a; // (k_stage1_summarize: the carry-in that matches `derived` errs iff a control character sits in a string of the resolved view; the other one always)
      if (derived ? true : !ok_view) { wflags |= WF_CTRL_IF_OUT; }
      if (derived ? !ok_view : true) { wflags |= WF_CTRL_IF_IN; }
      const u32 xw = span_finish(sx, buf, seg_start, SEG_BYTES, len, wc, true, resolved, derived);
      if ((!resolved || xw != 0u) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_INTERNAL); } // not plain: sums do not describe this input (the split pipeline does)
      if (uq.error) { utf8_error = 1u; }
    }
    if (lane == 0) {
      sh_wave[cur][wave][0] = count;
      sh_wave[cur][wave][1] = parity;
    }
    // ---- wave 0: the pending tile's prefix, while the other waves finish scanning (its predecessors published an iteration ago) ----
    u32 pB = 0, pS = 0;
    bool p_ok = false;
    if (wave == 0 && pend) { p_ok = direct_prefix(desc, ntiles, pend_tile, lane, pB, pS, org); }
    lds_writes_done();
    __syncthreads();
    // ---- every wave: where its segment lies inside the new tile (kept for the next iteration); wave 0: publish, broadcast ----
    u32 base_rel = 0, par_rel = 0, tile_count = 0, tile_par = 0;
#pragma unroll
    for (u32 v = 0; v < DIRECT_WAVES; v++) {
      const u32 c = sh_wave[cur][v][0], q = sh_wave[cur][v][1];
      if (v < wave) { base_rel += c; par_rel ^= q; }
      tile_count += c;
      tile_par ^= q;
    }
    if (wave == 0) {
      if (have) {
        if (lane == 0) {
          const u64 add = (1ull << 48) | (u64(tile_par & 1u) << 32) | u64(tile_count);
          desc_store(desc + tile, DIRECT_VALID | (u64(tile_par & 1u) << 32) | u64(tile_count));
          (void)__hip_atomic_fetch_add(G + tile / DIRECT_GROUP, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (pend && lane == 0) {
        if (p_ok) {
          if (pend_tile == ntiles - 1) { // the last tile knows the totals: n, unclosed string, sentinels (json_structural_indexer.h:284-286)
            const u32 total_out = pB + pend_tile_count, s_end = pS ^ pend_tile_par;
            u32 f = s_end ? SJGPU_F_UNCLOSED_STRING : 0u;
            if (u64(total_out) + 3 <= idx_words) {
              idx[total_out] = u32(len);
              idx[total_out + 1] = u32(len);
              idx[total_out + 2] = 0;
            } else {
              f |= SJGPU_F_IDX_OVERFLOW;
            }
            result->n = total_out;
            result->out_len = 0;
            if (f) { atomicOr(ctl_flags(ticket), f); }
          }
        } else {
          atomicOr(ctl_flags(ticket), SJGPU_F_INTERNAL);
        }
        sh_prefix[0] = pB;
        sh_prefix[1] = pS;
        sh_prefix[2] = p_ok ? 1u : 0u;
      }
      if (lane == 0) { sh_tile[cur ^ 1u] = have ? next_ticket : NO_TILE; } // the next iteration's ticket crosses with this barrier
    }
    // (not __syncthreads(): its release fence waits for the acknowledgement of the store and the add above -- a round trip to the device's coherence point
    // with eight waves standing by; what crosses here lives in LDS, whose writes are waited for, and leave_and_clean waits for everything global at the end)
    lds_writes_done();
    workgroup_barrier_lds_only();
    // ---- every wave: the pending segment's offsets from the registers, through the idle exchange buffer ----
    if (pend && pend_count != 0u && sh_prefix[2] != 0u) {
      u32 base = sh_prefix[0] + pend_base_rel;
      const u32 s = sh_prefix[1] ^ pend_par_rel;
      if ((pend_wflags & (s ? WF_CTRL_IF_IN : WF_CTRL_IF_OUT)) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_UNESCAPED_CTRL); }
      const u64 pend_start = org.begin + (u64(pend_tile) * DIRECT_WAVES + wave) * SEG_BYTES;
      bool overflow = false;
      emit_span4_adaptive<DIRECT_WINDOW>(keep_prev, u32(pend_start), lane, idx, idx_words, base, reinterpret_cast<u32 *>(xbuf[wave]), overflow, pend_count);
      if (__ballot(overflow) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_IDX_OVERFLOW); }
    } else if (pend && sh_prefix[2] != 0u) { // nothing to emit: the error bit of an empty segment still counts
      const u32 s = sh_prefix[1] ^ pend_par_rel;
      if ((pend_wflags & (s ? WF_CTRL_IF_IN : WF_CTRL_IF_OUT)) && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_UNESCAPED_CTRL); }
    }
    // ---- the segment scanned in this iteration becomes the pending one ----
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) { keep_prev[c] = keep[c]; }
    pend_tile = have ? tile : NO_TILE;
    pend_count = count;
    pend_wflags = mine ? wflags : 0u;
    pend_base_rel = base_rel;
    pend_par_rel = par_rel;
    pend_tile_count = tile_count;
    pend_tile_par = tile_par;
  }
  if (utf8_error && lane == 0) { atomicOr(ctl_flags(ticket), SJGPU_F_UTF8_ERROR); }
  leave_and_clean<64 * DIRECT_WAVES>(desc, direct_words(ntiles), ticket, result);
}

} // namespace

// ---- launchers ----------------------------------------------------------------------------------------
static inline void mark(hipEvent_t *ev, int k, hipStream_t stream) {
  if (ev) { (void)hipEventRecord(ev[k], stream); }
}

// clean: the workspace (result, descriptors, control words) is all zero on the device -- every kernel below leaves it that way -- so
// nothing is cleared; else (first use after something else wrote there, A/B switch SJGPU_FUSED_MEMSET) one memset in front of the kernel
static void clear_fused_workspace(scan_result_dev *result, uint64_t *desc, u32 ntiles, bool clean, hipStream_t stream) {
  static const bool always = std::getenv("SJGPU_FUSED_MEMSET") != nullptr; // A/B switch: the clear of rounds 1-4 in front of every call
  if (clean && !always) { return; }
  const size_t words = size_t(ntiles) + FUSED_CTL_WORDS;
  if (reinterpret_cast<uint64_t *>(result + 1) == desc) { // laid out back to back by the context: one clear
    (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev) + words * sizeof(u64), stream);
  } else {
    (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
    (void)hipMemsetAsync(desc, 0, words * sizeof(u64), stream);
  }
}

template <u32 WC>
static void launch_fused_wc(int op, const uint8_t *buf, uint64_t len, uint64_t *desc, void *out, uint64_t out_words,
                            scan_result_dev *result, scan_origin org, uint32_t max_workgroups, hipStream_t stream,
                            hipEvent_t *ev, uint64_t *trace, uint32_t trace_tiles, bool clean, uint8_t *tok = nullptr) {
  constexpr u64 tile_bytes = u64(FUSED_WAVES) * WC * CHUNK_BYTES;
  const u32 ntiles = u32((len - org.begin + tile_bytes - 1) / tile_bytes);
  u32 *ticket = reinterpret_cast<u32 *>(desc + ntiles);
  clear_fused_workspace(result, desc, ntiles, clean && !trace, stream);
  const u32 grid = ntiles < max_workgroups ? ntiles : max_workgroups;
  u64 *no_trace = nullptr;
  if (trace) {
    hipLaunchKernelGGL((k_fused<0, true, WC>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words,
                       result, trace, trace_tiles, org, static_cast<u8 *>(nullptr));
  } else if (op == 0 && tok) {
    hipLaunchKernelGGL((k_fused<0, false, WC, true>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words,
                       result, no_trace, 0u, org, tok);
  } else if (op == 0) {
    hipLaunchKernelGGL((k_fused<0, false, WC>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words,
                       result, no_trace, 0u, org, static_cast<u8 *>(nullptr));
  } else {
    hipLaunchKernelGGL((k_fused<1, false, WC>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words,
                       result, no_trace, 0u, org, static_cast<u8 *>(nullptr));
  }
  mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded; sjgpu_profile_read reports them as 0)
}

// test hook (tests/host/test_kernels_emu.cpp lowers it so that small documents take the large-input kernels); never changed by the library
uint64_t debug_fused_small_below = FUSED_SMALL_BELOW;

// k_stage1_direct: persistent workgroups of four waves, five per CU (max_workgroups = 8 per CU)
static const char *launch_direct(const uint8_t *buf, uint64_t len, uint64_t *desc, u32 *idx, uint64_t idx_words, scan_result_dev *result, scan_origin org,
                                 uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean, u32 cleared_tiles) {
  const u32 nt = u32((len - org.begin + DIRECT_TILE_BYTES - 1) / DIRECT_TILE_BYTES);
  const u32 words = direct_words(nt);
  u32 *tk = reinterpret_cast<u32 *>(desc + words);
  if (words != cleared_tiles) { clear_fused_workspace(result, desc, words, clean, stream); } // (only a dirty workspace is cleared at all; the control words lie behind THIS kernel's words)
  static const unsigned per_cu8 = []() { const char *v = std::getenv("SJGPU_DIRECT_WG"); const int x = v ? std::atoi(v) : 2; return (x >= 1 && x <= 8) ? unsigned(x) : 2u; }(); // A/B: workgroups per CU (two of eight waves)
  const u32 resident = max_workgroups * per_cu8 / 8u ? max_workgroups * per_cu8 / 8u : 1u;
  const u32 cap = (nt + 1) / 2; // every workgroup should own >= 2 tiles for the deferral to work
  const u32 grid = cap < resident ? (cap ? cap : 1u) : resident;
  hipLaunchKernelGGL(k_stage1_direct, dim3(grid), dim3(64 * DIRECT_WAVES), 0, stream, buf, len, desc, tk, nt, idx, idx_words, result, org);
  mark(ev, 1, stream);
  return "k_stage1_direct (8 waves, 128 KiB tiles, deferred emission)";
}

// returns the name of the scan kernel it launched (sjgpu_profile_kernel)
static const char *launch_fused(int op, const uint8_t *buf, uint64_t len, uint64_t *desc, void *out, uint64_t out_words,
                                scan_result_dev *result, scan_origin org, uint32_t max_workgroups, hipStream_t stream,
                                hipEvent_t *ev, uint64_t *trace = nullptr, uint32_t trace_tiles = 0, bool clean = false, uint8_t *tok = nullptr) {
  static const bool plain = std::getenv("SJGPU_FUSED_PLAIN") != nullptr; // A/B switch: the non-pipelined large-input kernel
  mark(ev, 0, stream); // slot 0 = everything this call enqueues (the clear, the scan kernel)
  if (len - org.begin <= debug_fused_small_below && !trace) {
    launch_fused_wc<1>(op, buf, len, desc, out, out_words, result, org, max_workgroups, stream, ev, trace, trace_tiles, clean, tok);
    return op == 0 ? (tok ? "k_fused<0, tokens> (16 KiB tiles)" : "k_fused<0> (16 KiB tiles)") : "k_fused<1> (16 KiB tiles)";
  } else if (trace || (plain && !tok)) {
    launch_fused_wc<FUSED_WAVE_CHUNKS>(op, buf, len, desc, out, out_words, result, org, max_workgroups, stream, ev, trace, trace_tiles, clean);
    return op == 0 ? "k_fused<0> (64 KiB tiles)" : "k_fused<1> (64 KiB tiles)";
  } else {
    static const bool onchip = []() { const char *v = std::getenv("SJGPU_MINIFY_ONCHIP"); return !v || v[0] != '0'; }(); // A/B: 0 = the re-reading kernel
    // minify: EIGHT waves per workgroup, 64 KiB tiles, two workgroups per CU (round 4; same reasoning and same measurement as for stage 1 below: what a tile
    // pays whatever its size, per byte, halves -- 543 -> 489 us per GiB of large_random, 168 -> 145 us per 256 MiB; sixteen waves, one workgroup per CU with
    // nobody to hide its look-back behind, are back at 540: profiles/r04_pipe_waves_ab.txt.  Round 2 had measured the 8-wave variant slower: it spilled then.)
    static const unsigned minify_waves = []() { const char *v = std::getenv("SJGPU_MINIFY_WAVES"); const int w = v ? std::atoi(v) : 8; return (w == 4 || w == 16) ? unsigned(w) : 8u; }(); // A/B
    const u32 onchip_waves = (op == 1 && onchip) ? minify_waves : 0u;
    // Stage 1: EIGHT waves per workgroup, 128 KiB tiles (round 4).  What a tile pays whatever its size -- a ticket, a look-back, a descriptor, three
    // barriers -- is served at ~17 ns per tile device-wide (scripts/micro/gather_lab.hip: a kernel of 16 384 trivial tiles takes 290 us with them, 130 us
    // without), a third of a 64 KiB tile's time at 4.5 TB/s; twice the tile at the same number of waves per CU halves it: 520 -> 466-478 us per GiB of
    // large_random, 173 -> 153-157 us per 256 MiB (profiles/r04_pipe_waves_ab.txt; SJGPU_PIPE_WAVES=4 brings the round-1 shape back).
    static const unsigned pipe_waves = []() { const char *v = std::getenv("SJGPU_PIPE_WAVES"); return v && std::atoi(v) == 4 ? 4u : 8u; }();
    const u32 s1_waves = (op == 0 && !trace) ? (tok ? 8u : pipe_waves) : FUSED_WAVES;
    const u64 tile_bytes = onchip_waves ? u64(onchip_waves) * ONCHIP_WAVE_CHUNKS * CHUNK_BYTES : u64(s1_waves) * FUSED_WAVE_CHUNKS * CHUNK_BYTES;
    const u32 ntiles = u32((len - org.begin + tile_bytes - 1) / tile_bytes);
    u32 *ticket = reinterpret_cast<u32 *>(desc + ntiles);
    clear_fused_workspace(result, desc, ntiles, clean, stream); // (a clean workspace -- every call but the first -- needs none: leave_and_clean)
    // half as many workgroups as tiles at most: every workgroup should own >= 2 tiles for the deferral to work
    const u32 cap = (ntiles + 1) / 2;
    const u32 grid = cap < max_workgroups ? cap : max_workgroups;
    // measured round 3 (profiles/r03_phase_priority.txt): emission at priority 3 (policy 4) is 2.4-2.9 % faster on dense output, nothing else moves
    static const unsigned prio = []() { const char *v = std::getenv("SJGPU_PRIO"); return v ? unsigned(std::atoi(v)) & 7u : 4u; }(); // A/B switch: phase_prio, 0 = off
    org.carry |= prio << 24;
    static const bool late_ticket = std::getenv("SJGPU_LATE_TICKET") != nullptr;                                       // A/B switch
    if (late_ticket) { org.carry |= CARRY_DEBUG_LATE_TICKET; }
    // The barrier at the top of the pipelined kernels' loop (rounds 1-5 needed it; since round 6 the kernels are correct without).  Measured with and
    // without, one process (profiles/r06_pipelined_ab.txt): minify 467 -> 456 us per GiB without it (a wave that has compacted its chunks starts loading at
    // once), stage 1 on dense output 471 -> 480 WITH it gone (its emission phases fall out of step), sparse output 331 -> 329.  So: stage 1 keeps it,
    // minify does not; env SJGPU_TOP_BARRIER=0 / 1 forces either.
    // tickets from two counters (draw_ticket): measured in one process (profiles/r06_pipelined_ab.txt): minify 456 -> 451 us per GiB, stage 1 472 -> 469 on large_random,
    // 333 -> 328 on NDJSON, 1104 -> 1124 on deep nesting (one offset per byte: its emission phases want the tiles in step).  So: minify takes two, stage 1 one;
    // env SJGPU_TWO_TICKETS=0 / 1 forces either.
    static const int two_tickets_env = []() { const char *v = std::getenv("SJGPU_TWO_TICKETS"); return v ? (v[0] != '0' ? 1 : 0) : -1; }();
    if (two_tickets_env >= 0 ? two_tickets_env == 1 : op == 1) { org.carry |= CARRY_DEBUG_TWO_TICKETS; }
    static const int top_barrier_env = []() { const char *v = std::getenv("SJGPU_TOP_BARRIER"); return v ? (v[0] != '0' ? 1 : 0) : -1; }();
    if (top_barrier_env >= 0 ? top_barrier_env == 1 : op == 0) { org.carry |= CARRY_DEBUG_TOP_BARRIER; }
    static const bool late_lookback = std::getenv("SJGPU_LATE_LOOKBACK") != nullptr; // A/B switch: the look-back behind the scan's barrier (rounds 1-5)
    if (late_lookback) { org.carry |= CARRY_DEBUG_LATE_LOOKBACK; }
    static const bool no_hint = std::getenv("SJGPU_NO_SPAN_HINT") != nullptr; // A/B switch
    if (no_hint) { org.carry |= CARRY_DEBUG_NO_SPAN_HINT; }
    static const bool queue_only = std::getenv("SJGPU_UTF8_QUEUE_ONLY") != nullptr; // A/B switch
    if (queue_only) { org.carry |= CARRY_DEBUG_QUEUE_UTF8; }
    if (onchip_waves) {
      const u32 resident_q = max_workgroups / (onchip_waves / 2u), resident = resident_q ? resident_q : 1u; // max_workgroups = 8 per CU; 33 KiB of LDS per four waves: sixteen waves per CU
      const u32 g = cap < resident ? cap : resident;
      if (onchip_waves == 8u) {
        hipLaunchKernelGGL((k_minify_onchip<8>), dim3(g), dim3(512), 0, stream, buf, len, desc, ticket, ntiles, static_cast<u8 *>(out), result, org);
      } else if (onchip_waves == 16u) {
        hipLaunchKernelGGL((k_minify_onchip<16>), dim3(g), dim3(1024), 0, stream, buf, len, desc, ticket, ntiles, static_cast<u8 *>(out), result, org);
      } else {
        hipLaunchKernelGGL((k_minify_onchip<4>), dim3(g), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, static_cast<u8 *>(out), result, org);
      }
      mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded -- two stream markers less per call; sjgpu_profile_read reports them as 0)
      return onchip_waves == 8u ? "k_minify_onchip<8>" : (onchip_waves == 16u ? "k_minify_onchip<16>" : "k_minify_onchip<4>");
    }
    // round 6: the one-pass kernel derived from the split pipeline's scan (k_stage1_direct); env SJGPU_DIRECT: 0 off, 1 blockIdx order, 2 tickets
    if (op == 0 && tok) { // the token stream: the default shape only (no A/B shapes)
      const u32 resident8 = max_workgroups >= 2u ? max_workgroups / 2u : 1u;
      hipLaunchKernelGGL((k_fused_pipelined<0, false, FUSED_WAVE_CHUNKS, 8, true>), dim3(cap < resident8 ? cap : resident8), dim3(512), 0, stream, buf, len, desc, ticket, ntiles, out,
                         out_words, result, org, static_cast<u64 *>(nullptr), tok);
      mark(ev, 1, stream);
      return "k_fused_pipelined<0, tokens> (8 waves, 128 KiB tiles)";
    }
    // round 6: the one-pass kernel for plain input (k_stage1_direct); env SJGPU_DIRECT=1 forces it here (AUTO: launch_stage1_direct below, by the context's hint)
    static const bool direct = []() { const char *v = std::getenv("SJGPU_DIRECT"); return v && v[0] != '0'; }();
    if (op == 0 && direct) { return launch_direct(buf, len, desc, static_cast<u32 *>(out), out_words, result, org, max_workgroups, stream, ev, clean, ntiles); }
    static const unsigned pipe_wc = []() { const char *v = std::getenv("SJGPU_PIPE_WC"); return v ? unsigned(std::atoi(v)) : 4u; }(); // A/B switch: 2 = 32 KiB tiles
    if (op == 0 && pipe_wc == 2u) {
      const u32 nt2 = u32((len - org.begin + FUSED_TILE_BYTES / 2 - 1) / (FUSED_TILE_BYTES / 2)); // (the clear above covered fewer, larger tiles: clear again)
      u32 *ticket2 = reinterpret_cast<u32 *>(desc + nt2);
      clear_fused_workspace(result, desc, nt2, false, stream);
      const u32 cap2 = (nt2 + 1) / 2;
      hipLaunchKernelGGL((k_fused_pipelined<0, false, 2>), dim3(cap2 < max_workgroups ? cap2 : max_workgroups), dim3(256), 0, stream, buf, len, desc, ticket2, nt2, out, out_words,
                         result, org);
      mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded -- two stream markers less per call; sjgpu_profile_read reports them as 0)
      return "k_fused_pipelined<0> (32 KiB tiles)";
    }
    if (op == 0 && s1_waves == 8u) {
      const u32 resident8 = max_workgroups >= 2u ? max_workgroups / 2u : 1u; // two workgroups of eight waves per CU
      hipLaunchKernelGGL((k_fused_pipelined<0, false, FUSED_WAVE_CHUNKS, 8>), dim3(cap < resident8 ? cap : resident8), dim3(512), 0, stream, buf, len, desc, ticket, ntiles, out,
                         out_words, result, org);
      mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded -- two stream markers less per call; sjgpu_profile_read reports them as 0)
      return "k_fused_pipelined<0> (8 waves, 128 KiB tiles)";
    }
    if (op == 0) {
      hipLaunchKernelGGL((k_fused_pipelined<0>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words, result, org);
    } else {
      hipLaunchKernelGGL((k_fused_pipelined<1>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, out, out_words, result, org);
    }
    mark(ev, 1, stream); // (a single kernel: slots 1 and 2 stay unrecorded; sjgpu_profile_read reports them as 0)
    return op == 0 ? "k_fused_pipelined<0>" : "k_fused_pipelined<1>";
  }
}

const char *launch_stage1_fused(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                scan_result_dev *result, scan_origin org, uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean, uint8_t *tok) {
  return launch_fused(0, buf, len, desc, idx, idx_words, result, org, max_workgroups, stream, ev, nullptr, 0, clean, tok);
}
// one traced run of the pipelined stage-1 kernel; trace holds *grid_out x PIPE_TRACE_ITERS x 8 stamps (zero = not reached)
uint32_t launch_stage1_pipelined_traced(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                        scan_result_dev *result, uint32_t max_workgroups, hipStream_t stream,
                                        uint64_t *trace, uint32_t max_records) {
  scan_origin org{0, 0, 0};
  const u32 ntiles = u32((len + FUSED_TILE_BYTES - 1) / FUSED_TILE_BYTES);
  u32 *ticket = reinterpret_cast<u32 *>(desc + ntiles);
  (void)hipMemsetAsync(result, 0, sizeof(scan_result_dev), stream);
  (void)hipMemsetAsync(desc, 0, (size_t(ntiles) + FUSED_CTL_WORDS) * sizeof(u64), stream);
  const u32 cap = (ntiles + 1) / 2;
  u32 grid = cap < max_workgroups ? cap : max_workgroups;
  if (grid > max_records / PIPE_TRACE_ITERS) { return 0; }
  hipLaunchKernelGGL((k_fused_pipelined<0, true>), dim3(grid), dim3(256), 0, stream, buf, len, desc, ticket, ntiles, idx, idx_words, result, org, trace);
  return grid;
}

void launch_stage1_fused_traced(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                scan_result_dev *result, uint32_t max_workgroups, hipStream_t stream, uint64_t *trace,
                                uint32_t trace_tiles) {
  launch_fused(0, buf, len, desc, idx, idx_words, result, scan_origin{0, 0, 0}, max_workgroups, stream, nullptr, trace, trace_tiles);
}
// the one-pass kernel for PLAIN input (k_stage1_direct): a call whose input is not plain reports SJGPU_F_INTERNAL and is re-run on the split pipeline
const char *launch_stage1_direct(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words, scan_result_dev *result, scan_origin org,
                                 uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean) {
  mark(ev, 0, stream);
  return launch_direct(buf, len, desc, idx, idx_words, result, org, max_workgroups, stream, ev, clean, 0xFFFFFFFFu);
}
const char *launch_minify_fused(const uint8_t *buf, uint64_t len, uint64_t *desc, uint8_t *dst, scan_result_dev *result,
                                scan_origin org, uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean) {
  return launch_fused(1, buf, len, desc, dst, 0, result, org, max_workgroups, stream, ev, nullptr, 0, clean);
}

} // namespace sjgpu
