// simdjson_amd/csrc/sjgpu_string_stream.hip -- SURVEY 8(f3): document::string_buf as a stream compaction of the document.
//
// The per-string walk of sjgpu_strings.hip gives one LANE one string: loads are gathers, lanes wait for the longest string of
// their wave, and the pass sits at 4 % of the HBM roofline (profiles/r02_pmc_strings.txt).  But the records of string_buf
// -- [u32 length][unescaped bytes][0], /root/reference/src/generic/stage2/tape_builder.h:415-433, stringparsing.h:150-193 --
// follow each other in document order, so the buffer is the DOCUMENT with the bytes outside strings dropped, the quotes
// replaced by 4 + 1 bytes and the escapes by what they stand for (sj_string_stream.h): the job has the shape of minify.
// A lane owns 64 bytes, a wave 4 KiB, a wave's share is one 16 KiB segment; loads are the 16-byte row loads of stage 1.
//   k_strs_count     per segment: output bytes, opening quotes and rejected escapes for both "starts inside / outside a string"
//   k_strs_resolve   a workgroup per 1024 segments: in-string state, output base and string ordinal in front of every segment; totals
//   k_strs_tokens + scan   which structurals are quotes (one bit each) and how many per tile of 4096 (skipped when the tape has counted them already)
//   k_strs_decide    the stream is taken iff every string is valid, every opening quote of the document is a structural (a quote
//                    glued to a scalar, a"b", is not: such documents are invalid and take the per-string path) and the buffer fits
//   k_strs_write     the bytes, through a per-wave LDS window (16-byte stores); where every string begins (by ordinal)
//   k_strs_finalize  per structural: its record offset (CSR, as before) and, for a string, the length word (the tape does this itself)
// Everything else -- a rejected escape, an unclosed string, unlisted quotes -- is left to k_strings<> (sjgpu_strings.hip), which
// then runs instead: same results as before for those documents, the fast path for all valid ones.
#include "sjgpu_device.h"
#include "sj_string_stream.h"

namespace sjgpu {
namespace {

constexpr u32 STRS_WAVES = 4; // waves (= segments) per workgroup

// The document's bytes for u_escape_byte (sj_string_stream.h): the value of a kept byte of a \\u escape is a function of at most ten bytes in
// front of it, which the workgroup has just loaded (L1 / L2).  Bytes at or beyond len read as 0x20, like load_block's.
// (Rounds 2-3 walked every lane's escapes with a decoder inside a divergent loop and parked the chunk in LDS for it; round 4 has no walk:
// which bytes stay is mask algebra -- unicode_masks --, and the values are computed one kept byte per lane.)
struct plain_doc {
  const u8 *buf;
  u32 len;
  __device__ __forceinline__ u32 byte(u32 pos) const { return pos < len ? u32(buf[pos]) : 0x20u; }
};

// The same for ONE kept byte of a \\u escape at position q: u_escape_byte reads q - 8 ... q + 2, so the sixteen bytes from q - 8 on, requested at once (two
// unaligned 8-byte loads), serve every byte() it asks for.  Through plain_doc the function's loads form a chain of two or three round trips to L1 / L2
// (the digits decide which bytes are read next), and a wave that patches a chunk waits for every one of them: 92 of k_strs_write's 222 us on the 256 MiB
// twitter-like text (session AS: the kernel with the patch rounds compiled out), one round trip with the window.
struct escape_window {
  u64 lo, hi;
  u32 base; // position of the window's first byte
  typedef u64 __attribute__((aligned(1))) u64_any;
  __device__ __forceinline__ escape_window(const u8 *buf, u32 q) : lo(*reinterpret_cast<const u64_any *>(buf + q - 8)), hi(*reinterpret_cast<const u64_any *>(buf + q)), base(q - 8u) {}
  __device__ __forceinline__ u32 byte(u32 pos) const {
    const u32 d = pos - base;
    return u32((d < 8u ? lo : hi) >> (8u * (d & 7u))) & 0xFFu;
  }
};

// per segment, from its bytes alone (hypothesis 0 = the segment starts outside a string)
struct strs_summary {
  u32 bytes0, bytes1; // output bytes if it starts outside / inside a string
  u32 opens0, quotes; // opening quotes if it starts outside; real quotes
  u32 flags;          // bit 0: odd number of quotes; bit 1 / 2: an escape the reference rejects inside a string if it starts outside / inside;
                      // bit 3: the look-back does not settle the segment's carries (segment_carries): the stream declines the document
  u32 pad[3];
};
struct strs_base {
  u32 out;       // output bytes in front of the segment
  u32 ordinal;   // strings that begin in front of it
  u32 in_string; // it starts inside a string
  u32 pad;
};

struct strs_carry {
  u32 e;    // first byte of the next chunk is escaped
  u32 s;    // inside a string (relative to the segment start in k_strs_count, absolute in k_strs_write)
  u_tops t; // what the \\u algebra of the chunk's last block hands to the next chunk's first (sj_string_stream.h)
};
struct strs_chunk {
  u64 quote, in_string; // real quotes; stage 1's in-string mask (opening quote included, closing excluded)
  string_block b;
  u64 k2, k3, k4; // kept bytes of \\u escapes (on their 2nd / 3rd / 4th hex digit): the values come from u_escape_byte
};

__device__ __forceinline__ u64 brev64(u64 x) { return u64(__brevll((unsigned long long)x)); }
// The carries of a segment from its look-back (lane i holds byte start - 1 - i): the escape bit like stage 1's spans (span_carry_assume), and
// what the \\u algebra of the 64 bytes in front would hand over -- those bytes are run through the same unicode_masks as a block of their own,
// whose own carry-in cannot reach its last ten positions.  Two things leave a question open: 64 backslashes in front (SPAN_B), and an escaped
// 'u' among the last ten bytes whose backslash run reaches the look-back's first byte (its escaped mask then depends on what lies further
// back).  Stage 1 carries such questions through its summaries (sj_xcarry.h); here an escaped byte changes what the OUTPUT holds, not one bit
// of a mask, so the stream simply leaves documents that raise one to the per-string kernels (sjgpu_strings.hip walk every string from its
// opening quote and never ask): same buffer, the slower road, for documents with a backslash run of 54 bytes and more that ends within ten
// bytes of a 16 KiB boundary.
__device__ __forceinline__ bool segment_carries(u64 start, u32 lane, u32 lookback, strs_carry &wc) {
  span_x sx;
  wc.e = span_carry_assume(start, lane, lookback, sx).e;
  bool ambiguous = sx.kind() == SPAN_B;
  wc.t = u_tops{0u, 0u, 0u, 0u};
  if (start == 0) { return ambiguous; }
  const u64 cu = brev64(__ballot(lookback == u32('u'))); // bit i = byte start - 64 + i
  if ((cu >> 54) == 0) { return ambiguous; }             // (wave-uniform) no 'u' that an escape reaching the boundary could begin with
  const u64 bs = brev64(__ballot(lookback == 0x5Cu));
  u64 unused;
  const u64 U = escaped_mask(bs, 0, unused) & cu, U_other = escaped_mask(bs, 1, unused) & cu;
  if ((U ^ U_other) >> 54) { ambiguous = true; }
  hex_classes h;
  h.hex = brev64(__ballot(byte_is_hex(lookback)));
  h.zero = brev64(__ballot(lookback == u32('0')));
  h.oct = brev64(__ballot(byte_is_octal(lookback)));
  h.d = brev64(__ballot(byte_is_d(lookback)));
  h.s8b = brev64(__ballot(byte_is_89ab(lookback)));
  h.scf = brev64(__ballot(byte_is_cdef(lookback)));
  u_tops out{0u, 0u, 0u, 0u};
  (void)unicode_masks(U, h, [&](u32 stage, u32 mine) -> u32 {
    if (stage == 0u) { out.a = mine; } else if (stage == 1u) { out.b = mine; } else if (stage == 2u) { out.c = mine; } else { out.d = mine; }
    return 0u;
  });
  wc.t = out;
  return ambiguous;
}

// one chunk: stage 1's escape and quote algebra (scan_chunk, sjgpu_device.h), then what the strings need on top of it
__device__ __forceinline__ strs_chunk string_chunk(const u32 (&w)[16], strs_carry &wc, u32 lane) {
  const planes P = transpose64(w);
  const classes c = classify(P);
  const u64 lt = lanemask_lt(lane);
  strs_chunk out;
  u64 escaped = 0;
  const bool any_backslash = (__ballot(c.backslash != 0) | u64(wc.e)) != 0; // wave-uniform
  if (any_backslash) {
    const bool all_bs = (c.backslash == ~0ull);
    const u32 own_out = all_bs ? 0u : (clz64(~c.backslash) & 1u);
    const u64 passm = __ballot(all_bs), setm = __ballot(own_out != 0);
    const u64 below = ~passm & lt;
    const u32 e_in = below ? u32((setm >> (63u - clz64(below))) & 1ull) : wc.e;
    u64 unused;
    escaped = escaped_mask(c.backslash, u64(e_in), unused);
    const u64 nonpass = ~passm;
    wc.e = nonpass ? u32((setm >> (63u - clz64(nonpass))) & 1ull) : wc.e;
  }
  out.quote = andn(c.quote, escaped);
  const u64 parm = __ballot((popc64(out.quote) & 1) != 0);
  const u32 s_in = __builtin_amdgcn_mbcnt_hi(u32(parm >> 32), __builtin_amdgcn_mbcnt_lo(u32(parm), wc.s)) & 1u; // set bits below my lane, the carry as the start value
  wc.s ^= u32(popc64(parm)) & 1u;
  out.in_string = prefix_xor(out.quote ^ u64(s_in)); // (the carry as a quote in front of bit 0)
  out.b = no_escapes(out.quote);
  out.k2 = 0; out.k3 = 0; out.k4 = 0;
  const bool pending = wc.t.any(); // wave-uniform: an escape that began in the chunk in front is not finished
  if (any_backslash || pending) { // a backslash in the last byte of a block escapes nothing INSIDE the block but is dropped
    const escape_classes ec = classify_escapes(P);
    out.b = simple_escapes(c.backslash, escaped, out.quote, ec);
    const u64 U = escaped & ec.u;
    u_tops next{0u, 0u, 0u, 0u};
    if (__ballot(U != 0) | u64(pending)) { // wave-uniform: a \\u escape somewhere in the chunk or reaching into it
      const u_tops in = wc.t;
      const u_masks um = unicode_masks(U, classify_hex(P), [&](u32 stage, u32 mine) -> u32 {
        u32 prev = u32(__shfl_up(int(mine), 1));
        const u32 last = readlane(mine, 63);
        if (stage == 0u) { next.a = last; if (lane == 0) { prev = in.a; } }
        else if (stage == 1u) { next.b = last; if (lane == 0) { prev = in.b; } }
        else if (stage == 2u) { next.c = last; if (lane == 0) { prev = in.c; } }
        else { next.d = last; if (lane == 0) { prev = in.d; } }
        return prev;
      });
      apply_unicode(out.b, um);
      out.k2 = um.k2; out.k3 = um.k3; out.k4 = um.k4;
    }
    wc.t = next;
  }
  return out;
}

// ---- the control block of the stream (written by pass 2) ------------------------------------------------------------------------------------
struct strs_ctrl {
  u32 n1_scan;   // entries of the ordinal scan (n + 1)
  u32 n1_old;    // entries of the per-string path's scan: n + 1 if it runs, else 0 (its scan kernels then do nothing)
  u32 go_stream; // the stream writes the buffer
  u32 go_old;    // the per-string kernels write it
  u32 opens;     // opening quotes of the document
  u32 bad;       // a rejected escape inside a string, or the document ends inside one
  u64 total;     // output bytes
};
__device__ __forceinline__ void strs_init(strings_result_dev *__restrict__ res, strs_ctrl *__restrict__ ctrl, u32 n1) {
  strings_result_dev r{};
  r.first_bad = 0xFFFFFFFFu; // NO_STRING
  *res = r;
  strs_ctrl c{};
  c.n1_scan = n1;
  *ctrl = c;
  u32 *tail = reinterpret_cast<u32 *>(ctrl) + sizeof(strs_ctrl) / 4; // the control block has 64 bytes
  for (u32 k = 0; k < (64 - sizeof(strs_ctrl)) / 4; k++) { tail[k] = 0u; }
}
// (a launch of its own only for a document without a segment: else the first thread of k_strs_count does it -- nothing reads either block before k_strs_resolve)
__global__ void k_strs_init(strings_result_dev *__restrict__ res, strs_ctrl *__restrict__ ctrl, u32 n1) {
  if (threadIdx.x == 0) { strs_init(res, ctrl, n1); }
}
// ---- pass 1: what every segment contributes, for both carry-ins ------------------------------------------------------------------------
__global__ __launch_bounds__(64 * STRS_WAVES) void k_strs_count(const u8 *__restrict__ buf, u64 len, u32 nseg, u32 allow_replacement, strs_summary *__restrict__ summ,
                                                               strings_result_dev *__restrict__ res, strs_ctrl *__restrict__ ctrl, u32 n1) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { strs_init(res, ctrl, n1); }
  (void)allow_replacement; // (a lone surrogate is flagged whatever the option says: the per-string road knows the replacement character)
  const u32 lane = threadIdx.x & 63u;
  const u32 seg = blockIdx.x * STRS_WAVES + (threadIdx.x >> 6);
  if (seg >= nseg) { return; }
  const u64 seg_start = u64(seg) * SEG_BYTES;
  const u32 lookback = lookback_issue(buf, seg_start, lane);
  strs_carry wc{0u, 0u, u_tops{0u, 0u, 0u, 0u}};
  bool ambiguous = false;
  u32 d0 = 0, dall = 0, o0 = 0, qall = 0;
  u64 bad0 = 0, bad1 = 0;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    if (cstart + CHUNK_BYTES <= len) { load_block_full(buf, pos, w); }
    else { load_block(buf, pos, len, w); }
    if (c == 0) { ambiguous = segment_carries(seg_start, lane, lookback, wc); }
    const strs_chunk m = string_chunk(w, wc, lane);
    d0 += u32(popc64(m.b.keep & m.in_string));
    dall += u32(popc64(m.b.keep));
    o0 += u32(popc64(m.quote & m.in_string));
    qall += u32(popc64(m.quote));
    bad0 |= (m.b.bad & m.in_string & ~m.quote) | m.b.bad_u;
    bad1 |= (m.b.bad & ~m.in_string & ~m.quote) | m.b.bad_u;
  }
  d0 = wave_sum(d0);
  dall = wave_sum(dall);
  o0 = wave_sum(o0);
  qall = wave_sum(qall);
  const u64 any0 = __ballot(bad0 != 0), any1 = __ballot(bad1 != 0);
  if (lane == 0) {
    strs_summary s;
    s.bytes0 = d0 + 4u * o0 + (qall - o0);
    s.bytes1 = (dall - d0) + 4u * (qall - o0) + o0;
    s.opens0 = o0;
    s.quotes = qall;
    s.flags = (qall & 1u) | (any0 ? 2u : 0u) | (any1 ? 4u : 0u) | (ambiguous ? 8u : 0u);
    s.pad[0] = 0; s.pad[1] = 0; s.pad[2] = 0;
    summ[seg] = s;
  }
}

constexpr u32 RES_THREADS = 1024, RES_WAVES = RES_THREADS / 64, RES_MAX_TILES = 256; // (a document has at most 4 GiB / 16 KiB = 262 144 segments)
// The segments in tiles of 1024, a thread per segment, a WORKGROUP PER TILE: the in-string state in front of every segment (a prefix XOR of the parities),
// which selects the segment's counts, then the prefix sums of those.  Rounds 3-6a walked the tiles with ONE workgroup: 1.8 us per tile, 30 us per 256 MiB
// document during which 255 CUs waited -- and neither fewer barriers nor summaries requested four tiles ahead moved it (sessions AN-AP): 16 waves of
// ~200 instructions on one CU are 3 000 cycles per tile.  Now a tile computes, from its own summaries, its totals for BOTH states it may start in, publishes
// them -- two 64-bit words in the padding of its first two summaries (k_strs_count has zeroed them; the top bit says "written": the data is the flag, as
// in sjgpu_fused.hip) -- and reads the words of the tiles in front of it (one thread each; workgroups are dispatched in order, so those run or have
// run); a walk over at most 255 pairs of totals gives the tile its state and its bases.  A tile holds at most 1024 x 2.5 x 16 KiB = 40 MiB (26 bits)
// and 8 M opening quotes (23 bits).
__device__ __forceinline__ u64 *tile_word(strs_summary *summ, u32 tile, u32 which) { return reinterpret_cast<u64 *>(&summ[u64(tile) * RES_THREADS + which].pad[1]); }
static_assert(sizeof(strs_summary) == 32 && offsetof(strs_summary, pad) == 20, "tile_word: an aligned 64-bit word inside the padding");
constexpr u64 TILE_WRITTEN = u64(1) << 63;
__device__ __forceinline__ void strs_decide(strs_ctrl *__restrict__ ctrl, const int *__restrict__ listed_ptr, u32 n, u64 out_cap, u32 *__restrict__ outq, strings_result_dev *__restrict__ res);
__global__ __launch_bounds__(RES_THREADS) void k_strs_resolve(strs_summary *__restrict__ summ, u32 nseg, strs_base *__restrict__ base, strs_ctrl *__restrict__ ctrl,
                                                          const int *__restrict__ listed_ptr, u32 n, u64 out_cap, u32 *__restrict__ outq, strings_result_dev *__restrict__ res) {
  __shared__ u32 sh_par[RES_WAVES];
  __shared__ u32 sh_bad[RES_WAVES];          // bit 0 / 1: a segment of the wave declines, the tile starting outside / inside a string
  __shared__ u64 sh_tot[2][RES_WAVES];       // per wave, for the tile starting outside / inside a string: bytes | opening quotes << 32
  __shared__ u64 sh_back[2][RES_MAX_TILES];  // the words of the tiles in front
  __shared__ u64 sh_front[2];                // what the walk found: state in front of the tile; bytes (high: unused) -- and opening quotes
  __shared__ u32 sh_state;
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const u32 tile = blockIdx.x, ntiles = gridDim.x;
  const u64 i = u64(tile) * RES_THREADS + t;
  uint4 counts{0u, 0u, 0u, 0u}; // bytes0, bytes1, opens0, quotes
  u32 flags = 0;
  if (i < nseg) {
    counts = *reinterpret_cast<const uint4 *>(summ + i);
    flags = summ[i].flags;
  }
  // the state of every segment relative to the tile's first
  const u64 odd = __ballot((flags & 1u) != 0);
  if (lane == 0) { sh_par[wave] = u32(popc64(odd)); }
  __syncthreads();
  u32 par_front = 0, par_total = 0;
  for (u32 w = 0; w < RES_WAVES; w++) {
    const u32 p = sh_par[w];
    par_front += w < wave ? p : 0u;
    par_total += p;
  }
  const u32 rel = (par_front + u32(popc64(odd & ((u64(1) << lane) - 1)))) & 1u;
  // what the segment adds if the TILE starts outside (h = 0: the segment's state is rel) or inside a string (h = 1)
  const u32 in_b = counts.y, in_o = counts.w - counts.z; // the segment starts inside a string
  const u32 b0 = rel ? in_b : counts.x, o0 = rel ? in_o : counts.z;
  const u32 b1 = rel ? counts.x : in_b, o1 = rel ? counts.z : in_o;
  const u32 ib0 = wave_incl_scan(b0), io0 = wave_incl_scan(o0), ib1 = wave_incl_scan(b1), io1 = wave_incl_scan(o1);
  // bit 1 / 2 of the flags: an escape the reference rejects inside a string if the SEGMENT starts outside / inside; bit 3: it declines anyway
  const u32 out_bad = ((flags >> 1) | (flags >> 3)) & 1u, in_bad = ((flags >> 2) | (flags >> 3)) & 1u;
  const u64 bad0 = __ballot((rel ? in_bad : out_bad) != 0), bad1 = __ballot((rel ? out_bad : in_bad) != 0);
  if (lane == 63) {
    sh_tot[0][wave] = u64(ib0) | (u64(io0) << 32);
    sh_tot[1][wave] = u64(ib1) | (u64(io1) << 32);
    sh_bad[wave] = (bad0 ? 1u : 0u) | (bad1 ? 2u : 0u);
  }
  __syncthreads();
  u64 tot0 = 0, tot1 = 0;
  u32 tile_bad = 0;
  if (t == 0) { // the tile's own words first: nobody behind it waits for what it is about to wait for
    for (u32 w = 0; w < RES_WAVES; w++) { tot0 += sh_tot[0][w]; tot1 += sh_tot[1][w]; tile_bad |= sh_bad[w]; }
    if (tile + 1 < ntiles) { // (a tile with a tile behind it is full: its second summary exists)
      const u64 a = (tot0 & 0x3FFFFFFull) | ((tot1 & 0x3FFFFFFull) << 26) | (u64(par_total & 1u) << 52) | (u64(tile_bad) << 53) | TILE_WRITTEN;
      const u64 b = (tot0 >> 32) | ((tot1 >> 32) << 23) | TILE_WRITTEN;
      __hip_atomic_store(tile_word(summ, tile, 0), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tile_word(summ, tile, 1), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (t < tile) { // the tiles in front: their words, when they are written
    u64 a, b;
    do { a = __hip_atomic_load(tile_word(summ, t, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(a & TILE_WRITTEN));
    do { b = __hip_atomic_load(tile_word(summ, t, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(b & TILE_WRITTEN));
    sh_back[0][t] = a;
    sh_back[1][t] = b;
  }
  __syncthreads();
  if (t == 0) {
    u32 s = 0, opens = 0, bad = 0;
    u64 bytes = 0;
    for (u32 w = 0; w < tile; w++) {
      const u64 a = sh_back[0][w], b = sh_back[1][w];
      bytes += s ? (a >> 26) & 0x3FFFFFFull : a & 0x3FFFFFFull;
      opens += u32(s ? (b >> 23) & 0x7FFFFFull : b & 0x7FFFFFull);
      bad |= u32(a >> (53u + s)) & 1u;
      s ^= u32(a >> 52) & 1u;
    }
    sh_state = s;
    sh_front[0] = bytes;
    sh_front[1] = opens;
    if (tile + 1 == ntiles) { // the totals of the document, the state behind its last byte
      const u64 own = s ? tot1 : tot0;
      ctrl->total = bytes + (own & 0xFFFFFFFFull);
      ctrl->opens = opens + u32(own >> 32);
      ctrl->bad = bad | ((tile_bad >> s) & 1u) | ((s ^ par_total) & 1u); // a segment that declines, or the document ends inside a string
      if (listed_ptr) { strs_decide(ctrl, listed_ptr, n, out_cap, outq, res); } // (the string tokens were counted in front: the tape's call)
    }
  }
  __syncthreads();
  const u32 s_tile = sh_state;
  const u32 s = rel ^ s_tile;
  u64 front = 0;
  for (u32 w = 0; w < RES_WAVES; w++) { front += w < wave ? sh_tot[s_tile][w] : u64(0); }
  const u32 mine_b = s_tile ? b1 : b0, mine_o = s_tile ? o1 : o0;
  const u32 excl_b = u32(front & 0xFFFFFFFFull) + (s_tile ? ib1 : ib0) - mine_b, excl_o = u32(front >> 32) + (s_tile ? io1 : io0) - mine_o;
  if (i < nseg) { base[i] = strs_base{u32(sh_front[0] + excl_b), u32(sh_front[1]) + excl_o, s, 0u}; }
}

// ---- the structural list: which tokens are strings ------------------------------------------------------------------------------------------
// One BIT per token and one count per tile of 4096 (sixteen rows of 256, one token per lane and row: a wave fetches the bytes of 64 consecutive tokens
// with one instruction); the ordinal of a token among the strings is the tile's prefix + a popcount, computed where it is needed (k_strs_finalize).
// Rounds 3-4a wrote an int per token, ran the generic three-kernel scan over that array and read it back: 1.0 GB of traffic for a 256 MiB twitter-like
// document (32 M tokens), 0.26 ms of the stand-alone string pass's 0.79.
constexpr u32 TOK_THREADS = 256, TOK_ROWS = 16, TOK_TILE = TOK_THREADS * TOK_ROWS, TOK_WAVES = TOK_THREADS / 64;
static_assert(TOK_TILE == 4096, "carve_strings_scratch (sjgpu_strings.hip) sizes the token pass's room for tiles of 4096");
static_assert(TOK_ROWS * TOK_WAVES == 64, "a tile's (row, wave) counts are scanned by one wave");
// qbits[tile * 64 + row * 4 + wave]: bit l = token tile * 4096 + row * 256 + wave * 64 + l is a string; count[tile] = strings of the tile
// (count has one more entry, zero: the exclusive scan over all of them leaves the number of string tokens there)
__global__ __launch_bounds__(TOK_THREADS) void k_strs_tokens(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, u32 n, u64 *__restrict__ qbits,
                                                            int *__restrict__ count, u32 tiles) {
  __shared__ u32 sh[TOK_WAVES];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u64 tile0 = u64(blockIdx.x) * TOK_TILE;
  u32 pos[TOK_ROWS];
#pragma unroll
  for (u32 row = 0; row < TOK_ROWS; row++) {
    const u64 i = tile0 + u64(row) * TOK_THREADS + tid;
    pos[row] = i < n ? idx[i] : 0xFFFFFFFFu;
  }
  u32 mine = 0; // strings of this wave's rows (wave-uniform)
#pragma unroll
  for (u32 row = 0; row < TOK_ROWS; row++) {
    const bool q = pos[row] < len && buf[pos[row]] == '"';
    const u64 m = __ballot(q);
    if (lane == 0) { qbits[u64(blockIdx.x) * 64 + row * TOK_WAVES + wave] = m; }
    mine += u32(popc64(m));
  }
  if (lane == 0) { sh[wave] = mine; }
  __syncthreads();
  if (tid == 0) {
    count[blockIdx.x] = int(sh[0] + sh[1] + sh[2] + sh[3]);
    if (blockIdx.x == 0) { count[tiles] = 0; }
  }
}

__device__ __forceinline__ void strs_decide(strs_ctrl *__restrict__ ctrl, const int *__restrict__ listed_ptr, u32 n, u64 out_cap, u32 *__restrict__ outq, strings_result_dev *__restrict__ res) {
  const u32 listed = u32(*listed_ptr); // string tokens of the list
  const bool ok = ctrl->bad == 0 && listed == ctrl->opens;
  if (ok && ctrl->total > out_cap) {
    res->overflow = 1;
    ctrl->go_stream = 0;
    ctrl->go_old = 0;
    ctrl->n1_old = 0;
    return;
  }
  ctrl->go_stream = ok ? 1u : 0u;
  ctrl->go_old = ok ? 0u : 1u;
  ctrl->n1_old = ok ? 0u : n + 1u;
  res->path = ok ? 1u : 2u;
  if (ok) {
    outq[listed] = u32(ctrl->total); // behind the last record
    res->bytes = ctrl->total;
    res->strings = listed;
  }
}
// (a launch of its own when the string tokens are counted BEHIND k_strs_resolve -- the stand-alone string pass; the tape has counted them before, and
// k_strs_resolve's last tile decides)
__global__ void k_strs_decide(strs_ctrl *__restrict__ ctrl, const int *__restrict__ listed_ptr, u32 n, u64 out_cap, u32 *__restrict__ outq, strings_result_dev *__restrict__ res) {
  strs_decide(ctrl, listed_ptr, n, out_cap, outq, res);
}

// ---- pass 3: the bytes ----------------------------------------------------------------------------------------------------------------------
// Half a chunk's output: at most 2.5 bytes per input byte ("" -> 5), plus the skew that lines the window up with the destination
constexpr u32 STRS_WINDOW_DATA = (CHUNK_BYTES / 4) * 5; // what 32 lanes can produce
constexpr u32 STRS_WINDOW = 16 + STRS_WINDOW_DATA + 16;
constexpr u32 STRS_STAGE_BYTES = STRS_WINDOW + 64; // + slack behind the window: the last lane's trailing dropped bytes land on the byte behind the chunk's output

// window offset of the lane's byte p: one slot per kept byte and closing quote in front of it, four per opening quote
struct window_map {
  u32 lane_off;
  u64 one, open;
  __device__ __forceinline__ u32 at(u32 p) const {
    const u64 below = (u64(1) << p) - 1;
    return lane_off + u32(popc64(one & below)) + 4u * u32(popc64(open & below));
  }
};
// The bytes whose value is not the document's -- escaped b f n r t, and the kept bytes of \\u escapes -- are PATCHED in the window after the
// scatter: the lanes that own such bytes list them -- one entry per run of neighbours: window offset, position in the chunk, length --, at most
// PATCH_PER_LANE each per round, and the wave works the list off 64 entries at a time.  Text without escapes lists nothing; the synthetic
// twitter-like text ~10 entries per chunk (one round, one pass); a string of nothing but \\uXXXX 683 (two rounds).
constexpr u32 PATCH_PER_LANE = 8, PATCH_LIST = 64 * PATCH_PER_LANE;
__device__ __forceinline__ u32 patch_entry(u32 window_offset, u32 chunk_offset, u32 more) { return (window_offset << 14) | (chunk_offset << 2) | more; } // more: bytes of the run behind its first

// the value of a listed byte at document position q (k_strs_write's patch rounds): escaped b f n r t, or a kept byte of an accepted \\u escape
template <class SRC> __device__ __forceinline__ u32 patched_byte(const SRC &src, u32 q) {
  if (src.byte(q - 1u) == u32('\\')) { return simple_escape_value(src.byte(q)); }
  const u32 k = src.byte(q - 2u) == u32('u') ? 2u : (src.byte(q - 3u) == u32('u') ? 3u : 4u); // (hex digits in between: none of them is a 'u')
  return u_escape_byte(src, q, k);
}

// (five waves per SIMD: what 29 KB of LDS per workgroup allow; left alone the allocator takes 105 registers, four waves)
__global__ __launch_bounds__(64 * STRS_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_strs_write(const u8 *__restrict__ buf, u64 len, u32 nseg, u32 allow_replacement,
                                                               const strs_base *__restrict__ base, const strs_ctrl *__restrict__ ctrl, u8 *__restrict__ out,
                                                               u32 *__restrict__ outq) {
  __shared__ __attribute__((aligned(16))) u8 sh_stage[STRS_WAVES][STRS_STAGE_BYTES];
  __shared__ u32 sh_patch[STRS_WAVES][PATCH_LIST];
  (void)allow_replacement;
  if (ctrl->go_stream == 0) { return; }
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 seg = blockIdx.x * STRS_WAVES + wave;
  if (seg >= nseg) { return; }
  u8 *const stage = sh_stage[wave];
  u32 *const plist = sh_patch[wave];
  const u64 seg_start = u64(seg) * SEG_BYTES;
  const plain_doc src{buf, u32(len)};
  const u32 lookback = lookback_issue(buf, seg_start, lane);
  const strs_base sb = base[seg];
  strs_carry wc{0u, sb.in_string, u_tops{0u, 0u, 0u, 0u}};
  u32 out_base = sb.out, ordinal = sb.ordinal;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    const u64 cstart = seg_start + u64(c) * CHUNK_BYTES;
    if (cstart >= len) { break; }
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u32 w[16];
    if (cstart + CHUNK_BYTES <= len) { load_block_full(buf, pos, w); }
    else { load_block(buf, pos, len, w); }
    if (c == 0) {
      (void)segment_carries(seg_start, lane, lookback, wc); // (no segment of a document this kernel writes is ambiguous: k_strs_decide)
    }
    const strs_chunk m = string_chunk(w, wc, lane);
    const u64 kept = m.b.keep & m.in_string;       // data bytes
    const u64 open = m.quote & m.in_string;         // 4 bytes each: the length, written by k_strs_finalize
    const u64 one = kept | andn(m.quote, m.in_string); // one byte each: data, and the 0 a closing quote turns into
    const u64 closing = andn(m.quote, m.in_string);
    const u32 cnt = u32(popc64(one)) + 4u * u32(popc64(open));
    const u32 incl = wave_incl_scan(cnt);
    const u32 total = readlane(incl, 63);
    const u32 nopen = u32(popc64(open));
    const u32 oincl = wave_incl_scan(nopen);
    if (total) {
      // where the strings of this lane begin (offsets in the buffer: the window plays no part)
      {
        const window_map whole{incl - cnt, one, open};
        u32 k = ordinal + (oincl - nopen);
        for (u64 t = open; t; t &= t - 1) { outq[k++] = out_base + whole.at(ctz64(t)); }
      }
      // The window holds what HALF a chunk can produce (32 lanes x 160 bytes: "" is five bytes of output for two of input); a chunk whose output fits -- every
      // chunk of a document that is not made of empty strings -- goes through it in one pass, the others in two, 32 lanes at a time.  (Through session AT the
      // window held a whole chunk's worst case, 10 KiB per wave: 49.5 KB of LDS per workgroup, three workgroups per CU where the registers allow five.)
      const bool halves = total > STRS_WINDOW_DATA; // wave-uniform
      const u32 first_of_upper = readlane(incl - cnt, 32);
#pragma unroll 1
      for (u32 pass = 0; pass < (halves ? 2u : 1u); pass++) {
      // (the masks pass through an empty asm: everything the sweep derives from them -- 128 bit tests, 16 masked dwords -- is invariant in this loop, and hoisted out of
      // it the kernel needs 256 registers)
      u32 one_lo = u32(one), one_hi = u32(one >> 32), open_lo = u32(open), open_hi = u32(open >> 32), closing_lo = u32(closing), closing_hi = u32(closing >> 32);
      asm volatile("" : "+v"(one_lo), "+v"(one_hi), "+v"(open_lo), "+v"(open_hi), "+v"(closing_lo), "+v"(closing_hi));
      const u64 one_p = u64(one_lo) | (u64(one_hi) << 32), open_p = u64(open_lo) | (u64(open_hi) << 32), closing_p = u64(closing_lo) | (u64(closing_hi) << 32);
      const bool active = !halves || (lane >> 5) == pass;
      const u32 pass_first = (halves && pass) ? first_of_upper : 0u;                                  // output of the chunk in front of this pass
      const u32 pass_total = halves ? (pass ? total - first_of_upper : first_of_upper) : total;
      if (pass_total == 0u) { continue; }
      const u32 pass_out = out_base + pass_first;
      const u32 skew = pass_out & 15u; // window offset and destination address agree modulo 16
      const window_map map{skew + (incl - cnt) - pass_first, one_p, open_p}; // (of the pass's lanes)
      // the bytes: one LDS store per input byte, EVERY byte at the lane's running offset -- a byte that is dropped does not advance the offset and is overwritten by
      // the lane's next byte that stays (stores of one lane land in program order); an opening quote lands in its own hole of four.  What is left over is the
      // lane's trailing dropped bytes at the offset where the NEXT lane's share begins: a lane whose share begins with a byte (not with a hole) stores that byte
      // once more behind the sweep (requested from the document in front of the sweep: the line is in L1 / L2).  Rounds 3-6a selected a dump byte per
      // dropped byte instead: 8 vector instructions per byte, 4 here.
      // (measured in round 5 and not kept: dword by dword the way minify compacts -- one v_perm_b32 per dword through an accumulator, OR-merged into a zeroed
      // window, an opening quote as a hole of one dword, chunks with two opening quotes in a dword on this road -- 228 -> 276 us per 256 MiB: its per-lane
      // branches (flush? hole?) and ds_or cost more than 64 branch-free byte stores; scripts/sessions/gpu_r5_x.sh)
      const u32 first_one = one_p ? ctz64(one_p) : 0u;
      u32 first_byte = 0;
      if (active && one_p) { first_byte = ((closing_p >> first_one) & 1u) ? 0u : (pos + first_one < len ? u32(buf[pos + first_one]) : 0x20u); }
      u8 *const at0 = stage + map.lane_off;
      if (active) {
        u8 *at = at0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          // a closing quote leaves as a zero byte: clear it in the dword (bit k of the nibble -> byte k: one multiply spreads the bits)
          const u32 close4 = u32(closing_p >> (4 * j)) & 0xFu;
          const u32 lsb = (close4 * 0x00204081u) & 0x01010101u;
          u32 fill = lsb << 8; // ... x 0xFF as a shift and a subtraction (the compiler folds them back into a 32-bit multiply, four issue slots, unless it loses sight of one)
          asm("" : "+v"(fill));
          const u32 wj = w[j] & ~(fill - lsb);
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const int i = 4 * j + b;
            const u32 is_one = u32(one_p >> i) & 1u, is_open = u32(open_p >> i) & 1u;
            *at = u8(wj >> (8 * b));
            at += is_one + 4u * is_open;
          }
        }
      }
      wave_lds_fence(); // (outside the branch: every lane of the wave passes every fence)
      if (active && one_p) { at0[4u * u32(popc64(open_p & ((u64(1) << first_one) - 1)))] = u8(first_byte); }
      wave_lds_fence();
      // the few bytes whose value is not the input's: escaped b f n r t, and what \\u escapes stand for -- listed by their owners, worked off one per lane
      const u64 k2 = m.k2 & kept, k3 = m.k3 & kept, k4 = m.k4 & kept, rm = m.b.remap & kept;
      // Listed per RUN of such bytes inside a lane (the two or three bytes a \\u escape leaves are neighbours, in the document and in the window): the wave pays
      // for the lane with the most trips, and a lane of three CJK escapes made nine trips -- and a second round -- with one entry per byte (session AS: the patch
      // rounds were 92 of the kernel's 222 us on the twitter-like text).  What a byte becomes is read off the DOCUMENT by whoever works the entry off: a backslash
      // in front of it -- one of b f n r t; else it sits on the 2nd, 3rd or 4th hex digit of an accepted escape, and the nearest 'u' in front says which.
      const u64 todo = active ? (k2 | k3 | k4 | rm) : u64(0);
      u64 starts = todo & ~(todo << 1);
      while (__ballot(starts != 0)) { // wave-uniform
        const u32 mine = min(u32(popc64(starts)), PATCH_PER_LANE);
        const u32 pincl = wave_incl_scan(mine);
        const u32 entries = readlane(pincl, 63);
        u32 slot = pincl - mine;
        for (u32 r = 0; r < mine; r++) {
          const u32 i = ctz64(starts);
          starts &= starts - 1;
          const u32 run = min(u32(ctz64(~(todo >> i))), 3u); // (an escape leaves at most three bytes)
          plist[slot++] = patch_entry(map.at(i), lane * BLOCK_BYTES + i, run - 1u);
        }
        wave_lds_fence();
        for (u32 e = lane; e < entries; e += 64) {
          const u32 entry = plist[e];
          const u32 run = (entry & 3u) + 1u, q = u32(cstart) + ((entry >> 2) & 0xFFFu);
          u8 *const at = stage + (entry >> 14);
          if (q >= 8u && u64(q) + 8u <= len) {
            const escape_window win(buf, q); // covers q - 8 ... q + 7: every byte the run's (at most three) values are functions of
            for (u32 b = 0; b < run; b++) { at[b] = u8(patched_byte(win, q + b)); }
          } else { // (the document's first and last bytes)
            for (u32 b = 0; b < run; b++) { at[b] = u8(patched_byte(src, q + b)); }
          }
        }
        wave_lds_fence();
      }
      wave_lds_fence();
      // the window leaves as 16-byte stores (emit_bytes' write-out); the 4-byte holes of the lengths carry whatever the window held
      u8 *const g0 = out + (u64(pass_out) - skew);
      const u32 end = skew + pass_total;
      const u32 v_first = (skew + 15u) >> 4, v_last = end >> 4;
      if (v_last > v_first) {
#pragma unroll 1
        for (u32 v = v_first + lane; v < v_last; v += 64) {
          *reinterpret_cast<uint4 *>(g0 + 16u * v) = *reinterpret_cast<const uint4 *>(stage + 16u * v);
        }
        if (skew + lane < 16u * v_first) { g0[skew + lane] = stage[skew + lane]; }
        if (16u * v_last + lane < end) { g0[16u * v_last + lane] = stage[16u * v_last + lane]; }
      } else {
#pragma unroll 1
        for (u32 i = skew + lane; i < end; i += 64) { g0[i] = stage[i]; }
      }
      wave_lds_fence();
      }
      out_base += total;
      ordinal += readlane(oincl, 63);
    }
  }
}

// ---- pass 4: per structural -------------------------------------------------------------------------------------------------------------
typedef u32 __attribute__((aligned(1))) u32_any; // gfx950 stores a dword at any byte address
// count: the exclusive prefixes of the tiles' string counts now (k_scan_partials)
__global__ __launch_bounds__(TOK_THREADS) void k_strs_finalize(const u64 *__restrict__ qbits, const int *__restrict__ count, u32 n, const u32 *__restrict__ outq,
                                                              const strs_ctrl *__restrict__ ctrl, u32 *__restrict__ offsets, u8 *__restrict__ out) {
  if (ctrl->go_stream == 0) { return; }
  __shared__ u32 sh_w[64];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u64 tile0 = u64(blockIdx.x) * TOK_TILE;
  if (wave == 0) { // the 64 (row, wave) counts of the tile in row-major order = list order: exclusive prefixes
    const u32 c = u32(popc64(qbits[u64(blockIdx.x) * 64 + lane]));
    sh_w[lane] = wave_incl_scan(c) - c;
  }
  __syncthreads();
  const u32 front = u32(count[blockIdx.x]);
  const u64 below = lanemask_lt(lane);
  u32 k[TOK_ROWS], at[TOK_ROWS], next[TOK_ROWS];
  u64 bits[TOK_ROWS];
#pragma unroll
  for (u32 row = 0; row < TOK_ROWS; row++) {
    const u64 i = tile0 + u64(row) * TOK_THREADS + tid;
    bits[row] = qbits[u64(blockIdx.x) * 64 + row * TOK_WAVES + wave];
    k[row] = front + sh_w[row * TOK_WAVES + wave] + u32(popc64(bits[row] & below)); // string tokens in front of token i
    at[row] = i <= n ? outq[k[row]] : 0u;
  }
#pragma unroll
  for (u32 row = 0; row < TOK_ROWS; row++) {
    const u64 i = tile0 + u64(row) * TOK_THREADS + tid;
    const bool is_string = i < n && ((bits[row] >> lane) & 1ull) != 0;
    next[row] = is_string ? outq[k[row] + 1] : 0u;
  }
#pragma unroll
  for (u32 row = 0; row < TOK_ROWS; row++) {
    const u64 i = tile0 + u64(row) * TOK_THREADS + tid;
    if (i > n) { continue; }
    offsets[i] = at[row]; // CSR: a structural that is no string has an empty record where the next one begins
    if (i < n && ((bits[row] >> lane) & 1ull)) { *reinterpret_cast<u32_any *>(out + at[row]) = next[row] - at[row] - 5u; }
  }
}

} // namespace

// scratch of the stream (carved by sjgpu_strings.hip): see strings_scratch in sjgpu_internal.h; clears `res` (first_bad = none) and the control block
void enqueue_string_stream(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, bool allow_replacement, uint8_t *out, uint64_t out_cap,
                           uint32_t *offsets, strings_result_dev *res, const strings_scratch &w, hipStream_t s, const int *listed) {
  strs_ctrl *ctrl = static_cast<strs_ctrl *>(w.ctrl);
  static_assert(sizeof(strs_ctrl) <= 64, "the control block has 64 bytes");
  static_assert(sizeof(strs_summary) == STRS_SUMMARY_BYTES && sizeof(strs_base) == STRS_BASE_BYTES, "strings_scratch_bytes counts on these");
  const u32 nseg = num_segments(len), n1 = n + 1;
  const u32 a = allow_replacement ? 1u : 0u;
  // the result and the control block in one small launch (rounds 3-4a: four memsets, i.e. four launches of the runtime's fill kernel)
  // (round 6: by k_strs_count's first thread -- a launch of its own, 5 us, only for a document without a segment)
  strs_summary *summ = static_cast<strs_summary *>(w.seg_summary);
  strs_base *base = static_cast<strs_base *>(w.seg_base);
  if (nseg) {
    hipLaunchKernelGGL(k_strs_count, dim3((nseg + STRS_WAVES - 1) / STRS_WAVES), dim3(64 * STRS_WAVES), 0, s, buf, len, nseg, a, summ, res, ctrl, n1);
  } else {
    hipLaunchKernelGGL(k_strs_init, dim3(1), dim3(64), 0, s, res, ctrl, n1);
  }
  const bool own_ordinals = listed == nullptr; // else the caller has counted the string tokens (launch_tape_front) and finishes the records itself
  const u32 tiles = u32((u64(n1) + TOK_TILE - 1) / TOK_TILE);
  int *const count = w.kord;                                                                               // tiles + 1 ints
  u64 *const qbits = reinterpret_cast<u64 *>(w.kord + ((size_t(tiles) + 1 + 63) & ~size_t(63)));           // 64 words per tile: n / 8 bytes (kord has 4 n)
  // (the tape's call knows the number of string tokens already: the last tile of k_strs_resolve decides, no launch for one thread)
  hipLaunchKernelGGL(k_strs_resolve, dim3(nseg ? (nseg + RES_THREADS - 1) / RES_THREADS : 1u), dim3(RES_THREADS), 0, s, summ, nseg, base, ctrl, listed, n, out_cap, w.outq, res);
  if (own_ordinals) {
    hipLaunchKernelGGL(k_strs_tokens, dim3(tiles), dim3(TOK_THREADS), 0, s, buf, len, idx, n, qbits, count, tiles);
    launch_scan_partials(count, tiles + 1, s);
    listed = count + tiles;
    hipLaunchKernelGGL(k_strs_decide, dim3(1), dim3(1), 0, s, ctrl, listed, n, out_cap, w.outq, res);
  }
  if (nseg) {
    hipLaunchKernelGGL(k_strs_write, dim3((nseg + STRS_WAVES - 1) / STRS_WAVES), dim3(64 * STRS_WAVES), 0, s, buf, len, nseg, a, base, ctrl, out, w.outq);
  }
  if (own_ordinals) {
    hipLaunchKernelGGL(k_strs_finalize, dim3(tiles), dim3(TOK_THREADS), 0, s, qbits, count, n, w.outq, ctrl, offsets, out);
  }
}

} // namespace sjgpu
