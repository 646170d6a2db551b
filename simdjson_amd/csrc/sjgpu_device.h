// simdjson_amd/csrc/sjgpu_device.h -- device-side building blocks shared by the split pipeline
// (sjgpu_kernels.hip) and the single-pass pipeline (sjgpu_fused.hip).  Everything here is WAVE-level:
// a wave64 owns one 4 KiB chunk (one 64-byte block per lane) and all cross-block carries are ballots,
// popcounts and DPP row operations; the only LDS use is the per-wave output window.
#ifndef SJGPU_DEVICE_H
#define SJGPU_DEVICE_H

#include "sjgpu.h"
#include "sjgpu_internal.h"
#include "sj_block.h"
#include "sj_xcarry.h"

// occupancy request whose arguments depend on a template parameter (the host compiler of the CPU tier parses attribute arguments it does not know)
#if defined(SJ_EMU)
#define SJ_WAVES_PER_EU(...)
#else
#define SJ_WAVES_PER_EU(...) __attribute__((amdgpu_waves_per_eu(__VA_ARGS__)))
#endif

namespace sjgpu {
namespace {

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 lanemask_lt(u32 lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ u32 readlane(u32 v, int l) { return u32(__builtin_amdgcn_readlane(int(v), l)); }
__device__ __forceinline__ u32 readlane_dyn(u32 v, u32 l) { return u32(__builtin_amdgcn_readlane(int(v), int(l))); }
// v_ffbl_b32 as the hardware has it: the lowest set bit, 0xFFFFFFFF for zero -- no select around it (__ffs defines the zero case
// and costs one), no undefined behaviour the compiler could build on (__builtin_ctz)
__device__ __forceinline__ u32 ffbl_raw(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return x ? u32(__builtin_ctz(x)) : 0xFFFFFFFFu;
#endif
}
__device__ __forceinline__ u32 clz64(u64 x) { return u32(__clzll((long long)x)); }      // x != 0
__device__ __forceinline__ u32 ctz64(u64 x) { return u32(__ffsll((long long)x) - 1); } // x != 0

// Inclusive prefix sum across the 64 lanes with DPP row operations (gfx9 family): Kogge-Stone inside
// each row of 16 (row_shr 1/2/4/8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows
// 2 and 3.  Six VALU adds, no LDS traffic, no ds_bpermute latency.
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false)); // row_shr:1
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false)); // row_shr:2
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false)); // row_shr:4
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false)); // row_shr:8
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false)); // row_bcast:15 -> rows 1,3
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false)); // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) { return readlane(wave_incl_scan(v), 63); }
__device__ __forceinline__ u32 wave_max(u32 v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const u32 other = u32(__shfl_xor(int(v), o));
    v = other > v ? other : v;
  }
  return v;
}

// LDS traffic of ONE wave is processed in program order, so a wave may read back what its own lanes
// wrote without a workgroup barrier; this only stops the compiler from reordering around the hand-off.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// "My LDS writes have been performed": s_waitcnt lgkmcnt(0).  __syncthreads() is a release fence + s_barrier, and on
// gfx950 the barrier itself does not wait for anything; the wait comes from the fence -- but hipcc 7.2 dropped it at
// the loop-top barrier of k_minify_onchip (the ticket written to LDS at the end of an iteration was not waited for
// across the back edge, three waves read the slot before the write had landed: scripts/lab/minify_debug.cpp found it,
// tests/test_build.py::test_every_barrier_waits_for_lds keeps looking).  Stores whose visibility a barrier at a loop
// top has to guarantee are followed by this.
__device__ __forceinline__ void lds_writes_done() {
#ifndef SJGPU_SELFTEST_DROP_LDS_WAIT // scripts/check_barriers.py must find the barriers this leaves unguarded
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
}

// A workgroup barrier for data that lives in LDS only: the caller has waited for its LDS writes (lds_writes_done), nothing global needs ordering.
// __syncthreads() would add a release fence, which waits for every outstanding global store and atomic of the wave to be acknowledged.
__device__ __forceinline__ void workgroup_barrier_lds_only() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
  __syncthreads();
#endif
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
// the same when the whole chunk is known to lie inside the input (wave-uniform fast path: four straight
// 16-byte loads per lane, no per-lane control flow, so the compiler can keep them in flight across compute)
__device__ __forceinline__ void load_block_full(const u8 *__restrict__ buf, u64 pos, u32 (&w)[16]) {
  const uint4 *p = reinterpret_cast<const uint4 *>(buf + pos);
  const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
  w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
}
// bit i set iff byte pos+i is real input
__device__ __forceinline__ u64 valid_mask(u64 pos, u64 len) {
  if (pos + BLOCK_BYTES <= len) { return ~0ull; }
  if (pos >= len) { return 0ull; }
  return (1ull << u32(len - pos)) - 1ull;
}

// ---- a whole chunk, streamed (round 5) ------------------------------------------------------------------------------------------------
// The input is read ONCE, front to back, and what the memory system does with a stream it will never see again decides the ceiling of every kernel here: a
// kernel that only moves stage 1's bytes reaches 5.0-5.1 TB/s with plain loads and 6.0-6.4 TB/s when the loads carry the non-temporal hint (global_load_dwordx4
// ... nt: the line is the first to leave the caches again) -- read-only 6.2 -> 7.0 (profiles/r05_mix_policy.txt, r05_stream_lab.txt; the same hint on the
// STORES costs 8 %).  But the hint only pays when ONE instruction consumes whole lines: load_block_full's pattern -- a lane's own 64 bytes, an instruction = a
// quarter of each of the chunk's 32 lines -- fetches every line four times with it (validate_utf8 181 -> 280 us per GiB).  So the chunk is requested COALESCED
// (lane i: 16 bytes at chunk + 1024 j + 16 i, an instruction = eight whole lines) and the block-per-lane order the scan needs is restored through 4 KiB of LDS
// per wave: four 16-byte stores, four 16-byte loads, the quarters of a block rotated by (block / 4) mod 4 so that neither side meets a bank conflict.
// xbuf: CHUNK_BYTES of LDS owned by this wave (16-byte aligned); the chunk lies wholly inside the input.  SJGPU_STREAM_LOADS=0 at compile time: plain loads.
#ifndef SJGPU_STREAM_LOADS
#define SJGPU_STREAM_LOADS 1
#endif
__device__ __forceinline__ u32 xbuf_slot(u32 block, u32 quarter) { return block * 4u + ((quarter + (block >> 2)) & 3u); } // 16-byte slots
// one 16-byte load of a stream read once (the instruction must cover whole lines: consecutive lanes, consecutive 16 bytes)
__device__ __forceinline__ uint4 load16_stream(const uint4 *p) {
#if defined(__HIP_DEVICE_COMPILE__) && SJGPU_STREAM_LOADS
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
// ... and one 16-byte store of a stream that is read once, by a later kernel (the masks: profiles/r05_split_lab.txt)
__device__ __forceinline__ void store16_stream(uint4 *p, const uint4 &v) {
#if defined(__HIP_DEVICE_COMPILE__) && SJGPU_STREAM_LOADS
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  const v4u x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<v4u *>(p));
#else
  *p = v;
#endif
}
struct chunk_request { uint4 r[4]; };
// the four loads of a chunk (issue early, consume with load_chunk_finish: the latency lies between the two)
__device__ __forceinline__ chunk_request load_chunk_issue(const u8 *__restrict__ buf, u64 cstart, u32 lane) {
  chunk_request q;
  const uint4 *src = reinterpret_cast<const uint4 *>(buf + cstart) + lane;
#pragma unroll
  for (u32 j = 0; j < 4; j++) {
    q.r[j] = load16_stream(src + 64u * j);
  }
  return q;
}
__device__ __forceinline__ void load_chunk_finish(const chunk_request &q, u32 lane, uint4 *__restrict__ xbuf, u32 (&w)[16]) {
#pragma unroll
  for (u32 j = 0; j < 4; j++) { // piece 64 j + lane = quarter lane % 4 of block 16 j + lane / 4
    xbuf[xbuf_slot(16u * j + (lane >> 2), lane & 3u)] = q.r[j];
  }
  wave_lds_fence();
#pragma unroll
  for (u32 k = 0; k < 4; k++) {
    const uint4 v = xbuf[xbuf_slot(lane, k)];
    w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
  }
  wave_lds_fence(); // (the next chunk's stores come behind these loads in the wave's LDS queue: in order)
}
__device__ __forceinline__ void load_chunk_stream(const u8 *__restrict__ buf, u64 cstart, u32 lane, uint4 *__restrict__ xbuf, u32 (&w)[16]) {
  load_chunk_finish(load_chunk_issue(buf, cstart, lane), lane, xbuf, w);
}

// ---- carries ----------------------------------------------------------------------------------------
struct wave_carry {
  u32 e;    // first byte of the next block is escaped
  u32 s;    // inside a string (relative or absolute, caller's choice)
  u32 p;    // previous byte is a non-quote scalar
};

// parity of the maximal backslash run ending at byte end-1 (0 if end == 0 or no run): a walk as long as the run.  Only k_docs (one
// workgroup per document of at most 64 KiB, sjgpu_small.hip) still walks; every tile kernel starts its spans from 64 bytes of look-back
// and carries what those leave open through its summaries (span_carry_assume below, sj_xcarry.h).  `end` is a multiple of 64.
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
// (UTF-8 needs no carry: blocks are validated one by one from the bytes in memory, see utf8_queue below)
// Split in two so that the ONE byte load it normally needs can be issued before the segment's first chunk is
// requested and consumed after: the two HBM latencies overlap instead of adding up.
__device__ __forceinline__ u32 lookback_issue(const u8 *__restrict__ buf, u64 start, u32 lane) {
  return (start > lane) ? u32(buf[start - 1 - lane]) : 0x20u; // lane i holds byte start-1-i (0x20 in front of the input)
}
// parity of the backslash run ending at byte start-1-skip, from m = ballot(lane's look-back byte is a backslash)
__device__ __forceinline__ u32 run_parity_from_mask(const u8 *__restrict__ buf, u64 start, u32 lane, u64 m, u32 skip) {
  const u64 inv = ~(m >> skip) & (~0ull >> skip); // bit i clear <=> byte start-1-skip-i is a backslash
  if (inv) { return ctz64(inv) & 1u; }
  // every byte we hold is a backslash (so start >= 64): keep walking from byte start-65
  return ((64u - skip) + backslash_run_parity(buf, start - 64, lane)) & 1u;
}
__device__ __forceinline__ wave_carry segment_carry_from(const u8 *__restrict__ buf, u64 start, u32 lane, u32 byte) {
  wave_carry c{0u, 0u, 0u};
  if (start == 0) { return c; }
  const u32 b1 = readlane(byte, 0);
  const u64 m = __ballot(byte == 0x5Cu);
  c.e = run_parity_from_mask(buf, start, lane, m, 0);
  if (b1 == 0x22u) {
    c.p = run_parity_from_mask(buf, start, lane, m, 1);
  } else {
    const bool ws = b1 == 0x20u || b1 == 0x09u || b1 == 0x0Au || b1 == 0x0Du;
    const u32 cur = b1 | 0x20u;
    const bool op = b1 < 0x80u && (cur == 0x2Cu || cur == 0x3Au || cur == 0x7Bu || cur == 0x7Du);
    c.p = (ws || op) ? 0u : 1u;
  }
  return c;
}

// ---- spans that ASSUME (sj_xcarry.h): the carry-in of a span without table and without walk ----------------------------------
// Every tile kernel (split and single-pass pipelines) derives a span's carries from the 64 bytes in front of it and nothing else.
// Two look-backs leave a question open -- 64 backslashes (kind B: is my first byte escaped?), a quote behind 63 backslashes (kind C:
// is that quote escaped?); the span then assumes "no", scans, and publishes an x word with its summary from which the resolve
// step (k_resolve_groups / k_resolve_segments, the look-back of the single-pass kernels) learns whether the assumption held and
// what to repair if not: the other in-string hypothesis, one candidate bit.  All of it is wave-uniform (scalar registers).
// (one word, so that it costs the kernels one scalar register across their chunk loops: they have none to spare)
struct span_x {
  u32 bits; // [1:0] kind (SPAN_EXACT / SPAN_B / SPAN_C); [2] kind B: everything scanned so far is backslashes; [3] / [4]: the span's last 64 bytes
            // ask kind B / C of its successor (span_note_tail behind the last chunk); [31:8] kind B: length L of the leading run once it has ended
  __device__ __forceinline__ u32 kind() const { return bits & 3u; }
  __device__ __forceinline__ u32 lead_open() const { return (bits >> 2) & 1u; }
  __device__ __forceinline__ u32 next_b() const { return (bits >> 3) & 1u; }
  __device__ __forceinline__ u32 next_c() const { return (bits >> 4) & 1u; }
  __device__ __forceinline__ u32 L() const { return bits >> 8; }
};
constexpr u32 SX_LEAD_OPEN = 4u, SX_NEXT_B = 8u, SX_NEXT_C = 16u;
__device__ __forceinline__ bool byte_is_scalar(u32 b) {
  const bool ws = b == 0x20u || b == 0x09u || b == 0x0Au || b == 0x0Du;
  const u32 cur = b | 0x20u;
  const bool op = b < 0x80u && (cur == 0x2Cu || cur == 0x3Au || cur == 0x7Bu || cur == 0x7Du);
  return !(ws || op);
}
// byte = lookback_issue(buf, start, lane); start is 0 or a multiple of the chunk size
__device__ __forceinline__ wave_carry span_carry_assume(u64 start, u32 lane, u32 byte, span_x &sx) {
  wave_carry c{0u, 0u, 0u};
  sx.bits = SPAN_EXACT;
  if (start == 0) { return c; }
  const u32 b1 = readlane(byte, 0);
  const u64 m = __ballot(byte == 0x5Cu);
  if (m == ~0ull) { // kind B: assume the run in front is even
    sx.bits = SPAN_B | SX_LEAD_OPEN;
    c.p = 1u; // a backslash is a scalar
    return c;
  }
  c.e = ctz64(~m) & 1u;
  if (b1 == 0x22u) {
    const u64 inv = ~(m >> 1) & (~0ull >> 1); // bit i clear <=> byte start-2-i is a backslash
    if (inv) { c.p = ctz64(inv) & 1u; }        // an escaped quote is a non-quote scalar
    else { sx.bits = SPAN_C; }                 // assume the quote is real: c.p = 0
  } else {
    c.p = byte_is_scalar(b1) ? 1u : 0u;
  }
  return c;
}
// per chunk, after its bytes are in w: the leading run of a kind-B span.  Costs nothing but a scalar test on the chunks of ordinary spans.
__device__ __forceinline__ void span_note_chunk(span_x &sx, const u32 (&w)[16], u32 chunk_off, u32 lane) {
  if ((sx.bits & SX_LEAD_OPEN) == 0u) { return; } // wave-uniform
  u32 acc = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) { acc |= w[j] ^ 0x5C5C5C5Cu; }
  const u64 other = __ballot(acc != 0u);
  if (other) {
    const u32 fl = ctz64(other);
    u32 pos = 0;
#pragma unroll
    for (int j = 15; j >= 0; j--) { // the lowest dword that holds something else wins
      const u32 x = w[j] ^ 0x5C5C5C5Cu;
      if (x) { pos = 4u * u32(j) + (u32(__ffs(int(x)) - 1) >> 3); }
    }
    sx.bits = (sx.bits & ~SX_LEAD_OPEN) | ((chunk_off + fl * BLOCK_BYTES + readlane_dyn(pos, fl)) << 8);
  }
}
// behind the scan of the span's LAST chunk: what its last 64 bytes are, read off the masks of lane 63 (the scan has them: two v_readlane where rounds
// 4-5a folded sixteen dwords on every lane) -- 64 backslashes ask kind B of the successor, 63 backslashes and a quote kind C
__device__ __forceinline__ void span_note_tail(span_x &sx, u64 backslash, u64 quote_raw) {
  const u32 blo = readlane(u32(backslash), 63);
  if (blo == ~0u) { // wave-uniform, rare
    const u32 bhi = readlane(u32(backslash >> 32), 63);
    if (bhi == ~0u) { sx.bits |= SX_NEXT_B; }
    else if (bhi == 0x7FFFFFFFu && (readlane(u32(quote_raw >> 32), 63) >> 31) != 0u) { sx.bits |= SX_NEXT_C; }
  }
}
// behind the span's last chunk: its x word.  span_bytes = the span's nominal size; scalars: the scan tracks the previous-scalar bit
// (stage 1; minify neither has nor needs kind C and the patched bit); resolved / derived: k_stage1_summarize's pinned segments.
__device__ __forceinline__ u32 span_finish(const span_x &sx, const u8 *__restrict__ buf, u64 start, u32 span_bytes, u64 len, const wave_carry &wc,
                                           bool scalars, bool resolved = false, u32 derived = 0) {
  if ((sx.bits & (3u | SX_NEXT_B | SX_NEXT_C)) == 0u) { return 0u; } // wave-uniform; nearly every span of ordinary input
  span_facts f;
  f.kind = sx.kind();
  f.bytes = span_bytes;
  f.lead_open = sx.lead_open();
  f.L = sx.L();
  f.quote_at_L = 0u; f.scalar_behind = 0u; f.scalar_first = 0u;
  f.next_b = sx.next_b();
  f.next_c = scalars ? sx.next_c() : 0u;
  f.e_end = wc.e;
  f.p_end = scalars ? wc.p : 0u;
  f.resolved = resolved ? 1u : 0u;
  f.derived = derived;
  if (f.kind == SPAN_B && !f.lead_open) {
    const u64 at = start + f.L; // bytes at or beyond len read as spaces
    f.quote_at_L = (at < len && buf[at] == 0x22u) ? 1u : 0u;
    if (scalars && f.quote_at_L && f.L + 1u < span_bytes) { f.scalar_behind = (at + 1 < len && byte_is_scalar(buf[at + 1])) ? 1u : 0u; }
  } else if (f.kind == SPAN_C && scalars) {
    f.scalar_first = (start < len && byte_is_scalar(buf[start])) ? 1u : 0u;
  } else if (f.kind == SPAN_C) {
    f.kind = SPAN_EXACT;
  }
#ifdef SJGPU_SELFTEST_NO_XW // tests/test_kernels_emu.py: with a part of the x words dropped (1 all, 2 the patched bit, 3 dep, 4 F) the adversarial documents must FAIL
  return span_xword(f) & ~(SJGPU_SELFTEST_NO_XW == 1 ? ~0u : (SJGPU_SELFTEST_NO_XW == 2 ? (0xFu << XW_D_SHIFT) : (SJGPU_SELFTEST_NO_XW == 3 ? XW_DEP : XW_F)));
#endif
  return span_xword(f);
}
// x = 1 and the effective hypothesis is the one the patch counts under: toggle the candidate bit of span-relative byte P.
// st[c] = the lane's structural bits of chunk c of the span (final: strings already masked out -- which is why d says when)
template <u32 NCH>
__device__ __forceinline__ void span_patch(u64 (&st)[NCH], u32 xw, u32 x, u32 se, u32 lane) {
  if (x && xw_d(xw, se) != 0) { // wave-uniform
    const u32 P = xw_patch_pos(xw);
    const u32 pc = P / CHUNK_BYTES, pl = (P / BLOCK_BYTES) & 63u;
#pragma unroll
    for (u32 c = 0; c < NCH; c++) {
      if (c == pc && lane == pl) { st[c] ^= 1ull << (P & 63u); }
    }
  }
}
// the count a summary promises under the state in front of it
__device__ __forceinline__ u32 xs_count(u32 c_out, u32 c_in, const xs_step &t) { return (t.se ? c_in : c_out) + u32(t.dcount); }

// ---- 64 summaries folded by one wave (k_resolve_groups: the segments of a group; the look-back of the single-pass kernels: the
// aggregates of 64 tiles) ------------------------------------------------------------------------------------------------------------
// x in front of every lane's summary as base ^ (free & X), X = x in front of the first one: the nearest summary in front of me whose
// successor's x does not depend on its own fixes the chain, the c bits from there on toggle it.  `before` = the lanes in front of the
// place asked about; REVERSED: lane 0 holds the LAST summary of the sequence (the look-back's windows), else the first.
struct lane_x {
  u32 base, free;
};
template <bool REVERSED>
__device__ __forceinline__ lane_x wave_x_chain(u32 xw, u64 before) {
  const u64 cm = __ballot((xw & XW_C) != 0u), dm = __ballot((xw & XW_DEP) != 0u);
  const u64 nd = ~dm & before;
  u64 range = before;
  if (nd) { range = REVERSED ? (before & ((2ull << ctz64(nd)) - 1ull)) : (before & ~((1ull << (63u - clz64(nd))) - 1ull)); }
  lane_x r;
  r.base = u32(popc64(cm & range)) & 1u;
  r.free = nd ? 0u : 1u;
  return r;
}
// The fold: every lane hands in one summary (lanes without one: the identity, {0, 0, 0, XW_IDENTITY}) and its two error bits (by
// effective hypothesis); out comes the summary of the sequence -- as a function of the four states in front of it, then compact
// (sj_xcarry.h: xs_compact) -- and the error bits under (s = 0, x = 0) and (s = 1, x = 0).
template <bool REVERSED>
__device__ __forceinline__ xs_sum wave_fold(const xs_sum &v, u32 lane, u32 e_out = 0, u32 e_in = 0, u32 *err_out = nullptr, u32 *err_in = nullptr) {
  const u64 before = REVERSED ? ~((2ull << lane) - 1ull) : lanemask_lt(lane);
  const u32 F = (v.xw >> 2) & 1u;
  const lane_x mine = wave_x_chain<REVERSED>(v.xw, before), after = wave_x_chain<REVERSED>(v.xw, ~0ull);
  u32 cnt[4], par[2], xo[2];
#pragma unroll
  for (u32 X = 0; X < 2; X++) {
    const u32 xin = mine.base ^ (mine.free & X);
    const u32 f = xin & F;
    const u64 qm = __ballot(((v.q & 1u) ^ f) != 0u);
    const u32 flipped = u32(popc64(qm & before)) & 1u; // in-string state in front of my summary, relative to the start of the sequence
    par[X] = u32(popc64(qm)) & 1u;
    xo[X] = after.base ^ (after.free & X);
#pragma unroll
    for (u32 S = 0; S < 2; S++) {
      const u32 se = S ^ flipped ^ f;
      cnt[S | (X << 1)] = wave_sum((se ? v.c_in : v.c_out) + u32(xin ? xw_d(v.xw, se) : 0));
      if (X == 0 && err_out) { // (the error bits of x = 1 are those of x = 0 under the other hypothesis: sj_xcarry.h)
        const u32 e = __ballot((se ? e_in : e_out) != 0u) ? 1u : 0u;
        if (S == 0) { *err_out = e; } else { *err_in = e; }
      }
    }
  }
  const u32 Fg = par[0] ^ par[1];
  xs_sum r;
  r.q = par[0];
  r.c_out = cnt[0];
  r.c_in = cnt[1];
  r.xw = xo[0] | ((xo[0] ^ xo[1]) << 1) | (Fg << 2) |
         xw_enc_d(int(Fg ? cnt[3] : cnt[2]) - int(cnt[0]), int(Fg ? cnt[2] : cnt[3]) - int(cnt[1])); // x = 1: effective hypothesis 0 is reached from S = Fg
  return r;
}

// ---- one chunk (64 blocks) through the scanner -----------------------------------------------------------
struct chunk_masks {
  u64 cand;        // structural candidates (strings ignored)
  u64 string_tail; // in_string ^ quote
  u64 in_string;   // includes opening quotes, excludes closing quotes
  u64 ws;          // whitespace bytes
  u64 ctrl;        // bytes <= 0x1F
  u64 backslash;   // the two raw classes span_note_tail asks of a span's last chunk
  u64 quote_raw;
};

// =====================================================================================================
// UTF-8: sparse and deferred.  Well-formedness of a 64-byte block is a function of its bytes and the three bytes in
// front of it, all of which sit in memory -- so blocks need not be validated in stream order, nor by the lane that
// scans them.  The scan only NOTES which blocks need a look: those that hold a non-ASCII byte, and those that
// follow a block whose last three bytes hold one (an open sequence ending in ASCII must be caught, too).  Their
// indexes queue up in a small per-wave LDS list; whenever 64 have gathered, the wave validates them with ALL lanes
// busy (utf8_drain: one block per lane, re-read through L2, 32 bytes at a time to keep the register footprint small).
// Mostly-ASCII text (one or two such blocks per 4 KiB chunk: NDJSON, twitter.json) pays ~10 VALU per chunk instead
// of the ~124 an in-line check of every chunk that holds any non-ASCII byte costs; pure ASCII pays one compare.
// Decides exactly what the reference's lookup algorithm decides (utf8_lookup4_algorithm.h:16-202), incl. the
// "sequence open at the end of the input" rule (:164-171).
// =====================================================================================================
// Dense non-ASCII text (CJK, Cyrillic, emoji-heavy: most blocks of a chunk noted) takes the other road: validating 64
// queued blocks costs a second transposition of each (~300 VALU per 64 blocks), validating a chunk in line from the bit
// planes the scan already holds ~124 -- so a chunk that notes more than UTF8_DENSE_FROM blocks is checked on the spot
// (utf8_dense_chunk) and queues nothing.
constexpr u32 UTF8Q_SLOTS = 128; // < 64 left over + <= 64 noted by one chunk
constexpr u32 UTF8_DENSE_FROM = 24;
struct utf8_queue {
  u32 *slots;  // LDS, UTF8Q_SLOTS words owned by this wave
  u32 count;   // wave-uniform
  u32 pending; // wave-uniform: the last three bytes in front of the next block hold a non-ASCII byte
  u32 error;   // wave-uniform, sticky
  const u8 *buf = nullptr; // the input, for the in-line check's look-back and end-of-input rule (null: always queue)
  u64 len = 0;
  u32 more = 0;            // the input continues behind len
  u32 dense_from = UTF8_DENSE_FROM; // chunks that note more blocks than this are checked in line
};
// state at the start of a span, from the look-back bytes (lane i holds byte start-1-i)
__device__ __forceinline__ u32 utf8_pending_from(u32 lookback_byte, u32 lane) {
  return (__ballot(lane < 3u && lookback_byte >= 0x80u) != 0) ? 1u : 0u;
}
// a chunk with many non-ASCII blocks, checked in line from its bit planes (what k_validate_utf8 does for every chunk)
template <class Q>
__device__ __forceinline__ void utf8_dense_chunk(Q &uq, const planes &P, u32 block0, u32 lane) {
  const utf8_leads L = utf8_classify(P);
  const u32 co = utf8_carry_out(L);
  u32 ci = __shfl_up(co, 1);
  u32 c0 = 0; // what the three bytes in front of the chunk demand: one value for the wave, computed on the scalar unit (as per-lane
              // code behind `lane == 0` its dozen compares were what pushed k_stage1_summarize into 20 B of scratch)
  if (block0) {
    const u32 b0 = u32(__builtin_amdgcn_readfirstlane(int(block0))); // wave-uniform by construction: lets the load be a scalar one
    const u32 pw = u32(__builtin_amdgcn_readfirstlane(int(*reinterpret_cast<const u32 *>(uq.buf + u64(b0) * BLOCK_BYTES - 4))));
    c0 = utf8_carry_from_bytes((pw >> 8) & 0xFFu, (pw >> 16) & 0xFFu, pw >> 24);
  }
  if (lane == 0) { ci = c0; }
  bool bad = utf8_errors(P, L, ci) != 0;
  // the input ends exactly with this chunk: a sequence still open there is an error (utf8_lookup4_algorithm.h:164-171);
  // an end inside the chunk is followed by 0x20 padding, which the position-wise test already rejects behind a lead
  if (!uq.more && (u64(block0) + 64u) * BLOCK_BYTES == uq.len && lane == 63 && (co & UTF8_CARRY_OPEN)) { bad = true; }
  if (__ballot(bad)) { uq.error = 1u; }
}
// one chunk: P = the lane's bit planes (plane 7 = non-ASCII positions), w15 = its last dword, block0 = index of lane 0's block
__device__ __forceinline__ void utf8_note_chunk(utf8_queue &uq, const planes &P, u32 w15, u32 block0, u32 lane) {
  const u64 m = __ballot(P.b[7] != 0);
  if (m | u64(uq.pending)) { // wave-uniform; ASCII chunks stop here
    const u64 t = __ballot((w15 & 0x80808000u) != 0); // bytes 61..63 of the block
    const u64 need = m | (t << 1) | u64(uq.pending);
    uq.pending = u32(t >> 63);
    if (uq.buf && u32(popc64(need)) > uq.dense_from) { // wave-uniform
      utf8_dense_chunk(uq, P, block0, lane);
      return;
    }
    const u32 k = __builtin_amdgcn_mbcnt_hi(u32(need >> 32), __builtin_amdgcn_mbcnt_lo(u32(need), 0u));
    if ((need >> lane) & 1ull) { uq.slots[uq.count + k] = block0 + lane; }
    uq.count += u32(popc64(need));
  }
}
// 32 bytes at p (a multiple of 32) as 8 dwords; bytes at or beyond len read as 0x20, nothing past len is touched
__device__ __forceinline__ void load_half(const u8 *__restrict__ buf, u64 p, u64 len, u32 (&w)[8]) {
  if (p + 32 <= len) {
    const uint4 *q = reinterpret_cast<const uint4 *>(buf + p);
    const uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
    const u32 rem = p < len ? u32(len - p) : 0u; // 0..31 real bytes
#pragma unroll 1
    for (u32 j = 0; j < 8; j++) {
      u32 v = 0x20202020u;
      for (u32 k = 0; k < 4; k++) {
        if (4 * j + k < rem) { v = (v & ~(0xFFu << (8 * k))) | (u32(buf[p + 4 * j + k]) << (8 * k)); }
      }
      w[j] = v;
    }
  }
}
// true iff block `block` (64 bytes at block * 64) is ill-formed given the three bytes in front of it; `more`: the
// input continues behind len (no end-of-input rule)
__device__ __forceinline__ bool utf8_check_block(const u8 *__restrict__ buf, u64 len, bool more, u32 block) {
  const u64 pos = u64(block) * BLOCK_BYTES;
  if (pos >= len) { return false; } // nothing but padding (noted because the block in front ends in a non-ASCII byte)
  u32 prev = 0x20202020u; // the dword in front of the half being checked
  if (pos >= 4) { prev = *reinterpret_cast<const u32 *>(buf + pos - 4); }
  u32 bad = 0;
#pragma unroll 1
  for (u32 h = 0; h < 2; h++) {
    const u64 p = pos + 32u * h;
    u32 w[8], q[8];
    load_half(buf, p, len, w);
    s2p32(w, q);
    const utf8_leads_t<u32> L = utf8_classify_planes<u32>(q);
    bad |= utf8_errors_planes<u32>(q, L, utf8_carry_from_bytes((prev >> 8) & 0xFFu, (prev >> 16) & 0xFFu, prev >> 24));
    // a sequence still open where the input ends (when len is not a multiple of 32 the padding has flagged it)
    if (!more && p + 32 >= len && (utf8_carry_out(L) & UTF8_CARRY_OPEN)) { bad |= 1u; }
    prev = w[7];
  }
  return bad != 0;
}
// validate the n NEWEST queued blocks (n <= 64 and n <= count), one per lane; the order of validation is immaterial, and
// taking them off the end means nothing has to move however long the list has grown
__device__ __forceinline__ void utf8_drain(utf8_queue &uq, const u8 *__restrict__ buf, u64 len, bool more, u32 lane, u32 n) {
  wave_lds_fence();
  bool bad = false;
  if (lane < n) { bad = utf8_check_block(buf, len, more, uq.slots[uq.count - n + lane]); }
  if (__ballot(bad)) { uq.error = 1u; }
  uq.count -= n;
}
// after every chunk: keep the list below 64 entries
__device__ __forceinline__ void utf8_drain_if_full(utf8_queue &uq, const u8 *__restrict__ buf, u64 len, bool more, u32 lane) {
  if (uq.count >= 64u) { utf8_drain(uq, buf, len, more, lane, 64u); }
}
// Kernels whose LDS is tight keep only the < 64 left-over entries between scans (`park`, 64 words per wave) and let
// the list grow in memory that is idle while the wave scans (its output window): utf8_resume before a span's first
// chunk, utf8_settle behind its last one (where few registers are live), no draining in between -- `room` must
// hold 63 + 64 entries per chunk of the span.
__device__ __forceinline__ void utf8_resume(utf8_queue &uq, u32 *room, const u32 *park, u32 lane) {
  if (lane < uq.count) { room[lane] = park[lane]; }
  uq.slots = room;
}
__device__ __forceinline__ void utf8_settle(utf8_queue &uq, u32 *park, const u8 *__restrict__ buf, u64 len, bool more, u32 lane) {
  while (uq.count >= 64u) { utf8_drain(uq, buf, len, more, lane, 64u); }
  wave_lds_fence();
  if (lane < uq.count) { park[lane] = uq.slots[lane]; }
  wave_lds_fence();
  uq.slots = park;
}
// at the end of the wave's work
__device__ __forceinline__ void utf8_drain_rest(utf8_queue &uq, const u8 *__restrict__ buf, u64 len, bool more, u32 lane) {
  utf8_drain_if_full(uq, buf, len, more, lane);
  if (uq.count) { utf8_drain(uq, buf, len, more, lane, uq.count); }
}

// ---- the same list with the BLOCKS parked in LDS (k_stage1_summarize, round 4) -----------------------------------------------------
// utf8_queue keeps the indexes of the noted blocks and fetches their bytes again when 64 have gathered -- by then ~130 KiB of
// other waves' tiles have gone through the 4 MiB L2 of the XCD, the lines are gone, and every noted block costs a 128-byte line
// of HBM traffic (amazon NDJSON: 11 % of the blocks, 0.23 GB per GiB on top of 1.07: profiles/r03_pmc_summary.txt).  The lane that
// notes a block HOLDS its 64 bytes; here it parks them (with the dword in front and the block's number: an 80-byte row) and the
// check reads LDS.  Rows are drained UTF8P_DRAIN_AT at a time (32 of 64 lanes busy); a chunk that notes more than UTF8P_DENSE_FROM blocks is checked in
// line as before, so at most UTF8P_DRAIN_AT - 1 + UTF8P_DENSE_FROM rows are ever parked: 48 rows = 3.75 KiB per wave, which with load_chunk_stream's 4 KiB
// makes 31 KiB per workgroup of k_stage1_summarize -- FIVE workgroups per CU.  (Rounds 4-5a: 48 / 24, 72 rows; with the exchange buffer that was 39 KiB
// and four workgroups: 291 against 277 us per GiB of NDJSON, profiles/r05_stream_ab.txt.  The kernel is bound by what it has in flight, not by the lanes
// a drain leaves idle.  Six workgroups would need 80 VGPRs, which this kernel only reaches by spilling: 355 us.)
constexpr u32 UTF8P_ROW_WORDS = 20; // [0..15] the block, [16] the dword in front of it, [17] its number; 80 bytes: rows stay 16-byte aligned
constexpr u32 UTF8P_DRAIN_AT = 32;
constexpr u32 UTF8P_DENSE_FROM = 16;
constexpr u32 UTF8P_ROWS = UTF8P_DRAIN_AT - 1 + UTF8P_DENSE_FROM + 1;
struct utf8_park {
  u32 *rows;   // LDS, UTF8P_ROWS x UTF8P_ROW_WORDS words owned by this wave
  u32 count;   // wave-uniform
  u32 pending; // wave-uniform: the last three bytes in front of the next block hold a non-ASCII byte
  u32 error;   // wave-uniform, sticky
  u32 prev_tail; // wave-uniform: the last dword in front of the next chunk
  const u8 *buf;
  u64 len;
  u32 more;
  u32 dense_from;
};
// state at the start of a span, from the look-back bytes (lane i holds byte start-1-i)
__device__ __forceinline__ void utf8_park_begin(utf8_park &uq, u32 lookback_byte, u32 lane) {
  uq.pending = utf8_pending_from(lookback_byte, lane);
  uq.prev_tail = readlane(lookback_byte, 3) | (readlane(lookback_byte, 2) << 8) | (readlane(lookback_byte, 1) << 16) | (readlane(lookback_byte, 0) << 24);
}
// true iff the parked block is ill-formed given the dword in front of it (utf8_check_block on a row)
__device__ __forceinline__ bool utf8_check_row(const u32 *row, u64 len, bool more) {
  const u64 pos = u64(row[17]) * BLOCK_BYTES;
  if (pos >= len) { return false; } // nothing but padding (noted because the block in front ends in a non-ASCII byte)
  u32 prev = row[16];
  u32 bad = 0;
#pragma unroll 1
  for (u32 h = 0; h < 2; h++) {
    const uint4 a = *reinterpret_cast<const uint4 *>(row + 8u * h), b = *reinterpret_cast<const uint4 *>(row + 8u * h + 4u);
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 q[8];
    s2p32(w, q);
    const utf8_leads_t<u32> L = utf8_classify_planes<u32>(q);
    bad |= utf8_errors_planes<u32>(q, L, utf8_carry_from_bytes((prev >> 8) & 0xFFu, (prev >> 16) & 0xFFu, prev >> 24));
    if (!more && pos + 32u * h + 32u >= len && (utf8_carry_out(L) & UTF8_CARRY_OPEN)) { bad |= 1u; } // a sequence still open where the input ends
    prev = w[7];
  }
  return bad != 0;
}
// validate the n NEWEST parked rows (n <= 64 and n <= count), one per lane
__device__ __forceinline__ void utf8_park_drain(utf8_park &uq, u32 lane, u32 n) {
  wave_lds_fence();
  bool bad = false;
  if (lane < n) { bad = utf8_check_row(uq.rows + (uq.count - n + lane) * UTF8P_ROW_WORDS, uq.len, uq.more != 0u); }
  if (__ballot(bad)) { uq.error = 1u; }
  uq.count -= n;
  wave_lds_fence();
}
// one chunk: P = the lane's bit planes, w = its 16 dwords (bytes beyond the input already read as spaces), block0 = number of lane 0's block
__device__ __forceinline__ void utf8_park_chunk(utf8_park &uq, const planes &P, const u32 (&w)[16], u32 block0, u32 lane) {
  const u64 m = __ballot(P.b[7] != 0);
  const u32 tail_prev = uq.prev_tail;
  uq.prev_tail = readlane(w[15], 63);
  if (m | u64(uq.pending)) { // wave-uniform; ASCII chunks stop here
    const u64 t = __ballot((w[15] & 0x80808000u) != 0); // bytes 61..63 of the block
    const u64 need = m | (t << 1) | u64(uq.pending);
    uq.pending = u32(t >> 63);
    if (uq.buf && u32(popc64(need)) > uq.dense_from) { // wave-uniform
      utf8_dense_chunk(uq, P, block0, lane);
      return;
    }
    u32 prev = u32(__shfl_up(int(w[15]), 1));
    if (lane == 0) { prev = tail_prev; }
    const u32 k = __builtin_amdgcn_mbcnt_hi(u32(need >> 32), __builtin_amdgcn_mbcnt_lo(u32(need), 0u));
    if ((need >> lane) & 1ull) {
      u32 *row = uq.rows + (uq.count + k) * UTF8P_ROW_WORDS;
      uint4 *r4 = reinterpret_cast<uint4 *>(row);
      r4[0] = make_uint4(w[0], w[1], w[2], w[3]);
      r4[1] = make_uint4(w[4], w[5], w[6], w[7]);
      r4[2] = make_uint4(w[8], w[9], w[10], w[11]);
      r4[3] = make_uint4(w[12], w[13], w[14], w[15]);
      row[16] = prev;
      row[17] = block0 + lane;
    }
    uq.count += u32(popc64(need));
    if (uq.count >= UTF8P_DRAIN_AT) { utf8_park_drain(uq, lane, uq.count < 64u ? uq.count : 64u); } // (never more than 47 parked: one round leaves none)
  }
}

// uq (WANT_UTF8 only): the wave's UTF-8 list; block0 = index of lane 0's block (byte offset of the chunk / 64)
__device__ __forceinline__ void utf8_note(utf8_queue &uq, const planes &P, const u32 (&w)[16], u32 block0, u32 lane) { utf8_note_chunk(uq, P, w[15], block0, lane); }
__device__ __forceinline__ void utf8_note(utf8_park &uq, const planes &P, const u32 (&w)[16], u32 block0, u32 lane) { utf8_park_chunk(uq, P, w, block0, lane); }
template <bool WANT_STRUCTURALS, bool WANT_UTF8, class Q = utf8_queue>
__device__ __forceinline__ chunk_masks scan_chunk(const u32 (&w)[16], wave_carry &wc, u32 lane, Q *uq = nullptr, u32 block0 = 0) {
  const planes P = transpose64(w);
  const u64 lt = lanemask_lt(lane);
  chunk_masks out;
  if (WANT_UTF8) { utf8_note(*uq, P, w, block0, lane); }
  const classes c = classify(P);

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
  // in-string parity: prefix XOR over lanes by ballot + "set bits below my lane" (v_mbcnt, the carry as its start value)
  const u64 parm = __ballot((popc64(q.quote) & 1) != 0);
  const u32 s_in = __builtin_amdgcn_mbcnt_hi(u32(parm >> 32), __builtin_amdgcn_mbcnt_lo(u32(parm), wc.s)) & 1u;
  wc.s ^= u32(popc64(parm)) & 1u;
  // previous-scalar: bit 63 of the left neighbour comes over as bit 31 of one DPP move (wave_shr:1; lane 0 keeps the carry), and the
  // shift by one byte position is two funnel shifts -- five instructions where ballot, per-lane 64-bit shift and select took eight
  u64 follows = 0;
  if (WANT_STRUCTURALS) {
    const u32 nq_lo = u32(q.nonquote_scalar), nq_hi = u32(q.nonquote_scalar >> 32);
    const u32 left_hi = u32(__builtin_amdgcn_update_dpp(int(wc.p << 31), int(nq_hi), 0x138, 0xf, 0xf, false));
    follows = (u64(funnel_shift_right(nq_hi, nq_lo, 31)) << 32) | u64(funnel_shift_right(nq_lo, left_hi, 31));
    wc.p = readlane(nq_hi, 63) >> 31;
  }
  const block_masks m = finish_block_follows(c, q, s_in, follows);
  out.cand = m.cand;
  out.string_tail = m.string_tail;
  out.in_string = m.in_string;
  out.ws = c.ws;
  out.ctrl = c.ctrl;
  out.backslash = c.backslash;
  out.quote_raw = c.quote;

  return out;
}


// =====================================================================================================
// output: bitmap -> ascending u32 offsets, through a per-wave LDS window, leaving as 16-byte stores
// =====================================================================================================
constexpr u32 EMIT_WINDOW = 1536; // offsets per window (multiple of 4); denser chunks take several rounds
// LDS words a window needs: + skew so that LDS slot and destination agree modulo 16 bytes, + a dump slot where
// empty extraction chains park their (ignored) stores
constexpr u32 emit_stage_words(u32 window) { return window + 8; }
constexpr u32 EMIT_STAGE_WORDS = emit_stage_words(EMIT_WINDOW);

// One chunk: lane owns `structural` (64 bits) for the block at byte offset pos32; the wave appends the
// set positions to idx[base...], advancing base.  idx must be 16-byte aligned.
template <u32 WINDOW = EMIT_WINDOW>
__device__ __forceinline__ void emit_indices(u64 structural, u32 pos32, u32 lane, u32 *__restrict__ idx, u64 idx_words,
                                             u32 &base, u32 *__restrict__ stage, bool &overflow) {
  constexpr u32 EMIT_WINDOW = WINDOW; // shadows the default inside this function
  constexpr u32 EMIT_DUMP_SLOT = WINDOW + 4;
  u32 lo = u32(structural), hi = u32(structural >> 32);
  const u32 cnt = u32(__popc(lo) + __popc(hi));
  const u32 incl = wave_incl_scan(cnt);
  const u32 total = readlane(incl, 63);
  if (total == 0) { return; }
  if (u64(base) + total > idx_words) { // caller's array too small: drop, flag (SJGPU_F_IDX_OVERFLOW)
    overflow = true;
    base += total;
    return;
  }
  u32 off = incl - cnt;
  const u32 skew = base & 3u;
#pragma unroll 1
  for (u32 w0 = 0; w0 < total; w0 += EMIT_WINDOW) { // wave-uniform; one round unless the chunk is very dense
    if (total <= EMIT_WINDOW) {
      // Common case, everything fits one window: extract the low and the high half of every lane's mask in the
      // SAME uniform loop (two independent ffs/clear chains per lane, trip count = the wave's largest half
      // popcount).  A chain that has run dry keeps rewriting its last slot with the same value, a chain that
      // was empty from the start writes to a dump slot, so there is no divergent control flow at all.
      const u32 nlo = u32(__popc(lo)), nhi = u32(__popc(hi));
      u32 mx = nlo > nhi ? nlo : nhi;
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x111, 0xf, 0xf, false)));
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x112, 0xf, 0xf, false)));
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x114, 0xf, 0xf, false)));
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x118, 0xf, 0xf, false)));
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x142, 0xa, 0xf, false)));
      mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x143, 0xc, 0xf, false)));
      const u32 trips = readlane(mx, 63);
      // six instructions and one LDS store per chain and trip (see emit_span): a chain that has run dry stores to the dump slot
      u32 a_lo = 4u * (off + skew), a_hi = 4u * (off + skew + nlo); // byte addresses
      const u32 pos_hi = pos32 + 32u;
      char *const stage_b = reinterpret_cast<char *>(stage);
#pragma unroll 1 // (ffbl_raw is inline assembly: convergent to the compiler, which then refuses to unroll a loop of unknown length)
      for (u32 t = 0; t < trips; t++) {
        *reinterpret_cast<u32 *>(stage_b + (lo ? a_lo : 4u * EMIT_DUMP_SLOT)) = pos32 + ffbl_raw(lo);
        *reinterpret_cast<u32 *>(stage_b + (hi ? a_hi : 4u * EMIT_DUMP_SLOT)) = pos_hi + ffbl_raw(hi);
        lo &= lo - 1;
        hi &= hi - 1;
        a_lo += 4u;
        a_hi += 4u;
      }
    } else if (total > 2u * EMIT_WINDOW) {
      // Dense chunk (minified small objects, arrays of digits, nesting: up to one offset per byte).  The per-lane
      // extraction below would run 64 iterations per window with only the window's ~20 source lanes active; here
      // the wave walks over the source blocks whose offsets fall into the window and expands ONE block per step
      // with all 64 lanes: lane j owns bit j, its slot is the block's first slot + the number of set bits below j.
      const u32 lim = w0 + EMIT_WINDOW;
      const u32 rel = skew - w0;
      const u32 first = incl - cnt;                 // my block's first element
      const u32 value0 = pos32 - 63u * lane;         // + 64 * L = (byte offset of block L) + lane
      const u32 mlo_mine = u32(structural), mhi_mine = u32(structural >> 32);
      for (u64 rem = __ballot(cnt != 0 && incl > w0 && first < lim); rem; rem &= rem - 1) { // wave-uniform
        const u32 L = ctz64(rem);
        const u32 mlo = readlane_dyn(mlo_mine, L), mhi = readlane_dyn(mhi_mine, L);
        const u32 e = readlane_dyn(first, L) + __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        const bool mine = ((lane < 32u ? mlo >> lane : mhi >> (lane - 32u)) & 1u) != 0;
        if (mine && e >= w0 && e < lim) { stage[e + rel] = value0 + 64u * L; }
      }
    } else {
    const u32 lim = w0 + EMIT_WINDOW;
    const u32 rel = skew - w0; // stage slot of element e is e + rel (mod 2^32; e >= w0 whenever we store)
    while (lo && off < lim) {
      stage[off + rel] = pos32 + u32(__ffs(int(lo)) - 1);
      lo &= lo - 1;
      off++;
    }
    if (lo == 0) {
      while (hi && off < lim) {
        stage[off + rel] = pos32 + 32u + u32(__ffs(int(hi)) - 1);
        hi &= hi - 1;
        off++;
      }
    }
    }
    wave_lds_fence();
    const u32 here = min(EMIT_WINDOW, total - w0);
    u32 *const g0 = idx + (u64(base) + w0 - skew); // 16-byte aligned
    const u32 end = skew + here;                   // stage slots [skew, end) are live
    const u32 v_first = (skew + 3u) >> 2, v_last = end >> 2; // whole 4-word vectors [v_first, v_last)
    if (v_last > v_first) {
#pragma unroll 1
      for (u32 v = v_first + lane; v < v_last; v += 64) {
        *reinterpret_cast<uint4 *>(g0 + 4u * v) = *reinterpret_cast<const uint4 *>(stage + 4u * v);
      }
      if (skew + lane < 4u * v_first) { g0[skew + lane] = stage[skew + lane]; }
      if (4u * v_last + lane < end) { g0[4u * v_last + lane] = stage[4u * v_last + lane]; }
    } else {
#pragma unroll 1
      for (u32 i = skew + lane; i < end; i += 64) { g0[i] = stage[i]; }
    }
    wave_lds_fence();
  }
  base += total;
}

// A wave's whole SPAN at once: FOUR consecutive chunks whose offsets fit one window together (sparse output: NDJSON,
// pretty-printed text).  One prefix scan over packed counts, ONE extraction loop with eight independent chains per lane
// (two halves x four chunks), one hand-over to the store loop -- instead of four scans, four loops and four hand-overs
// whose fixed latencies dominated the emission of sparse chunks (profiles/r02_pipelined_phase_trace.txt: 1.6 us per
// chunk of ~245 offsets).  Returns false without touching anything when the span does not fit the window (the caller
// then emits chunk by chunk).  m[c] = lane's structural bits of chunk c, pos0 = byte offset of lane 0's block in chunk 0.
template <u32 WINDOW, u32 NCH>
__device__ __forceinline__ bool emit_span(const u64 *m, u32 pos0, u32 lane, u32 *__restrict__ idx, u64 idx_words, u32 &base,
                                          u32 *__restrict__ stage, bool &overflow) {
  static_assert(NCH == 2 || NCH == 4, "two or four chunks");
  constexpr u32 DUMP = WINDOW + 4;
  constexpr u32 NH = 2 * NCH;
  u32 h[NH], n[NH];
#pragma unroll
  for (u32 c = 0; c < NCH; c++) {
    h[2 * c] = u32(m[c]);
    h[2 * c + 1] = u32(m[c] >> 32);
    n[2 * c] = u32(__popc(h[2 * c]));
    n[2 * c + 1] = u32(__popc(h[2 * c + 1]));
  }
  // scans over pairs of 16-bit fields (a chunk holds at most 4096 offsets)
  const u32 p01 = (n[0] + n[1]) | ((n[2] + n[3]) << 16);
  const u32 i01 = wave_incl_scan(p01);
  const u32 t01 = readlane(i01, 63);
  u32 T[4] = {t01 & 0xFFFFu, t01 >> 16, 0u, 0u};
  u32 p23 = 0, i23 = 0;
  if (NCH == 4) {
    p23 = (n[NH - 4] + n[NH - 3]) | ((n[NH - 2] + n[NH - 1]) << 16);
    i23 = wave_incl_scan(p23);
    const u32 t23 = readlane(i23, 63);
    T[2] = t23 & 0xFFFFu;
    T[3] = t23 >> 16;
  }
  const u32 total = T[0] + T[1] + T[2] + T[3];
  if (total > WINDOW) { return false; }
  if (total == 0) { return true; }
  if (u64(base) + total > idx_words) {
    overflow = true;
    base += total;
    return true;
  }
  const u32 skew = base & 3u;
  const u32 e01 = i01 - p01, e23 = i23 - p23; // exclusive, per field
  u32 a[NH];
  a[0] = skew + (e01 & 0xFFFFu);
  a[2] = skew + T[0] + (e01 >> 16);
  if (NCH == 4) {
    a[NH - 4] = skew + T[0] + T[1] + (e23 & 0xFFFFu);
    a[NH - 2] = skew + T[0] + T[1] + T[2] + (e23 >> 16);
  }
  u32 mx = 0;
#pragma unroll
  for (u32 c = 0; c < NCH; c++) {
    a[2 * c + 1] = a[2 * c] + n[2 * c];
    mx = max(mx, max(n[2 * c], n[2 * c + 1]));
  }
  // (an empty chain needs no slot of its own: the loop below sends every store of a chain without bits to the dump slot -- rounds 2-5a also
  // pre-selected it here, sixteen instructions per span for nothing)
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x111, 0xf, 0xf, false)));
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x112, 0xf, 0xf, false)));
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x114, 0xf, 0xf, false)));
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x118, 0xf, 0xf, false)));
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x142, 0xa, 0xf, false)));
  mx = max(mx, u32(__builtin_amdgcn_update_dpp(0, int(mx), 0x143, 0xc, 0xf, false)));
  const u32 trips = readlane(mx, 63);
  const u32 lane_pos = pos0 + lane * BLOCK_BYTES;
  // Six instructions and one LDS store per chain and trip (it was eight: profiles/r04_emit_chains.txt): a chain that has run dry stores
  // (garbage) to the dump slot instead of rewriting its last slot, so the value needs no select and the slot advances unconditionally.
  u32 vbase[NH];
#pragma unroll
  for (u32 k = 0; k < NH; k++) {
    vbase[k] = lane_pos + (k >> 1) * CHUNK_BYTES + (k & 1u) * 32u;
    a[k] *= 4u; // byte addresses from here on: no shift per store
  }
  // two trips per round: the second store of a chain takes its address from the first one's register plus an IMMEDIATE offset (ds_write_b32 ... offset:4;
  // the dump slot has a neighbour for it), so a chain's slot advances once per round -- 44 VALU instructions per trip instead of 48 (round 5)
  char *const stage_c = reinterpret_cast<char *>(stage);
  u32 t = 0;
#pragma unroll 1
  for (; t + 2u <= trips; t += 2u) {
#pragma unroll
    for (u32 k = 0; k < NH; k++) {
      const u32 bits = h[k];
      *reinterpret_cast<u32 *>(stage_c + (bits ? a[k] : 4u * DUMP)) = vbase[k] + ffbl_raw(bits); // (an empty chain's value lands in the dump slot)
      const u32 rest = bits & (bits - 1u);
      *reinterpret_cast<u32 *>(stage_c + (rest ? a[k] : 4u * DUMP) + 4u) = vbase[k] + ffbl_raw(rest);
      h[k] = rest & (rest - 1u);
      a[k] += 8u;
    }
  }
  if (t < trips) {
#pragma unroll
    for (u32 k = 0; k < NH; k++) {
      const u32 bits = h[k];
      *reinterpret_cast<u32 *>(stage_c + (bits ? a[k] : 4u * DUMP)) = vbase[k] + ffbl_raw(bits);
    }
  }
  wave_lds_fence();
  u32 *const g0 = idx + (u64(base) - skew); // 16-byte aligned
  const u32 end = skew + total;            // stage slots [skew, end) are live
  const u32 v_first = (skew + 3u) >> 2, v_last = end >> 2;
  if (v_last > v_first) {
#pragma unroll 1
    for (u32 q = v_first + lane; q < v_last; q += 64) {
      *reinterpret_cast<uint4 *>(g0 + 4u * q) = *reinterpret_cast<const uint4 *>(stage + 4u * q);
    }
    if (skew + lane < 4u * v_first) { g0[skew + lane] = stage[skew + lane]; }
    if (4u * v_last + lane < end) { g0[4u * v_last + lane] = stage[4u * v_last + lane]; }
  } else {
#pragma unroll 1
    for (u32 i = skew + lane; i < end; i += 64) { g0[i] = stage[i]; }
  }
  wave_lds_fence();
  base += total;
  return true;
}
// a span of four chunks: in one piece, else as two pairs, else chunk by chunk (st[c] = lane's structural bits of chunk c,
// zero for chunks beyond the input; span_pos = byte offset of the span).  span_count = the number of offsets in the span,
// which every caller knows from its summary: a dense span (large_random: ~5 000 against a window of 1 280) goes straight
// to the per-chunk path instead of counting itself three times on the way there.
template <u32 WINDOW>
__device__ __forceinline__ void emit_span4_adaptive(const u64 (&st)[4], u32 span_pos, u32 lane, u32 *__restrict__ idx, u64 idx_words, u32 &base,
                                                    u32 *__restrict__ stage, bool &overflow, u32 span_count) {
  if (span_count <= WINDOW && emit_span<WINDOW, 4>(st, span_pos, lane, idx, idx_words, base, stage, overflow)) { return; }
  const bool pairs = span_count <= 2u * WINDOW; // wave-uniform; beyond that at least one pair cannot fit
#pragma unroll
  for (u32 half = 0; half < 2; half++) {
    const u32 pos = span_pos + half * 2u * CHUNK_BYTES;
    if (pairs && emit_span<WINDOW, 2>(st + 2 * half, pos, lane, idx, idx_words, base, stage, overflow)) { continue; }
    emit_indices<WINDOW>(st[2 * half], pos + lane * BLOCK_BYTES, lane, idx, idx_words, base, stage, overflow);
    emit_indices<WINDOW>(st[2 * half + 1], pos + CHUNK_BYTES + lane * BLOCK_BYTES, lane, idx, idx_words, base, stage, overflow);
  }
}

// =====================================================================================================
// output: byte compaction (minify) of one chunk through a per-wave LDS window
// =====================================================================================================
constexpr u32 MINIFY_STAGE_BYTES = CHUNK_BYTES + 32;
constexpr u32 MINIFY_LUT_WORDS = 16;

// v_perm_b32 selector that packs the bytes of a dword whose bit is set in k4 to the low end (0x0C = zero byte)
__device__ __forceinline__ u32 compaction_selector(u32 k4) {
  u32 sel = 0x0C0C0C0Cu, at = 0;
#pragma unroll
  for (u32 b = 0; b < 4; b++) {
    if (k4 & (1u << b)) {
      sel = (sel & ~(0xFFu << (8 * at))) | (b << (8 * at));
      at++;
    }
  }
  return sel;
}
// every wave that calls emit_bytes shares one 16-entry table in LDS (fill it once, then wave_lds_fence / barrier)
__device__ __forceinline__ void init_compaction_lut(u32 *lut, u32 lane) {
  if (lane < MINIFY_LUT_WORDS) { lut[lane] = compaction_selector(lane); }
}

// lane keeps the bytes of its block whose bit is set in `keep`; the wave appends them to dst[base...].
// dst must be 16-byte aligned.  Dword-granular: each input dword is compacted with ONE v_perm_b32 (selector
// from the LDS table), streamed through a 64-bit shift accumulator and OR-merged into the zeroed window, so a
// lane issues <= 17 LDS operations per block instead of one per byte (and neighbouring lanes, whose output
// ranges share a dword, need no ordering).  The stage must be all-zero on entry and is left all-zero.
// REZERO = false: the caller overwrites the window before its next use (k_minify_onchip parks input bytes in it).
template <bool REZERO = true>
__device__ __forceinline__ void emit_bytes(const u32 (&w)[16], u64 keep, u32 lane, u8 *__restrict__ dst, u32 &base,
                                           u8 *__restrict__ stage, const u32 *__restrict__ lut) {
  const u32 cnt = u32(popc64(keep));
  const u32 incl = wave_incl_scan(cnt);
  const u32 total = readlane(incl, 63);
  if (total == 0) { return; }
  const u32 skew = base & 15u; // stage offset and destination address agree modulo 16
  const u32 o = skew + (incl - cnt);
  u32 *const stage_w = reinterpret_cast<u32 *>(stage);
  u32 dw = o >> 2, fill = o & 3u;
  u64 acc = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const u32 k4 = u32(keep >> (4 * j)) & 0xFu;
    const u32 packed = byte_perm(0u, w[j], lut[k4]);
    acc |= u64(packed) << (8u * fill);
    fill += u32(__popc(k4));
    if (fill >= 4u) {
      atomicOr(&stage_w[dw], u32(acc)); // ds_or_b32, result unused
      dw++;
      acc >>= 32;
      fill -= 4u;
    }
  }
  if (fill) { atomicOr(&stage_w[dw], u32(acc)); }
  wave_lds_fence();
  u8 *const g0 = dst + (u64(base) - skew);
  const u32 end = skew + total;
  const u32 v_first = (skew + 15u) >> 4, v_last = end >> 4; // whole 16-byte vectors [v_first, v_last)
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
  if (REZERO) { // what was used: vectors [0, ceil(end/16))
#pragma unroll 1
    for (u32 v = lane; 16u * v < end; v += 64) { *reinterpret_cast<uint4 *>(stage + 16u * v) = make_uint4(0, 0, 0, 0); }
    wave_lds_fence();
  }
  base += total;
}

// zero a fresh per-wave minify window (call once before the first emit_bytes)
__device__ __forceinline__ void clear_minify_stage(u8 *stage, u32 lane) {
  for (u32 v = lane; 16u * v < MINIFY_STAGE_BYTES; v += 64) { *reinterpret_cast<uint4 *>(stage + 16u * v) = make_uint4(0, 0, 0, 0); }
  wave_lds_fence();
}

} // namespace
} // namespace sjgpu
#endif
