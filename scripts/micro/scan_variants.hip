// scripts/micro/scan_variants.hip -- what bounds the stage-1 SCAN (k_stage1_summarize shape) on gfx950?
// Findings of the round-2 run (profiles/r02_scan_variants.txt, in-line UTF-8 at the time): loads + OR + one 8-byte
// store per lane and chunk reach 4.6 TB/s whatever the block -> address map, k_validate_utf8's loads alone 5.9; the scan
// on L2-resident input runs at 5.6-8.5 TB/s: the split pipeline's summarize kernel is bound by its memory path, not by
// its ~330 VALU instructions per chunk; prefetching the next chunk: +0..4 %; persistent waves: -3 %.
// Round-1 PMC (profiles/r01_pmc_stage1.txt): VALU ~50 % busy, HBM at 3.8 of 6.3 TB/s, waves parked on s_waitcnt 61 % of
// their life.  Each variant below removes or changes ONE thing; all read the same 1 GiB buffer with one wave per 16 KiB
// segment unless the name says otherwise.  Diagnostics only (not product code); timing = hipEvents, best of N.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I simdjson_amd/csrc scripts/micro/scan_variants.hip \
//         -Lsimdjson_amd/lib -lsjcorpus -Wl,-rpath,$PWD/simdjson_amd/lib -o scripts/micro/scan_variants.bin
#include "../../simdjson_amd/csrc/sjgpu_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" size_t sjc_amazon_ndjson(void *, size_t, size_t, uint64_t, uint64_t *);
extern "C" size_t sjc_large_random(void *, size_t, size_t, uint64_t, uint64_t *);

namespace sjgpu {
namespace {

// ---- shared tail of every "scan" variant: what summarize does with a chunk's masks (resolved / unresolved) ----
struct seg_acc {
  u32 n_a = 0, n_b = 0;
  u64 ctrl_a = 0, ctrl_b = 0;
  bool resolved = false;
  u32 derived = 0;
  u64 flip = 0;
};
__device__ __forceinline__ void fold_chunk(seg_acc &A, const chunk_masks &m, u32 c, u32 lane, u64 *mask0, u64 *mask1, u64 slot) {
  if (c == 0) {
    const u64 cm = __ballot(m.ctrl != 0);
    if (cm) {
      const u32 lc = ctz64(cm);
      u32 v = 0;
      if (lane == lc) { v = u32(m.in_string >> ctz64(m.ctrl)) & 1u; }
      A.derived = readlane_dyn(v, lc);
      A.resolved = true;
      A.flip = A.derived ? ~0ull : 0ull;
    }
  }
  if (A.resolved) {
    const u64 structural = m.cand & ~(m.string_tail ^ A.flip);
    A.n_a += u32(popc64(structural));
    A.ctrl_a |= m.ctrl & (m.in_string ^ A.flip);
    mask0[slot] = structural;
  } else {
    A.n_a += u32(popc64(m.cand));
    A.n_b += u32(popc64(m.cand & m.string_tail));
    A.ctrl_a |= m.ctrl & m.in_string;
    A.ctrl_b |= m.ctrl & ~m.in_string;
    mask0[slot] = m.cand;
    mask1[slot] = m.string_tail;
  }
}
__device__ __forceinline__ void finish_seg(const seg_acc &A, const wave_carry &wc, u32 lane, seg_summary *summ, u32 seg) {
  const u32 ta = wave_sum(A.n_a), tb = wave_sum(A.n_b);
  const bool any_a = __ballot(A.ctrl_a != 0) != 0, any_b = __ballot(A.ctrl_b != 0) != 0;
  u32 flags = wc.s ? SF_PARITY : 0u;
  seg_summary s;
  s.count_if_out = A.resolved ? ta : ta - tb;
  s.count_if_in = A.resolved ? ta : tb;
  if (any_a) { flags |= SF_CTRL_IF_OUT; }
  if (any_b) { flags |= SF_CTRL_IF_IN; }
  s.flags = flags | (A.resolved ? SF_RESOLVED : 0u);
  s.pad = 0;
  if (lane == 0) { summ[seg] = s; }
}

// segment carry with the escape-table byte requested EARLY (together with the look-back byte), not behind vmcnt(0)
__device__ __forceinline__ wave_carry carry_pre(const u8 *buf, u64 start, u32 lane, u32 byte, u32 esc_val, const u8 *esc) {
  wave_carry c{0u, 0u, 0u};
  if (start == 0) { return c; }
  const u32 b1 = readlane(byte, 0);
  const u64 m = __ballot(byte == 0x5Cu);
  c.e = (esc_val != ESC_PASS) ? (esc_val & 1u) : escape_lookup(esc, start / SEG_BYTES, lane);
  if (b1 == 0x22u) {
    c.p = run_parity_from_mask(buf, start, lane, m, 1, esc);
  } else {
    const bool ws = b1 == 0x20u || b1 == 0x09u || b1 == 0x0Au || b1 == 0x0Du;
    const u32 cur = b1 | 0x20u;
    const bool op = b1 < 0x80u && (cur == 0x2Cu || cur == 0x3Au || cur == 0x7Bu || cur == 0x7Du);
    c.p = (ws || op) ? 0u : 1u;
  }
  return c;
}

// MODE 1: esc byte hoisted.  MODE 2: + next chunk prefetched into a second register set (fully unrolled).
// WRAP != 0: segment index wrapped into a WRAP-segment window (L2-resident input: compute ceiling of the scan).
template <int MODE, u32 WRAP, int THREADS>
__device__ __forceinline__ void scan_body(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0, u64 *__restrict__ mask1,
                                          seg_summary *__restrict__ summ, const u8 *__restrict__ esc, u32 nseg) {
  const u32 lane = threadIdx.x & 63u;
  const u32 seg_out = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (seg_out >= nseg) { return; }
  const u32 seg = WRAP ? (seg_out % WRAP) : seg_out;
  const u64 seg_start = u64(seg) * SEG_BYTES;
  const u32 lookback = lookback_issue(buf, seg_start, lane);
  const u32 esc_val = esc[seg];
  wave_carry wc{0u, 0u, 0u};
  __shared__ u32 uq_slots[THREADS / 64][UTF8Q_SLOTS];
  utf8_queue uq{uq_slots[threadIdx.x >> 6], 0u, 0u, 0u};
  seg_acc A;
  if (MODE == 1) {
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      const u64 pos = seg_start + u64(c) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES;
      u32 w[16];
      load_block_full(buf, pos, w);
      if (c == 0) { wc = carry_pre(buf, seg_start, lane, lookback, esc_val, esc); }
      const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, 0u);
      fold_chunk(A, m, c, lane, mask0, mask1, (u64(seg_out) * SEG_BYTES + u64(c) * CHUNK_BYTES) / BLOCK_BYTES + lane);
      utf8_drain_if_full(uq, buf, len, false, lane);
    }
  } else {
    u32 w[16], wn[16];
    load_block_full(buf, seg_start + u64(lane) * BLOCK_BYTES, w);
    wc = carry_pre(buf, seg_start, lane, lookback, esc_val, esc);
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      if (c + 1 < SEG_CHUNKS) { load_block_full(buf, seg_start + u64(c + 1) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, wn); }
      const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, 0u);
      fold_chunk(A, m, c, lane, mask0, mask1, (u64(seg_out) * SEG_BYTES + u64(c) * CHUNK_BYTES) / BLOCK_BYTES + lane);
      utf8_drain_if_full(uq, buf, len, false, lane);
#pragma unroll
      for (int j = 0; j < 16; j++) { w[j] = wn[j]; }
    }
  }
  utf8_drain_rest(uq, buf, len, false, lane);
  finish_seg(A, wc, lane, summ, seg_out);
}
template <int MODE, u32 WRAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_scan(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0, u64 *__restrict__ mask1,
                                                   seg_summary *__restrict__ summ, const u8 *__restrict__ esc, u32 nseg) {
  scan_body<MODE, WRAP, THREADS>(buf, len, mask0, mask1, summ, esc, nseg);
}
// the same bodies with the register budget forced down (more waves per SIMD, possibly spills)
#define SCAN_OCC(NAME, MODE, W)                                                                                              \
  __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) void NAME(                                     \
      const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0, u64 *__restrict__ mask1, seg_summary *__restrict__ summ, \
      const u8 *__restrict__ esc, u32 nseg) {                                                                                \
    scan_body<MODE, 0, 64>(buf, len, mask0, mask1, summ, esc, nseg);                                                          \
  }
SCAN_OCC(k_scan1_occ6, 1, 6)
SCAN_OCC(k_scan1_occ8, 1, 8)
SCAN_OCC(k_scan2_occ5, 2, 5)
SCAN_OCC(k_scan2_occ6, 2, 6)

// persistent waves: grid-stride over segments, next segment's first chunk (and its look-back bytes) prefetched
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_scan_persistent(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0,
                                                              u64 *__restrict__ mask1, seg_summary *__restrict__ summ,
                                                              const u8 *__restrict__ esc, u32 nseg) {
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (THREADS / 64);
  u32 w[16], wn[16];
  __shared__ u32 uq_slots[THREADS / 64][UTF8Q_SLOTS];
  utf8_queue uq{uq_slots[threadIdx.x >> 6], 0u, 0u, 0u};
  u32 seg = wave;
  if (seg >= nseg) { return; }
  u32 lookback = lookback_issue(buf, u64(seg) * SEG_BYTES, lane);
  u32 esc_val = esc[seg];
  load_block_full(buf, u64(seg) * SEG_BYTES + u64(lane) * BLOCK_BYTES, w);
  for (; seg < nseg; seg += nwaves) {
    const u64 seg_start = u64(seg) * SEG_BYTES;
    wave_carry wc = carry_pre(buf, seg_start, lane, lookback, esc_val, esc);
    seg_acc A;
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      if (c + 1 < SEG_CHUNKS) {
        load_block_full(buf, seg_start + u64(c + 1) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, wn);
      } else if (seg + nwaves < nseg) {
        const u64 nstart = u64(seg + nwaves) * SEG_BYTES;
        lookback = lookback_issue(buf, nstart, lane);
        esc_val = esc[seg + nwaves];
        load_block_full(buf, nstart + u64(lane) * BLOCK_BYTES, wn);
      }
      const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, 0u);
      fold_chunk(A, m, c, lane, mask0, mask1, (seg_start + u64(c) * CHUNK_BYTES) / BLOCK_BYTES + lane);
      utf8_drain_if_full(uq, buf, len, false, lane);
#pragma unroll
      for (int j = 0; j < 16; j++) { w[j] = wn[j]; }
    }
    finish_seg(A, wc, lane, summ, seg);
  }
}

// ---- memory-path only: the same loads (64-byte lane stride), an OR-reduce, the same 8-byte mask store ----
// MAP 0: one wave per 16 KiB segment, chunk by chunk (summarize's map); MAP 1: grid-stride over chunks (validate_utf8's map);
// MAP 2: like 0 but all four chunks' loads issued before any is consumed
template <int MAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_loads(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0, u32 nseg) {
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (THREADS / 64);
  if (MAP == 1) {
    const u64 nchunks = u64(nseg) * SEG_CHUNKS;
    for (u64 ch = wave; ch < nchunks; ch += nwaves) {
      u32 w[16];
      load_block_full(buf, ch * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, w);
      u32 x = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) { x |= w[j]; }
      mask0[ch * 64 + lane] = x;
    }
  } else if (MAP == 0) {
    if (wave >= nseg) { return; }
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      u32 w[16];
      load_block_full(buf, u64(wave) * SEG_BYTES + u64(c) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, w);
      u32 x = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) { x |= w[j]; }
      mask0[(u64(wave) * SEG_CHUNKS + c) * 64 + lane] = x;
    }
  } else {
    if (wave >= nseg) { return; }
    u32 w[SEG_CHUNKS][16];
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) { load_block_full(buf, u64(wave) * SEG_BYTES + u64(c) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, w[c]); }
#pragma unroll
    for (u32 c = 0; c < SEG_CHUNKS; c++) {
      u32 x = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) { x |= w[c][j]; }
      mask0[(u64(wave) * SEG_CHUNKS + c) * 64 + lane] = x;
    }
  }
}

// ---- loads + transposition + classification only (no cross-lane traffic) ----
template <int WHAT> // 0: transpose only (xor of planes), 1: + classify, 2: + utf8 ascii test + escapes/quotes/prefix_xor (per-lane part)
__global__ __launch_bounds__(64) void k_lane_math(const u8 *__restrict__ buf, u64 len, u64 *__restrict__ mask0, u32 nseg) {
  const u32 lane = threadIdx.x & 63u;
  const u32 seg = blockIdx.x;
  for (u32 c = 0; c < SEG_CHUNKS; c++) {
    u32 w[16];
    load_block_full(buf, u64(seg) * SEG_BYTES + u64(c) * CHUNK_BYTES + u64(lane) * BLOCK_BYTES, w);
    const planes P = transpose64(w);
    u64 r;
    if (WHAT == 0) {
      r = P.b[0] ^ P.b[1] ^ P.b[2] ^ P.b[3] ^ P.b[4] ^ P.b[5] ^ P.b[6] ^ P.b[7];
    } else {
      const classes k = classify(P);
      r = k.backslash ^ k.quote ^ k.ws ^ k.op ^ k.ctrl;
      if (WHAT == 2) {
        u64 nxt;
        const u64 escaped = escaped_mask(k.backslash, 0, nxt);
        const quote_scalar q = quotes_and_scalars(k, escaped);
        const block_masks m = finish_block(k, q, 0, 0);
        r = m.cand & ~m.string_tail;
      }
    }
    mask0[(u64(seg) * SEG_CHUNKS + c) * 64 + lane] = r;
  }
}

} // namespace
} // namespace sjgpu

using namespace sjgpu;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  const size_t target = size_t(1) << 30;
  const int reps = 12;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<uint8_t> host(target + (1 << 20));
  uint8_t *buf = nullptr;
  u64 *mask0 = nullptr, *mask1 = nullptr;
  seg_summary *summ = nullptr;
  uint8_t *esc = nullptr;
  CK(hipMalloc(&buf, target + (1 << 20)));
  CK(hipMalloc(&mask0, target / 8 + 4096));
  CK(hipMalloc(&mask1, target / 8 + 4096));
  CK(hipMalloc(&summ, (target / SEG_BYTES + 4096) * sizeof(seg_summary)));
  CK(hipMalloc(&esc, ESC_TABLE_BYTES));
  CK(hipMemset(esc, 0, ESC_TABLE_BYTES));
  for (int kind = 0; kind < 2; kind++) {
    uint64_t units = 0;
    size_t L = kind == 0 ? sjc_amazon_ndjson(host.data(), host.size(), target, 1000, &units) : sjc_large_random(host.data(), host.size(), target, 1000, &units);
    if (L == 0) { printf("corpus generation failed\n"); return 1; }
    L &= ~size_t(SEG_BYTES - 1); // whole segments only: every variant may use the branch-free loads
    CK(hipMemcpy(buf, host.data(), L, hipMemcpyHostToDevice));
    const u32 nseg = u32(L / SEG_BYTES);
    printf("==== %s, %zu bytes, %u segments\n", kind == 0 ? "amazon_ndjson" : "large_random", L, nseg);
    launch_escape_table(buf, 0, L, esc, nullptr);
    CK(hipDeviceSynchronize());
    auto time = [&](const char *name, auto launch) {
      float best = 1e9f, sum = 0;
      for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, nullptr);
        launch();
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms; }
        if (ms < best) { best = ms; }
      }
      const hipError_t e = hipGetLastError();
      printf("%-64s best %7.4f ms  avg %7.4f ms  %6.0f GB/s%s\n", name, best, sum / (reps - 2), double(L) / (best * 1e-3) / 1e9,
             e == hipSuccess ? "" : hipGetErrorString(e));
      fflush(stdout);
    };
    scan_origin org{0, 0, 0, esc};
    scan_origin org_noesc{0, 0, 0, nullptr};
    time("A0 k_stage1_summarize (product), escape table", [&] { hipLaunchKernelGGL(k_stage1_summarize, dim3((nseg + 3) / 4), dim3(256), 0, nullptr, buf, u64(L), mask0, mask1, summ, org, nseg); });
    time("A1 k_stage1_summarize (product), no table (byte walk)", [&] { hipLaunchKernelGGL(k_stage1_summarize, dim3((nseg + 3) / 4), dim3(256), 0, nullptr, buf, u64(L), mask0, mask1, summ, org_noesc, nseg); });
    time("B1 esc byte hoisted, 64-thread WGs", [&] { hipLaunchKernelGGL((k_scan<1, 0, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("B2 esc byte hoisted, 256-thread WGs (4 segments each)", [&] { hipLaunchKernelGGL((k_scan<1, 0, 256>), dim3((nseg + 3) / 4), dim3(256), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("C1 + next chunk prefetched (double buffer), 64-thread WGs", [&] { hipLaunchKernelGGL((k_scan<2, 0, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("C2 + next chunk prefetched, 256-thread WGs", [&] { hipLaunchKernelGGL((k_scan<2, 0, 256>), dim3((nseg + 3) / 4), dim3(256), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("B3 esc hoisted, forced 6 waves/SIMD", [&] { hipLaunchKernelGGL(k_scan1_occ6, dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("B4 esc hoisted, forced 8 waves/SIMD", [&] { hipLaunchKernelGGL(k_scan1_occ8, dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("C3 prefetch, forced 5 waves/SIMD", [&] { hipLaunchKernelGGL(k_scan2_occ5, dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("C4 prefetch, forced 6 waves/SIMD", [&] { hipLaunchKernelGGL(k_scan2_occ6, dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    for (u32 wpc : {16u, 20u, 24u, 32u}) {
      char name[96];
      snprintf(name, sizeof name, "D  persistent waves + prefetch, %u waves per CU", wpc);
      time(name, [&] { hipLaunchKernelGGL((k_scan_persistent<64>), dim3(256 * wpc), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    }
    time("E1 scan, input wrapped into 1 MiB (L2-resident): compute ceiling", [&] { hipLaunchKernelGGL((k_scan<1, 64, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("E2 same with prefetch", [&] { hipLaunchKernelGGL((k_scan<2, 64, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, mask1, summ, esc, nseg); });
    time("F0 loads + OR + mask store, segment map (summarize's)", [&] { hipLaunchKernelGGL((k_loads<0, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("F1 loads + OR + mask store, grid-stride chunk map, 8192 waves", [&] { hipLaunchKernelGGL((k_loads<1, 64>), dim3(8192), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("F2 loads + OR + mask store, segment map, 4 chunks in flight", [&] { hipLaunchKernelGGL((k_loads<2, 64>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("F3 loads + OR + mask store, segment map, 256-thread WGs", [&] { hipLaunchKernelGGL((k_loads<0, 256>), dim3((nseg + 3) / 4), dim3(256), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("G0 loads + transposition", [&] { hipLaunchKernelGGL((k_lane_math<0>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("G1 loads + transposition + classify", [&] { hipLaunchKernelGGL((k_lane_math<1>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("G2 loads + transposition + classify + per-lane string algebra", [&] { hipLaunchKernelGGL((k_lane_math<2>), dim3(nseg), dim3(64), 0, nullptr, buf, u64(L), mask0, nseg); });
    time("H  k_validate_utf8 (product)", [&] { hipLaunchKernelGGL(k_validate_utf8, dim3(8192), dim3(64), 0, nullptr, buf, u64(L), reinterpret_cast<scan_result_dev *>(summ), u64(0), 0u); });
  }
  return 0;
}
