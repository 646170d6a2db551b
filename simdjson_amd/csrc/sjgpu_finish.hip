// simdjson_amd/csrc/sjgpu_finish.hip -- what the reference does with the structural list AFTER the scan, on the device:
//
//   * document boundaries of the streaming modes: "the last value that directly follows another value", and whether the
//     brackets behind it balance            (/root/reference/src/generic/stage1/find_next_document_index.h:39-98)
//   * RFC 7464 record separators: drop them, re-insert the scalar starts the scanner glued to them, cut at the last one
//                                            (find_next_document_index.h:126-267)
//   * comma-delimited documents: drop the commas at nesting depth 0, cut behind the last one   (:288-369)
//   * the nesting depth in front of every structural (bracket prefix scan): the first data-parallel slice of stage 2 --
//     what the reference's tape builder carries as `depth` while it walks the list serially
//                                            (/root/reference/src/generic/stage2/json_iterator.h, tape_builder.h:108-123)
//
// The reference walks the list backwards / compacts it in place on one core.  Here every step is a map over the list, a
// prefix scan (depth, output slot) or a reduction (last boundary, last separator) -- so it runs where the list lies, needs
// no PCIe round trip per batch, and an NDJSON / RS / comma stream sharded over several GPUs can find its own cuts.
// Behaviour is pinned by stage1_finish.cpp (the host implementation, itself pinned against the reference): the two must
// leave the same words in the index array (tests/test_gpu_parity.py::test_device_finish_*).
#include "sjgpu_device.h"

namespace sjgpu {
namespace {

constexpr u32 FIN_THREADS = 256, FIN_PER_THREAD = 16, FIN_BLOCK = FIN_THREADS * FIN_PER_THREAD; // 4096 structurals per workgroup

// structural classes
enum : u32 { C_OTHER = 0, C_OBJ_OPEN = 1, C_ARR_OPEN = 2, C_OBJ_CLOSE = 3, C_ARR_CLOSE = 4, C_COLON = 5, C_COMMA = 6, C_RS = 7 };
// Branch-free: a chain of selects.  (Rounds 2-4 had a `switch` here, which hipcc compiles into a tree of DIVERGENT branches -- per entry and row a dozen
// s_cbranch with exec-mask juggling: the depth scan's code pass took 235 us for 65 M entries whether it gathered its bytes out of the document or read them
// from the token stream, profiles/r05_token_stream.txt: it was bound by its branches, not by its gather.)
__device__ __forceinline__ u32 classify_byte(u32 c) {
  u32 k = C_OTHER;
  k = c == u32('{') ? u32(C_OBJ_OPEN) : k;
  k = c == u32('[') ? u32(C_ARR_OPEN) : k;
  k = c == u32('}') ? u32(C_OBJ_CLOSE) : k;
  k = c == u32(']') ? u32(C_ARR_CLOSE) : k;
  k = c == u32(':') ? u32(C_COLON) : k;
  k = c == u32(',') ? u32(C_COMMA) : k;
  k = c == 0x1Eu ? u32(C_RS) : k;
  return k;
}
__device__ __forceinline__ bool is_open(u32 k) { return k == C_OBJ_OPEN || k == C_ARR_OPEN; }
__device__ __forceinline__ bool is_close(u32 k) { return k == C_OBJ_CLOSE || k == C_ARR_CLOSE; }
__device__ __forceinline__ bool is_sep(u32 k) { return k == C_COLON || k == C_COMMA; }
__device__ __forceinline__ bool is_ws_byte(u32 c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

// ---- block-level exclusive scan of one int per thread (256 threads) ----------------------------------------------------
__device__ __forceinline__ int block_excl_scan256(int v, int *sh /*[8]*/, int &total) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int incl = int(wave_incl_scan(u32(v)));
  if (lane == 63) { sh[wave] = incl; }
  __syncthreads();
  int base = 0;
  for (u32 w = 0; w < wave; w++) { base += sh[w]; }
  total = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return base + incl - v;
}

// ---- generic in-place exclusive scan of an int array, three kernels per level --------------------------------------------
__global__ __launch_bounds__(FIN_THREADS) void k_scan_blocks(int *__restrict__ a, const u32 *__restrict__ n_ptr, int *__restrict__ partial) {
  __shared__ int sh[8];
  const u32 n = *n_ptr;
  if (u64(blockIdx.x) * FIN_BLOCK >= n) { if (threadIdx.x == 0) { partial[blockIdx.x] = 0; } return; }
  if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) {
    // 16-byte aligned array: four rows of 1024 ints, a thread owns four CONSECUTIVE ints of each row, so that the lanes of a
    // wave read and write consecutive 16-byte pieces (16 consecutive ints per thread put the lanes 64 bytes apart: 1.4 TB/s)
    int before = 0; // sum of the rows in front
#pragma unroll 1
    for (u32 row = 0; row < FIN_PER_THREAD / 4; row++) {
      const u64 e0 = u64(blockIdx.x) * FIN_BLOCK + u64(row) * (FIN_THREADS * 4) + u64(threadIdx.x) * 4;
      int v[4] = {0, 0, 0, 0};
      if (e0 + 3 < n) {
        const int4 q = *reinterpret_cast<const int4 *>(a + e0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        for (u32 j = 0; j < 4; j++) { if (e0 + j < n) { v[j] = a[e0 + j]; } }
      }
      int total;
      const int run = before + block_excl_scan256(v[0] + v[1] + v[2] + v[3], sh, total);
      const int4 o = make_int4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
      if (e0 + 3 < n) {
        *reinterpret_cast<int4 *>(a + e0) = o;
      } else {
        const int w[4] = {o.x, o.y, o.z, o.w};
        for (u32 j = 0; j < 4; j++) { if (e0 + j < n) { a[e0 + j] = w[j]; } }
      }
      before += total;
    }
    if (threadIdx.x == 0) { partial[blockIdx.x] = before; }
    return;
  }
  const u64 base = u64(blockIdx.x) * FIN_BLOCK + u64(threadIdx.x) * FIN_PER_THREAD;
  int v[FIN_PER_THREAD], sum = 0;
#pragma unroll
  for (u32 j = 0; j < FIN_PER_THREAD; j++) {
    v[j] = (base + j < n) ? a[base + j] : 0;
    sum += v[j];
  }
  int total;
  int run = block_excl_scan256(sum, sh, total);
#pragma unroll
  for (u32 j = 0; j < FIN_PER_THREAD; j++) {
    if (base + j < n) { a[base + j] = run; }
    run += v[j];
  }
  if (threadIdx.x == 0) { partial[blockIdx.x] = total; }
}
// one workgroup: exclusive scan of up to 2^20 partials in place (a second level on top would follow the same pattern;
// 2^20 blocks of 4096 cover the whole 32-bit index range)
__global__ __launch_bounds__(1024) void k_scan_partials(int *__restrict__ partial, u32 nblocks) {
  __shared__ int sh[1024];
  const u32 per = (nblocks + 1023) / 1024;
  const u32 lo = min(threadIdx.x * per, nblocks), hi = min(lo + per, nblocks);
  int sum = 0;
  for (u32 i = lo; i < hi; i++) { sum += partial[i]; }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (u32 d = 1; d < 1024; d <<= 1) {
    const int t = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = sh[threadIdx.x] - sum;
  for (u32 i = lo; i < hi; i++) {
    const int x = partial[i];
    partial[i] = run;
    run += x;
  }
}
__global__ __launch_bounds__(FIN_THREADS) void k_scan_add(int *__restrict__ a, const u32 *__restrict__ n_ptr, const int *__restrict__ partial) {
  const u32 n = *n_ptr;
  const int add = partial[blockIdx.x];
  if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) { // the layout of k_scan_blocks' aligned path: coalesced 16-byte pieces
#pragma unroll
    for (u32 row = 0; row < FIN_PER_THREAD / 4; row++) {
      const u64 e0 = u64(blockIdx.x) * FIN_BLOCK + u64(row) * (FIN_THREADS * 4) + u64(threadIdx.x) * 4;
      if (e0 + 3 < n) {
        int4 q = *reinterpret_cast<const int4 *>(a + e0);
        q.x += add; q.y += add; q.z += add; q.w += add;
        *reinterpret_cast<int4 *>(a + e0) = q;
      } else {
        for (u32 j = 0; j < 4; j++) { if (e0 + j < n) { a[e0 + j] += add; } }
      }
    }
    return;
  }
  const u64 base = u64(blockIdx.x) * FIN_BLOCK + u64(threadIdx.x) * FIN_PER_THREAD;
#pragma unroll
  for (u32 j = 0; j < FIN_PER_THREAD; j++) {
    if (base + j < n) { a[base + j] += add; }
  }
}

// ---- maps ---------------------------------------------------------------------------------------------------------------
// delta[i] = +1 / -1 / 0 for an opening / closing bracket / anything else (nesting depth as ONE counter, the way the
// reference's comma filter and the tape builder count it)
__global__ __launch_bounds__(FIN_THREADS) void k_bracket_delta(const u8 *__restrict__ buf, const u32 *__restrict__ idx, const u32 *__restrict__ n_ptr,
                                                             int *__restrict__ delta) {
  const u32 n = *n_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  if (i >= n) { return; }
  const u32 k = classify_byte(buf[idx[i]]);
  delta[i] = is_open(k) ? 1 : (is_close(k) ? -1 : 0);
}

// ---- the boundary search of the streaming modes -------------------------------------------------------------------------------------------
// The reference walks the list BACKWARDS and stops at the first hit (find_next_document_index.h:39-98): O(last document).  Rounds 2-3 ran a map over
// the whole list instead -- 65 M gathers and 254 000 workgroups to find what the last thousand entries of an NDJSON list settle (0.55 ms per GiB of
// amazon NDJSON, bench.py leg next_f2_finish_device).  Now a bounded grid walks the list from its END in chunks of 1024 entries and a workgroup
// leaves as soon as a boundary behind its chunk is known; the balance runs from that boundary on, over as many chunks as there are; the last
// workgroup to arrive resolves.  One launch prepares the state (it replaced three memsets).  A list without any boundary (one large document) still
// costs the whole map -- as it does the reference.
constexpr u32 LB_PER = 4, LB_CHUNK = FIN_THREADS * LB_PER, LB_GRID_MAX = 1024, LB_HEAD = 32;
__global__ void k_finish_init(finish_state *__restrict__ st, u32 n, u32 len) {
  finish_state z{};
  z.n_in = n; z.n_cur = n; z.n_report = n;
  z.next_start = len;
  *st = z; // verdict = FIN_SEARCH, all reductions at their identity
}
// Last "value directly following a value" among structurals [1, n): atomicMax of its list index + 1 (0 = none).
// Chunks first_chunk, first_chunk + 1, ... counted from the END of the list, at most chunk_limit of them: the search is two launches -- the last
// LB_HEAD chunks by as many workgroups, then the rest, whose workgroups find the answer waiting and leave (all workgroups of ONE launch start before
// the first has fetched a byte: a thousand of them posting candidates to one word cost 29 us where 3 are needed).
__global__ __launch_bounds__(FIN_THREADS) void k_last_boundary(const u8 *__restrict__ buf, const u32 *__restrict__ idx, const u32 *__restrict__ n_ptr,
                                                             finish_state *__restrict__ st, u32 first_chunk, u32 chunk_limit) {
  const u32 n = *n_ptr;
  if (n < 2) { return; }
  const u32 all_chunks = (n + LB_CHUNK - 1) / LB_CHUNK;
  const u64 stop = u64(first_chunk) + chunk_limit;
  const u32 nchunks = stop < all_chunks ? u32(stop) : all_chunks;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(idx) & 15u) == 0;
  for (u64 c = u64(first_chunk) + blockIdx.x; c < nchunks; c += gridDim.x) {
    const u64 lo = (u64(all_chunks) - 1u - c) * LB_CHUNK; // chunks are counted from the end, aligned to the list's start
    // a boundary at or behind this chunk's end is known: nothing in this chunk -- or in the ones this workgroup would take next -- can beat it
    if (u64(__hip_atomic_load(&st->boundary_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > lo + LB_CHUNK) { return; }
    const u64 i0 = lo + u64(threadIdx.x) * LB_PER;
    u32 best = 0; // highest boundary of this thread, + 1
    if (i0 < n) {
      u32 pos[LB_PER], cls[LB_PER + 1];
      if (i0 + LB_PER <= n && aligned16) {
        const uint4 q = *reinterpret_cast<const uint4 *>(idx + i0);
        pos[0] = q.x; pos[1] = q.y; pos[2] = q.z; pos[3] = q.w;
      } else {
#pragma unroll
        for (u32 j = 0; j < LB_PER; j++) { pos[j] = idx[i0 + j < n ? i0 + j : i0]; }
      }
      const u32 before = idx[i0 ? i0 - 1 : 0]; // the entry in front of the thread's first (entry 0 has none: never a boundary)
      cls[0] = classify_byte(buf[before]);
#pragma unroll
      for (u32 j = 0; j < LB_PER; j++) { cls[j + 1] = classify_byte(buf[pos[j]]); }
#pragma unroll
      for (u32 j = 0; j < LB_PER; j++) {
        const u64 i = i0 + j;
        const u32 cur = cls[j + 1], prev = cls[j];
        const bool b = i >= 1 && i < n && !is_sep(cur) && !is_close(cur) && !is_open(prev) && !is_sep(prev);
        best = b ? u32(i) + 1u : best;
      }
    }
    const u32 cand = wave_max(best);
    if (cand && (threadIdx.x & 63u) == 0 && cand > __hip_atomic_load(&st->boundary_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      atomicMax(&st->boundary_plus1, cand);
    }
  }
}
// complete_prefix(v, n) of stage1_finish.cpp from the two reductions
__device__ __forceinline__ void resolve_prefix(finish_state *__restrict__ st, u32 n) {
  if (st->verdict != FIN_SEARCH) { return; } // the filter has already decided (k_after_filter / k_count_below)
  const int ob = __hip_atomic_load(&st->obj_balance, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int ab = __hip_atomic_load(&st->arr_balance, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool balanced = ob == 0 && ab == 0;
  st->keep = (n == 0) ? 0u : (balanced ? n : (st->boundary_plus1 ? st->boundary_plus1 - 1 : 0u));
}
// Bracket balance of the structurals from the last boundary (or 0) to n, braces and square brackets separately; the workgroup that arrives
// last resolves (st->arrivals counts them).
__global__ __launch_bounds__(FIN_THREADS) void k_tail_balance(const u8 *__restrict__ buf, const u32 *__restrict__ idx, const u32 *__restrict__ n_ptr,
                                                            finish_state *__restrict__ st) {
  const u32 n = *n_ptr;
  const u32 from = st->boundary_plus1 ? st->boundary_plus1 - 1 : 0;
  // the workgroups that have entries to count (the first always "has"): the others leave without a word -- arrivals at ONE counter are served
  // one after the other, a thousand of them took 20 us
  const u32 span_groups = n > from ? (n - from + FIN_THREADS - 1) / FIN_THREADS : 0u;
  const u32 active = span_groups == 0 ? 1u : (span_groups < gridDim.x ? span_groups : gridDim.x);
  if (blockIdx.x >= active) { return; }
  int o = 0, a = 0;
  for (u64 i = u64(from) + u64(blockIdx.x) * FIN_THREADS + threadIdx.x; i < n; i += u64(active) * FIN_THREADS) {
    const u32 k = classify_byte(buf[idx[i]]);
    o += (k == C_OBJ_OPEN) - (k == C_OBJ_CLOSE);
    a += (k == C_ARR_OPEN) - (k == C_ARR_CLOSE);
  }
  const int so = int(wave_sum(u32(o))), sa = int(wave_sum(u32(a)));
  if ((threadIdx.x & 63u) == 0) {
    if (so) { atomicAdd(&st->obj_balance, so); }
    if (sa) { atomicAdd(&st->arr_balance, sa); }
  }
  __syncthreads(); // every wave's contribution has been issued
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&st->arrivals, 1u) == active - 1u) {
      __threadfence();
      resolve_prefix(st, n);
    }
  }
}

// ---- nesting depth: a map that leaves 2 bits per entry, a scan of tile totals, a pass that writes ---------------------------------------------
// Rounds 2-3: k_bracket_delta wrote an int per entry and the generic scan read and wrote that array twice more (0.49 ms per 65 M entries of
// amazon NDJSON).  What the shapes cost was measured in isolation (scripts/micro/gather_lab.hip, profiles/r04_gather_lab.txt, 64 M entries over
// 1 GiB): the bare map -- list word, token byte, something out -- takes 264-279 us (5.8 TB/s: the whole document comes in, line by line, to yield
// one byte per 16); a single pass with decoupled look-back pays 17 ns per TILE for its ticket and its walk whatever the tile does (290 us for
// 16 384 tiles of 4096: 0.42-0.58 ms measured, no gain; tiles of 16 384 need 214 VGPRs and ran at 0.74 ms); four consecutive entries per thread
// (16-byte list loads) make a wave's gather reach over 4 KiB with two lanes per line and come back to every line four times: +95 us.  So: the map
// keeps its one entry per lane and row, leaves the deltas as 2-bit codes (one dword per thread and tile: 0.25 B per entry instead of 4) and the
// tile totals; one workgroup scans the totals; the writing pass turns codes into depths with ballots.
constexpr u32 DP_ROWS = 16, DP_TILE = DP_ROWS * FIN_THREADS;
static_assert(DP_TILE == FIN_BLOCK, "k_scan_partials' callers count tiles of FIN_BLOCK entries");
// codes[tile * 256 + thread]: bits 2 r = the entry of row r (tile * 4096 + r * 256 + thread): 1 opens, 2 closes; partial[tile] = opens - closes
// TOK: the entries' bytes come from the token stream stage 1 wrote beside the list (sjgpu_stage1_tokens_device: tok[e] = buf[idx[e]]) -- one byte per
// entry, coalesced, instead of the list word AND a 128-byte line of the document per entry (the gather that held this pass at 0.21 of the roofline)
template <bool TOK>
__global__ __launch_bounds__(FIN_THREADS) void k_depth_codes(const u8 *__restrict__ buf, const u32 *__restrict__ idx, u32 n, u32 *__restrict__ codes,
                                                           int *__restrict__ partial, const u8 *__restrict__ tok) {
  __shared__ int sh[FIN_THREADS / 64];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u64 tile0 = u64(blockIdx.x) * DP_TILE;
  u32 pos[DP_ROWS]; // TOK: the byte itself
#pragma unroll
  for (u32 row = 0; row < DP_ROWS; row++) {
    const u64 e = tile0 + u64(row) * FIN_THREADS + tid;
    pos[row] = e < n ? (TOK ? u32(tok[e]) : idx[e]) : 0xFFFFFFFFu;
  }
  u32 c = 0;
  int sum = 0; // of this wave's entries (wave-uniform)
#pragma unroll
  for (u32 row = 0; row < DP_ROWS; row++) {
    u32 code = 0;
    if (pos[row] != 0xFFFFFFFFu) {
      const u32 ch = (TOK ? pos[row] : u32(buf[pos[row]])) | 0x20u; // '[' | 0x20 = '{', ']' | 0x20 = '}', and no other byte maps onto either
      code = ch == u32('{') ? 1u : (ch == u32('}') ? 2u : 0u);
    }
    c |= code << (2u * row);
    sum += int(popc64(__ballot(code == 1u))) - int(popc64(__ballot(code == 2u)));
  }
  codes[u64(blockIdx.x) * FIN_THREADS + tid] = c;
  if (lane == 0) { sh[wave] = sum; }
  __syncthreads();
  if (tid == 0) { partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3]; }
}
// depth[e] for e in [0, n]: partial holds the exclusive prefixes of the tile totals now
__global__ __launch_bounds__(FIN_THREADS) void k_depth_write(const u32 *__restrict__ codes, const int *__restrict__ partial, u32 n, int *__restrict__ depth) {
  constexpr u32 WAVES = FIN_THREADS / 64;
  static_assert(DP_ROWS * WAVES == 64, "the (row, wave) totals of a tile are scanned by one wave");
  __shared__ int sh_w[DP_ROWS * WAVES];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u64 n1 = u64(n) + 1, tile0 = u64(blockIdx.x) * DP_TILE;
  const u32 c = codes[u64(blockIdx.x) * FIN_THREADS + tid];
  const int front = partial[blockIdx.x];
  const u64 below = lanemask_lt(lane);
  int mine[DP_ROWS]; // opens - closes among the lanes in front of this one, per row
#pragma unroll
  for (u32 row = 0; row < DP_ROWS; row++) {
    const u32 code = (c >> (2u * row)) & 3u;
    const u64 opens = __ballot(code == 1u), closes = __ballot(code == 2u);
    mine[row] = int(popc64(opens & below)) - int(popc64(closes & below));
    if (lane == 0) { sh_w[row * WAVES + wave] = int(popc64(opens)) - int(popc64(closes)); }
  }
  __syncthreads();
  if (wave == 0) { // the 64 (row, wave) totals in row-major order = list order: exclusive prefixes inside the tile
    const int x = sh_w[lane];
    sh_w[lane] = int(wave_incl_scan(u32(x))) - x;
  }
  __syncthreads();
#pragma unroll
  for (u32 row = 0; row < DP_ROWS; row++) {
    const u64 e = tile0 + u64(row) * FIN_THREADS + tid;
    if (e < n1) { depth[e] = front + sh_w[row * WAVES + wave] + mine[row]; }
  }
}

// ---- comma-delimited: keep flag per structural, last root comma ------------------------------------------------------------
// depth[i] = nesting depth in front of structural i (exclusive scan of k_bracket_delta); keep[i] = 0 for a comma at depth 0
__global__ __launch_bounds__(FIN_THREADS) void k_comma_flags(const u8 *__restrict__ buf, const u32 *__restrict__ idx, const u32 *__restrict__ n_ptr,
                                                           const int *__restrict__ depth, int *__restrict__ keep, finish_state *__restrict__ st) {
  const u32 n = *n_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  bool root = false;
  if (i < n) {
    root = buf[idx[i]] == ',' && depth[i] == 0;
    keep[i] = root ? 0 : 1;
  }
  const u64 m = __ballot(root);
  if (m && (threadIdx.x & 63u) == 0) {
    atomicAdd(&st->separators, u32(popc64(m)));
    const u32 cand = u32(u64(blockIdx.x) * FIN_THREADS + (threadIdx.x & ~63u) + 63u - clz64(m)) + 1u;
    if (cand > __hip_atomic_load(&st->last_sep_index_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicMax(&st->last_sep_index_plus1, cand); }
  }
}

// ---- record separators -----------------------------------------------------------------------------------------------------
// first byte behind p that is neither whitespace nor RS (or len); counts the RS bytes passed and remembers the last one
__device__ __forceinline__ u32 value_behind(const u8 *__restrict__ buf, u32 len, u32 p, u32 &rs_passed, u32 &last_rs) {
  u32 v = p + 1;
  for (; v < len; v++) {
    const u32 c = buf[v];
    if (c == 0x1E) { rs_passed++; last_rs = v; }
    else if (!is_ws_byte(c)) { break; }
  }
  return v;
}
// keep[i] = 1 if list slot i yields an output entry, out_pos[i] = the entry: a non-RS structural as it is; the head of a
// run of record separators yields the start of the value behind the run when the scanner glued that value to the RS
__global__ __launch_bounds__(FIN_THREADS) void k_rs_flags(const u8 *__restrict__ buf, u32 len, const u32 *__restrict__ idx, const u32 *__restrict__ n_ptr,
                                                        int *__restrict__ keep, u32 *__restrict__ out_pos, finish_state *__restrict__ st) {
  const u32 n = *n_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  if (i >= n) { return; }
  const u32 pos = idx[i];
  if (buf[pos] != 0x1E) {
    keep[i] = 1;
    out_pos[i] = pos;
    return;
  }
  keep[i] = 0;
  out_pos[i] = 0;
  // absorbed: the record separator in front of this one reaches past it (only whitespace / RS in between)
  if (i > 0) {
    const u32 prev = idx[i - 1];
    if (buf[prev] == 0x1E) {
      bool only_blank = true;
      for (u32 q = prev + 1; q < pos; q++) {
        const u32 c = buf[q];
        if (c != 0x1E && !is_ws_byte(c)) { only_blank = false; break; }
      }
      if (only_blank) { return; }
    }
  }
  u32 rs = 1, last_rs = pos;
  const u32 value = value_behind(buf, len, pos, rs, last_rs);
  atomicAdd(&st->separators, rs);
  atomicMax(&st->last_sep_pos_plus1, last_rs + 1u);
  if (value < len) {
    const u32 k = classify_byte(buf[value]);
    const bool is_operator = k != C_OTHER && k != C_RS;
    u64 m = i + 1;
    while (m < n && idx[m] < value) { m++; } // the separators this run absorbed
    const bool scanner_has_it = m < n && idx[m] == value;
    if (!is_operator && !scanner_has_it) {
      keep[i] = 1;
      out_pos[i] = value;
    }
  }
}

// ---- compaction ------------------------------------------------------------------------------------------------------------
// slot[i] = exclusive scan of keep; src = the entries (idx itself or k_rs_flags' out_pos); tmp[slot[i]] = src[i] for kept i.
// keep_flags holds the flags as they were BEFORE the scan turned the array into slots (a copy).
__global__ __launch_bounds__(FIN_THREADS) void k_scatter(const u32 *__restrict__ src, const int *__restrict__ slot, const u8 *__restrict__ keep_flags,
                                                       const u32 *__restrict__ n_ptr, u32 *__restrict__ tmp) {
  const u32 n = *n_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  if (i < n && keep_flags[i]) { tmp[slot[i]] = src[i]; }
}
__global__ __launch_bounds__(FIN_THREADS) void k_flags_to_bytes(const int *__restrict__ keep, const u32 *__restrict__ n_ptr, u8 *__restrict__ flags) {
  const u32 n = *n_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  if (i < n) { flags[i] = u8(keep[i]); }
}
__global__ __launch_bounds__(FIN_THREADS) void k_copy_back(const u32 *__restrict__ tmp, const u32 *__restrict__ count_ptr, u32 *__restrict__ idx) {
  const u32 n = *count_ptr;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  if (i < n) { idx[i] = tmp[i]; }
}
// kept = slot[n-1] + keep[n-1]; then what filter_root_commas / filter_record_separators decide from the reductions
// (stage1_finish.cpp).  mode_final: 1 for the *_FINAL modes.  Leaves st->n_cur = the list length the boundary search runs on.
__global__ void k_after_filter(finish_state *__restrict__ st, const int *__restrict__ slot, const u8 *__restrict__ keep_flags, const u32 *__restrict__ idx,
                               const u32 *__restrict__ src, u32 len, int is_rs, int mode_final) {
  const u32 n = st->n_in;
  const u32 kept = n ? u32(slot[n - 1]) + u32(keep_flags[n - 1]) : 0u;
  st->kept = kept;
  st->next_start = len;
  st->verdict = FIN_SEARCH; // run the boundary search on n_cur entries
  st->n_cur = kept;
  st->n_report = kept;
  if (kept == 0) { st->verdict = FIN_KEEP_GIVEN; st->keep = 0; return; }
  if (is_rs) {
    if (st->separators == 0) {
      if (!mode_final) { st->verdict = FIN_KEEP_GIVEN; st->keep = 0; }
      return;
    }
    if (mode_final) { st->verdict = FIN_KEEP_GIVEN; st->keep = kept; return; }
    st->next_start = st->last_sep_pos_plus1 - 1;
    if (st->separators < 2) { st->verdict = FIN_TOO_LARGE; return; }
    st->verdict = FIN_COUNT_BELOW; // keep = entries of the compacted list in front of the last record separator
    return;
  }
  if (mode_final) { return; }
  if (st->separators == 0) { st->verdict = FIN_TOO_LARGE; return; }
  const u32 li = st->last_sep_index_plus1 - 1; // list index of the last root comma
  st->next_start = idx[li] + 1; // idx is still the un-compacted list here
  const u32 k = u32(slot[li]);  // kept entries in front of it
  (void)src;
  if (k == 0) { st->verdict = FIN_KEEP_GIVEN; st->keep = 0; return; }
  st->n_cur = k;
  st->n_report = k;
}
// entries of the compacted list (tmp) below a position: the list is ascending, so a count is a search
__global__ __launch_bounds__(FIN_THREADS) void k_count_below(const u32 *__restrict__ tmp, finish_state *__restrict__ st) {
  if (st->verdict != FIN_COUNT_BELOW) { return; }
  const u32 n = st->kept, limit = st->next_start;
  const u64 i = u64(blockIdx.x) * FIN_THREADS + threadIdx.x;
  const bool below = i < n && tmp[i] < limit;
  const u64 m = __ballot(below);
  if (m && (threadIdx.x & 63u) == 0) { atomicAdd(&st->keep, u32(popc64(m))); }
}

} // namespace

// ---- host side: enqueue the kernels of one finish ------------------------------------------------------------------------------
static inline u32 blocks_for(u64 n, u32 per) { return u32((n + per - 1) / per); }

// in-place exclusive scan of up to 2^20 ints by one workgroup
void launch_scan_partials(int *a, u32 count, hipStream_t s) { hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, a, count); }

void enqueue_scan(int *a, u32 n_max, const u32 *n_ptr, int *partial, hipStream_t s) {
  const u32 nb = blocks_for(n_max, FIN_BLOCK);
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(FIN_THREADS), 0, s, a, n_ptr, partial);
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partial, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(FIN_THREADS), 0, s, a, n_ptr, partial);
  }
}

size_t finish_workspace_bytes(uint32_t n) {
  const size_t nn = size_t(n) + 64;
  // state | scan array (int) | out_pos / tmp (u32) x2 | keep flags (u8) | partials
  return 256 + nn * 4 + nn * 4 + nn * 4 + nn + (size_t(blocks_for(nn, FIN_BLOCK)) + 64) * 4 + 256;
}

// boundary search over the first st->n_cur entries of `list`
static void enqueue_prefix_search(const uint8_t *buf, const uint32_t *list, uint32_t n_max, finish_state *st, hipStream_t s) {
  if (n_max == 0) { return; }
  const u32 chunks = blocks_for(n_max, LB_CHUNK), grid = chunks < LB_GRID_MAX ? chunks : LB_GRID_MAX;
  hipLaunchKernelGGL(k_last_boundary, dim3(chunks < LB_HEAD ? chunks : LB_HEAD), dim3(FIN_THREADS), 0, s, buf, list, &st->n_cur, st, 0u, LB_HEAD);
  if (chunks > LB_HEAD) {
    const u32 rest = chunks - LB_HEAD;
    hipLaunchKernelGGL(k_last_boundary, dim3(rest < LB_GRID_MAX ? rest : LB_GRID_MAX), dim3(FIN_THREADS), 0, s, buf, list, &st->n_cur, st, LB_HEAD, 0xFFFFFFFFu);
  }
  hipLaunchKernelGGL(k_tail_balance, dim3(grid), dim3(FIN_THREADS), 0, s, buf, list, &st->n_cur, st);
}

void launch_finish(int mode, const uint8_t *buf, uint64_t len, uint32_t *idx, uint32_t n, void *workspace, hipStream_t s) {
  uint8_t *w = static_cast<uint8_t *>(workspace);
  finish_state *st = reinterpret_cast<finish_state *>(w);
  const size_t nn = size_t(n) + 64;
  int *scan = reinterpret_cast<int *>(w + 256);
  u32 *out_pos = reinterpret_cast<u32 *>(w + 256 + nn * 4);
  u32 *tmp = reinterpret_cast<u32 *>(w + 256 + nn * 8);
  u8 *flags = w + 256 + nn * 12;
  int *partial = reinterpret_cast<int *>(w + 256 + nn * 12 + ((nn + 255) & ~size_t(255)));
  hipLaunchKernelGGL(k_finish_init, dim3(1), dim3(1), 0, s, st, n, u32(len));
  const bool rs = mode == SJGPU_JSON_SEQUENCE_PARTIAL || mode == SJGPU_JSON_SEQUENCE_FINAL;
  const bool comma = mode == SJGPU_COMMA_DELIMITED_PARTIAL || mode == SJGPU_COMMA_DELIMITED_FINAL;
  const bool final_batch = mode == SJGPU_STREAMING_FINAL || mode == SJGPU_JSON_SEQUENCE_FINAL || mode == SJGPU_COMMA_DELIMITED_FINAL;
  const u32 nb = blocks_for(n, FIN_THREADS);
  if (!rs && !comma) {
    enqueue_prefix_search(buf, idx, n, st, s);
    return;
  }
  if (comma) {
    hipLaunchKernelGGL(k_bracket_delta, dim3(nb), dim3(FIN_THREADS), 0, s, buf, idx, &st->n_in, scan);
    enqueue_scan(scan, n, &st->n_in, partial, s);
    int *keep = reinterpret_cast<int *>(out_pos); // the slot array of the compaction (out_pos is unused in this mode)
    hipLaunchKernelGGL(k_comma_flags, dim3(nb), dim3(FIN_THREADS), 0, s, buf, idx, &st->n_in, scan, keep, st);
    hipLaunchKernelGGL(k_flags_to_bytes, dim3(nb), dim3(FIN_THREADS), 0, s, keep, &st->n_in, flags);
    enqueue_scan(keep, n, &st->n_in, partial, s);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(FIN_THREADS), 0, s, idx, keep, flags, &st->n_in, tmp);
    hipLaunchKernelGGL(k_after_filter, dim3(1), dim3(1), 0, s, st, keep, flags, idx, idx, u32(len), 0, final_batch ? 1 : 0);
  } else {
    int *keep = scan;
    hipLaunchKernelGGL(k_rs_flags, dim3(nb), dim3(FIN_THREADS), 0, s, buf, u32(len), idx, &st->n_in, keep, out_pos, st);
    hipLaunchKernelGGL(k_flags_to_bytes, dim3(nb), dim3(FIN_THREADS), 0, s, keep, &st->n_in, flags);
    enqueue_scan(keep, n, &st->n_in, partial, s);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(FIN_THREADS), 0, s, out_pos, keep, flags, &st->n_in, tmp);
    hipLaunchKernelGGL(k_after_filter, dim3(1), dim3(1), 0, s, st, keep, flags, idx, out_pos, u32(len), 1, final_batch ? 1 : 0);
    hipLaunchKernelGGL(k_count_below, dim3(nb), dim3(FIN_THREADS), 0, s, tmp, st);
  }
  hipLaunchKernelGGL(k_copy_back, dim3(nb), dim3(FIN_THREADS), 0, s, tmp, &st->kept, idx);
  // the boundary search of the modes that still need one runs on the compacted list; the resolution's answer is only
  // taken when the verdict says so (finish_state::verdict)
  enqueue_prefix_search(buf, idx, n, st, s);
}

// depth[i] = nesting depth in front of structural i, i in [0, n]  (depth[n] = depth behind the last one); scratch holds
// depth_scan_scratch_bytes(n) bytes: the tile totals and one dword of codes per thread and tile
size_t depth_scan_scratch_bytes(uint32_t n) {
  const size_t tiles = blocks_for(u64(n) + 1, DP_TILE);
  return 256 + ((tiles + 64) * 4 + 255) / 256 * 256 + tiles * FIN_THREADS * 4;
}
void launch_depth_scan(const uint8_t *buf, const uint32_t *idx, uint32_t n, int32_t *depth, void *scratch, hipStream_t s, const uint8_t *tok) {
  const u32 tiles = blocks_for(u64(n) + 1, DP_TILE);
  int *partial = reinterpret_cast<int *>(static_cast<u8 *>(scratch) + 256);
  u32 *codes = reinterpret_cast<u32 *>(static_cast<u8 *>(scratch) + 256 + ((size_t(tiles) + 64) * 4 + 255) / 256 * 256);
  if (tok) { hipLaunchKernelGGL(k_depth_codes<true>, dim3(tiles), dim3(FIN_THREADS), 0, s, buf, idx, n, codes, partial, tok); }
  else { hipLaunchKernelGGL(k_depth_codes<false>, dim3(tiles), dim3(FIN_THREADS), 0, s, buf, idx, n, codes, partial, tok); }
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partial, tiles);
  hipLaunchKernelGGL(k_depth_write, dim3(tiles), dim3(FIN_THREADS), 0, s, codes, partial, n, depth);
}

} // namespace sjgpu
