// scripts/micro/gather_lab.hip -- what bounds a pass over a SPARSE structural list (finish, depth scan, the tape's token front): the list pass
// reads 4 B of list and ONE byte of document per structural -- through 128-byte lines -- and writes 4 B.  Variants of the same pass over a synthetic
// list (one structural per ~16 bytes of a 1 GiB document, the density of amazon NDJSON), timed in one process so that they can be compared:
//   map1        one entry per thread: list word, token byte, int out                       (k_bracket_delta of rounds 2-3)
//   map1 -g     the same without the gather (what the list and the output alone cost)
//   map1 b      the same with a BYTE out
//   one <flags> the one-pass scan (k_depth_onepass): tiles of 4096, ticket, look-back;  -t no ticket, -l no look-back, -g no gather
//   one4        the one-pass scan with four CONSECUTIVE entries per thread (16-byte list loads, the first version)
//   lds         map1, the token bytes taken from an LDS copy of the document bytes the workgroup's entries span (coalesced 16-byte loads)
// hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_lab.hip -o build/gather_lab.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

__device__ __forceinline__ int delta_of(u32 c) { return (c == '{' || c == '[') ? 1 : ((c == '}' || c == ']') ? -1 : 0); }
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));
  v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ u32 readlane(u32 v, int l) { return u32(__builtin_amdgcn_readlane(int(v), l)); }

template <bool GATHER, bool BYTE_OUT>
__global__ __launch_bounds__(256) void k_map1(const u8 *__restrict__ buf, const u32 *__restrict__ idx, u32 n, void *__restrict__ out) {
  const u64 i = u64(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) { return; }
  const u32 p = idx[i];
  const int d = delta_of(GATHER ? u32(buf[p]) : (p & 0xFFu));
  if (BYTE_OUT) { static_cast<u8 *>(out)[i] = u8(d); } else { static_cast<int *>(out)[i] = d; }
}

constexpr u64 AGG = 1ull << 62, INCL = 2ull << 62;
__device__ __forceinline__ int lookback(const u64 *__restrict__ desc, u32 tile, u32 lane) {
  int acc = 0;
  for (int end = int(tile);;) {
    const int t = end - 1 - int(lane);
    const u64 d = t >= 0 ? __hip_atomic_load(desc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INCL;
    const u32 status = u32(d >> 62);
    const u64 incl = __ballot(status == 2u), valid = __ballot(status != 0u);
    const u32 k = incl ? u32(__ffsll((long long)incl) - 1) : 64u;
    const u64 upto = k >= 63u ? ~0ull : ((2ull << k) - 1ull);
    if (~valid & upto) { __builtin_amdgcn_s_sleep(2); continue; }
    acc += int(readlane(wave_incl_scan((upto >> lane) & 1ull ? u32(d) : 0u), 63));
    if (k != 64u) { return acc; }
    end -= 64;
  }
}
// PER = consecutive entries per thread and row (1 or 4); tile = 4096 entries
template <u32 PER, bool GATHER, bool LOOKBACK, bool TICKET>
__global__ __launch_bounds__(256) void k_one(const u8 *__restrict__ buf, const u32 *__restrict__ idx, u32 n, int *__restrict__ depth, u64 *__restrict__ desc,
                                             u32 *__restrict__ ticket) {
  constexpr u32 ROWS = 16 / PER;
  __shared__ int sh_w[ROWS * 4];
  __shared__ u32 sh_tile;
  __shared__ int sh_front;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  u32 tile = blockIdx.x;
  if (TICKET) {
    if (tid == 0) { sh_tile = atomicAdd(ticket, 1u); }
    __syncthreads();
    tile = sh_tile;
  }
  const u64 tile0 = u64(tile) * 4096;
  u32 pos[ROWS][PER];
#pragma unroll
  for (u32 row = 0; row < ROWS; row++) {
    const u64 e0 = tile0 + u64(row) * (256 * PER) + u64(tid) * PER;
    if (PER == 4 && e0 + 3 < n) {
      const uint4 q = *reinterpret_cast<const uint4 *>(idx + e0);
      pos[row][0] = q.x; pos[row][1 % PER] = q.y; pos[row][2 % PER] = q.z; pos[row][3 % PER] = q.w;
    } else {
#pragma unroll
      for (u32 j = 0; j < PER; j++) { pos[row][j] = e0 + j < n ? idx[e0 + j] : 0u; }
    }
  }
  u32 codes = 0;
  int incl[ROWS];
#pragma unroll
  for (u32 row = 0; row < ROWS; row++) {
    const u64 e0 = tile0 + u64(row) * (256 * PER) + u64(tid) * PER;
    int sum = 0;
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
      int d = 0;
      if (e0 + j < n) { d = delta_of(GATHER ? u32(buf[pos[row][j]]) : (pos[row][j] & 0xFFu)); }
      codes |= u32(d + 1) << (2u * (row * PER + j));
      sum += d;
    }
    incl[row] = int(wave_incl_scan(u32(sum)));
    if (lane == 63) { sh_w[row * 4 + wave] = incl[row]; }
  }
  __syncthreads();
  if (wave == 0) {
    const int x = lane < ROWS * 4 ? sh_w[lane < ROWS * 4 ? lane : 0] : 0, inc = int(wave_incl_scan(u32(x)));
    if (lane < ROWS * 4) { sh_w[lane] = inc - x; }
    const int total = int(readlane(u32(inc), 63));
    int front = 0;
    if (LOOKBACK) {
      if (lane == 0 && tile != 0) { __hip_atomic_store(desc + tile, AGG | u64(u32(total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      front = tile ? lookback(desc, tile, lane) : 0;
      if (lane == 0) { __hip_atomic_store(desc + tile, INCL | u64(u32(front + total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if (lane == 0) { sh_front = front; }
  }
  __syncthreads();
  const int front = sh_front;
#pragma unroll
  for (u32 row = 0; row < ROWS; row++) {
    const u64 e0 = tile0 + u64(row) * (256 * PER) + u64(tid) * PER;
    int d[PER], own = 0;
#pragma unroll
    for (u32 j = 0; j < PER; j++) { d[j] = int((codes >> (2u * (row * PER + j))) & 3u) - 1; own += d[j]; }
    int run = front + sh_w[row * 4 + wave] + incl[row] - own;
    if (PER == 4 && e0 + 3 < n) {
      *reinterpret_cast<int4 *>(depth + e0) = make_int4(run, run + d[0], run + d[0] + d[1 % PER], run + d[0] + d[1 % PER] + d[2 % PER]);
    } else {
#pragma unroll
      for (u32 j = 0; j < PER; j++) { if (e0 + j < n) { depth[e0 + j] = run; } run += d[j]; }
    }
  }
}
// map1 with the document bytes staged through LDS: a workgroup's 1024 entries span [first, last]; it copies that range 16 KiB at a time (16-byte
// loads, coalesced) and every thread picks its bytes out of the copy
__global__ __launch_bounds__(256) void k_lds(const u8 *__restrict__ buf, const u32 *__restrict__ idx, u32 n, int *__restrict__ out) {
  constexpr u32 PER = 4, STAGE = 16384;
  __shared__ uint4 sh[STAGE / 16];
  const u64 e0 = u64(blockIdx.x) * (256 * PER);
  u32 pos[PER];
  int d[PER];
#pragma unroll
  for (u32 j = 0; j < PER; j++) { const u64 e = e0 + j * 256 + threadIdx.x; pos[j] = e < n ? idx[e] : 0xFFFFFFFFu; d[j] = 0; }
  const u64 last_e = e0 + 256 * PER - 1 < n ? e0 + 256 * PER - 1 : u64(n) - 1;
  const u32 first = idx[e0] & ~15u, last = idx[last_e];
  for (u32 base = first; base <= last; base += STAGE) {
    __syncthreads();
    for (u32 o = threadIdx.x * 16; o < STAGE && base + o <= last; o += 256 * 16) { sh[o / 16] = *reinterpret_cast<const uint4 *>(buf + base + o); }
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
      const u32 r = pos[j] - base;
      if (r < STAGE) { d[j] = delta_of(reinterpret_cast<const u8 *>(sh)[r]); }
    }
  }
#pragma unroll
  for (u32 j = 0; j < PER; j++) { const u64 e = e0 + j * 256 + threadIdx.x; if (e < n) { out[e] = d[j]; } }
}

int main() {
  const size_t len = size_t(1) << 30;
  const u32 n = u32(len / 16);
  u8 *buf;
  u32 *idx, *ticket;
  int *out;
  u64 *desc;
  CK(hipMalloc(&buf, len + 4096));
  CK(hipMalloc(&idx, size_t(n) * 4 + 64));
  CK(hipMalloc(&out, size_t(n) * 4 + 64));
  const u32 tiles = (n + 4095) / 4096;
  CK(hipMalloc(&desc, size_t(tiles) * 8 + 64));
  ticket = reinterpret_cast<u32 *>(desc + tiles);
  {
    std::vector<u8> h(len);
    std::vector<u32> hi(n);
    u64 x = 88172645463325252ull;
    for (size_t i = 0; i < len; i++) { h[i] = 'a'; }
    for (u32 i = 0; i < n; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const u32 p = i * 16u + u32(x & 7u);
      hi[i] = p;
      h[p] = "{}[]:,\"\"1t"[(x >> 8) % 10];
    }
    CK(hipMemcpy(buf, h.data(), len, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, hi.data(), size_t(n) * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double moved = double(n) * 8 + double(len); // list + whole document (every line is touched) + int out
  printf("# %u entries over %zu bytes of document; GB/s counts 8 B per entry + the whole document (every 128-byte line holds entries)\n", n, len);
  auto timeit = [&](const char *name, auto launch) -> int {
    float best = 1e9f;
    for (int r = 0; r < 6; r++) {
      CK(hipMemsetAsync(desc, 0, size_t(tiles) * 8 + 64, 0));
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) { best = ms; }
    }
    printf("%-34s %8.1f us  %7.0f GB/s\n", name, best * 1e3, moved / best / 1e6);
    return 0;
  };
  const u32 g1 = (n + 255) / 256;
  for (int rep = 0; rep < 2; rep++) {
    timeit("map1", [&] { hipLaunchKernelGGL((k_map1<true, false>), dim3(g1), dim3(256), 0, 0, buf, idx, n, (void *)out); });
    timeit("map1 -g", [&] { hipLaunchKernelGGL((k_map1<false, false>), dim3(g1), dim3(256), 0, 0, buf, idx, n, (void *)out); });
    timeit("map1 b", [&] { hipLaunchKernelGGL((k_map1<true, true>), dim3(g1), dim3(256), 0, 0, buf, idx, n, (void *)out); });
    timeit("one", [&] { hipLaunchKernelGGL((k_one<1, true, true, true>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one -t", [&] { hipLaunchKernelGGL((k_one<1, true, true, false>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one -l", [&] { hipLaunchKernelGGL((k_one<1, true, false, true>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one -l -t", [&] { hipLaunchKernelGGL((k_one<1, true, false, false>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one -g", [&] { hipLaunchKernelGGL((k_one<1, false, true, true>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one -g -l -t", [&] { hipLaunchKernelGGL((k_one<1, false, false, false>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one4", [&] { hipLaunchKernelGGL((k_one<4, true, true, true>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("one4 -l -t", [&] { hipLaunchKernelGGL((k_one<4, true, false, false>), dim3(tiles), dim3(256), 0, 0, buf, idx, n, out, desc, ticket); });
    timeit("lds", [&] { hipLaunchKernelGGL(k_lds, dim3((n + 1023) / 1024), dim3(256), 0, 0, buf, idx, n, out); });
  }
  return 0;
}
