// scripts/micro/stream_lab.hip -- how a stage-1 wave should fetch its 4 KiB chunk.  One wave = one chunk at a time (lane L works on the 64 bytes of block L, as
// in the product kernels), K dependent VALU instructions of stand-in work per chunk, `ratio` bytes written per byte read (16-byte coalesced stores).  Load paths:
//   V0  what the kernels do: four 16-byte loads per lane at a lane stride of 64 bytes (every instruction touches all 32 lines of the chunk, a quarter of each)
//   V1  coalesced plain loads (an instruction = 1 KiB contiguous) into registers, block-per-lane order restored through a swizzled LDS buffer
//   V2  the same with the non-temporal hint on the loads
//   V3  global_load_lds_dwordx4 nt straight into the LDS buffer (the swizzle is in the global address), the NEXT chunk requested as soon as this one is in registers
//   V4  V3 without nt
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_lab.hip -o scripts/micro/stream_lab.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 4;

__device__ __forceinline__ uint32_t slot_of(uint32_t b, uint32_t q) { return b * 4u + ((q + (b >> 2)) & 3u); } // 16-byte slot of quarter q of block b (of the 64 in a chunk)

template <int MODE>
__global__ __launch_bounds__(64 * WAVES) void k_lab(const v4 *__restrict__ in, v4 *__restrict__ out, uint32_t nchunks, uint32_t out_vecs, int K) {
  __shared__ v4 lds[WAVES][256];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t stride = gridDim.x * WAVES;
  v4 *const mine = lds[wave];
  uint32_t c = blockIdx.x * WAVES + wave;
  auto request = [&](uint32_t chunk) { // V3 / V4: LDS slot j * 64 + lane receives the piece that belongs there
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      const uint32_t s = j * 64u + lane, b = s >> 2, q = ((s & 3u) - (b >> 2)) & 3u;
      const v4 *src = in + size_t(chunk) * 256u + b * 4u + q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(mine + j * 64u), 16, 0, MODE == 3 ? 2 : 0);
    }
  };
  if ((MODE == 3 || MODE == 4) && c < nchunks) { request(c); }
  for (; c < nchunks; c += stride) {
    v4 w[4];
    if (MODE == 0) {
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) { w[q] = in[size_t(c) * 256u + lane * 4u + q]; }
    } else if (MODE == 1 || MODE == 2) {
      v4 r[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const v4 *src = in + size_t(c) * 256u + j * 64u + lane;
        r[j] = MODE == 2 ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t p = j * 64u + lane;
        mine[slot_of(p >> 2, p & 3u)] = r[j];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) { w[q] = mine[slot_of(lane, q)]; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the chunk has landed
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) { w[q] = mine[slot_of(lane, q)]; }
      __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): it is in registers
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (c + stride < nchunks) { request(c + stride); }
    }
    v4 acc = w[0] ^ w[1] ^ w[2] ^ w[3];
    uint32_t x = acc.x ^ acc.y, y = acc.z ^ acc.w;
#pragma unroll 8
    for (int k = 0; k < K; k += 2) { x = (x << 1) ^ y; y = (y >> 1) + x; }
    v4 *dst = out + size_t(c) * out_vecs;
    for (uint32_t o = lane; o < out_vecs; o += 64) { dst[o] = v4{x, y, o, c}; }
  }
}

int main() {
  const size_t in_bytes = size_t(1) << 30;
  const uint32_t nchunks = uint32_t(in_bytes / 4096);
  v4 *in, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, size_t(3) << 29));
  CK(hipMemset(in, 1, in_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  typedef void (*kern)(const v4 *, v4 *, uint32_t, uint32_t, int);
  struct var { const char *name; kern k; };
  const var vars[] = {{"V0 strided (product)", k_lab<0>}, {"V1 coalesced + LDS", k_lab<1>}, {"V2 coalesced nt + LDS", k_lab<2>}, {"V3 load_lds nt, ahead", k_lab<3>}, {"V4 load_lds, ahead", k_lab<4>}};
  struct mix { const char *name; double ratio; };
  const mix mixes[] = {{"read only", 0.0}, {"masks 0.125", 0.125}, {"NDJSON 0.22", 0.2175}, {"large_random 1.22", 1.2174}};
  printf("# 1 GiB in 4 KiB chunks, one wave per chunk at a time, 4 waves per workgroup; ms = median of 7 interleaved rounds; GB/s = (read + written) / time\n");
  for (int K : {0, 300}) {
    for (const mix &m : mixes) {
      const uint32_t out_vecs = uint32_t(m.ratio * 256 + 0.5);
      const double bytes = double(in_bytes) + double(nchunks) * out_vecs * 16;
      for (int grid : {256 * 6, 256 * 8}) {
        float t[5][7];
        for (int rep = 0; rep < 7; rep++) {
          for (int v = 0; v < 5; v++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(vars[v].k, dim3(grid), dim3(64 * WAVES), 0, 0, in, out, nchunks, out_vecs, K);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t[v][rep], e0, e1));
          }
        }
        for (int v = 0; v < 5; v++) {
          for (int a = 0; a < 7; a++) { for (int b = a + 1; b < 7; b++) { if (t[v][b] < t[v][a]) { float x = t[v][a]; t[v][a] = t[v][b]; t[v][b] = x; } } }
          printf("K %3d  %-18s grid %5d  %-22s %7.3f ms  %6.0f GB/s\n", K, m.name, grid, vars[v].name, t[v][3], bytes / (t[v][3] * 1e-3) / 1e9);
        }
      }
    }
  }
  return 0;
}
