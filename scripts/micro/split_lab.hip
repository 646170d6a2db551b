// scripts/micro/split_lab.hip -- the memory traffic of the split pipeline and nothing else, call after call over the same buffers: K1 reads the input (1 GiB,
// coalesced 16-byte loads, plain or nt) and writes the masks (1/8 of it); K2 reads the masks (plain or nt) and writes `ratio` bytes of output per input byte.
// What this asks: where does the time go when the input is streamed -- session K/L saw the scan kernel gain and the emit kernel lose.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/split_lab.hip -o scripts/micro/split_lab.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int LD, int ST> // K1: per 16 KiB segment (one wave): read 1024 vectors, write 128 vectors of masks.  ST: 0 plain, 1 sc1, 2 nt
__global__ __launch_bounds__(256) void k1(const v4 *__restrict__ in, v4 *__restrict__ masks, uint32_t nseg) {
  const uint32_t lane = threadIdx.x & 63u, seg = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (seg >= nseg) { return; }
  v4 acc = {0, 0, 0, 0};
  for (uint32_t c = 0; c < 4; c++) {
    v4 r[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      const v4 *src = in + size_t(seg) * 1024u + c * 256u + j * 64u + lane;
      r[j] = LD ? __builtin_nontemporal_load(src) : *src;
    }
    acc ^= r[0] ^ r[1] ^ r[2] ^ r[3];
  }
#pragma unroll
  for (uint32_t h = 0; h < 2; h++) {
    v4 *dst = masks + size_t(seg) * 128u + h * 64u + lane;
    const v4 val = acc + v4{h, h, h, h};
    if (ST == 0) { *dst = val; }
    else if (ST == 1) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory"); }
    else { __builtin_nontemporal_store(val, dst); }
  }
}
template <int LD, int ST> // K2: per segment (one wave): read its 128 vectors of masks, write out_vecs vectors.  ST: 0 plain, 2 nt
__global__ __launch_bounds__(64) void k2(const v4 *__restrict__ masks, v4 *__restrict__ out, uint32_t nseg, uint32_t out_vecs) {
  const uint32_t lane = threadIdx.x, seg = blockIdx.x;
  const v4 *src = masks + size_t(seg) * 128u + lane;
  const v4 a = LD ? __builtin_nontemporal_load(src) : src[0], b = LD ? __builtin_nontemporal_load(src + 64) : src[64];
  const v4 acc = a ^ b;
  v4 *dst = out + size_t(seg) * out_vecs;
  for (uint32_t o = lane; o < out_vecs; o += 64) {
    const v4 val = acc + v4{o, o, o, o};
    if (ST == 0) { dst[o] = val; } else { __builtin_nontemporal_store(val, dst + o); }
  }
}

int main() {
  const size_t in_bytes = size_t(1) << 30;
  const uint32_t nseg = uint32_t(in_bytes / 16384);
  v4 *in, *masks, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&masks, in_bytes / 8));
  CK(hipMalloc(&out, size_t(3) << 29));
  CK(hipMemset(in, 1, in_bytes));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  typedef void (*kern1)(const v4 *, v4 *, uint32_t);
  typedef void (*kern2)(const v4 *, v4 *, uint32_t, uint32_t);
  struct var { const char *name; kern1 a; kern2 b; };
  const var vars[] = {{"plain / plain", k1<0, 0>, k2<0, 0>}, {"nt input / plain", k1<1, 0>, k2<0, 0>}, {"nt input / nt masks", k1<1, 0>, k2<1, 0>},
                      {"nt input, sc1 masks / plain", k1<1, 1>, k2<0, 0>}, {"nt input, sc1 masks / nt masks", k1<1, 1>, k2<1, 0>}, {"nt input, nt mask st / nt masks", k1<1, 2>, k2<1, 0>},
                      {"nt input / nt masks, nt out", k1<1, 0>, k2<1, 2>}, {"plain / nt masks", k1<0, 0>, k2<1, 0>}};
  const int NV = 8;
  struct mix { const char *name; double ratio; };
  const mix mixes[] = {{"escape_heavy ~0", 0.002}, {"NDJSON 0.22", 0.2175}, {"twitter 0.49", 0.49}, {"large_random 1.22", 1.2174}};
  printf("# per call: K1 reads 1 GiB + writes 128 MiB of masks, K2 reads the masks + writes ratio x 1 GiB; us = median of 9 interleaved rounds of 4 calls; [K1, K2] by events\n");
  for (const mix &m : mixes) {
    const uint32_t out_vecs = uint32_t(m.ratio * 1024 + 0.5);
    float t[NV][9], t1[NV][9];
    for (int rep = 0; rep < 9; rep++) {
      for (int v = 0; v < NV; v++) {
        float sum = 0, sum1 = 0;
        for (int call = 0; call < 5; call++) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(vars[v].a, dim3((nseg + 3) / 4), dim3(256), 0, 0, in, masks, nseg);
          CK(hipEventRecord(e1, 0));
          hipLaunchKernelGGL(vars[v].b, dim3(nseg), dim3(64), 0, 0, masks, out, nseg, out_vecs);
          CK(hipEventRecord(e2, 0));
          CK(hipEventSynchronize(e2));
          float a, b;
          CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
          if (call) { sum += b; sum1 += a; } // (the first call of a variant meets the caches the variant before left)
        }
        t[v][rep] = sum / 4; t1[v][rep] = sum1 / 4;
      }
    }
    for (int v = 0; v < NV; v++) {
      for (int a = 0; a < 9; a++) { for (int b = a + 1; b < 9; b++) { if (t[v][b] < t[v][a]) { float x = t[v][a]; t[v][a] = t[v][b]; t[v][b] = x; x = t1[v][a]; t1[v][a] = t1[v][b]; t1[v][b] = x; } } }
      printf("%-20s %-34s %7.1f us   [K1 %6.1f, K2 %6.1f]\n", m.name, vars[v].name, 1e3 * t[v][4], 1e3 * t1[v][4], 1e3 * (t[v][4] - t1[v][4]));
    }
  }
  return 0;
}
