// scripts/micro/pcie_overlap.hip -- what the host-buffer (plug-in) path can hope for on this box:
// pageable vs pinned vs registered copies in each direction, both directions at once, and the price of
// hipHostRegister.  hipcc --offload-arch=gfx950 -O2 pcie_overlap.hip -o pcie_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const size_t N = size_t(1) << 30;
  char *d_a, *d_b;
  CK(hipMalloc(&d_a, N)); CK(hipMalloc(&d_b, N));
  char *pageable_in = static_cast<char *>(aligned_alloc(4096, N)), *pageable_out = static_cast<char *>(aligned_alloc(4096, N));
  memset(pageable_in, 1, N); memset(pageable_out, 2, N);
  char *pinned_in, *pinned_out;
  double t0 = now();
  CK(hipHostMalloc(&pinned_in, N)); 
  printf("hipHostMalloc 1 GiB: %.1f ms\n", (now() - t0) * 1e3);
  CK(hipHostMalloc(&pinned_out, N));
  memset(pinned_in, 1, N); memset(pinned_out, 2, N);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto rate = [&](const char *name, auto fn, double bytes) {
    fn(); CK(hipDeviceSynchronize());
    double best = 1e9;
    for (int i = 0; i < 3; i++) { double t = now(); fn(); CK(hipDeviceSynchronize()); best = std::min(best, now() - t); }
    printf("%-58s %7.2f ms  %6.1f GB/s\n", name, best * 1e3, bytes / best / 1e9);
  };
  rate("H2D pageable 1 GiB (hipMemcpyAsync)", [&] { CK(hipMemcpyAsync(d_a, pageable_in, N, hipMemcpyHostToDevice, s1)); }, N);
  rate("D2H pageable 1 GiB (hipMemcpyAsync)", [&] { CK(hipMemcpyAsync(pageable_out, d_b, N, hipMemcpyDeviceToHost, s2)); }, N);
  rate("H2D pinned 1 GiB", [&] { CK(hipMemcpyAsync(d_a, pinned_in, N, hipMemcpyHostToDevice, s1)); }, N);
  rate("D2H pinned 1 GiB", [&] { CK(hipMemcpyAsync(pinned_out, d_b, N, hipMemcpyDeviceToHost, s2)); }, N);
  rate("H2D + D2H pinned, two streams (2 GiB moved)", [&] {
    CK(hipMemcpyAsync(d_a, pinned_in, N, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(pinned_out, d_b, N, hipMemcpyDeviceToHost, s2)); }, 2.0 * N);
  rate("H2D pageable + D2H pinned, one thread (2 GiB moved)", [&] {
    CK(hipMemcpyAsync(pinned_out, d_b, N, hipMemcpyDeviceToHost, s2));
    CK(hipMemcpyAsync(d_a, pageable_in, N, hipMemcpyHostToDevice, s1)); }, 2.0 * N);
  rate("H2D pageable + D2H pageable, two threads (2 GiB moved)", [&] {
    std::thread t([&] { CK(hipMemcpyAsync(pageable_out, d_b, N, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
    CK(hipMemcpyAsync(d_a, pageable_in, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
    t.join(); }, 2.0 * N);
  // chunked: 16 MiB pieces, pageable H2D from one thread while pinned D2H pieces are in flight
  rate("64 x 16 MiB: H2D pageable chunk, D2H pinned chunk behind it", [&] {
    const size_t C = size_t(16) << 20;
    for (size_t o = 0; o < N; o += C) {
      CK(hipMemcpyAsync(d_a + o, pageable_in + o, C, hipMemcpyHostToDevice, s1));
      CK(hipMemcpyAsync(pinned_out + o, d_b + o, C, hipMemcpyDeviceToHost, s2));
    } }, 2.0 * N);
  rate("64 x 16 MiB: H2D pinned chunk, D2H pinned chunk behind it", [&] {
    const size_t C = size_t(16) << 20;
    for (size_t o = 0; o < N; o += C) {
      CK(hipMemcpyAsync(d_a + o, pinned_in + o, C, hipMemcpyHostToDevice, s1));
      CK(hipMemcpyAsync(pinned_out + o, d_b + o, C, hipMemcpyDeviceToHost, s2));
    } }, 2.0 * N);
  for (size_t sz : {size_t(16) << 20, size_t(256) << 20, size_t(1) << 30}) {
    double t = now();
    CK(hipHostRegister(pageable_out, sz, hipHostRegisterDefault));
    double reg = now() - t;
    t = now();
    CK(hipHostUnregister(pageable_out));
    printf("hipHostRegister %4zu MiB: %.2f ms, unregister %.2f ms\n", sz >> 20, reg * 1e3, (now() - t) * 1e3);
  }
  CK(hipHostRegister(pageable_out, N, hipHostRegisterDefault));
  rate("D2H into REGISTERED malloc memory 1 GiB", [&] { CK(hipMemcpyAsync(pageable_out, d_b, N, hipMemcpyDeviceToHost, s2)); }, N);
  rate("H2D pageable + D2H registered, one thread (2 GiB moved)", [&] {
    CK(hipMemcpyAsync(pageable_out, d_b, N, hipMemcpyDeviceToHost, s2));
    CK(hipMemcpyAsync(d_a, pageable_in, N, hipMemcpyHostToDevice, s1)); }, 2.0 * N);
  // small copies: latency of the single-shot path
  for (size_t sz : {size_t(64) << 10, size_t(1) << 20}) {
    double t = now();
    for (int i = 0; i < 200; i++) { CK(hipMemcpyAsync(d_a, pageable_in, sz, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
    double a = (now() - t) / 200;
    t = now();
    for (int i = 0; i < 200; i++) { CK(hipMemcpyAsync(d_a, pinned_in, sz, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
    printf("H2D %5zu KiB + sync: pageable %.1f us, pinned %.1f us\n", sz >> 10, a * 1e6, (now() - t) / 200 * 1e6);
  }
  return 0;
}
