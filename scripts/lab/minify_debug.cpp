// scripts/lab/minify_debug.cpp -- where does a single-pass minify go wrong?  Runs launch_minify_fused (the internal
// launcher, exported by libsjgpu.so) over one input again and again and checks EVERY tile descriptor the kernel left
// behind (inclusive prefix: in-string bit + output cursor behind the tile) against a scalar walk over the input.
// Diagnostics only (this is the tool that traced round 2's "one run in two is wrong" to a wave reading the ticket slot
// before the write had landed: hipcc had dropped the LDS wait in front of the loop-top barrier).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/lab/minify_debug.cpp -I include -I simdjson_amd/csrc -Lsimdjson_amd/lib \
//       -lsjgpu -lsjcorpus -Loracle/_ref -lsjref -Wl,-rpath,'$ORIGIN/../../simdjson_amd/lib' -Wl,-rpath,'$ORIGIN/../../oracle/_ref' \
//       -o build/lab/minify_debug
//   SJGPU_MINIFY_ONCHIP=4 build/lab/minify_debug <MiB> <repetitions> <tile bytes: 32768 | 65536> [minified_twitter | amazon_ndjson]
#include "sjgpu_internal.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
size_t sjc_twitter_like(uint8_t *, size_t, size_t, uint64_t, uint64_t *);
size_t sjc_amazon_ndjson(uint8_t *, size_t, size_t, uint64_t, uint64_t *);
int sjref_minify(const char *, const uint8_t *, size_t, uint8_t *, size_t *);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

int main(int argc, char **argv) {
  const size_t target = argc > 1 ? size_t(std::atoll(argv[1])) << 20 : size_t(1) << 30;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 10;
  const size_t tile_bytes = argc > 3 ? size_t(std::atoll(argv[3])) : 32768; // must match the kernel SJGPU_MINIFY_ONCHIP selects
  const std::string kind = argc > 4 ? argv[4] : "minified_twitter";
  std::vector<uint8_t> in(target + (1 << 20));
  uint64_t units = 0;
  size_t n = 0;
  if (kind == "minified_twitter") {
    std::vector<uint8_t> t(target + (1 << 20));
    const size_t m = sjc_twitter_like(t.data(), t.size(), target, 1000, &units);
    sjref_minify("haswell", t.data(), m, in.data(), &n);
  } else {
    n = sjc_amazon_ndjson(in.data(), in.size(), target, 1000, &units);
  }
  in.resize(n);
  const size_t ntiles = (n + tile_bytes - 1) / tile_bytes;
  // expected state behind every tile
  std::vector<uint8_t> exp_s(ntiles);
  std::vector<uint64_t> exp_kept(ntiles);
  {
    bool in_str = false, esc = false;
    uint64_t kept = 0;
    for (size_t i = 0; i < n; i++) {
      const uint8_t b = in[i];
      if (in_str) {
        kept++;
        if (esc) { esc = false; }
        else if (b == '\\') { esc = true; }
        else if (b == '"') { in_str = false; }
      } else if (b == '"') { in_str = true; kept++; }
      else if (!(b == ' ' || b == '\t' || b == '\n' || b == '\r')) { kept++; }
      if ((i + 1) % tile_bytes == 0 || i + 1 == n) { exp_s[i / tile_bytes] = in_str; exp_kept[i / tile_bytes] = kept; }
    }
    std::printf("%s: %zu bytes, %zu tiles of %zu, kept %llu, ends in string %d\n", kind.c_str(), n, ntiles, tile_bytes,
                (unsigned long long)kept, int(in_str));
  }
  uint8_t *d_in, *d_out, *d_esc, *d_ws;
  CK(hipMalloc(&d_in, n + 64));
  CK(hipMalloc(&d_out, n + 64));
  CK(hipMalloc(&d_esc, sjgpu::ESC_TABLE_BYTES));
  CK(hipMemset(d_esc, 0, sjgpu::ESC_TABLE_BYTES));
  CK(hipMalloc(&d_ws, sizeof(sjgpu::scan_result_dev) + (ntiles + 1) * 8));
  CK(hipMemcpy(d_in, in.data(), n, hipMemcpyHostToDevice));
  auto *result = reinterpret_cast<sjgpu::scan_result_dev *>(d_ws);
  auto *desc = reinterpret_cast<uint64_t *>(result + 1);
  std::vector<uint64_t> h_desc(ntiles + 1);
  int wrong_runs = 0;
  for (int rep = 0; rep < reps; rep++) {
    sjgpu::scan_origin org{0, 0, 0, d_esc};
    const char *name = sjgpu::launch_minify_fused(d_in, n, desc, d_out, result, org, 2048, nullptr, nullptr);
    CK(hipDeviceSynchronize());
    sjgpu::scan_result_dev r;
    CK(hipMemcpy(&r, result, sizeof r, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_desc.data(), desc, (ntiles + 1) * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, first = ntiles, bad_s = 0;
    for (size_t t = 0; t < ntiles; t++) {
      const uint64_t d = h_desc[t];
      const bool ok = (d >> 62) == 2 && ((d >> 32) & 1) == exp_s[t] && uint32_t(d) == uint32_t(exp_kept[t]);
      if (!ok) {
        bad++;
        if (((d >> 32) & 1) != exp_s[t]) { bad_s++; }
        if (first == ntiles) { first = t; }
      }
    }
    std::printf("rep %2d %s: out_len %llu flags %u tickets %u | wrong tiles %zu (in-string bit wrong in %zu), first %zu", rep, name,
                (unsigned long long)r.out_len, r.flags, uint32_t(h_desc[ntiles]), bad, bad_s, first);
    if (bad) {
      wrong_runs++;
      std::printf("\n");
      for (size_t t = first > 2 ? first - 2 : 0; t < ntiles && t < first + 6; t++) {
        const uint64_t d = h_desc[t];
        std::printf("    tile %zu: status %u s %u cursor %u | expected s %u cursor %llu (delta %lld)\n", t, unsigned(d >> 62), unsigned((d >> 32) & 1),
                    uint32_t(d), exp_s[t], (unsigned long long)exp_kept[t], (long long)uint32_t(d) - (long long)exp_kept[t]);
      }
      // where does the error change again?
      size_t shown = 0;
      long long prev_delta = 0;
      bool prev_s = false;
      for (size_t t = first; t < ntiles && shown < 12; t++) {
        const uint64_t d = h_desc[t];
        const long long delta = (long long)uint32_t(d) - (long long)exp_kept[t];
        const bool sw = ((d >> 32) & 1) != exp_s[t];
        if (t == first || delta != prev_delta || sw != prev_s) {
          std::printf("    change at tile %zu: cursor delta %lld, in-string bit %s\n", t, delta, sw ? "WRONG" : "right");
          shown++;
        }
        prev_delta = delta;
        prev_s = sw;
      }
    } else {
      std::printf(" -- all descriptors right\n");
    }
  }
  std::printf("%d of %d runs wrong\n", wrong_runs, reps);
  return 0;
}
