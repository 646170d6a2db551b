// scripts/lab/perf_lab.cpp -- development loop on the GPU box: every workload x op x pipeline through the C-ABI
// (libsjgpu.so), parity at FULL size against the real reference (oracle/_ref/libsjref.so: exact compare of all n+3
// index words / all minified bytes), GPU time per call from libsjgpu's own HIP events.  One run = a few seconds, no
// Python start-up.  Diagnostics, not product code; bench.py is the contract, this is the stopwatch next to the bench.
//   g++ -O2 -std=c++17 scripts/lab/perf_lab.cpp -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
//       -Lsimdjson_amd/lib -lsjgpu -lsjcorpus -Loracle/_ref -lsjref -L/opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,'$ORIGIN/../../simdjson_amd/lib' -Wl,-rpath,'$ORIGIN/../../oracle/_ref' -o build/lab/perf_lab
#include "sjgpu.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
size_t sjc_large_random(uint8_t *, size_t, size_t, uint64_t, uint64_t *);
size_t sjc_amazon_ndjson(uint8_t *, size_t, size_t, uint64_t, uint64_t *);
size_t sjc_twitter_like(uint8_t *, size_t, size_t, uint64_t, uint64_t *);
int sjref_available(const char *);
int sjref_stage1(const char *, const uint8_t *, size_t, int, size_t, uint32_t *, uint32_t *);
int sjref_minify(const char *, const uint8_t *, size_t, uint8_t *, size_t *);
int sjref_validate_utf8(const char *, const uint8_t *, size_t);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

static std::vector<uint8_t> make(const std::string &kind, size_t target) {
  std::vector<uint8_t> v(target + (1 << 20));
  uint64_t units = 0;
  size_t n = 0;
  if (kind == "large_random") { n = sjc_large_random(v.data(), v.size(), target, 1000, &units); }
  else if (kind == "amazon_ndjson") { n = sjc_amazon_ndjson(v.data(), v.size(), target, 1000, &units); }
  else if (kind == "twitter_like") { n = sjc_twitter_like(v.data(), v.size(), target, 1000, &units); }
  else if (kind == "deep_nesting") { n = target; std::fill(v.begin(), v.begin() + n / 2, uint8_t('[')); std::fill(v.begin() + n / 2, v.begin() + n, uint8_t(']')); }
  else if (kind == "escape_heavy") { // strings that are backslash runs of many lengths (the shape of corpus.escape_heavy)
    static const int runs[] = {1, 2, 3, 4, 7, 8, 62, 63, 64, 65, 66, 126, 127, 128, 129, 130, 1, 1, 2, 2, 4094, 4095, 4096, 4097, 4098,
                               16382, 16383, 16384, 16385, 16386, 65535, 65536, 65537, 3, 5, 9, 17, 33, 1, 2, 1, 2};
    v[n++] = '[';
    while (n < target) {
      for (int r : runs) {
        if (n + size_t(r) + 8 >= v.size()) { break; }
        v[n++] = '"';
        std::memset(v.data() + n, '\\', size_t(r));
        n += size_t(r);
        v[n++] = '"';
        if (r % 2) { v[n++] = 'x'; v[n++] = '"'; }
        v[n++] = ',';
      }
    }
    v[n - 1] = ']';
  } else if (kind == "minified_twitter") { // no control characters at all: every span keeps both in-string hypotheses
    std::vector<uint8_t> t(target + (1 << 20));
    size_t m = sjc_twitter_like(t.data(), t.size(), target, 1000, &units), out = 0;
    sjref_minify("haswell", t.data(), m, v.data(), &out);
    n = out;
  } else if (kind == "cjk_text") { // array of strings of 3-byte characters: every block holds non-ASCII bytes
    v[n++] = '[';
    while (n + 400 < target) {
      v[n++] = '"';
      for (int k = 0; k < 100; k++) { v[n++] = 0xE6; v[n++] = uint8_t(0x97 + (k & 7)); v[n++] = uint8_t(0xA5 + (k % 23)); }
      v[n++] = '"'; v[n++] = ','; v[n++] = '\n';
    }
    v[n++] = '0'; v[n++] = ']';
  }
  v.resize(n);
  return v;
}

int main(int argc, char **argv) {
  size_t target = size_t(1) << 30;
  std::vector<std::string> kinds = {"amazon_ndjson", "large_random", "twitter_like", "escape_heavy", "deep_nesting", "minified_twitter", "cjk_text"};
  std::vector<std::string> ops = {"stage1", "minify", "validate_utf8"};
  std::vector<int> pipelines = {SJGPU_PIPELINE_FUSED, SJGPU_PIPELINE_SPLIT};
  int reps = 10;
  bool check = true, trace = false;
  int stress = 0;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--size" && i + 1 < argc) { target = std::strtoull(argv[++i], nullptr, 10); }
    else if (a == "--kinds" && i + 1 < argc) { kinds.clear(); std::string s = argv[++i]; size_t p = 0; while (p <= s.size()) { size_t q = s.find(',', p); if (q == std::string::npos) q = s.size(); kinds.push_back(s.substr(p, q - p)); p = q + 1; } }
    else if (a == "--ops" && i + 1 < argc) { ops.clear(); std::string s = argv[++i]; size_t p = 0; while (p <= s.size()) { size_t q = s.find(',', p); if (q == std::string::npos) q = s.size(); ops.push_back(s.substr(p, q - p)); p = q + 1; } }
    else if (a == "--fused-only") { pipelines = {SJGPU_PIPELINE_FUSED}; }
    else if (a == "--auto") { pipelines = {SJGPU_PIPELINE_AUTO}; }
    else if (a == "--reps" && i + 1 < argc) { reps = std::atoi(argv[++i]); }
    else if (a == "--no-check") { check = false; }
    else if (a == "--trace") { trace = true; }
    else if (a == "--stress" && i + 1 < argc) { stress = std::atoi(argv[++i]); }
  }
  const char *impl = sjref_available("icelake") ? "icelake" : "haswell";
  sjgpu_ctx *ctx = nullptr;
  int rc = sjgpu_ctx_create(0, target + (2 << 20), &ctx);
  if (rc) { std::printf("sjgpu_ctx_create failed: %d\n", rc); return 1; }
  uint8_t *d_buf = nullptr, *d_out = nullptr;
  uint32_t *d_idx = nullptr;
  CK(hipMalloc(&d_buf, target + (2 << 20)));
  CK(hipMalloc(&d_idx, (target + (2 << 20) + 16) * sizeof(uint32_t)));
  CK(hipMalloc(&d_out, target + (2 << 20) + 64));
  std::vector<uint32_t> ref_idx, got_idx;
  std::vector<uint8_t> ref_out, got_out;
  int failures = 0;
  std::printf("%-18s %-14s %-6s %9s %9s %9s %7s  %s\n", "workload", "op", "pipe", "gpu ms", "in GB/s", "alg GB/s", "frac", "parity vs reference");
  for (const std::string &kind : kinds) {
    std::vector<uint8_t> host = make(kind, target);
    const size_t L = host.size();
    CK(hipMemcpy(d_buf, host.data(), L, hipMemcpyHostToDevice));
    uint32_t ref_n = 0;
    int ref_err = 0, ref_merr = 0, ref_utf8 = 1;
    size_t ref_mlen = 0;
    if (check) {
      ref_idx.assign(L + 80, 0);
      ref_err = sjref_stage1(impl, host.data(), L, 0, 0, ref_idx.data(), &ref_n);
      ref_out.assign(L + 64, 0);
      ref_merr = sjref_minify(impl, host.data(), L, ref_out.data(), &ref_mlen);
      ref_utf8 = sjref_validate_utf8(impl, host.data(), L);
    }
    if (trace) { // phase budget of the pipelined single-pass kernel (in-kernel wall-clock stamps, 100 MHz)
      const uint32_t max_records = 2048 * 32;
      std::vector<uint64_t> t(size_t(max_records) * 8);
      uint32_t wgs = 0;
      for (int rep = 0; rep < 2; rep++) { rc = sjgpu_debug_trace_pipelined(ctx, d_buf, L, d_idx, L + 3, t.data(), max_records, &wgs); }
      if (rc) { std::printf("trace failed: %d\n", rc); }
      else {
        static const char *names[7] = {"ticket", "scan (wave 0)", "wait for the other waves", "publish + look-back", "broadcast", "emit (wave 0)", "park masks"};
        double sum[7] = {0, 0, 0, 0, 0, 0, 0}, iter_sum = 0;
        uint64_t nrec = 0, t_min = ~0ull, t_max = 0;
        for (uint32_t wgi = 0; wgi < wgs; wgi++) {
          for (uint32_t it = 0; it < 32; it++) {
            const uint64_t *r = &t[(size_t(wgi) * 32 + it) * 8];
            if (!r[0] || !r[7] || !r[4]) { continue; } // incomplete record (first / last iterations skip phases)
            bool ok = true;
            for (int k = 0; k < 7; k++) { if (r[k + 1] < r[k]) { ok = false; } }
            if (!ok) { continue; }
            for (int k = 0; k < 7; k++) { sum[k] += double(r[k + 1] - r[k]) / 100.0; }
            iter_sum += double(r[7] - r[0]) / 100.0;
            nrec++;
            t_min = std::min(t_min, r[0]);
            t_max = std::max(t_max, r[7]);
          }
        }
        std::printf("%-18s pipelined-kernel trace: %u workgroups, %llu complete iterations, span %.1f us\n", kind.c_str(), wgs, (unsigned long long)nrec, double(t_max - t_min) / 100.0);
        for (int k = 0; k < 7; k++) { std::printf("    %-28s mean %7.2f us\n", names[k], sum[k] / double(nrec ? nrec : 1)); }
        std::printf("    %-28s mean %7.2f us\n", "iteration", iter_sum / double(nrec ? nrec : 1));
      }
    }
    for (const std::string &op : ops) {
      for (int pl : pipelines) {
        if (op == "validate_utf8" && pl == SJGPU_PIPELINE_SPLIT && pipelines.size() > 1) { continue; }
        sjgpu_set_pipeline(ctx, pl);
        auto call = [&]() -> int {
          if (op == "stage1") { return sjgpu_stage1_device(ctx, d_buf, L, d_idx, L + 3, nullptr); }
          if (op == "minify") { return sjgpu_minify_device(ctx, d_buf, L, d_out, nullptr); }
          return sjgpu_validate_utf8_device(ctx, d_buf, L, nullptr);
        };
        sjgpu_scan_result res{};
        if (stress && check && op != "validate_utf8") { // every repetition compared in full: is the kernel deterministic?
          int bad = 0;
          for (int r = 0; r < stress; r++) {
            rc = call();
            sjgpu_result(ctx, nullptr, &res);
            bool ok = rc == 0;
            if (ok && op == "stage1") {
              ok = res.n == ref_n && (res.flags & ~SJGPU_F_UTF8_ERROR) == 0;
              if (ok) {
                got_idx.resize(size_t(res.n) + 3);
                CK(hipMemcpy(got_idx.data(), d_idx, got_idx.size() * 4, hipMemcpyDeviceToHost));
                ok = std::memcmp(got_idx.data(), ref_idx.data(), got_idx.size() * 4) == 0;
              }
            } else if (ok) {
              ok = res.out_len == ref_mlen && res.flags == 0;
              if (ok) {
                got_out.resize(res.out_len);
                CK(hipMemcpy(got_out.data(), d_out, res.out_len, hipMemcpyDeviceToHost));
                ok = std::memcmp(got_out.data(), ref_out.data(), res.out_len) == 0;
              }
            }
            if (!ok) { bad++; std::printf("  stress %s %s rep %d: MISMATCH n %u out_len %llu flags %u\n", kind.c_str(), op.c_str(), r, res.n, (unsigned long long)res.out_len, res.flags); }
          }
          std::printf("%-18s %-14s %-6s stress: %d of %d repetitions wrong\n", kind.c_str(), op.c_str(), pl == SJGPU_PIPELINE_FUSED ? "fused" : "split", bad, stress);
          failures += bad;
          continue;
        }
        for (int w = 0; w < 2; w++) { rc = call(); sjgpu_result(ctx, nullptr, &res); }
        if (rc != 0) { std::printf("%-18s %-14s call failed: %d (%s)\n", kind.c_str(), op.c_str(), rc, sjgpu_last_error(ctx)); failures++; continue; }
        const int used = sjgpu_last_pipeline(ctx);
        sjgpu_profile_enable(ctx, 1);
        for (int r = 0; r < reps; r++) { call(); }
        double ms[4] = {0, 0, 0, 0};
        uint32_t calls = 0;
        sjgpu_profile_read(ctx, ms, &calls);
        sjgpu_profile_enable(ctx, 0);
        sjgpu_result(ctx, nullptr, &res);
        const double gpu_ms = (ms[0] + ms[1] + ms[2]) / std::max(1u, calls);
        double alg = double(L);
        std::string verdict = "-";
        if (op == "stage1") {
          alg += 4.0 * (double(res.n) + 3);
          if (check) {
            const int err = sjgpu_stage1_error_from_flags(res.n, res.flags);
            bool ok = err == ref_err && (res.flags & (SJGPU_F_INTERNAL | SJGPU_F_IDX_OVERFLOW)) == 0;
            if (ok && (err == 0 || err == 11)) {
              ok = res.n == ref_n;
              if (ok) {
                got_idx.resize(size_t(res.n) + 3);
                CK(hipMemcpy(got_idx.data(), d_idx, got_idx.size() * 4, hipMemcpyDeviceToHost));
                ok = std::memcmp(got_idx.data(), ref_idx.data(), got_idx.size() * 4) == 0;
              }
            }
            verdict = ok ? "exact (n = " + std::to_string(res.n) + ", err " + std::to_string(err) + ")" : "MISMATCH: err " + std::to_string(err) + " vs " + std::to_string(ref_err) + ", n " + std::to_string(res.n) + " vs " + std::to_string(ref_n) + ", flags " + std::to_string(res.flags);
            if (!ok) { failures++; }
          }
        } else if (op == "minify") {
          alg += double(res.out_len);
          if (check) {
            const int err = (res.flags & SJGPU_F_UNCLOSED_STRING) ? 15 : 0;
            bool ok = err == ref_merr && res.out_len == ref_mlen && (res.flags & SJGPU_F_INTERNAL) == 0;
            if (ok) {
              got_out.resize(res.out_len);
              CK(hipMemcpy(got_out.data(), d_out, res.out_len, hipMemcpyDeviceToHost));
              ok = std::memcmp(got_out.data(), ref_out.data(), res.out_len) == 0;
            }
            verdict = ok ? "exact (" + std::to_string(res.out_len) + " bytes)" : "MISMATCH: len " + std::to_string(res.out_len) + " vs " + std::to_string(ref_mlen) + ", flags " + std::to_string(res.flags);
            if (!ok) { failures++; }
          }
        } else if (check) {
          const int okv = (res.flags & SJGPU_F_UTF8_ERROR) ? 0 : 1;
          verdict = okv == ref_utf8 ? "same verdict (" + std::to_string(okv) + ")" : "MISMATCH";
          if (okv != ref_utf8) { failures++; }
        }
        std::printf("%-18s %-14s %-6s %9.4f %9.1f %9.1f %7.4f  %s", kind.c_str(), op.c_str(), used == 1 ? "fused" : "split", gpu_ms, L / gpu_ms / 1e6,
                    alg / gpu_ms / 1e6, alg / gpu_ms / 1e6 / 8000.0, verdict.c_str());
        if (used != 1 && op != "validate_utf8") { std::printf("  [slots %.4f %.4f %.4f]", ms[0] / calls, ms[1] / calls, ms[2] / calls); }
        std::printf("\n");
        std::fflush(stdout);
      }
    }
  }
  std::printf("%s\n", failures ? "LAB: FAILURES" : "LAB: all parity checks passed");
  sjgpu_ctx_destroy(ctx);
  return failures ? 1 : 0;
}
