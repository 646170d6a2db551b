// tests/host/test_mgpu_emu.cpp -- CPU tier: sjgpu_mgpu_* (simdjson_amd/csrc/sjgpu_mgpu.hip: ONE host buffer scanned by SEVERAL GPUs, one host thread per
// device, one bit per shard exchanged through host memory) with DISTINCT devices on a box that has none.
//
// The GPU tier lists the one device a box has several times (tests/test_gpu_parity.py::test_mgpu_*): the sharding, the phases and the concatenation run,
// but a call made under the wrong current device cannot be told from a right one.  Here sjgpu_mgpu.hip is compiled as C++ against tests/host/emu with
// -DSJ_EMU_DEVICES=3: every thread has a current device, every allocation and stream belongs to the device that was current when it was made, and every
// HIP call that names one checks that it runs under its own device (tests/host/emu/hip/hip_runtime.h).  The per-device contexts behind the driver are
// MOCKS of the C-ABI entry points it calls (sjgpu_ctx_create, sjgpu_stage1_shard_device, sjgpu_minify_shard_device, sjgpu_result, ...): they check the
// same discipline -- the context's device is current, buffers and stream belong to it -- and answer from the oracle (test infrastructure), with the
// shard semantics of include/sjgpu.h ("one large document sharded across GPUs"): offsets relative to the shard, SJGPU_F_UNCLOSED_STRING = the shard ENDS
// inside a string.  The host-side pieces the driver shares with the product -- sjgpu_clean_cut, sjgpu_trim_partial_utf8, sjgpu_stage1_finish_host -- are
// the product's own (stage1_finish.cpp, compiled in).  Compared with: the oracle over the WHOLE document.
// Usage: test_mgpu_emu <seed> <documents>
#include "sjgpu.h"
#include "sj_oracle.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ---- the device bookkeeping of the emulator's several-device mode ------------------------------------------------------------------
namespace sj_emu {
int device_violations = 0;
static std::mutex g_m;
static std::map<const uint8_t *, std::pair<size_t, int>> g_allocs;
int &current_device() {
  static thread_local int d = 0;
  return d;
}
void note_alloc(const void *p, size_t n, int device) {
  std::lock_guard<std::mutex> lk(g_m);
  g_allocs[static_cast<const uint8_t *>(p)] = {n, device};
}
void drop_alloc(const void *p) {
  std::lock_guard<std::mutex> lk(g_m);
  g_allocs.erase(static_cast<const uint8_t *>(p));
}
int device_of(const void *p) {
  std::lock_guard<std::mutex> lk(g_m);
  const uint8_t *q = static_cast<const uint8_t *>(p);
  auto it = g_allocs.upper_bound(q);
  if (it == g_allocs.begin()) { return -1; }
  --it;
  return (q < it->first + it->second.first) ? it->second.second : -1;
}
void device_violation(const char *what, int expected, int current) {
  std::lock_guard<std::mutex> lk(g_m);
  device_violations++;
  std::fprintf(stderr, "DEVICE VIOLATION: %s (belongs to device %d, current device %d)\n", what, expected, current);
}
} // namespace sj_emu

// ---- mock contexts: the C-ABI entry points sjgpu_mgpu.hip calls, answered by the oracle --------------------------------------------
struct sjgpu_ctx {
  int device;
  size_t capacity;
  sjgpu_scan_result last;
};
static int g_live_contexts = 0;
static void check_under(const sjgpu_ctx *ctx, const void *a, const void *b, void *stream, const char *who) {
  if (sj_emu::current_device() != ctx->device) { sj_emu::device_violation(who, ctx->device, sj_emu::current_device()); }
  if (a && sj_emu::device_of(a) != ctx->device) { sj_emu::device_violation("input buffer of a shard call is not the context's device's memory", ctx->device, sj_emu::device_of(a)); }
  if (b && sj_emu::device_of(b) != ctx->device) { sj_emu::device_violation("output buffer of a shard call is not the context's device's memory", ctx->device, sj_emu::device_of(b)); }
  if (stream && static_cast<sj_emu::stream_rec *>(stream)->device != ctx->device) { sj_emu::device_violation("stream of a shard call belongs to another device", ctx->device, static_cast<sj_emu::stream_rec *>(stream)->device); }
}
extern "C" {
int sjgpu_ctx_create(int device, size_t capacity, sjgpu_ctx **out) {
  if (device < 0 || device >= SJ_EMU_DEVICES) { return SJGPU_E_NO_DEVICE; }
  *out = new sjgpu_ctx{device, capacity, {0, 0, 0}};
  std::lock_guard<std::mutex> lk(sj_emu::g_m);
  g_live_contexts++;
  return 0;
}
void sjgpu_ctx_destroy(sjgpu_ctx *ctx) {
  if (!ctx) { return; }
  delete ctx;
  std::lock_guard<std::mutex> lk(sj_emu::g_m);
  g_live_contexts--;
}
int sjgpu_set_capacity(sjgpu_ctx *ctx, size_t capacity) { ctx->capacity = capacity; return 0; }
size_t sjgpu_capacity(const sjgpu_ctx *ctx) { return ctx->capacity; }
int sjgpu_result(sjgpu_ctx *ctx, void *stream, sjgpu_scan_result *out) {
  check_under(ctx, nullptr, nullptr, stream, "sjgpu_result under another device");
  *out = ctx->last;
  return 0;
}
// a shard that begins inside a string = the same bytes behind an opening quote (the byte in front of a clean cut is whitespace or an operator:
// no escape, no scalar carries over), minus that quote in the output
int sjgpu_stage1_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *idx_dev, size_t idx_words, void *stream) {
  check_under(ctx, buf_dev, idx_dev, stream, "sjgpu_stage1_shard_device under another device");
  if (len > ctx->capacity) { return 1; }
  std::vector<uint8_t> tmp;
  if (in_string) { tmp.push_back('"'); }
  tmp.insert(tmp.end(), static_cast<const uint8_t *>(buf_dev), static_cast<const uint8_t *>(buf_dev) + len);
  std::vector<uint32_t> idx(tmp.size() + 1);
  uint32_t flags = 0;
  uint32_t n = sjo_scan(tmp.data(), tmp.size(), idx.data(), &flags);
  uint32_t first = 0;
  if (in_string) {
    if (n == 0 || idx[0] != 0) { return 24; }
    first = 1;
  }
  uint32_t *out = static_cast<uint32_t *>(idx_dev);
  uint32_t f = (flags & 1u ? SJGPU_F_UNCLOSED_STRING : 0u) | (flags & 2u ? SJGPU_F_UNESCAPED_CTRL : 0u) | (flags & 4u ? SJGPU_F_UTF8_ERROR : 0u);
  if (size_t(n - first) + 3 > idx_words) { f |= SJGPU_F_IDX_OVERFLOW; }
  else { for (uint32_t i = first; i < n; i++) { out[i - first] = idx[i] - first; } }
  ctx->last = {n - first, f, 0};
  return 0;
}
int sjgpu_minify_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *dst_dev, void *stream) {
  check_under(ctx, buf_dev, dst_dev, stream, "sjgpu_minify_shard_device under another device");
  std::vector<uint8_t> tmp;
  if (in_string) { tmp.push_back('"'); }
  tmp.insert(tmp.end(), static_cast<const uint8_t *>(buf_dev), static_cast<const uint8_t *>(buf_dev) + len);
  std::vector<uint32_t> idx(tmp.size() + 1);
  uint32_t flags = 0;
  (void)sjo_scan(tmp.data(), tmp.size(), idx.data(), &flags);
  const bool open = (flags & 1u) != 0;
  if (open) { tmp.push_back('"'); } // the oracle voids the output of a document that ends inside a string: close it, drop the quote again
  std::vector<uint8_t> dst(tmp.size() + 1);
  size_t n = 0;
  if (sjo_minify(tmp.data(), tmp.size(), dst.data(), &n) != 0) { return 24; }
  const size_t lo = in_string ? 1 : 0, hi = open ? n - 1 : n;
  std::memcpy(dst_dev, dst.data() + lo, hi - lo);
  ctx->last = {0, open ? SJGPU_F_UNCLOSED_STRING : 0u, uint64_t(hi - lo)};
  return 0;
}
int sjgpu_validate_utf8(sjgpu_ctx *, const uint8_t *buf, size_t len, int *ok) {
  *ok = sjo_validate_utf8(buf, len);
  return 0;
}
} // extern "C"

// ---- documents ------------------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 1;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return uint32_t(rng_state >> 33);
}
static uint32_t rnd_below(uint32_t n) { return n ? rnd() % n : 0; }
typedef std::vector<uint8_t> bytes;
static void put(bytes &d, const char *s) { d.insert(d.end(), s, s + std::strlen(s)); }
static void value(bytes &d, int depth) {
  const uint32_t r = rnd_below(depth > 4 ? 4 : 7);
  if (r == 0) { put(d, "-12.5e3"); }
  else if (r == 1) { put(d, "null"); }
  else if (r == 2 || r == 3) { // strings with clean-cut bait inside them: spaces, commas, brackets, escaped quotes, multi-byte characters
    d.push_back('"');
    const uint32_t n = rnd_below(rnd_below(4) ? 40 : 3000);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t k = rnd_below(16);
      if (k == 0) { put(d, "\\\""); } else if (k == 1) { put(d, " , "); } else if (k == 2) { put(d, "] {"); } else if (k == 3) { put(d, "\xe2\x82\xac"); } else if (k == 4) { put(d, "\\\\"); }
      else { d.push_back(uint8_t('a' + rnd_below(26))); }
    }
    d.push_back('"');
  } else if (r == 4) {
    d.push_back('[');
    const uint32_t n = rnd_below(8);
    for (uint32_t i = 0; i < n; i++) { if (i) { put(d, ", "); } value(d, depth + 1); }
    d.push_back(']');
  } else {
    d.push_back('{');
    const uint32_t n = rnd_below(6);
    for (uint32_t i = 0; i < n; i++) { if (i) { put(d, ",\n "); } put(d, "\"key "); d.push_back(uint8_t('0' + i)); put(d, "\": "); value(d, depth + 1); }
    d.push_back('}');
  }
}
static bytes make_document(size_t target, uint32_t kind) {
  bytes d;
  if (kind == 0) { d.push_back('['); }
  while (d.size() < target) {
    value(d, 0);
    put(d, kind == 0 ? ",\n" : "\n");
  }
  if (kind == 0) { put(d, "0]"); }
  if (kind == 2) { put(d, "\"ends inside a string , "); }          // UNCLOSED_STRING
  if (kind == 3) { d[d.size() / 2] = 0xFF; }                         // UTF8_ERROR somewhere in the middle
  if (kind == 4) { d.insert(d.begin() + long(d.size() / 3), {'"', 'a', 0x01, 'b', '"', ' '}); } // maybe a control character inside a string (or it closes one)
  return d;
}

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; std::fprintf(stderr, "MISMATCH: " __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)

int main(int argc, char **argv) {
  rng_state = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
  const int ndocs = argc > 2 ? std::atoi(argv[2]) : 40;
  static const std::vector<std::vector<int>> lists = {{0, 1}, {1, 0}, {0, 1, 2}, {2, 0, 1, 2}, {1}, {2, 2, 1}};
  size_t calls = 0, total_bytes = 0;
  for (int doc = 0; doc < ndocs; doc++) {
    const bytes d = make_document(2000 + rnd_below(200000), rnd_below(5));
    const size_t len = d.size();
    total_bytes += len;
    // the whole document through the oracle
    std::vector<uint32_t> oidx(len + 3);
    std::vector<uint8_t> omin(len + 1);
    size_t omin_len = 0;
    const int omin_err = sjo_minify(d.data(), len, omin.data(), &omin_len);
    const int outf = sjo_validate_utf8(d.data(), len);
    for (const std::vector<int> &devs : lists) {
      sjgpu_mgpu *m = nullptr;
      CHECK(sjgpu_mgpu_create(devs.data(), int(devs.size()), &m) == 0 && m && sjgpu_mgpu_count(m) == int(devs.size()), "create with %zu devices", devs.size());
      if (!m) { continue; }
      for (int mode : {int(SJGPU_REGULAR), int(SJGPU_STREAMING_FINAL), int(SJGPU_STREAMING_PARTIAL)}) {
        uint32_t on = 0;
        std::fill(oidx.begin(), oidx.end(), 0u);
        const int oerr = sjo_stage1(d.data(), len, mode, len, oidx.data(), &on);
        std::vector<uint32_t> idx(len + 3, 0xEEEEEEEEu);
        uint32_t n = 0, next = 0;
        const int err = sjgpu_mgpu_stage1(m, d.data(), len, mode, idx.data(), len + 3, &n, &next);
        calls++;
        CHECK(err == oerr, "doc %d, %zu devices, mode %d: error %d vs oracle %d", doc, devs.size(), mode, err, oerr);
        if (err == 0 && oerr == 0) {
          size_t bad = 0;
          const size_t words = size_t(on) + (mode == SJGPU_REGULAR ? 3 : 0);
          while (bad < words && idx[bad] == oidx[bad]) { bad++; }
          CHECK(n == on && bad == words, "doc %d, %zu devices, mode %d: n %u vs %u; first difference at word %zu of %zu: %u vs %u", doc, devs.size(), mode, n, on, bad, words,
                bad < words ? idx[bad] : 0u, bad < words ? oidx[bad] : 0u);
        }
      }
      std::vector<uint8_t> dst(len + 1, 0xEE);
      size_t dst_len = 99;
      const int merr = sjgpu_mgpu_minify(m, d.data(), len, dst.data(), &dst_len);
      calls++;
      CHECK(merr == omin_err && dst_len == omin_len && std::memcmp(dst.data(), omin.data(), omin_len) == 0, "doc %d, %zu devices: minify %d / %zu vs oracle %d / %zu", doc,
            devs.size(), merr, dst_len, omin_err, omin_len);
      int ok = -1;
      CHECK(sjgpu_mgpu_validate_utf8(m, d.data(), len, &ok) == 0 && ok == outf, "doc %d, %zu devices: validate_utf8 %d vs oracle %d", doc, devs.size(), ok, outf);
      calls++;
      sjgpu_mgpu_destroy(m);
    }
  }
  CHECK(g_live_contexts == 0, "%d contexts were never destroyed", g_live_contexts);
  {
    std::lock_guard<std::mutex> lk(sj_emu::g_m);
    CHECK(sj_emu::g_allocs.empty(), "%zu device allocations were never freed", sj_emu::g_allocs.size());
  }
  CHECK(sj_emu::device_violations == 0, "%d calls ran under the wrong device", sj_emu::device_violations);
  // the checker is not vacuous: a copy on device 1's stream while device 0 is current must be found
  {
    const int before = sj_emu::device_violations;
    (void)hipSetDevice(1);
    hipStream_t s1 = nullptr;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    void *p1 = nullptr;
    (void)hipMalloc(&p1, 64);
    (void)hipSetDevice(0);
    uint8_t host[64] = {0};
    std::fprintf(stderr, "(the two violations that follow are the self-test of the checker)\n");
    (void)hipMemcpyAsync(p1, host, 64, hipMemcpyHostToDevice, s1);
    CHECK(sj_emu::device_violations >= before + 2, "the device checker missed a copy made under the wrong device");
    (void)hipSetDevice(1);
    (void)hipFree(p1);
    (void)hipStreamDestroy(s1);
    (void)hipSetDevice(0);
  }
  std::printf("%d documents, %zu bytes, %zu calls over %zu device lists (up to %d distinct devices): %d mismatches\n", ndocs, total_bytes, calls, lists.size(), SJ_EMU_DEVICES, failures);
  return failures ? 1 : 0;
}
