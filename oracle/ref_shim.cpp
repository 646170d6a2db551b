// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the *real* reference (simdjson v4.6.1, compiled from the sources where they
// lie under /root/reference by oracle/Makefile; outputs only into oracle/_ref/).  It gives the
// tests and bench.py's `cpu_baseline` leg ctypes access to the reference's own x86 kernels
// (`icelake`, `haswell`, `westmere`, `fallback`) for the three entry points of the hot path:
//
//   * dom_parser_implementation::stage1   (/root/reference/include/simdjson/internal/dom_parser_implementation.h:80)
//   * implementation::minify              (/root/reference/include/simdjson/implementation.h:116)
//   * implementation::validate_utf8       (/root/reference/include/simdjson/implementation.h:128)
//
// Kernels are addressed by name through get_available_implementations() so the global "active
// implementation" pointer is never touched.  Nothing under simdjson_amd/ may link or load this.
#include "simdjson.h"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>

using simdjson::internal::dom_parser_implementation;

namespace {
const simdjson::implementation *find_impl(const char *name) {
  auto impl = simdjson::get_available_implementations()[name];
  if (!impl || !impl->supported_by_runtime_system()) { return nullptr; }
  return impl;
}
struct ref_parser {
  std::unique_ptr<dom_parser_implementation> p;
  size_t capacity;
};
} // namespace

extern "C" {

// 1 if the named reference kernel exists and the host CPU can run it.
int sjref_available(const char *impl_name) { return find_impl(impl_name) != nullptr; }

// Persistent parser (mirrors benchmark/benchmarker.h:315-346: allocate once, time only stage1()).
void *sjref_parser_create(const char *impl_name, size_t capacity) {
  auto impl = find_impl(impl_name);
  if (!impl) { return nullptr; }
  auto *rp = new ref_parser();
  rp->capacity = capacity;
  if (impl->create_dom_parser_implementation(capacity, 1024, rp->p) != simdjson::SUCCESS) {
    delete rp;
    return nullptr;
  }
  return rp;
}
void sjref_parser_destroy(void *h) { delete static_cast<ref_parser *>(h); }

// Runs the reference stage1 and returns its error_code.  Afterwards *n_out is
// parser.n_structural_indexes and idx_out (if non-null, >= n+3 words) holds idx[0..n+2].
int sjref_parser_stage1(void *h, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out,
                        uint32_t *n_out, uint32_t *next_structural_index_out) {
  auto *rp = static_cast<ref_parser *>(h);
  auto err = rp->p->stage1(buf, len, static_cast<simdjson::stage1_mode>(mode));
  if (n_out) { *n_out = rp->p->n_structural_indexes; }
  if (next_structural_index_out) { *next_structural_index_out = rp->p->next_structural_index; }
  if (idx_out && rp->p->structural_indexes) {
    std::memcpy(idx_out, rp->p->structural_indexes.get(),
                (size_t(rp->p->n_structural_indexes) + 3) * sizeof(uint32_t));
  }
  return int(err);
}

// One-shot convenience: fresh parser with capacity `capacity` (0 => len), so stale state never leaks.
int sjref_stage1(const char *impl_name, const uint8_t *buf, size_t len, int mode, size_t capacity,
                 uint32_t *idx_out, uint32_t *n_out) {
  void *h = sjref_parser_create(impl_name, capacity ? capacity : (len ? len : 1));
  if (!h) { return -1; }
  int err = sjref_parser_stage1(h, buf, len, mode, idx_out, n_out, nullptr);
  sjref_parser_destroy(h);
  return err;
}

int sjref_minify(const char *impl_name, const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1; }
  size_t n = 0;
  auto err = impl->minify(buf, len, dst, n);
  *dst_len = n;
  return int(err);
}

int sjref_validate_utf8(const char *impl_name, const uint8_t *buf, size_t len) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1; }
  return impl->validate_utf8(reinterpret_cast<const char *>(buf), len) ? 1 : 0;
}

// ---- strings ---------------------------------------------------------------------------------------------
// dom_parser_implementation::parse_string (/root/reference/include/simdjson/internal/dom_parser_implementation.h:124) of the
// named kernel: src behind the opening quote, inside a buffer with SIMDJSON_PADDING readable bytes behind the closing quote;
// dst with room for the string + SIMDJSON_PADDING.  Returns the unescaped length, or -1 for nullptr.
long sjref_parse_string(const char *impl_name, const uint8_t *src, uint8_t *dst, int allow_replacement) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -2; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(64, 16, p) != simdjson::SUCCESS) { return -2; }
  uint8_t *end = p->parse_string(src, dst, allow_replacement != 0);
  return end ? long(end - dst) : -1;
}

// A full dom parse (stage 1 + stage 2) with the named kernel; copies document::string_buf -- [u32 length][bytes][0] per
// string, document order (src/generic/stage2/tape_builder.h:415-433) -- up to the end of the last record the tape points
// at.  buf must be padded.  Returns the parse's error_code; *used_out / *strings_out from the tape walk.
int sjref_dom_string_buf(const char *impl_name, const uint8_t *buf, size_t len, uint8_t *out, size_t out_cap, uint64_t *used_out,
                         uint32_t *strings_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(len ? len : 1, 1024, p) != simdjson::SUCCESS) { return -2; }
  simdjson::dom::document doc;
  if (doc.allocate(len) != simdjson::SUCCESS) { return -2; }
  auto err = p->parse(buf, len, doc);
  uint64_t used = 0;
  uint32_t strings = 0;
  if (err == simdjson::SUCCESS) {
    const uint64_t end = doc.tape[0] & 0xFFFFFFFFFFFFFFull; // the root entry points behind the last entry
    for (uint64_t i = 1; i + 1 < end; i++) {
      const uint64_t v = doc.tape[i];
      const char type = char(v >> 56);
      if (type == '"') {
        const uint64_t at = v & 0xFFFFFFFFFFFFFFull;
        uint32_t l;
        std::memcpy(&l, doc.string_buf.get() + at, 4);
        if (at + 5 + l > used) { used = at + 5 + l; }
        strings++;
      } else if (type == 'l' || type == 'u' || type == 'd') {
        i++; // the value sits in the next slot
      }
    }
    if (used <= out_cap && out) { std::memcpy(out, doc.string_buf.get(), used); }
  }
  if (used_out) { *used_out = used; }
  if (strings_out) { *strings_out = strings; }
  return int(err);
}

// A full dom parse with the named kernel, handing back what stage 2 left: the tape (doc.tape[0 .. tape[0] payload)) and the string
// buffer up to the end of the last record.  buf must be padded (SIMDJSON_PADDING readable bytes behind len).  max_depth as
// dom::parser's.  Returns the parse's error_code; the outputs are only filled on SUCCESS.
int sjref_dom_parse(const char *impl_name, const uint8_t *buf, size_t len, uint32_t max_depth, uint64_t *tape_out, size_t tape_cap, uint64_t *tape_words_out,
                    uint8_t *str_out, size_t str_cap, uint64_t *str_bytes_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(len ? len : 1, max_depth, p) != simdjson::SUCCESS) { return -2; }
  simdjson::dom::document doc;
  if (doc.allocate(len) != simdjson::SUCCESS) { return -2; }
  auto err = p->parse(buf, len, doc);
  if (tape_words_out) { *tape_words_out = 0; }
  if (str_bytes_out) { *str_bytes_out = 0; }
  if (err != simdjson::SUCCESS) { return int(err); }
  const uint64_t words = doc.tape[0] & 0xFFFFFFFFFFFFFFull;
  uint64_t used = 0;
  for (uint64_t i = 1; i + 1 < words; i++) {
    const uint64_t v = doc.tape[i];
    const char type = char(v >> 56);
    if (type == '"') {
      const uint64_t at = v & 0xFFFFFFFFFFFFFFull;
      uint32_t l;
      std::memcpy(&l, doc.string_buf.get() + at, 4);
      if (at + 5 + l > used) { used = at + 5 + l; }
    } else if (type == 'l' || type == 'u' || type == 'd') {
      i++;
    }
  }
  if (tape_words_out) { *tape_words_out = words; }
  if (str_bytes_out) { *str_bytes_out = used; }
  if (tape_out && words <= tape_cap) { std::memcpy(tape_out, doc.tape.get(), words * sizeof(uint64_t)); }
  if (str_out && used <= str_cap) { std::memcpy(str_out, doc.string_buf.get(), used); }
  return 0;
}

// stage 2 alone (dom_parser_implementation::stage2, tape_builder::parse_document<false>) after one stage 1: best seconds over
// `iters` runs after a warm one; negative on failure.  What bench.py times beside the device tape builder.
double sjref_bench_stage2(const char *impl_name, const uint8_t *buf, size_t len, int iters, int *err_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1.0; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(len ? len : 1, 1024, p) != simdjson::SUCCESS) { return -2.0; }
  simdjson::dom::document doc;
  if (doc.allocate(len) != simdjson::SUCCESS) { return -2.0; }
  auto err = p->stage1(buf, len, simdjson::stage1_mode::regular);
  if (err != simdjson::SUCCESS) { if (err_out) { *err_out = int(err); } return -3.0; }
  double best = 1e300;
  for (int it = 0; it < iters + 1; it++) {
    auto t0 = std::chrono::steady_clock::now();
    err = p->stage2(doc);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (it > 0 && dt < best) { best = dt; }
    if (err != simdjson::SUCCESS) { break; }
  }
  if (err_out) { *err_out = int(err); }
  return best;
}

// dom::parser::parse the way the reference's benchmarker times it (/root/reference/benchmark/benchmarker.h:315-346: one parser and
// one document allocated up front, the same buffer parsed again and again): stage 1 + stage 2 of the named kernel on one core, best
// seconds over `iters` runs after a warm one; negative on failure.  bench.py's plugin_host_path.dom_parse leg times the plug-in's
// two roads beside it.
double sjref_bench_parse(const char *impl_name, const uint8_t *buf, size_t len, int iters, int *err_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1.0; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(len ? len : 1, 1024, p) != simdjson::SUCCESS) { return -2.0; }
  simdjson::dom::document doc;
  if (doc.allocate(len) != simdjson::SUCCESS) { return -2.0; }
  double best = 1e300;
  simdjson::error_code err = simdjson::SUCCESS;
  for (int it = 0; it < iters + 1; it++) {
    auto t0 = std::chrono::steady_clock::now();
    err = p->parse(buf, len, doc);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (it > 0 && dt < best) { best = dt; }
    if (err != simdjson::SUCCESS) { break; }
  }
  if (err_out) { *err_out = int(err); }
  return best;
}

// On-Demand's raw key comparison: ondemand::raw_json_string::unsafe_is_equal(length, target) of the builtin kernel
// (/root/reference/include/simdjson/generic/ondemand/raw_json_string-inl.h:66-69).  raw points behind the opening quote.
int sjref_raw_key_equal(const uint8_t *raw, size_t length, const uint8_t *target, size_t m) {
  simdjson::ondemand::raw_json_string r(raw);
  return r.unsafe_is_equal(length, std::string_view(reinterpret_cast<const char *>(target), m)) ? 1 : 0;
}

// The string work of stage 2 alone, the way tape_builder does it (visit_string, tape_builder.h:187-205 with :415-433): for every
// structural that is a quote, parse_string behind a 4-byte length slot, NUL behind it.  buf must be padded; out needs
// 5 (len + 1) / 3 + 64 bytes.  Returns the best seconds over `iters` runs (after one warm run), negative on failure (incl. a
// string the kernel rejects); *used_out / *strings_out describe the buffer it leaves in out.
double sjref_bench_parse_strings(const char *impl_name, const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, uint8_t *out, int iters,
                                 uint64_t *used_out, uint32_t *strings_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1.0; }
  std::unique_ptr<dom_parser_implementation> p;
  if (impl->create_dom_parser_implementation(64, 16, p) != simdjson::SUCCESS) { return -2.0; }
  double best = 1e300;
  uint64_t used = 0;
  uint32_t strings = 0;
  for (int it = 0; it < iters + 1; it++) {
    auto t0 = std::chrono::steady_clock::now();
    uint8_t *at = out;
    strings = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (idx[i] >= len || buf[idx[i]] != '"') { continue; }
      uint8_t *dst = p->parse_string(buf + idx[i] + 1, at + sizeof(uint32_t), false);
      if (!dst) { return -3.0; }
      const uint32_t l = uint32_t(dst - (at + sizeof(uint32_t)));
      std::memcpy(at, &l, sizeof(uint32_t));
      *dst = 0;
      at = dst + 1;
      strings++;
    }
    used = uint64_t(at - out);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (it > 0 && dt < best) { best = dt; }
  }
  if (used_out) { *used_out = used; }
  if (strings_out) { *strings_out = strings; }
  return best;
}

// ---- timing legs for bench.py's cpu_baseline (reference convention: best-of-N, input bytes only,
// ---- /root/reference/benchmark/benchmarker.h:315-346,418) -------------------------------------
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// which: 0 = stage1(regular), 1 = minify, 2 = validate_utf8.  Returns best seconds over `iters`
// runs (after one warm run), or a negative number on failure.  `scratch` must hold len bytes for minify.
double sjref_bench(const char *impl_name, int which, const uint8_t *buf, size_t len, int iters,
                   uint8_t *scratch, int *err_out) {
  auto impl = find_impl(impl_name);
  if (!impl) { return -1.0; }
  std::unique_ptr<dom_parser_implementation> p;
  if (which == 0 && impl->create_dom_parser_implementation(len, 1024, p) != simdjson::SUCCESS) { return -2.0; }
  double best = 1e300;
  int err = 0;
  for (int it = 0; it < iters + 1; it++) {
    double t0 = now_s();
    if (which == 0) {
      err = int(p->stage1(buf, len, simdjson::stage1_mode::regular));
    } else if (which == 1) {
      size_t n = 0;
      err = int(impl->minify(buf, len, scratch, n));
    } else {
      err = impl->validate_utf8(reinterpret_cast<const char *>(buf), len) ? 0 : int(simdjson::UTF8_ERROR);
    }
    double dt = now_s() - t0;
    if (it > 0 && dt < best) { best = dt; }
  }
  if (err_out) { *err_out = err; }
  return best;
}

} // extern "C"
