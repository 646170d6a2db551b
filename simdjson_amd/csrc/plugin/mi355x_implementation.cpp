// simdjson_amd/csrc/plugin/mi355x_implementation.cpp -- see mi355x_implementation.h.
#include "mi355x_implementation.h"

#include "sjgpu.h"

#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace simdjson {
namespace mi355x {
namespace {

using builtin_parser = simdjson::SIMDJSON_BUILTIN_IMPLEMENTATION::dom_parser_implementation;

std::atomic<int> g_device{0};
// test hooks (debug_set_test_hooks): relaxed loads of two process-wide atomics -- rounds 3-5 asked the ENVIRONMENT on every parse() / validate_utf8(), a
// libc walk and a race with any setenv of the host program inside functions whose reference counterparts are re-entrant and lock-free
std::atomic<int> g_hook_stage2_decline{0}, g_hook_utf8_fail_attempts{0};

// libsjgpu's infrastructure codes -> simdjson::error_code (library must not abort or print,
// /root/reference/src/implementation.cpp:307)
error_code map_error(int rc) noexcept {
  if (rc >= 0) { return error_code(rc); }
  switch (rc) {
  case SJGPU_E_NO_DEVICE: return UNSUPPORTED_ARCHITECTURE;
  case SJGPU_E_NOMEM: return MEMALLOC;
  case SJGPU_E_BADARG: return CAPACITY; // len > 0xFFFFFFFF or similar
  default: return UNEXPECTED_ERROR;
  }
}

// minify() / validate_utf8() have no parser object (include/simdjson/implementation.h:116,128) and the reference's are
// re-entrant: every call borrows a context from libsjgpu's pool (sjgpu_ctx_create takes a parked one: microseconds),
// so concurrent callers never wait for each other, and a context only ever allocates what its operation uses.
struct borrowed_ctx {
  sjgpu_ctx *ctx = nullptr;
  int rc;
  borrowed_ctx() noexcept { rc = sjgpu_ctx_create(g_device.load(), 0xFFFFFFFFu, &ctx); }
  ~borrowed_ctx() { sjgpu_ctx_destroy(ctx); }
  borrowed_ctx(const borrowed_ctx &) = delete;
  borrowed_ctx &operator=(const borrowed_ctx &) = delete;
};

// SJGPU_DEVICES=0,1,2,3: documents of SJGPU_MGPU_FROM_MB megabytes and more (default 256) are cut into one shard per listed
// device and scanned by all of them at once, each over its own PCIe link (sjgpu_mgpu_*, include/sjgpu.h).  Read once.
struct device_list {
  std::vector<int> devices;
  size_t from_bytes = size_t(256) << 20;
  device_list() noexcept {
    if (const char *v = std::getenv("SJGPU_DEVICES")) {
      for (const char *p = v; *p;) {
        char *end = nullptr;
        const long d = std::strtol(p, &end, 10);
        if (end == p) { break; }
        devices.push_back(int(d));
        p = (*end == ',') ? end + 1 : end;
      }
    }
    if (const char *v = std::getenv("SJGPU_MGPU_FROM_MB")) { from_bytes = size_t(std::strtoull(v, nullptr, 10)) << 20; }
    if (devices.size() < 2) { devices.clear(); }
  }
};
const device_list &listed_devices() noexcept {
  static const device_list *l = new (std::nothrow) device_list();
  return *l;
}

// dom::parser::parse of documents of SJGPU_STAGE2_FROM_KB kilobytes and more (default 1024; 0 = never) runs stage 2 on the device as well
// (sjgpu_parse: the structural list never crosses PCIe, the tape and the string buffer come back instead of it).  Shorter documents,
// _number_as_string parsers and nesting limits beyond 4095 keep the reference's CPU stage 2.  Read when a parser is made.
size_t device_stage2_from() noexcept {
  // 1 MiB: bench.py's plugin_host_path.dom_parse sweep (round 4, twitter-like documents, wall time per parse): the device road costs
  // ~0.28 ms + 0.045 ms per MiB since its call became optimistic (17 launches; 0.40 ms + ... with 27 launches and 11 memsets: the threshold was
  // 2 MiB then), stage 1 on the GPU + the reference's stage 2 ~0.05 ms + 0.31 ms per MiB -- they cross near 0.9 MiB
  // (0.6 MiB: 0.29 vs 0.21 ms; 1 MiB: 0.32 vs 0.34; 4 MiB: 0.44 vs 1.34 ms; 256 MiB: 12.8 vs 81.2 ms, the reference alone 101.4 ms)
  size_t kb = 1024;
  if (const char *v = std::getenv("SJGPU_STAGE2_FROM_KB")) { kb = size_t(std::strtoull(v, nullptr, 10)); }
  return kb ? kb << 10 : ~size_t(0);
}

class dom_parser_implementation final : public internal::dom_parser_implementation {
public:
  dom_parser_implementation() noexcept : stage2_from_(device_stage2_from()) {}
  ~dom_parser_implementation() override {
    sjgpu_mgpu_destroy(mgpu_);
    sjgpu_ctx_destroy(ctx_);
  }

  // stage 1 on the GPU, then the reference's own stage 2 (src/haswell.cpp:159-163 shape)
  simdjson_warn_unused error_code parse(const uint8_t *buf, size_t len, dom::document &doc) noexcept final {
    // (2.4e9: the device's string records use 32-bit offsets, sjgpu_stage2_device answers CAPACITY beyond it -- such documents keep the road below)
    if (len >= stage2_from_ && len <= size_t(2400000000u) && ctx_ && !_number_as_string && _max_depth >= 1 && _max_depth <= 4095 && len <= _capacity &&
        doc.capacity() >= len && doc.tape && doc.string_buf) {
      // what dom::document::allocate reserved (include/simdjson/dom/document-inl.h:48-56)
      const size_t cap = doc.capacity();
      const size_t tape_words = SIMDJSON_ROUNDUP_N(cap + 3, 64), string_bytes = SIMDJSON_ROUNDUP_N(5 * (cap / 3) + SIMDJSON_PADDING, 64);
      buf_ = buf;
      len_ = len;
      uint64_t tw = 0, sb = 0;
      int rc = sjgpu_parse(ctx_, buf, len, uint32_t(_max_depth), doc.tape.get(), tape_words, doc.string_buf.get(), string_bytes, &tw, &sb);
      if (g_hook_stage2_decline.load(std::memory_order_relaxed)) { rc = SJGPU_E_HIP; } // test hook (debug_set_test_hooks, plugin_test 2c): the device road fails after it has run
      n_structural_indexes = 0; // the list stayed on the device
      next_structural_index = 0;
      // A verdict about the DOCUMENT is final (SUCCESS, TAPE_ERROR, STRING_ERROR, ...).  What says something about the device road instead
      // -- a negative SJGPU_E_* (HIP / allocation failure, overflow of a workspace), CAPACITY (a limit of the device kernels: len <=
      // _capacity was checked above), MEMALLOC, UNEXPECTED_ERROR (a stage 1 that gave up twice) -- must not cost the caller a parse the
      // reference would have delivered: such a document takes stage1() + the reference's stage 2 below, like a short one.
      if (!(rc < 0 || rc == int(CAPACITY) || rc == int(MEMALLOC) || rc == int(UNEXPECTED_ERROR))) { return map_error(rc); }
      device_stage2_declined_++;
    }
    auto error = stage1(buf, len, stage1_mode::regular);
    if (error) { return error; }
    return stage2(doc);
  }

  simdjson_warn_unused error_code stage1(const uint8_t *buf, size_t len, stage1_mode mode) noexcept final {
    buf_ = buf;
    len_ = len;
    if (!ctx_) { return UNINITIALIZED; }
    // the array holds ROUNDUP(capacity,64)+9 words (generic/dom_parser_implementation.h:63-78)
    const size_t words = SIMDJSON_ROUNDUP_N(_capacity, 64) + 9;
    if (len > _capacity) { return CAPACITY; }
    const device_list &dl = listed_devices();
    if (!dl.devices.empty() && len >= dl.from_bytes) { // every listed GPU takes a shard
      if (!mgpu_ && sjgpu_mgpu_create(dl.devices.data(), int(dl.devices.size()), &mgpu_) != 0) { mgpu_ = nullptr; }
      if (mgpu_) {
        return map_error(sjgpu_mgpu_stage1(mgpu_, buf, len, int(mode), structural_indexes.get(), words, &n_structural_indexes, &next_structural_index));
      }
    }
    return map_error(sjgpu_stage1(ctx_, buf, len, int(mode), structural_indexes.get(), words, &n_structural_indexes,
                                  &next_structural_index));
  }

  simdjson_warn_unused error_code stage2(dom::document &doc) noexcept final { return with_inner([&] { return inner_->stage2(doc); }); }
  simdjson_warn_unused error_code stage2_next(dom::document &doc) noexcept final {
    return with_inner([&] { return inner_->stage2_next(doc); });
  }
  simdjson_warn_unused uint8_t *parse_string(const uint8_t *src, uint8_t *dst, bool allow_replacement) const noexcept final {
    return inner_->parse_string(src, dst, allow_replacement);
  }
  simdjson_warn_unused uint8_t *parse_wobbly_string(const uint8_t *src, uint8_t *dst) const noexcept final {
    return inner_->parse_wobbly_string(src, dst);
  }

  simdjson_warn_unused error_code set_capacity(size_t capacity) noexcept final {
    if (capacity > SIMDJSON_MAXSIZE_BYTES) { return CAPACITY; }
    const size_t words = SIMDJSON_ROUNDUP_N(capacity, 64) + 9;
    structural_indexes.reset(new (std::nothrow) uint32_t[words]);
    if (!structural_indexes) { _capacity = 0; return MEMALLOC; }
    structural_indexes[0] = 0;
    n_structural_indexes = 0;
    int rc = ctx_ ? sjgpu_set_capacity(ctx_, capacity) : sjgpu_ctx_create(g_device.load(), capacity, &ctx_);
    if (rc != 0) { _capacity = 0; return map_error(rc); }
    if (inner_) {
      auto error = inner_->set_capacity(capacity);
      if (error) { _capacity = 0; return error; }
    }
    _capacity = capacity;
    return SUCCESS;
  }

  simdjson_warn_unused error_code set_max_depth(size_t max_depth) noexcept final {
    if (!inner_) { return UNINITIALIZED; }
    auto error = inner_->set_max_depth(max_depth);
    _max_depth = error ? 0 : max_depth;
    return error;
  }

  // stage 2 / string parsing live in the reference's builtin CPU kernel
  error_code create_inner(size_t capacity, size_t max_depth) noexcept {
    return simdjson::builtin_implementation()->create_dom_parser_implementation(capacity, max_depth, inner_);
  }

private:
  // Lend our stage-1 output to the CPU kernel object for the duration of one stage-2 call.
  template <class F> error_code with_inner(F &&call) noexcept {
    auto *in = static_cast<builtin_parser *>(inner_.get());
    in->buf = buf_;
    in->len = len_;
    in->_number_as_string = _number_as_string;
    in->_unpadded = _unpadded;
    in->n_structural_indexes = n_structural_indexes;
    in->next_structural_index = next_structural_index;
    std::swap(in->structural_indexes, structural_indexes);
    error_code error = call();
    std::swap(in->structural_indexes, structural_indexes);
    next_structural_index = in->next_structural_index;
    return error;
  }

  sjgpu_ctx *ctx_ = nullptr;
  sjgpu_mgpu *mgpu_ = nullptr; // made on the first document long enough for SJGPU_DEVICES
  std::unique_ptr<internal::dom_parser_implementation> inner_{};
  const uint8_t *buf_ = nullptr;
  size_t len_ = 0;
  size_t stage2_from_;
  size_t device_stage2_declined_ = 0; // parse() calls the device road handed back (a limit or a failure of the road, not a verdict on the document)
};

class implementation final : public simdjson::implementation {
public:
  implementation() noexcept
      : simdjson::implementation("mi355x", "AMD Instinct MI355X (gfx950): stage 1 / minify / validate_utf8 in HIP via libsjgpu", 0) {}

  // pattern of src/haswell.cpp:23-35: nothrow new, MEMALLOC on failure, then set_capacity / set_max_depth
  simdjson_warn_unused error_code create_dom_parser_implementation(
      size_t capacity, size_t max_depth, std::unique_ptr<internal::dom_parser_implementation> &dst) const noexcept final {
    auto *p = new (std::nothrow) dom_parser_implementation();
    if (!p) { return MEMALLOC; }
    dst.reset(p);
    if (auto err = p->create_inner(capacity, max_depth)) { dst.reset(); return err; }
    if (auto err = p->set_capacity(capacity)) { dst.reset(); return err; }
    if (auto err = p->set_max_depth(max_depth)) { dst.reset(); return err; }
    return SUCCESS;
  }

  simdjson_warn_unused error_code minify(const uint8_t *buf, size_t len, uint8_t *dst, size_t &dst_len) const noexcept final {
    dst_len = 0;
    borrowed_ctx b;
    if (b.rc != 0) { return map_error(b.rc); }
    size_t n = 0;
    const int rc = sjgpu_minify(b.ctx, buf, len, dst, &n); // any length: beyond 4 GiB - 1 it goes piece by piece
    dst_len = n;
    return map_error(rc);
  }

  // "true if and only if the string is valid UTF-8" (include/simdjson/implementation.h:118-128) -- and no error channel.  A failure of the ROAD
  // (no context to borrow, a HIP call or an allocation that failed) is not a verdict about the bytes, so it is not answered with `false` before
  // the road has been tried three ways: a pooled context; a FRESH one after the pool's parked memory has been given back (sjgpu_pool_trim);
  // the input in 16 MiB pieces, which needs 16 MiB of device memory whatever len is.  Only when all three fail -- a device that is gone --
  // does the call say `false`: there is no CPU path in this backend to ask instead.
  simdjson_warn_unused bool validate_utf8(const char *buf, size_t len) const noexcept final {
    // Only what another attempt can change is retried: a HIP call or an allocation that failed (SJGPU_E_HIP, SJGPU_E_NOMEM / MEMALLOC).  A bad argument
    // or a capacity answer would come back the same three times -- and the second attempt gives back every parked context of the process first.
    const int induced = g_hook_utf8_fail_attempts.load(std::memory_order_relaxed); // test hook (debug_set_test_hooks, plugin_test 5b): the first N attempts fail as if HIP had
    for (int attempt = 0; attempt < 3; attempt++) {
      if (attempt == 1) { (void)sjgpu_pool_trim(); }
      borrowed_ctx b;
      int ok = 0;
      int rc = b.rc;
      if (rc == 0) {
        rc = attempt < 2 ? sjgpu_validate_utf8(b.ctx, reinterpret_cast<const uint8_t *>(buf), len, &ok)
                         : sjgpu_validate_utf8_pieces(b.ctx, reinterpret_cast<const uint8_t *>(buf), len, size_t(16) << 20, &ok);
      }
      if (attempt < induced) { rc = SJGPU_E_HIP; }
      if (rc == 0) { return ok != 0; }
      validate_utf8_retries_.fetch_add(1, std::memory_order_relaxed);
      if (!(rc == SJGPU_E_HIP || rc == SJGPU_E_NOMEM || rc == int(MEMALLOC))) { break; } // not transient: the same call would fail the same way
    }
    return false;
  }
  mutable std::atomic<size_t> validate_utf8_retries_{0}; // attempts that failed for a reason other than the bytes
};

} // namespace

const simdjson::implementation *get_implementation() noexcept {
  static const implementation *singleton = new (std::nothrow) implementation(); // never destroyed (protected non-virtual dtor)
  return singleton;
}

bool available() noexcept { return sjgpu_device_count() > 0; }

void debug_set_test_hooks(int stage2_decline, int utf8_fail_attempts) noexcept {
  g_hook_stage2_decline.store(stage2_decline, std::memory_order_relaxed);
  g_hook_utf8_fail_attempts.store(utf8_fail_attempts, std::memory_order_relaxed);
}

void register_stream(const uint8_t *buf, size_t len) noexcept { (void)sjgpu_stream_register(buf, len); }
void unregister_stream(const uint8_t *buf) noexcept { (void)sjgpu_stream_unregister(buf); }
void unregister_stream(const uint8_t *buf, size_t len) noexcept { (void)sjgpu_stream_unregister_len(buf, len); }

// ---- pinned_padded_string ---------------------------------------------------------------------------------------------------------------
pinned_padded_string::pinned_padded_string(size_t length) noexcept {
  const size_t total = length + SIMDJSON_PADDING;
  if (total < length) { return; } // overflow: data() stays null, like allocate_padded_buffer
  data_ = static_cast<char *>(sjgpu_host_alloc(total));
  if (!data_) { return; }
  std::memset(data_ + length, 0, SIMDJSON_PADDING);
  size_ = length;
}
pinned_padded_string::pinned_padded_string(const char *data, size_t length) noexcept : pinned_padded_string(length) {
  if (data_ && data && length) { std::memcpy(data_, data, length); }
}
pinned_padded_string &pinned_padded_string::operator=(pinned_padded_string &&o) noexcept {
  if (this != &o) {
    sjgpu_host_free(data_);
    size_ = o.size_;
    data_ = o.data_;
    o.size_ = 0;
    o.data_ = nullptr;
  }
  return *this;
}
pinned_padded_string::~pinned_padded_string() noexcept { sjgpu_host_free(data_); }

simdjson_result<pinned_padded_string> load_pinned(std::string_view path) noexcept {
  const std::string name(path); // fopen wants a terminator
  std::FILE *fp = std::fopen(name.c_str(), "rb");
  if (!fp) { return IO_ERROR; }
  if (std::fseek(fp, 0, SEEK_END) < 0) { std::fclose(fp); return IO_ERROR; }
  const long llen = std::ftell(fp);
  if (llen < 0 || llen == LONG_MAX) { std::fclose(fp); return IO_ERROR; }
  const size_t len = size_t(llen);
  pinned_padded_string s(len);
  if (!s.data()) { std::fclose(fp); return MEMALLOC; }
  std::rewind(fp);
  const size_t got = std::fread(s.data(), 1, len, fp);
  if (std::fclose(fp) != 0 || got != len) { return IO_ERROR; }
  return s;
}

simdjson::error_code activate(int device) noexcept {
  if (device < 0 || device >= sjgpu_device_count()) { return UNSUPPORTED_ARCHITECTURE; }
  g_device.store(device);
  simdjson::get_active_implementation() = get_implementation();
  return SUCCESS;
}

} // namespace mi355x
} // namespace simdjson
