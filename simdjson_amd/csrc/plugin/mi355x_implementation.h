// simdjson_amd/csrc/plugin/mi355x_implementation.h -- simdjson::implementation plug-in for MI355X.
//
// The host side of the drop-in boundary, in the reference's own language (C++), mirroring the two
// plug-in classes of /root/reference:
//   simdjson::implementation                     include/simdjson/implementation.h:45-160
//   simdjson::internal::dom_parser_implementation include/simdjson/internal/dom_parser_implementation.h:48-242
// Stage 1, minify and validate_utf8 go to the GPU through the C-ABI of include/sjgpu.h; stage 2 and
// string parsing (out of scope) are delegated UNMODIFIED to the reference's builtin
// CPU kernel, exactly as every in-tree kernel reuses src/generic/stage2.  There is no CPU path for
// stage 1 here: if the GPU is unusable, activation fails and the registry is left untouched.
#ifndef SIMDJSON_MI355X_IMPLEMENTATION_H
#define SIMDJSON_MI355X_IMPLEMENTATION_H

#include "simdjson.h"

namespace simdjson {
namespace mi355x {

/** The singleton kernel object ("mi355x"); never null. Usable only if available() is true. */
const simdjson::implementation *get_implementation() noexcept;

/** True iff libsjgpu sees at least one HIP device. */
bool available() noexcept;

/**
 * Make dom::parser / ondemand::parser / parse_many / minify() / validate_utf8() use the GPU backend:
 * simdjson::get_active_implementation() = get_implementation()   (doc/implementation-selection.md,
 * "Manually Selecting"; include/simdjson/implementation.h:226).  Returns UNSUPPORTED_ARCHITECTURE and
 * changes nothing when no GPU is usable.
 */
simdjson::error_code activate(int device = 0) noexcept;

/**
 * parse_many / iterate_many hand stage 1 one window of the stream at a time (1 MB by default,
 * include/simdjson/dom/document_stream-inl.h:285-317); a GPU launch and a PCIe round trip per window lose to a CPU kernel.
 * Tell the backend where the whole stream lies and it scans 32 MiB spans once and cuts the windows out of them (sjgpu_stream_register,
 * include/sjgpu.h): call register_stream(buf, len) before parse_many(buf, len) and unregister_stream(buf) when the stream is done.
 * The bytes must stay valid and unchanged in between.  The in-tree build does this from document_stream::start() by itself.
 */
void register_stream(const uint8_t *buf, size_t len) noexcept;
void unregister_stream(const uint8_t *buf) noexcept;
/** The same, naming the registration that leaves by its length (two streams over one base: without it the LONGEST is assumed to have left). */
void unregister_stream(const uint8_t *buf, size_t len) noexcept;

/**
 * TEST HOOKS (tests/plugin/plugin_test.cpp; never needed by a user).  stage2_decline != 0: the device road of parse() reports a HIP failure AFTER it ran, so
 * that the fall-back to stage 1 + the reference's stage 2 can be exercised; utf8_fail_attempts = N: the first N attempts of every validate_utf8() call fail
 * as if HIP had.  Both zero (the initial state): no effect.  Process-wide, stored in atomics; the product's calls read them with one relaxed load.
 */
void debug_set_test_hooks(int stage2_decline, int utf8_fail_attempts) noexcept;

/**
 * A document buffer in PAGE-LOCKED host memory -- SURVEY.md 8(f).1's "pinned-memory padded_string allocator", the C++ face of
 * sjgpu_host_alloc (include/sjgpu.h).  What it replaces: simdjson::padded_string over internal::allocate_padded_buffer
 * (/root/reference/include/simdjson/padded_string-inl.h:34-56: `new char[length + SIMDJSON_PADDING]`, padding zeroed).  Ordinary memory has to be
 * pinned page by page by the HIP runtime on every upload of a buffer it has not seen before, which halves the upload rate (DESIGN.md
 * section 5, host-buffer path); a buffer allocated here is locked once, when it is made.  Same shape as padded_string where it matters to a
 * parser: size() bytes of content, SIMDJSON_PADDING zero bytes behind them, converts to padded_string_view, so
 *     auto json = simdjson::mi355x::load_pinned("big.json");        // or pinned_padded_string(data, length)
 *     dom::element doc = parser.parse(json.value());                 // no copy, no realloc: the view says the padding is there
 * Move-only; the memory goes back with sjgpu_host_free.  data() == nullptr after a failed allocation (like padded_string).
 */
class pinned_padded_string {
public:
  pinned_padded_string() noexcept = default;
  /** length bytes of uninitialised content + zeroed padding */
  explicit pinned_padded_string(size_t length) noexcept;
  /** a copy of data[0..length) */
  pinned_padded_string(const char *data, size_t length) noexcept;
  pinned_padded_string(std::string_view sv) noexcept : pinned_padded_string(sv.data(), sv.size()) {}
  pinned_padded_string(pinned_padded_string &&o) noexcept : size_(o.size_), data_(o.data_) { o.size_ = 0; o.data_ = nullptr; }
  pinned_padded_string &operator=(pinned_padded_string &&o) noexcept;
  pinned_padded_string(const pinned_padded_string &) = delete;
  pinned_padded_string &operator=(const pinned_padded_string &) = delete;
  ~pinned_padded_string() noexcept;

  size_t size() const noexcept { return size_; }
  size_t length() const noexcept { return size_; }
  const char *data() const noexcept { return data_; }
  char *data() noexcept { return data_; }
  const uint8_t *u8data() const noexcept { return reinterpret_cast<const uint8_t *>(data_); }
  /** content + padding: what dom::parser::parse / ondemand::parser::iterate take without copying */
  operator padded_string_view() const noexcept { return padded_string_view(data_, size_, size_ + SIMDJSON_PADDING); }
  operator std::string_view() const noexcept { return std::string_view(data_, size_); }

private:
  size_t size_ = 0;
  char *data_ = nullptr;
};

/** padded_string::load (/root/reference/include/simdjson/padded_string-inl.h:175-231) into page-locked memory: IO_ERROR / MEMALLOC as there. */
simdjson_result<pinned_padded_string> load_pinned(std::string_view path) noexcept;

} // namespace mi355x
} // namespace simdjson

#endif
