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

} // namespace mi355x
} // namespace simdjson

#endif
