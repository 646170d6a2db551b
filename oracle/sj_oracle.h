/* oracle/sj_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, byte-at-a-time restatement of simdjson's stage 1 (structural indexing), minify and
 * validate_utf8 as the reference's x86 SIMD kernels (icelake == haswell == westmere) behave.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
 * the checker.  Parity status: PINNED -- tests/test_oracle_vs_reference.py checks this file
 * against the real reference (oracle/_ref/libsjref.so, built from /root/reference by
 * oracle/Makefile) and against the known answers committed under tests/golden/.
 */
#ifndef SJ_ORACLE_H
#define SJ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* simdjson::error_code values used on this path (/root/reference/include/simdjson/error.h:19-53) */
enum {
  SJO_SUCCESS = 0,
  SJO_CAPACITY = 1,
  SJO_STRING_ERROR = 5,
  SJO_UTF8_ERROR = 11,
  SJO_EMPTY = 13,
  SJO_UNESCAPED_CHARS = 14,
  SJO_UNCLOSED_STRING = 15,
  SJO_UNEXPECTED_ERROR = 24
};

/* simdjson::stage1_mode (/root/reference/include/simdjson/internal/dom_parser_implementation.h:22-27) */
enum {
  SJO_REGULAR = 0,
  SJO_STREAMING_PARTIAL = 1,
  SJO_STREAMING_FINAL = 2,
  SJO_JSON_SEQUENCE_PARTIAL = 3,
  SJO_JSON_SEQUENCE_FINAL = 4,
  SJO_COMMA_DELIMITED_PARTIAL = 5,
  SJO_COMMA_DELIMITED_FINAL = 6
};

/* Raw scan: writes ascending structural offsets to idx (room for len words),
 * returns their count; flags: bit0 unclosed string, bit1 unescaped control char inside a string,
 * bit2 invalid UTF-8. */
uint32_t sjo_scan(const uint8_t *buf, size_t len, uint32_t *idx, uint32_t *flags);

/* Full stage1 protocol.  idx needs room for len+3 words.  *n is the parser's
 * n_structural_indexes: read-modify-write exactly where the reference touches it. */
int sjo_stage1(const uint8_t *buf, size_t len, int mode, size_t capacity, uint32_t *idx, uint32_t *n);

/* minify (App. A.7): dst needs len bytes. */
int sjo_minify(const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len);

/* validate_utf8 (App. A.8): 1 = well-formed. */
int sjo_validate_utf8(const uint8_t *buf, size_t len);

/* host-side helpers of finish(), exposed so tests can pin them one by one */
size_t sjo_trim_partial_utf8(const uint8_t *buf, size_t len);
uint32_t sjo_find_next_document_index(const uint8_t *buf, const uint32_t *idx, uint32_t n);

/* ---- the strings of a document, unescaped (stage 2's string work) ------------------------------------
 * sjo_parse_string: byte-at-a-time restatement of stringparsing::parse_string
 * (/root/reference/src/generic/stage2/stringparsing.h:150-193 with handle_unicode_codepoint :50-96, escape_map :22-43,
 * jsoncharutils::hex_to_u32_nocheck / codepoint_to_utf8).  src points BEHIND the opening quote, end is one past the last
 * readable byte (bytes at or beyond it read as 0x20, the way a space-padded buffer reads); writes the unescaped bytes to
 * dst (may be NULL: measure only) and returns their number, or -1 where the reference returns nullptr (bad escape, bad
 * \u hex, lone or unpaired surrogate unless allow_replacement) or the closing quote is missing. */
long sjo_parse_string(const uint8_t *src, const uint8_t *end, uint8_t *dst, int allow_replacement);
/* sjo_string_buffer: what dom stage 2 leaves in document::string_buf (tape_builder.h:415-433): for every structural that is
 * a quote, in document order, [u32 length][unescaped bytes][0].  offsets (n words, may be NULL): offset of structural i's
 * record, 0xFFFFFFFF for the others.  Returns 0 or SJO_STRING_ERROR; *first_bad = index of the first structural whose
 * string is invalid (0xFFFFFFFF if none) -- records are produced for the valid strings either way, invalid ones take no
 * room (the reference stops at the first one, so nothing behind it is observable there). */
int sjo_string_buffer(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, int allow_replacement, uint8_t *out,
                      size_t out_cap, uint32_t *offsets, uint64_t *bytes, uint32_t *strings, uint32_t *first_bad);

/* ---- stage 2 of one document -- the DOM tape (sj_oracle_stage2.c) --------------------------------------------
 * The reference's serial walk over the structural list with the tape builder as visitor
 * (/root/reference/src/generic/stage2/json_iterator.h:121-244, tape_builder.h:142-441, format /root/reference/doc/tape.md).
 * idx[0..n) = the structurals of buf[0..len) as stage 1 left them, idx[n] = len (its first sentinel); max_depth as
 * dom::parser's (DEFAULT_MAX_DEPTH = 1024).  tape needs up to len + 3 words, string_buf 5 (len / 3) + 64 bytes (what
 * dom::document::allocate reserves, include/simdjson/dom/document-inl.h:29-67).  Returns the reference's error_code: SUCCESS,
 * EMPTY, TAPE_ERROR 3, DEPTH_ERROR 4, STRING_ERROR 5, T/F/N_ATOM_ERROR 6/7/8, NUMBER_ERROR 9, BIGINT_ERROR 10 (or CAPACITY
 * when a buffer was too small).  *tape_words / *string_bytes = what was produced (complete only on SUCCESS). */
int sjo_stage2(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, uint64_t *tape, size_t tape_cap,
               uint8_t *string_buf, size_t string_cap, uint64_t *tape_words, uint64_t *string_bytes);

/* On-Demand's raw key comparison (raw_json_string::unsafe_is_equal, value_iterator::find_field_raw), one key / a whole list */
int sjo_raw_key_equal(const uint8_t *raw, size_t length, const uint8_t *target, size_t m);
uint32_t sjo_match_keys(const uint8_t *buf, size_t len, const uint32_t *idx, uint32_t n, const uint8_t *targets, const uint32_t *lens, uint32_t K,
                        uint32_t *out);

/* FNV-1a-64 over the n+3 index words (little-endian bytes): the digest SURVEY App. B quotes. */
uint64_t sjo_fnv1a64(const void *data, size_t nbytes);

/* best-of-iters seconds for which = 0 stage1 / 1 minify / 2 validate_utf8 (bench.py "port" leg) */
double sjo_bench(int which, const uint8_t *buf, size_t len, int iters, void *scratch);

#ifdef __cplusplus
}
#endif
#endif
