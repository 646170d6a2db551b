/* include/sjgpu.h -- C-ABI of libsjgpu: simdjson's stage 1 (structural indexing), minify and
 * validate_utf8 on AMD Instinct MI355X (gfx950, hand-written HIP).
 *
 * This is the drop-in boundary below simdjson's plug-in classes.  Each entry point replaces one
 * virtual of the reference (citations relative to /root/reference):
 *
 *   sjgpu_stage1         internal::dom_parser_implementation::stage1(buf, len, stage1_mode)
 *                        include/simdjson/internal/dom_parser_implementation.h:80
 *                        (behaviour: src/generic/stage1/json_structural_indexer.h:193-397)
 *   sjgpu_minify         implementation::minify(buf, len, dst, dst_len)      include/simdjson/implementation.h:116
 *   sjgpu_validate_utf8  implementation::validate_utf8(buf, len)             include/simdjson/implementation.h:128
 *   sjgpu_ctx_create / sjgpu_set_capacity / sjgpu_ctx_destroy
 *                        implementation::create_dom_parser_implementation    include/simdjson/implementation.h:97-101
 *                        dom_parser_implementation::set_capacity / dtor      internal/dom_parser_implementation.h:154,170
 *
 * The *_device variants take buffers that already live in HBM (device-resident pipelines, bench.py,
 * multi-GPU NDJSON shards) and are asynchronous on the caller's HIP stream.
 *
 * Conventions: plain pointers and sizes only.  Return value >= 0 is a simdjson::error_code
 * (include/simdjson/error.h:19-53: 0 SUCCESS, 1 CAPACITY, 2 MEMALLOC, 11 UTF8_ERROR, 13 EMPTY,
 * 14 UNESCAPED_CHARS, 15 UNCLOSED_STRING, 24 UNEXPECTED_ERROR); < 0 is an infrastructure error the
 * plug-in shim maps to UNSUPPORTED_ARCHITECTURE / MEMALLOC / UNEXPECTED_ERROR.  Nothing throws,
 * prints or aborts.  One context per parser object; a context serves one call at a time and may be
 * used from any thread.  There is NO CPU fallback inside this library: without a usable GPU every
 * compute entry point fails with SJGPU_E_NO_DEVICE.
 */
#ifndef SJGPU_H
#define SJGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SJGPU_E_NO_DEVICE (-1) /* no HIP device / runtime unusable */
#define SJGPU_E_HIP       (-2) /* a HIP call failed; see sjgpu_last_error() */
#define SJGPU_E_NOMEM     (-3) /* device or pinned-host allocation failed */
#define SJGPU_E_BADARG    (-4) /* null/misaligned pointer, len > 0xFFFFFFFF, unknown mode */
#define SJGPU_E_OVERFLOW  (-5) /* caller's index buffer too small for n+3 words */
#define SJGPU_E_PEER      (-6) /* sjgpu_comm_*: another rank of the communicator failed; nothing was exchanged */

/* simdjson::stage1_mode (internal/dom_parser_implementation.h:22-27) */
enum sjgpu_stage1_mode {
  SJGPU_REGULAR = 0,
  SJGPU_STREAMING_PARTIAL = 1,
  SJGPU_STREAMING_FINAL = 2,
  SJGPU_JSON_SEQUENCE_PARTIAL = 3,
  SJGPU_JSON_SEQUENCE_FINAL = 4,
  SJGPU_COMMA_DELIMITED_PARTIAL = 5,
  SJGPU_COMMA_DELIMITED_FINAL = 6
};

/* scan flags (device result) */
#define SJGPU_F_UNCLOSED_STRING 1u /* input ends inside a string */
#define SJGPU_F_UNESCAPED_CTRL  2u /* byte <= 0x1F inside a string */
#define SJGPU_F_UTF8_ERROR      4u /* not well-formed UTF-8 */
#define SJGPU_F_IDX_OVERFLOW    8u /* index buffer too small; indices beyond it were dropped */
#define SJGPU_F_INTERNAL        16u /* single-pass pipeline gave up (bounded spin expired); result invalid */
#define SJGPU_F_RANGE_CARRY     32u /* ranges only (`more` set): the backslash run that ends this range makes the next range's first byte
                                       escaped (or its leading quote's predecessor): hand it on with the in-string bit, see sjgpu_stage1_range_device */

typedef struct sjgpu_ctx sjgpu_ctx;

/* What a device-resident call produced (valid after sjgpu_result()). */
typedef struct sjgpu_scan_result {
  uint32_t n;       /* stage1: number of structural indexes written (sentinels follow at idx[n..n+2]) */
  uint32_t flags;   /* SJGPU_F_* */
  uint64_t out_len; /* minify: bytes written to dst (0 if unclosed string) */
} sjgpu_scan_result;

int sjgpu_device_count(void); /* number of usable HIP devices, 0 if none */

/* Context = per-parser device workspace for documents up to `capacity` bytes on HIP device `device`. */
int sjgpu_ctx_create(int device, size_t capacity, sjgpu_ctx **out);
void sjgpu_ctx_destroy(sjgpu_ctx *ctx);
int sjgpu_set_capacity(sjgpu_ctx *ctx, size_t capacity);
/* sjgpu_ctx_destroy PARKS a context (its stream, page-locked blocks and small workspaces are reused by the next
 * sjgpu_ctx_create on that device: the reference's tests make 100 000 parsers); up to 64 stay parked.  sjgpu_pool_trim frees
 * them all and returns how many there were -- for orderly shutdown, or to get the memory back. */
int sjgpu_pool_trim(void);
size_t sjgpu_capacity(const sjgpu_ctx *ctx);
const char *sjgpu_last_error(const sjgpu_ctx *ctx); /* text of the last HIP failure ("" if none) */

/* ---- host-buffer entry points (what the simdjson plug-in shim binds) --------------------------------
 * buf: len readable bytes, no padding required (the reference's stage 1 never reads past len,
 * src/generic/stage1/buf_block_reader.h:99-104).
 * idx_out: the parser's structural_indexes array, idx_words >= len+3 words available
 * (include/simdjson/generic/dom_parser_implementation.h:63-78 allocates ROUNDUP(capacity,64)+9).
 * *n_io / *next_io: the parser's n_structural_indexes / next_structural_index, written exactly where
 * the reference writes them (json_structural_indexer.h:264,287: n = count, next = 0 once the scan
 * has passed the UNCLOSED_STRING / UNESCAPED_CHARS exits; untouched on the earlier exits). */
int sjgpu_stage1(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words,
                 uint32_t *n_io, uint32_t *next_io);
/* dst: len writable bytes; never written beyond dst+len.  UNCLOSED_STRING => *dst_len = 0. */
int sjgpu_minify(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len);
/* *ok = 1 iff buf[0..len) is well-formed UTF-8 (len == 0 => 1). */
int sjgpu_validate_utf8(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, int *ok);
/* The same verdict with the input taken in pieces of at most piece_bytes (cut in front of a character's first byte, so every piece
 * is well-formed or not by itself; 0 = the default, 1 GiB): what sjgpu_validate_utf8 does beyond its piece size, with the size in the
 * caller's hand.  A piece needs piece_bytes of device memory, not len: the road a caller takes when the one-piece call failed for
 * want of memory (the plug-in's implementation::validate_utf8 has no error channel -- include/simdjson/implementation.h:118-128 -- and
 * retries this way before it answers). */
int sjgpu_validate_utf8_pieces(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, size_t piece_bytes, int *ok);

/* ---- device-resident entry points --------------------------------------------------------------------
 * buf_dev: device pointer, 16-byte aligned, len readable bytes (nothing is read past len).
 * stream: the caller's hipStream_t (NULL = HIP's default stream, which is what PyTorch's default
 * stream is); work is ordered with whatever else the caller enqueued there.  Calls only enqueue;
 * fetch the outcome with sjgpu_result(), which waits for that stream.
 * sjgpu_stage1_device: regular-mode scan; writes idx_dev[0..n) ascending byte offsets and the three
 * sentinels idx_dev[n]=len, idx_dev[n+1]=len, idx_dev[n+2]=0 (json_structural_indexer.h:284-286);
 * needs idx_words >= n+3 (len+3 always suffices).  idx_dev and dst_dev must be 16-byte aligned like buf_dev (the
 * offsets leave as 16-byte stores); a misaligned pointer is SJGPU_E_BADARG.  Error precedence is applied by the caller from
 * `flags` (helper below). */
int sjgpu_stage1_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words, void *stream);
int sjgpu_minify_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *dst_dev, void *stream);
int sjgpu_validate_utf8_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *stream);
int sjgpu_result(sjgpu_ctx *ctx, void *stream, sjgpu_scan_result *out);
/* sjgpu_stage1_device with the TOKEN-BYTE STREAM beside the offsets: tok_dev[i] = buf_dev[idx_dev[i]] for i < n (tok_bytes >= idx_words).
 * Every consumer of the structural list dereferences it -- the reference's stage 2 walks `buf[*next_structural++]`
 * (/root/reference/src/generic/stage2/json_iterator.h:246-288), find_next_document_index reads the byte under every index
 * (src/generic/stage1/find_next_document_index.h:39-98) -- and on a GPU that dereference is a gather through 128-byte lines: a pass over the
 * list of a sparse document fetches the whole document again to pick one byte per structural.  Stage 1 holds those bytes when it decides what
 * is structural: here they leave with the offsets (compacted by the scan kernel per 16 KiB segment, copied behind the output cursor by the
 * emission kernel; +1 B written per structural) and list passes read ONE coalesced byte per entry: sjgpu_depth_scan_tokens_device below.
 * tok_dev: 16-byte aligned like the other device pointers.
 * Opt-in because it costs stage 1 (split pipeline only; the byte compaction adds ~40 % to the scan kernel's instructions: DESIGN.md
 * section 4b has the measured cost and what the consumers get back).  Same result / flags / list as sjgpu_stage1_device. */
int sjgpu_stage1_tokens_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words, void *tok_dev, size_t tok_bytes,
                               void *stream);

/* Pipeline selection (default SJGPU_PIPELINE_AUTO; the environment variable SJGPU_PIPELINE = "split" |
 * "fused" | "auto" sets the default of new contexts):
 *   SPLIT  summarize -> resolve -> emit: four launches, a 64-bit mask per block round-trips through HBM, every
 *          kernel runs at full occupancy with no inter-workgroup waiting -- fastest on medium inputs;
 *   FUSED  one kernel, chained scan between tiles: reads every byte once, one launch -- fastest on small inputs
 *          and, in its pipelined form (look-back + emission of a tile deferred behind the scan of the next), on
 *          large ones;
 *   AUTO   FUSED up to 5 MiB (16 KiB tiles); beyond that SPLIT, except
 *            stage 1 from 512 MiB (pipelined, eight waves per workgroup, 128 KiB tiles) when the output is DENSE: the previous
 *              whole-document scan of more than 5 MiB through this context produced at least 0.2 offsets per byte (a new context
 *              assumes dense; the figure is taken when that scan's result is read, so it follows the documents a context sees).
 *              Sparse output (NDJSON, pretty-printed text) stays SPLIT at every size: 20 % faster at 1 GiB (profiles/r05_pipeline_sweep.txt);
 *            minify from 192 MiB (k_minify_onchip reads its input once, the split pipeline twice), whatever the density.
 *          (A/B switches, read once per process: SJGPU_PIPE_WAVES=4 / SJGPU_MINIFY_WAVES=4 bring back the four-wave
 *          shapes of rounds 1-3 -- 64 KiB / 32 KiB tiles.)
 * All produce identical bytes; a single-pass call that raises SJGPU_F_INTERNAL is re-run split by the
 * host-buffer entry points, device-resident callers see the flag in sjgpu_result(). */
#define SJGPU_PIPELINE_SPLIT 0
#define SJGPU_PIPELINE_FUSED 1
#define SJGPU_PIPELINE_AUTO  2
int sjgpu_set_pipeline(sjgpu_ctx *ctx, int pipeline);
int sjgpu_last_pipeline(const sjgpu_ctx *ctx); /* SJGPU_PIPELINE_SPLIT / _FUSED: what the last enqueued scan used */

/* Diagnostics: one single-pass stage-1 call whose first trace_tiles tiles record 8 wall-clock stamps each
 * (100 MHz ticks: loop top, ticket, wave-0 scanned, all scanned, look-back done, prefix broadcast, wave-0
 * emitted, unused) into trace_host[trace_tiles*8].  Synchronous. */
int sjgpu_debug_trace_stage1(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words,
                             uint64_t *trace_host, uint32_t trace_tiles);

/* The same for the pipelined single-pass kernel (the large-input one): thread 0 of every workgroup stamps the first 32
 * iterations of its loop -- loop top, ticket known, wave 0 scanned, all waves scanned, aggregate published + look-back
 * done, prefix broadcast, wave 0 emitted, masks parked -- into trace_host[workgroup][32][8] (zero = not reached);
 * max_records >= 32 x workgroups (2048 workgroups at most).  Synchronous. */
int sjgpu_debug_trace_pipelined(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *idx_dev, size_t idx_words,
                                uint64_t *trace_host, uint32_t max_records, uint32_t *workgroups_out);

/* Diagnostics: which kernels produced the string buffer of the context's last sjgpu_parse_strings_device / sjgpu_stage2_device /
 * sjgpu_parse call: 1 = the stream compaction of the document (every string valid and listed: sjgpu_string_stream.hip),
 * 2 = the per-string walk (a string the reference rejects, an unclosed string, or a quote glued to a scalar), 0 = none yet. */
int sjgpu_debug_string_path(const sjgpu_ctx *ctx);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).  While enabled,
 * each *_device call brackets every kernel it enqueues with events (up to 4096 calls are retained);
 * sjgpu_profile_read waits for the stream, adds the elapsed milliseconds per kernel slot into
 * ms_sum[0..2] (split pipeline: summarize, resolve, emit; single pass: slot 0 = everything the
 * call enqueues -- workspace clears, the scan kernel; validate_utf8: slot 0), stores the number of
 * calls accumulated and resets.  sjgpu_profile_kernel names the dominant kernel of the last enqueued call. */
int sjgpu_profile_enable(sjgpu_ctx *ctx, int on);
int sjgpu_profile_read(sjgpu_ctx *ctx, double *ms_sum, uint32_t *calls);
const char *sjgpu_profile_kernel(const sjgpu_ctx *ctx); /* e.g. "k_fused_pipelined<0>", "k_stage1_summarize+k_stage1_emit" */

/* regular-mode error_code from a scan result (json_structural_indexer.h:249-294,395-396):
 * UNCLOSED_STRING > UNESCAPED_CHARS > EMPTY > UTF8_ERROR > SUCCESS. */
int sjgpu_stage1_error_from_flags(uint32_t n, uint32_t flags);

/* Host post-pass of finish() for all seven modes on host-resident arrays: sentinels, streaming
 * truncation, RS / root-comma filtering (json_structural_indexer.h:249-397,
 * src/generic/stage1/find_next_document_index.h:39-369).  idx holds the n_raw raw structurals and
 * has room for n_raw+3 words; len is the (already trimmed) scanned length. */
int sjgpu_stage1_finish_host(const uint8_t *buf, size_t len, int mode, uint32_t *idx, uint32_t n_raw, uint32_t flags,
                             uint32_t *n_io, uint32_t *next_io);
/* streaming modes: length after dropping a trailing partial UTF-8 character (…indexer.h:156-174) */
size_t sjgpu_trim_partial_utf8(const uint8_t *buf, size_t len);


/* ---- many small documents in one launch ---------------------------------------------------------------------------
 * The reference scans one document per stage1() call (dom::parser::parse, ondemand::parser::iterate); a caller with a
 * batch of small documents (log lines, API payloads, one NDJSON record each) pays a launch and a PCIe round trip per
 * document that way.  sjgpu_stage1_many scans them all in ONE launch, one workgroup per document (regular mode): the
 * documents are gathered into one page-locked block, the kernel reads and writes that block across PCIe (blocks beyond
 * 2 MiB are staged through HBM with one copy each way), and every document gets its own error code, n and list --
 * exactly what sjgpu_stage1(..., SJGPU_REGULAR, ...) would have produced for it.  Documents of up to a few hundred KiB;
 * longer ones belong to sjgpu_stage1.  sjgpu_stage1 itself takes the same one-workgroup kernel for documents up to
 * 64 KiB (one launch, one wait, no device-side staging; SJGPU_SMALL_DOCS=0 turns that off). */
typedef struct sjgpu_doc {
  const uint8_t *buf;  /* in: the document */
  size_t len;
  uint32_t *idx_out;   /* in: room for len + 3 words; out: idx[0..n+2] */
  size_t idx_words;
  uint32_t n;          /* out: n_structural_indexes (0 when error is UNCLOSED_STRING / UNESCAPED_CHARS / EMPTY / CAPACITY) */
  int error;           /* out: simdjson::error_code of this document */
} sjgpu_doc;
int sjgpu_stage1_many(sjgpu_ctx *ctx, sjgpu_doc *docs, size_t count);

/* ---- the structural list after the scan, on the device -------------------------------------------------------------
 * sjgpu_stage1_finish_device: finish() of a streaming mode (json_structural_indexer.h:295-394 with
 * find_next_document_index.h:39-369) for a list that lives in HBM: idx_dev[0..n_raw) are the raw structurals of
 * buf_dev[0..len) as sjgpu_stage1_device left them (sentinels behind them), flags the scan's flags; len is the length
 * after sjgpu_trim_partial_utf8.  Record separators / root commas are filtered out of idx_dev in place, the last complete
 * document is found with a reduction and a bracket balance instead of the reference's backward walk, and idx_dev ends up
 * holding exactly the words sjgpu_stage1() would have delivered (idx[0 .. *n_io + 2]).  Returns the error_code of the
 * mode (SUCCESS, EMPTY, CAPACITY, UTF8_ERROR, UNESCAPED_CHARS); *next_start_out = where the next batch begins for the
 * partial json_sequence / comma_delimited modes.  Waits for the stream (it reads one small state back).
 * sjgpu_stage1() uses the same kernels for streaming-mode documents beyond the small-document path (SJGPU_FINISH=host
 * keeps the host walk of stage1_finish.cpp, =device forces the device path).
 * sjgpu_depth_scan_device: depth_dev[i] (int32, n + 1 entries) = number of containers open in front of structural i --
 * the `depth` the reference's stage 2 carries while it walks the list (src/generic/stage2/json_iterator.h), as a
 * bracket prefix scan; depth_dev[n] = depth behind the last structural (0 for a balanced document).  Asynchronous. */
/* The list passes (finish, depth scan, strings, stage 2, the staged form of sjgpu_stage1_many) share per-context scratch memory that grows on
 * demand: calls on ONE context must be ordered on one stream (or separated by a wait) -- two of them in flight on different
 * streams would use the same scratch.  Use one context per concurrent pipeline, as the reference uses one parser per thread. */
int sjgpu_stage1_finish_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int mode, void *idx_dev, uint32_t n_raw,
                               uint32_t flags, void *stream, uint32_t *n_io, uint32_t *next_start_out);
int sjgpu_depth_scan_device(sjgpu_ctx *ctx, const void *buf_dev, const void *idx_dev, uint32_t n, void *depth_dev, void *stream);
/* the same from the token stream of sjgpu_stage1_tokens_device: neither the document nor the list is read */
int sjgpu_depth_scan_tokens_device(sjgpu_ctx *ctx, const void *tok_dev, uint32_t n, void *depth_dev, void *stream);

/* ---- the strings of a document, unescaped ----------------------------------------------------------
 * What the reference's stage 2 does at every quote of the structural list: stringparsing::parse_string
 * (src/generic/stage2/stringparsing.h:150-193; the virtual dom_parser_implementation::parse_string,
 * include/simdjson/internal/dom_parser_implementation.h:124) into document::string_buf as [u32 length][unescaped bytes][0]
 * (src/generic/stage2/tape_builder.h:187-205, :415-433) -- here for all strings of the list at once: as a stream compaction of
 * the document when every string is valid (the records follow each other in document order: sjgpu_string_stream.hip), else one
 * lane per structural (measure, exclusive scan, write: sjgpu_strings.hip); same bytes either way.  idx_dev[0..n] is the list sjgpu_stage1_device left for buf_dev[0..len)
 * (regular mode, no error) INCLUDING its first sentinel idx_dev[n] = len, which bounds the last token.  string_buf_dev receives the records in document order, byte for byte the reference's
 * document::string_buf (5 (len + 1) / 3 bytes always suffice; the reference allocates ROUNDUP(5 len / 3 + 64, 64));
 * offsets_dev (n + 1 words, may be NULL): offsets_dev[i] = where structural i's record begins -- for a string exactly the
 * payload of the reference's tape entry -- offsets_dev[i + 1] - offsets_dev[i] = its size (0: not a string, or an invalid
 * one), offsets_dev[n] = *bytes_out.  allow_replacement as in the reference (On-Demand's option; dom passes false).
 * Returns SUCCESS, STRING_ERROR (5) when the reference would reject a string (*first_bad_out = list index of the first one;
 * the records of the valid strings are written all the same), CAPACITY for len > 2.4e9, SJGPU_E_OVERFLOW when
 * string_buf_bytes is too small, other negatives as usual.  Waits for the stream (reads 24 bytes back). */
int sjgpu_parse_strings_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, int allow_replacement,
                               void *string_buf_dev, size_t string_buf_bytes, void *offsets_dev, void *stream, uint64_t *bytes_out,
                               uint32_t *strings_out, uint32_t *first_bad_out);

/* ---- On-Demand's raw key comparison, every key of a resident document at once -----------------------------
 * ondemand's object lookup (value_iterator::find_field_raw, include/simdjson/generic/ondemand/value_iterator-inl.h:132, :229) walks
 * the fields of an object and compares each key's RAW bytes -- escapes not resolved -- with the wanted name:
 * raw_json_string::unsafe_is_equal(length, target) (raw_json_string-inl.h:66-69).  Here for ALL keys of the list and K wanted names
 * in one pass: match_dev[i] (n words) = index of the first name equal to structural i if that structural is a key (a string whose
 * next structural is ':'), 0xFFFFFFFF otherwise.  names: K byte strings back to back (host memory), name_lens[k] their lengths
 * (K <= 256, 64 KiB in total).  idx_dev[0..n] as sjgpu_stage1_device left it, first sentinel included.  *matches_out = keys that
 * matched.  Waits for the stream (reads 4 bytes back). */
int sjgpu_match_keys_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, const uint8_t *names, const uint32_t *name_lens,
                            uint32_t K, void *match_dev, void *stream, uint32_t *matches_out);

/* ---- stage 2 of a resident document: the DOM tape --------------------------------------------------------
 * What dom_parser_implementation::stage2(dom::document &) (include/simdjson/internal/dom_parser_implementation.h:94) leaves in
 * the document -- behaviour: json_iterator::walk_document (src/generic/stage2/json_iterator.h:121-244) with the tape builder as
 * visitor (src/generic/stage2/tape_builder.h:142-441; format doc/tape.md) -- for a document and structural list that live in
 * HBM, without walking: tape positions and nesting depths are prefix sums over the list, brackets find their partners through
 * a stable sort by nesting level, every token checks the walk's rule for itself, and the reference's error is the smallest
 * offending list index (sjgpu_tape.hip).  Numbers are converted exactly like the reference's (Eisel-Lemire, exact big-integer
 * decision beyond 19 digits: sj_number.h); the strings come from the string pass above.
 * idx_dev[0..n] = the list sjgpu_stage1_device left for buf_dev[0..len) (regular mode, no error), first sentinel included;
 * buf_dev 16-byte aligned like every device entry point (SJGPU_E_BADARG otherwise).
 * max_depth: dom::parser's (DEFAULT_MAX_DEPTH 1024; at most 4095 here).  tape_dev: room for tape_cap_words 64-bit words
 * (len + 3 always suffice; the reference allocates ROUNDUP(len + 3, 64)), 8-byte aligned; string_buf_dev as for
 * sjgpu_parse_strings_device.  Returns the reference's error_code: SUCCESS, EMPTY (n == 0), TAPE_ERROR 3, DEPTH_ERROR 4,
 * STRING_ERROR 5, T/F/N_ATOM_ERROR 6/7/8, NUMBER_ERROR 9, BIGINT_ERROR 10 (numbers beyond 64 bits: _number_as_string is not
 * offered here) -- the one the reference's serial walk meets FIRST -- or SJGPU_E_OVERFLOW / other negatives.  On SUCCESS
 * tape_dev[0 .. *tape_words_out) and string_buf_dev[0 .. *string_bytes_out) are word for word what the reference's dom parse
 * leaves in dom::document::tape / string_buf.  Waits for the stream (reads 56 bytes back).
 * The call is optimistic: it enqueues the string pass's stream compaction alone and the bracket sort in one pass; a document
 * the stream declines (a string the reference rejects, quotes glued to scalars) or one nested 64 deep and more says so in the
 * results read back, and the launches run a second time with the per-string kernels / the sort's second pass enqueued. */
int sjgpu_stage2_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, uint32_t max_depth, void *tape_dev,
                        size_t tape_cap_words, void *string_buf_dev, size_t string_buf_bytes, void *stream, uint64_t *tape_words_out,
                        uint64_t *string_bytes_out);
/* the same with the token stream of sjgpu_stage1_tokens_device (tok_dev[i] = buf_dev[idx_dev[i]], i < n; NULL = sjgpu_stage2_device).  Round 5: the tape's token
 * front read one coalesced byte per token instead of gathering it out of the document.  Since round 6 the front stages the document's bytes in LDS for the
 * numbers and the atoms and takes the token bytes from the same window (k_tok_stage): the stream is accepted and NOT read -- same results, same time as
 * sjgpu_stage2_device; the consumer that still profits from the stream is sjgpu_depth_scan_tokens_device */
int sjgpu_stage2_tokens_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, const void *idx_dev, uint32_t n, const void *tok_dev, uint32_t max_depth,
                               void *tape_dev, size_t tape_cap_words, void *string_buf_dev, size_t string_buf_bytes, void *stream, uint64_t *tape_words_out,
                               uint64_t *string_bytes_out);
/* dom_parser_implementation::parse(buf, len, doc) (include/simdjson/internal/dom_parser_implementation.h:64) for HOST buffers:
 * upload, stage 1, stage 2 on the device, tape and string buffer copied into the caller's arrays (the document's
 * doc.tape / doc.string_buf).  The structural list never leaves the device.  Same error codes as the reference's parse
 * (stage 1's first: CAPACITY, EMPTY, UNCLOSED_STRING, UNESCAPED_CHARS, UTF8_ERROR; then stage 2's). */
int sjgpu_parse(sjgpu_ctx *ctx, const uint8_t *buf, size_t len, uint32_t max_depth, uint64_t *tape_out, size_t tape_cap_words,
                uint8_t *string_buf_out, size_t string_buf_bytes, uint64_t *tape_words_out, uint64_t *string_bytes_out);

/* ---- one large document sharded across GPUs ("general inputs") -------------------------
 * The reference has no counterpart: its stage 1 is one serial pass whose carries (json_escape_scanner.h:50-71
 * next_is_escaped, json_string_scanner.h:62-85 prev_in_string, json_scanner.h:128-157 prev_scalar,
 * utf8_lookup4_algorithm.h:164-171 prev_incomplete) run through the whole buffer.  Cut the buffer where all of
 * them except the in-string bit are provably zero, exchange ONE bit per shard, and every shard scans alone:
 *   1. host: cut k = sjgpu_clean_cut(buf, len, k*len/G)   (byte cut-1 is ASCII whitespace or , : [ ] { })
 *   2. rank r: sjgpu_string_parity_device(shard r); sjgpu_result().n = 1 iff it holds an odd number of
 *      unescaped quotes; all-gather the G bits; in_string(r) = XOR of the bits of shards < r
 *   3. rank r: sjgpu_stage1_shard_device / sjgpu_minify_shard_device with in_string(r).
 * sjgpu_result() after step 3: n / out_len of the shard; SJGPU_F_UNCLOSED_STRING = "the shard ENDS inside a
 * string" (an error only for the last shard); the other flags as usual.  Offsets are shard-relative; the
 * sentinels written behind them hold the shard's length.  Concatenating the shards' outputs in rank order
 * gives exactly the whole-buffer result. */
size_t sjgpu_clean_cut(const uint8_t *buf, size_t len, size_t target); /* first cut >= target, or len */
int sjgpu_string_parity_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, void *stream);
int sjgpu_stage1_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *idx_dev,
                              size_t idx_words, void *stream);
int sjgpu_minify_shard_device(sjgpu_ctx *ctx, const void *buf_dev, size_t len, int in_string, void *dst_dev,
                              void *stream);

/* ---- ONE host buffer, SEVERAL GPUs of the node, one process (sjgpu_mgpu.hip) ------------------------------------------
 * The C++ face of SURVEY.md 8(e) for callers whose document and index array live in host memory (the simdjson plug-in,
 * dom::parser / parse_many windows of hundreds of megabytes): the buffer is cut at clean cuts into one shard per
 * listed device, every device uploads, scans and downloads its own shard over its own PCIe link (one host thread per
 * device), and the shards exchange exactly one bit each -- through host memory, no collective.  Results are
 * bit-identical to sjgpu_stage1 / sjgpu_minify / sjgpu_validate_utf8 on one device, for every stage1_mode.
 * devices[]: HIP device numbers, one shard per entry (an entry may repeat: that is how a one-GPU box tests this).
 * The plug-in uses it when the environment lists devices: SJGPU_DEVICES=0,1,2,3 (documents of SJGPU_MGPU_FROM_MB
 * megabytes and more, default 256).  Shards that STAY in HBM are one process per GPU: simdjson_amd/sharded.py. */
typedef struct sjgpu_mgpu sjgpu_mgpu;
int sjgpu_mgpu_create(const int *devices, int count, sjgpu_mgpu **out);
void sjgpu_mgpu_destroy(sjgpu_mgpu *m);
int sjgpu_mgpu_count(const sjgpu_mgpu *m);
int sjgpu_mgpu_stage1(sjgpu_mgpu *m, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words,
                      uint32_t *n_io, uint32_t *next_io);
int sjgpu_mgpu_minify(sjgpu_mgpu *m, const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len);
int sjgpu_mgpu_validate_utf8(sjgpu_mgpu *m, const uint8_t *buf, size_t len, int *ok);

/* ---- the index concatenation across GPUs, one process per GPU (sjgpu_comm.hip) ------------------------------
 * NDJSON / parse_many shards are scanned independently (no collective in the scan); the only exchange the path has is the
 * variable-length gather of the per-shard structural lists to a consumer -- over RCCL / xGMI, device-resident in and out.
 * Global position of a structural = byte base of its shard + its shard-relative offset, the reference's own convention for
 * batches (include/simdjson/dom/document_stream-inl.h:250: batch_start + structural_indexes[i]).
 *   sjgpu_comm_unique_id   rank 0 makes the 128-byte id and hands it to the other ranks by any means (file, socket, MPI,
 *                          torch.distributed); sjgpu_comm_create is collective (ncclCommInitRank) and binds the rank to `device`.
 *   sjgpu_comm_gather_indices  collective.  Every rank: idx_dev[0..n) shard-relative u32 offsets in HBM, base = byte offset of its
 *                          shard.  ncclAllGather of (n, base, room at the root), then -- once the root has room: a root that must grow
 *                          its staging array does so FIRST and tells all ranks (one more word); if it cannot, every rank returns,
 *                          SJGPU_E_NOMEM at the root and SJGPU_E_PEER elsewhere, nothing posted -- every rank sends exactly n words
 *                          to `root` (ncclSend / ncclRecv,
 *                          all senders at once: xGMI is point to point), and the root writes base + offset as 64-bit words, shards
 *                          in rank order, into out_dev[0 .. *total_out) (out_cap_words >= the sum; SJGPU_E_OVERFLOW otherwise, after
 *                          the receives have drained).  counts_out (world entries, may be NULL): n of every rank.  Asynchronous on
 *                          `stream` except for the 24 * world bytes of counts the host needs to size the receives.
 *   RCCL is opened when the first of these calls arrives (dlopen: SJGPU_RCCL_LIB, librccl.so.1, $ROCM_PATH/lib); nothing else in
 *   the library needs it, and a box without it only loses these entry points (SJGPU_E_HIP, sjgpu_comm_last_error(NULL) says why). */
typedef struct sjgpu_comm sjgpu_comm;
#define SJGPU_COMM_ID_BYTES 128
int sjgpu_comm_unique_id(void *id_out, size_t id_bytes);
int sjgpu_comm_create(int rank, int world, const void *id, size_t id_bytes, int device, sjgpu_comm **out);
void sjgpu_comm_destroy(sjgpu_comm *comm);
const char *sjgpu_comm_last_error(const sjgpu_comm *comm);
int sjgpu_comm_ranks(const sjgpu_comm *comm); /* ncclCommCount: the ranks RCCL itself sees in the communicator; -1 on failure */
int sjgpu_comm_gather_indices(sjgpu_comm *comm, const void *idx_dev, uint32_t n, uint64_t base, int root, void *out_dev, size_t out_cap_words,
                              uint64_t *total_out, uint64_t *counts_out, void *stream);

/* ---- ranges of ONE resident buffer, one after the other (overlap the upload of batch k+1 with
 * the scan of batch k -- the GPU analogue of the reference's stage1_worker, dom/document_stream-inl.h:16-85).
 * Scans bytes [begin, end) of buf_dev; bytes [0, begin) must already be resident (escapes, the previous-scalar bit
 * and UTF-8 state are read from them), bytes beyond end need not be.  begin is a multiple of 1 MiB.  `more` != 0:
 * more ranges follow (no end-of-input checks).  in_string and n_before / out_before come from sjgpu_result() of
 * the previous range: (flags & (SJGPU_F_UNCLOSED_STRING | SJGPU_F_RANGE_CARRY)), n / out_len -- 0, 0 for the first range
 * (bit 0 of in_string: the range begins inside a string; SJGPU_F_RANGE_CARRY: it begins behind a backslash run of 64 bytes or
 * more whose parity the previous range knows -- /root/reference/src/generic/stage1/json_escape_scanner.h:50-71's one carried
 * bit, handed from call to call).  Offsets stay
 * relative to byte 0 and are appended at idx_dev[n_before...]; result.n / out_len are running totals.
 * in_string is exactly `flags & (SJGPU_F_UNCLOSED_STRING | SJGPU_F_RANGE_CARRY)` of the previous range's result (0 for the first range);
 * any other bit is SJGPU_E_BADARG.  (Through round 3 it was "non-zero = inside a string": a caller that passes 2, -1 or a bool other than 1,
 * or that forwards only bit 0 and so drops the escape carry of a backslash run across the cut, has to be changed -- INTEGRATION.md.)
 * sjgpu_stage1() and sjgpu_minify() drive exactly this for host buffers of 64 MiB and more (env
 * SJGPU_STREAM_FROM_MB / SJGPU_STREAM_CHUNK_MB), with the device-to-host copies on a second thread. */
int sjgpu_stage1_range_device(sjgpu_ctx *ctx, const void *buf_dev, size_t begin, size_t end, int more, int in_string,
                              uint32_t n_before, void *idx_dev, size_t idx_words, void *stream);
int sjgpu_minify_range_device(sjgpu_ctx *ctx, const void *buf_dev, size_t begin, size_t end, int more, int in_string,
                              uint32_t out_before, void *dst_dev, void *stream);

/* ---- windows of one stream (parse_many) -------------------------------------------------------------------------------
 * document_stream calls stage1 once per window of ONE buffer (include/simdjson/dom/document_stream-inl.h:285-317; 1 MB windows by
 * default); a launch and a PCIe round trip per megabyte lose to a CPU kernel.  The interface only ever sees a window, so the
 * integrator names the stream: between sjgpu_stream_register(base, len) and sjgpu_stream_unregister(base) the bytes
 * [base, base + len) must stay valid and unchanged (the range is page-locked meanwhile unless SJGPU_STREAM_PIN=0).  sjgpu_stage1
 * calls in the streaming_partial / streaming_final modes whose buffer lies inside a registered stream are then answered from a
 * span of 32 MiB scanned once -- identical results, microseconds per window (profiles / bench.py plugin_host_path).  The in-tree
 * patch registers in document_stream::start(); out of tree: simdjson::mi355x::register_stream.
 * Registrations are COUNTED per base address: registering a base again (a second stream over the same buffer) adds a reference, every
 * registration needs its own sjgpu_stream_unregister, and the range stays page-locked until the last one.  While several are alive the
 * extent served from spans is the SHORTEST length any of them named: a window beyond it takes the ordinary path -- same results, one upload
 * per window.  sjgpu_stream_unregister_len(base, len) says WHICH registration leaves (the one made with that len; if none was, the longest),
 * and the extent becomes the shortest of those that stay; sjgpu_stream_unregister(base) cannot say, so the longest is assumed to have left
 * (never unsafe, but a short-lived short registration then caps the extent for the life of the long one).  A caller that registers a base
 * twice and unregisters once leaks the reference (and the page-lock) until it unregisters again. */
int sjgpu_stream_register(const uint8_t *base, size_t len);
int sjgpu_stream_unregister(const uint8_t *base);
int sjgpu_stream_unregister_len(const uint8_t *base, size_t len);
size_t sjgpu_debug_stream_extent(const uint8_t *base); /* tests: the extent currently served from spans for the registration(s) made AT base; 0 = none */

/* ---- page-locked host memory ("a pinned-memory padded_string allocator") ----------------------
 * The host-buffer entry points accept any memory.  Ordinary (pageable) memory has to be pinned page by page by the
 * runtime on every call it has not seen before -- measured here ~25 GB/s per direction, 47-57 GB/s when the pages
 * are already locked (profiles/r01_host_path_overlap.txt).  An integrator who parses many documents allocates the
 * document buffer (the reference's padded_string, include/simdjson/padded_string.h) with sjgpu_host_alloc, or
 * registers long-lived arrays such as dom_parser_implementation::structural_indexes once with
 * sjgpu_host_register (about 19 ms per GiB), and unregisters before freeing them. */
void *sjgpu_host_alloc(size_t bytes);            /* NULL on failure */
void sjgpu_host_free(void *p);
int sjgpu_host_register(void *p, size_t bytes);  /* 0 or a negative SJGPU_E_* */
int sjgpu_host_unregister(void *p);

#ifdef __cplusplus
}
#endif
#endif /* SJGPU_H */
