// simdjson_amd/csrc/sjgpu_internal.h -- launch interface between the C-ABI (sjgpu_capi*.hip, sjgpu_ctx.h) and the
// gfx950 kernels (sjgpu_kernels.hip).  Geometry:
//
//   block   = 64 input bytes                = one lane
//   chunk   = 64 blocks = 4 KiB             = one wave64 pass
//   segment = SEG_CHUNKS chunks = 16 KiB    = one wave's (= one workgroup's) contiguous share
//
// Device workspace per context (all sized from `capacity`, resident in HBM between calls):
//   masks  2 planes x [ceil(cap/64)] x u64  per block: plane 0 = final structural mask (resolved segments) or
//                                  candidates; plane 1 = string_tail (unresolved segments only)
//   summ   [nseg + ngroups] x seg_summary  per-segment carry summary (quote parity, counts for both
//                                  in-string hypotheses, error bits), then the same per group of 64 segments
//   pref   [ngroups] x seg_prefix  per-group resolved carry-in (in-string bit, output base)
//   result                          one scan_result, mirrored to pinned host memory
#ifndef SJGPU_INTERNAL_H
#define SJGPU_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sjgpu {

constexpr uint32_t BLOCK_BYTES = 64;
constexpr uint32_t CHUNK_BYTES = 4096;
constexpr uint32_t SEG_CHUNKS = 4;
constexpr uint32_t SEG_BYTES = CHUNK_BYTES * SEG_CHUNKS;

// summary flags
constexpr uint32_t SF_PARITY = 1u;     // odd number of unescaped quotes in the segment
constexpr uint32_t SF_CTRL_IF_OUT = 2u; // control char inside a string if the segment starts OUTSIDE a string
constexpr uint32_t SF_CTRL_IF_IN = 4u;  // ... if it starts INSIDE a string
constexpr uint32_t SF_UTF8 = 8u;        // UTF-8 error in the segment
constexpr uint32_t SF_RESOLVED = 16u;   // the segment fixed its own in-string carry-in (first control character): its
                                        // mask plane 0 is final and both counts are equal
constexpr uint32_t SF_TOKENS = 32u;     // launch_stage1 with a token stream: the segment's structural bytes lie in its staging area
constexpr uint32_t SF_SPARSE = 64u;     // round 6: at most SPARSE_MAX candidates in the segment: the start of its plane-0 area holds them as a LIST of words
                                        // (bits 0-13 the candidate's byte offset inside the segment, bit 31 its string_tail bit) instead of 2 x 2 KiB of planes
constexpr uint32_t SPARSE_MAX = 32u;    // one 128-byte line of list words per sparse segment

struct seg_summary {
  uint32_t count_if_out; // structurals (stage1) / kept bytes (minify) if the segment starts outside a string
  uint32_t count_if_in;  // ... inside a string
  uint32_t flags;
  uint32_t xw;           // the x word of the summary (sj_xcarry.h): what a wrong look-back assumption changes, and what the successor's x is
};
struct seg_prefix {
  uint32_t base; // exclusive prefix of counts = first output slot of the segment
  uint32_t in_string; // bit 0: inside a string; bit 1: x, "the look-back assumption of this segment / group is wrong"
};
struct scan_result_dev {
  uint32_t n;
  uint32_t flags;
  uint64_t out_len;
};

inline uint32_t num_segments(uint64_t len) { return uint32_t((len + SEG_BYTES - 1) / SEG_BYTES); }
// split pipeline's resolve is two-level: 64 segments (1 MiB of input) form a group
constexpr uint32_t RESOLVE_GROUP = 64;
inline uint32_t num_groups(uint64_t len) { return (num_segments(len) + RESOLVE_GROUP - 1) / RESOLVE_GROUP; }

// single-pass pipeline (sjgpu_fused.hip): tile = one 256-thread workgroup = 4 waves x 4 chunks = 64 KiB;
// workspace = one 8-byte descriptor per tile + the ticket word behind them.
constexpr uint32_t FUSED_WAVE_CHUNKS = 4;
constexpr uint32_t FUSED_TILE_BYTES = 4 * FUSED_WAVE_CHUNKS * CHUNK_BYTES;
// k_minify_onchip (bytes of the pending tile wait in LDS): two chunks per wave, 4 or 8 waves per workgroup
constexpr uint32_t ONCHIP_WAVE_CHUNKS = 2;
constexpr uint32_t ONCHIP_TILE_BYTES = 4 * ONCHIP_WAVE_CHUNKS * CHUNK_BYTES;
// inputs up to FUSED_SMALL_BELOW use 16 KiB tiles (one chunk per wave): 4x the parallelism, 1/4 of the per-tile latency
constexpr uint64_t FUSED_SMALL_BELOW = uint64_t(8) << 20;
inline uint32_t num_fused_tiles(uint64_t len) { // descriptor words a context needs for documents up to len
  const uint64_t big = (len + ONCHIP_TILE_BYTES - 1) / ONCHIP_TILE_BYTES; // the smallest tile any large-input kernel uses
  const uint64_t small_len = len < FUSED_SMALL_BELOW ? len : FUSED_SMALL_BELOW;
  const uint64_t small = (small_len + FUSED_TILE_BYTES / FUSED_WAVE_CHUNKS - 1) / (FUSED_TILE_BYTES / FUSED_WAVE_CHUNKS);
  return uint32_t(big > small ? big : small);
}

// ---- launchers (sjgpu_kernels.hip); every one only enqueues on `stream` ---------------------------------
// ev: nullptr, or PROFILE_EVENTS events recorded around the kernels: ev[k], ev[k+1] bracket slot k (slot 0 includes the
// escape-table launch and the workspace clears in front of the first scan kernel; unused slots measure zero).
constexpr int PROFILE_SLOTS = 3;
constexpr int PROFILE_EVENTS = PROFILE_SLOTS + 1;
// carry: state in front of byte 0 when the buffer is a shard of a larger document; 0 for a whole document.
// Shards are cut by sjgpu_clean_cut, so the in-string bit is the ONLY state that crosses a cut.
constexpr uint32_t CARRY_IN_STRING = 1u; // the first byte scanned is inside a string
constexpr uint32_t CARRY_SHARD = 2u;     // minify: report out_len even if the scan ends inside a string
constexpr uint32_t CARRY_MORE = 4u;      // the scan does not end at the end of the input: no "sequence open at EOF" check
constexpr uint32_t CARRY_X = 8u;         // x in front of the first span (sj_xcarry.h): SJGPU_F_RANGE_CARRY of the range in front
constexpr uint32_t CARRY_DEBUG_LATE_TICKET = 0x100u; // A/B switch of the pipelined kernel (env SJGPU_LATE_TICKET)
constexpr uint32_t CARRY_DEBUG_NO_SPAN_HINT = 0x200u; // A/B switch: emission counts the span itself (env SJGPU_NO_SPAN_HINT)
constexpr uint32_t CARRY_DEBUG_LATE_LOOKBACK = 0x800u; // A/B switch of the pipelined kernels: the pending tile's look-back BEHIND the scan's barrier, as in rounds 1-5 (env SJGPU_LATE_LOOKBACK)
constexpr uint32_t CARRY_DEBUG_TOP_BARRIER = 0x1000u;  // A/B switch of the pipelined kernels: the third barrier, at the loop top, as in rounds 1-5 (env SJGPU_TOP_BARRIER)
constexpr uint32_t CARRY_DEBUG_TWO_TICKETS = 0x2000u;  // A/B switch of the pipelined kernels: tickets from two counters, even / odd tiles by the workgroup's parity (env SJGPU_TWO_TICKETS)
constexpr uint32_t CARRY_DEBUG_QUEUE_UTF8 = 0x400u;   // A/B switch: dense non-ASCII chunks are queued like sparse ones (env SJGPU_UTF8_QUEUE_ONLY)
// A scan covers bytes [begin, len) of a buffer whose bytes [0, begin) are resident too (the look-back of escapes,
// previous scalar and UTF-8 state reads them); begin is a multiple of RANGE_ALIGN.  Offsets stay relative to byte 0
// and are appended at output slot base0.  A whole document is {0, 0, 0}.
struct scan_origin {
  uint64_t begin;
  uint32_t base0;
  uint32_t carry;
};
constexpr uint64_t RANGE_ALIGN = uint64_t(1) << 20; // one resolve group = 16 large tiles = 64 small tiles
// tokstage / tok (both or neither): the token-byte stream beside the offsets -- tok[i] = buf[idx[i]] for i < n.  tokstage: scratch of num_segments(len -
// org.begin) * SEG_BYTES bytes, 16-byte aligned (the segments' structural bytes wait there between the two kernels); tok: room for n bytes
void launch_stage1(const uint8_t *buf, uint64_t len, uint4 *masks, seg_summary *summ, seg_prefix *pref, uint32_t *idx,
                   uint64_t idx_words, scan_result_dev *result, scan_origin org, hipStream_t stream, hipEvent_t *ev, uint8_t *tokstage = nullptr,
                   uint8_t *tok = nullptr);
void launch_minify(const uint8_t *buf, uint64_t len, seg_summary *summ, seg_prefix *pref, uint8_t *dst,
                   scan_result_dev *result, scan_origin org, hipStream_t stream, hipEvent_t *ev);
// result->n = parity; workspace: one byte per 16 KiB segment of the buffer (the per-segment parities and x words, folded by a second launch)
void launch_string_parity(const uint8_t *buf, uint64_t len, scan_result_dev *result, uint8_t *workspace, hipStream_t stream);
// (rounds 1-3 built an escape table in front of every large scan -- one byte per 16 KiB boundary: is the byte behind it escaped? -- by walking
// back over the backslash run in front; round 4 carries that bit through the summaries instead, sj_xcarry.h, and the table is gone.)
// one byte per 16 KiB segment of a 4 GiB buffer: the per-segment scratch of launch_string_parity
constexpr size_t SEGMENT_BYTES_TABLE = ((uint64_t(1) << 32) / SEG_BYTES + 1024) & ~size_t(1023);
// bytes [begin, len) of buf; begin > 0 (a multiple of 4 KiB): a later range of a resident buffer, the flags add up in *result;
// more: the input continues behind len
void launch_validate_utf8(const uint8_t *buf, uint64_t len, scan_result_dev *result, hipStream_t stream, hipEvent_t *ev,
                          uint64_t begin = 0, bool more = false);
// single-pass variants: desc holds num_fused_tiles(capacity) + FUSED_WORKSPACE_EXTRA_WORDS words; only profile slot 0 is used.
// clean: *result, the descriptors and the control words behind them are all zero on the device.  Every (un-traced) single-pass kernel leaves them
// that way when it ends -- its last workgroup to leave clears what the call used (sjgpu_fused.hip: leave_and_clean) -- so a caller that cleared the
// whole workspace once, when it allocated it, passes true from then on and a call is ONE dispatch; with false the launcher clears in front of the kernel.
constexpr uint32_t FUSED_WORKSPACE_EXTRA_WORDS = 18; // control words: [ticket, workgroups gone][flags, -], fourteen words of distance, [second ticket counter] (its own 128-byte line)
const char *launch_stage1_fused(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                scan_result_dev *result, scan_origin org, uint32_t max_workgroups, hipStream_t stream,
                                hipEvent_t *ev, bool clean = false, uint8_t *tok = nullptr); // tok: the token-byte stream beside the offsets (round 6: gathered at emission) // -> name of the scan kernel launched
// round 6: the one-pass kernel for PLAIN input (every 16 KiB segment pins its own string state at a control character of its first chunk, no span assumed its
// escape carry): additive prefixes, emission deferred by one tile (sjgpu_fused.hip: k_stage1_direct).  Input that is not plain: SJGPU_F_INTERNAL in the result.
// Needs num_direct_words(len) + FUSED_WORKSPACE_EXTRA_WORDS descriptor words (never more than num_fused_tiles(len) from 8 MiB on).
const char *launch_stage1_direct(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words, scan_result_dev *result, scan_origin org,
                                 uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean = false);
void launch_stage1_fused_traced(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                scan_result_dev *result, uint32_t max_workgroups, hipStream_t stream, uint64_t *trace,
                                uint32_t trace_tiles); // 8 wall_clock64 stamps (100 MHz) per tile
// ---- the structural list after the scan (sjgpu_finish.hip) ----------------------------------------------------------------
enum : uint32_t { FIN_SEARCH = 0, FIN_KEEP_GIVEN = 1, FIN_TOO_LARGE = 2, FIN_COUNT_BELOW = 3 };
struct finish_state {            // device-resident, zero-initialised; read back by the host once the kernels are done
  uint32_t n_in, n_cur, n_report; // structurals handed in / the boundary search runs on / the reference's `n` after the filter
  uint32_t next_start;            // where the next batch begins (len unless a separator says otherwise)
  uint32_t kept, keep, verdict;   // compacted length; structurals of complete documents; FIN_*
  uint32_t separators, last_sep_index_plus1, last_sep_pos_plus1; // root commas / record separators seen
  uint32_t boundary_plus1;        // list index + 1 of the last value that directly follows a value
  int32_t obj_balance, arr_balance;
  uint32_t arrivals;              // workgroups of k_tail_balance that have added their part (the last one resolves)
  uint32_t pad[2];
};
size_t finish_workspace_bytes(uint32_t n);
// modes 1..6 of sjgpu_stage1_mode on the n raw structurals idx[0..n) (n > 0) of buf[0..len): filters the list in place
// (json_sequence / comma_delimited) and leaves the decision in the finish_state at the start of `workspace`
void launch_finish(int mode, const uint8_t *buf, uint64_t len, uint32_t *idx, uint32_t n, void *workspace, hipStream_t stream);
// depth[i] = nesting depth in front of structural i for i in [0, n], depth[n] = behind the last; scratch: depth_scan_scratch_bytes(n)
size_t depth_scan_scratch_bytes(uint32_t n);
// tok (optional): the token stream of the list (tok[i] = buf[idx[i]]); with it neither buf nor idx is read
void launch_depth_scan(const uint8_t *buf, const uint32_t *idx, uint32_t n, int32_t *depth, void *scratch, hipStream_t stream, const uint8_t *tok = nullptr);

// ---- one workgroup per document (sjgpu_small.hip) ---------------------------------------------------------------------
// in_off: byte offset of the document in the input block (a multiple of 64); out_off: first output unit of the document in
// the output block (stage 1: words, a multiple of 4, room for len + 3; minify: bytes, a multiple of 16, room for len)
struct doc_desc {
  uint64_t in_off;
  uint64_t out_off;
  uint32_t len;
  uint32_t flags;
};
constexpr uint32_t DOC_KEEP_UNCLOSED = 1u; // minify: report out_len even if the document ends inside a string
// documents up to this size take the one-workgroup kernel when they come one by one (sjgpu_stage1 & co.)
constexpr size_t DOCS_SINGLE_MAX = size_t(64) << 10;
// op 0 stage 1, 1 minify, 2 validate_utf8; docs == nullptr: one document, described by `single`.  in_base / out_base /
// results may be device memory or page-locked host memory (the kernel then moves the bytes across PCIe itself).
void launch_docs(int op, const uint8_t *in_base, const doc_desc *docs, doc_desc single, uint32_t count, void *out_base,
                 scan_result_dev *results, hipStream_t stream);

uint32_t launch_stage1_pipelined_traced(const uint8_t *buf, uint64_t len, uint64_t *desc, uint32_t *idx, uint64_t idx_words,
                                        scan_result_dev *result, uint32_t max_workgroups, hipStream_t stream,
                                        uint64_t *trace, uint32_t max_records); // -> workgroups launched (32 records of 8 stamps each)
const char *launch_minify_fused(const uint8_t *buf, uint64_t len, uint64_t *desc, uint8_t *dst, scan_result_dev *result,
                                scan_origin org, uint32_t max_workgroups, hipStream_t stream, hipEvent_t *ev, bool clean = false);

// ---- strings (sjgpu_strings.hip) -------------------------------------------------------------------------------
struct strings_result_dev {
  uint64_t bytes;     // string buffer bytes used
  uint32_t strings;   // records written
  uint32_t first_bad; // index of the first structural whose string the reference rejects, 0xFFFFFFFF = none
  uint32_t overflow;  // a record did not fit into the caller's buffer
  uint32_t path;      // which kernels wrote the buffer: 1 = the stream compaction (sjgpu_string_stream.hip), 2 = the per-string walk
};
// scratch of one string pass, carved from one allocation of strings_scratch_bytes(n, len) bytes (256-byte aligned pieces)
constexpr size_t STRS_SUMMARY_BYTES = 32, STRS_BASE_BYTES = 16;
struct strings_scratch {
  void *ctrl;        // 64 bytes: scan lengths, which path writes the buffer, totals (sjgpu_string_stream.hip: strs_ctrl)
  int *partial;      // block sums of the scans
  int *kord;         // n + 2 ints of room: string tokens per tile of 4096 structurals (then their prefixes) and one bit per structural (sjgpu_string_stream.hip)
  uint32_t *outq;    // n + 2: where the k-th string's record begins
  void *seg_summary; // per 16 KiB segment of the document
  void *seg_base;
  size_t bytes;
};
strings_scratch carve_strings_scratch(void *base, uint32_t n, uint64_t len);
size_t strings_scratch_bytes(uint32_t n, uint64_t len);
// the buffer as a stream compaction of the document (sjgpu_string_stream.hip); leaves in the control block whether the
// per-string kernels have to run instead
void enqueue_string_stream(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, bool allow_replacement, uint8_t *out, uint64_t out_cap,
                           uint32_t *offsets, strings_result_dev *res, const strings_scratch &w, hipStream_t s, const int *listed);
// What the string pass hands to a caller that finishes the records itself (the tape): where the k-th string's record begins, and the device
// flag that says whether the stream compaction wrote the buffer (then the length words are still missing) or the per-string kernels did.
struct strings_handoff {
  const uint32_t *outq;
  const uint32_t *go_stream;
};
// listed (optional): device pointer to the number of string tokens of the list, counted by the caller (launch_tape_front) -- the pass then
// neither counts them itself nor writes offsets / length words for the stream's records (the caller does, from the handoff: the k-th string
// token's record begins at outq[k])
// roads: which of the two roads to the buffer are enqueued.  STRINGS_BOTH: the stream compaction and, behind it, the per-string kernels, which return at once
// when the stream took the document (the decision is taken on the device: stand-alone sjgpu_parse_strings_device).  STRINGS_STREAM_ONLY: the stream alone --
// a document it declines comes back with res->path == 2 and NOTHING written, and the caller enqueues the pass again with STRINGS_WALK_ONLY (the verdict
// overridden: the per-string kernels write).  sjgpu_stage2_device takes that optimistic route: five launches that nearly always do nothing are not enqueued.
enum : int { STRINGS_BOTH = 0, STRINGS_STREAM_ONLY = 1, STRINGS_WALK_ONLY = 2 };
strings_handoff launch_parse_strings(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, bool allow_replacement, uint8_t *out, uint64_t out_cap,
                                     uint32_t *offsets, strings_result_dev *res, void *scratch, hipStream_t s, const int *listed = nullptr, int roads = STRINGS_BOTH);
// ---- the tape (sjgpu_tape.hip) ----------------------------------------------------------------------------------
struct tape_result_dev {
  uint64_t error_key;   // smallest (list index << 8 | rank << 4 | error_code) over all offending tokens, ~0 = none (sj_tape_rules.h)
  uint64_t tape_words;  // words of the finished tape (both root words included)
  uint32_t slow_numbers; // number tokens that took the big-integer decision
  uint32_t overflow;    // the caller's tape was too small
  uint32_t max_level;   // the highest nesting level among the brackets and commas: 64 and more needs the sort's second pass (launch_tape's `deep`)
  uint32_t pad;
};
constexpr uint32_t TAPE_ONE_PASS_LEVELS = 64; // levels one pass of the sort tells apart
size_t tape_workspace_bytes(uint32_t n, uint64_t len);
// stage 2 of buf[0..len) from its structural list idx[0..n] (idx[n] = len), in two halves around the string pass: launch_tape_front leaves
// the token bytes and the block totals of the per-token counters in the workspace and returns the number of string tokens (device);
// launch_tape writes the reference's tape (and the length words of the string records when the stream compaction wrote them).
// workspace: tape_workspace_bytes(n, len), its first bytes are the tape_result_dev afterwards
// tok (optional): the token stream of the list (tok[i] = buf[idx[i]], sjgpu_stage1_tokens_device): the front then reads it instead of gathering
const int *launch_tape_front(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, void *workspace, hipStream_t s,
                             const uint8_t *tok = nullptr);
// deep: enqueue the sort's second pass (documents nested 64 deep and more).  Without it a deeper document comes back with max_level >= 64 in the result and
// a tape that is NOT valid: the caller runs the three launches again with deep = true (sjgpu_stage2_device: optimistic, five launches saved on nearly every call).
void launch_tape(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, uint32_t max_depth, const uint32_t *str_offsets, strings_handoff strs,
                 uint8_t *string_buf, uint64_t *tape, uint64_t tape_cap, void *workspace, hipStream_t s, bool deep = true);
// On-Demand's raw key comparison over the whole list (sjgpu_strings.hip); names_block: [u32 lens[K]][name bytes back to back] in device memory
void launch_match_keys(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, const uint8_t *names_block, uint32_t K, uint32_t *out, uint32_t *matches,
                       hipStream_t s);
// in-place exclusive scan of a[0 .. *n_ptr) (n_max >= *n_ptr sizes the grid); partial: blocks_for(n_max, 4096) + 64 ints (sjgpu_finish.hip)
void enqueue_scan(int *a, uint32_t n_max, const uint32_t *n_ptr, int *partial, hipStream_t s);
// in-place exclusive scan of a[0 .. count), count <= 2^20, by one workgroup (sjgpu_finish.hip: k_scan_partials)
void launch_scan_partials(int *a, uint32_t count, hipStream_t s);

} // namespace sjgpu
#endif
