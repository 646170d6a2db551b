// simdjson_amd/csrc/sj_tape_rules.h -- the reference's stage-2 walk as rules a token can check by ITSELF.
//
// json_iterator::walk_document (/root/reference/src/generic/stage2/json_iterator.h:121-244) is a state machine over the structural
// list that stops at the first token it does not expect.  Which token it expects depends on three things only: the token in front,
// whether that token was a key, and the kind of the innermost open container -- so "the first error of the serial walk" equals
// "the smallest list index whose token breaks its local rule", and every token can be judged alone once a prefix scan has
// delivered its nesting depth and a bracket match the kind of its container (sjgpu_tape.hip).  An error is reported as a KEY
// (list index, rank inside the token, error code) whose minimum over all tokens is the reference's answer.
//
// Also here: how many tape words a token writes (tape_builder.h / tape_writer.h, /root/reference/doc/tape.md).
// Host + device: sjgpu_tape.hip and tests/host/test_tape_model.cpp use the same functions.
#ifndef SJGPU_SJ_TAPE_RULES_H
#define SJGPU_SJ_TAPE_RULES_H

#include "sj_number.h"

namespace sjgpu {

enum : u32 { CTX_NONE = 0, CTX_OBJECT = 1, CTX_ARRAY = 2 };

// error key: smaller = reported first.  rank 0: the token is not what the walk expects (TAPE_ERROR), 1: nesting too deep
// (DEPTH_ERROR, checked after the token was accepted: json_iterator.h:165-166, :206-207), 2: the token's own content (string,
// number, atom: the visitor runs last).
SJ_HD u64 error_key(u64 index, u32 rank, u32 code) { return (index << 8) | (u64(rank) << 4) | code; }
constexpr u64 NO_ERROR_KEY = ~u64(0);
SJ_HD u32 error_code_of(u64 key) { return key == NO_ERROR_KEY ? u32(SJ_SUCCESS) : u32(key & 0xFu); }

SJ_HD bool is_open_char(u32 c) { return c == '{' || c == '['; }
SJ_HD bool is_close_char(u32 c) { return c == '}' || c == ']'; }
// what visit_primitive hands to parse_number INSIDE a container: `(*value - '0') < 10` in int arithmetic holds for every byte
// below ':' (json_iterator.h:345-347) -- a ',' or '!' in a value position is a NUMBER_ERROR, not a TAPE_ERROR
SJ_HD bool takes_number_path(u32 c, bool root) { return root ? (c - '0' <= 9u || c == '-') : (c <= '9' && c != '"'); }
// a token the walk accepts where a value is expected (its content may still be rejected by the visitor)
SJ_HD bool starts_value(u32 c, bool root) { return c == '"' || is_open_char(c) || takes_number_path(c, root) || c == 't' || c == 'f' || c == 'n'; }

// tape words of token c (0 for ':' and ','; numbers take the type word and the value word)
SJ_HD u32 tape_slots(u32 c, bool root) {
  if (c == ':' || c == ',') { return 0; }
  if (c != '"' && !is_open_char(c) && !is_close_char(c) && takes_number_path(c, root)) { return 2; }
  return 1;
}

// 0, or SJ_TAPE_ERROR / SJ_DEPTH_ERROR with its rank in *rank.
//   i: list index; c / prev / prev2 / next: the bytes at idx[i], idx[i-1], idx[i-2], idx[i+1] (0 where there is none);
//   ctx_prev / ctx_prev2: kind of the container a ',' at i-1 / i-2 sits in (CTX_*; only read when that token is a ',');
//   depth: containers open in front of token i.
SJ_HD u32 token_grammar_error(u64 i, u32 c, u32 prev, u32 prev2, u32 next, u32 ctx_prev, u32 ctx_prev2, long long depth, u32 max_depth, u32 *rank) {
  *rank = 0;
  bool ok;
  if (i == 0) { // the root value (json_iterator.h:133-152, visit_root_primitive :313-340)
    ok = starts_value(c, true);
  } else if (depth <= 0) { // the root value has ended: "more than one JSON value at the root" (:236-239)
    ok = false;
  } else if (prev == '{') { // object_begin (:162-175)
    ok = c == '"' || c == '}';
  } else if (prev == '[') { // array_begin / array_value (:203-219)
    ok = c == ']' || starts_value(c, false);
  } else if (prev == ':') { // object_field (:177-186)
    ok = starts_value(c, false);
  } else if (prev == ',') { // object_continue (:189-197) wants a key, array_continue (:222-223) a value
    ok = ctx_prev == CTX_OBJECT ? c == '"' : (ctx_prev == CTX_ARRAY && starts_value(c, false));
  } else {
    const bool prev_is_key = prev == '"' && (prev2 == '{' || (prev2 == ',' && ctx_prev2 == CTX_OBJECT));
    ok = prev_is_key ? c == ':' : (c == ',' || is_close_char(c)); // a value has ended (:188-201, :221-226); bracket kinds are matched elsewhere
  }
  if (!ok) { return SJ_TAPE_ERROR; }
  // a non-empty container one level too deep (:165-166, :206-207); empty ones are written without descending (:146-147 ...)
  if (is_open_char(c) && next != (c == '{' ? u32('}') : u32(']')) && depth + 1 >= (long long)max_depth) {
    *rank = 1;
    return SJ_DEPTH_ERROR;
  }
  return 0;
}

// A ',' where the walk expects a VALUE (behind '[', ':' or an array's ',') is not a separator: visit_primitive hands it to
// parse_number like every byte below ':' and the document ends with NUMBER_ERROR (content rank), e.g. "[ ,1]" or {"a":,}.
SJ_HD bool comma_in_value_position(u64 i, u32 prev, u32 ctx_prev) {
  return i > 0 && (prev == '[' || prev == ':' || (prev == ',' && ctx_prev == CTX_ARRAY));
}

// tape words (/root/reference/doc/tape.md, tape_writer.h)
SJ_HD u64 tape_word(u32 type, u64 payload) { return (u64(type) << 56) | payload; }

} // namespace sjgpu
#endif
