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
//   (first / int depth: the form the kernels call -- 32-bit arithmetic only; depths beyond +-2^31 do not occur, the list has < 2^32 entries)
SJ_HD u32 token_rule(bool first, u32 c, u32 prev, u32 prev2, u32 next, u32 ctx_prev, u32 ctx_prev2, int depth, u32 max_depth, u32 *rank);
SJ_HD u32 token_grammar_error(u64 i, u32 c, u32 prev, u32 prev2, u32 next, u32 ctx_prev, u32 ctx_prev2, long long depth, u32 max_depth, u32 *rank) {
  const int d = depth > 0x7FFFFFFFll ? 0x7FFFFFFF : (depth < -0x7FFFFFFFll ? -0x7FFFFFFF : int(depth));
  return token_rule(i == 0, c, prev, prev2, next, ctx_prev, ctx_prev2, d, max_depth, rank);
}
SJ_HD u32 token_rule(bool first, u32 c, u32 prev, u32 prev2, u32 next, u32 ctx_prev, u32 ctx_prev2, int depth, u32 max_depth, u32 *rank) {
  *rank = 0;
  bool ok;
  if (first) { // the root value (json_iterator.h:133-152, visit_root_primitive :313-340)
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
  if (is_open_char(c) && next != (c == '{' ? u32('}') : u32(']')) && max_depth <= 0x7FFFFFFFu && depth >= int(max_depth) - 1) { // depth + 1 >= max_depth
    *rank = 1;
    return SJ_DEPTH_ERROR;
  }
  return 0;
}

// ---- the token front: what one token contributes to the prefix sums, and where it is listed -----------------------------------------------
// (sjgpu_tape.hip: k_tok_classify / k_tok_apply; tests/host/test_tape_model.cpp runs the same functions)
struct tok_packed {
  u32 a, b, c; // a: tape words | sort flag << 16;  b: string | opening << 16;  c: closing | number << 16  (block sums of 4096 tokens fit the fields)
};
SJ_HD tok_packed tok_contribution(u32 ch, bool root) {
  const u32 open = is_open_char(ch) ? 1u : 0u, close = is_close_char(ch) ? 1u : 0u;
  tok_packed p;
  const u32 slots = tape_slots(ch, root); // two words: a number token (visit_primitive's number path)
  p.a = slots | ((open | close | (ch == ',' ? 1u : 0u)) << 16);
  p.b = (ch == '"' ? 1u : 0u) | (open << 16);
  p.c = close | ((slots == 2u ? 1u : 0u) << 16);
  return p;
}
// one-word tokens that are neither strings nor brackets (atoms, and bytes that are no token) among tokens with these counts:
// tape words = strings + rest + 2 numbers + opens + closes
SJ_HD int one_word_rest(int words, int strings, int numbers, int opens, int closes) { return words - strings - 2 * numbers - opens - closes; }
// which list a token goes to: the value kernels of its kind take it from there
enum : u32 { LIST_NONE = 0, LIST_NUMBERS = 1, LIST_STRINGS = 2, LIST_REST = 3 };
SJ_HD u32 value_list_of(const tok_packed &p) {
  if (p.c >> 16) { return LIST_NUMBERS; }
  if (p.b & 0xFFFFu) { return LIST_STRINGS; }
  return ((p.a & 0xFFFFu) == 1u && !(p.a >> 16)) ? u32(LIST_REST) : u32(LIST_NONE); // (brackets carry the sort flag, ':' and ',' no word)
}
SJ_HD u64 list_entry(u32 tape_position, u32 token) { return (u64(tape_position) << 32) | token; }
// what an element of the sort is, kept in the four bits of its 16-bit key the level (<= 4095) leaves free: the passes behind the sort
// then never have to look the token's byte up again.  A comma also carries what it can say about its two followers before anybody knows its
// container -- would an object be content with them, would an array (comma_fine_bits, below) -- so that k_tape_match, which learns the container's
// kind, has nothing to fetch for it unless the document is broken there.  What travels WITH the key (the sort's 32-bit payload): a bracket's tape
// position (the two bracket words are written from payloads alone), a comma's list index (for the error key of the rare broken follower).
constexpr u32 KIND_SHIFT = 12, KIND_OPEN_OBJECT = 1, KIND_OPEN_ARRAY = 2, KIND_CLOSE_OBJECT = 3, KIND_CLOSE_ARRAY = 4, KIND_COMMA = 8; // commas: 8 ... 11
constexpr u32 COMMA_FINE_IN_OBJECT = 1, COMMA_FINE_IN_ARRAY = 2;
SJ_HD u32 sort_kind(u32 ch, u32 comma_bits) {
  return ch == '{' ? KIND_OPEN_OBJECT : (ch == '[' ? KIND_OPEN_ARRAY : (ch == '}' ? KIND_CLOSE_OBJECT : (ch == ']' ? KIND_CLOSE_ARRAY : KIND_COMMA + comma_bits)));
}
SJ_HD bool kind_is_open(u32 kind) { return kind == KIND_OPEN_OBJECT || kind == KIND_OPEN_ARRAY; }
SJ_HD bool kind_is_comma(u32 kind) { return kind >= KIND_COMMA; }
SJ_HD u32 sort_key(u32 level, u32 ch, u32 comma_bits) { return level | (sort_kind(ch, comma_bits) << KIND_SHIFT); }

// ---- the same rule from tables ---------------------------------------------------------------------------------------------------------
// token_rule spelled out is ~60 boolean operations on per-lane conditions, which the GPU compiler turns into as many scalar mask
// instructions per token: the kernel that only applies the rule was bound by their issue (profiles/r03_pmc_summary.txt: 250 M scalar
// against 100 M vector instructions per call).  Here the bytes are looked up: what a byte IS (props), which state the walk is in behind a
// byte (state_behind), and which properties each state accepts (accepts) -- three small tables a workgroup builds in LDS from the very
// predicates above (so they cannot drift apart), a handful of selects on small integers, one AND.  tests/host/test_tape_rules.cpp compares
// the two forms on every combination of bytes, container kinds and depths that can make a difference.
enum : u32 { P_QUOTE = 1, P_LBRACE = 2, P_LBRACKET = 4, P_RBRACE = 8, P_RBRACKET = 16, P_COLON = 32, P_COMMA = 64, P_VALUE_AT_ROOT = 128, P_VALUE_INSIDE = 256 };
enum : u32 { ST_ROOT = 0, ST_DONE = 1, ST_BEHIND_LBRACE = 2, ST_BEHIND_LBRACKET = 3, ST_BEHIND_COLON = 4, ST_BEHIND_COMMA = 5 /* + CTX_* */, ST_BEHIND_KEY = 8,
              ST_BEHIND_VALUE = 9, ST_BEHIND_QUOTE = 10 /* key or value: decided by the token in front of it */, ST_COUNT = 11 };
SJ_HD u32 byte_props_of(u32 c) {
  return (c == '"' ? P_QUOTE : 0u) | (c == '{' ? P_LBRACE : 0u) | (c == '[' ? P_LBRACKET : 0u) | (c == '}' ? P_RBRACE : 0u) | (c == ']' ? P_RBRACKET : 0u) |
         (c == ':' ? P_COLON : 0u) | (c == ',' ? P_COMMA : 0u) | (starts_value(c, true) ? P_VALUE_AT_ROOT : 0u) | (starts_value(c, false) ? P_VALUE_INSIDE : 0u);
}
SJ_HD u32 state_behind_byte(u32 prev) {
  return prev == '{' ? ST_BEHIND_LBRACE : (prev == '[' ? ST_BEHIND_LBRACKET : (prev == ':' ? ST_BEHIND_COLON : (prev == ',' ? ST_BEHIND_COMMA : (prev == '"' ? ST_BEHIND_QUOTE : ST_BEHIND_VALUE))));
}
SJ_HD u32 state_accepts(u32 st) {
  switch (st) {
  case ST_ROOT: return P_VALUE_AT_ROOT;
  case ST_BEHIND_LBRACE: return P_QUOTE | P_RBRACE;
  case ST_BEHIND_LBRACKET: return P_RBRACKET | P_VALUE_INSIDE;
  case ST_BEHIND_COLON: return P_VALUE_INSIDE;
  case ST_BEHIND_COMMA + CTX_OBJECT: return P_QUOTE;
  case ST_BEHIND_COMMA + CTX_ARRAY: return P_VALUE_INSIDE;
  case ST_BEHIND_KEY: return P_COLON;
  case ST_BEHIND_VALUE: return P_COMMA | P_RBRACE | P_RBRACKET;
  default: return 0u; // ST_DONE, a ',' that sits in no container
  }
}
struct rule_tables {
  const unsigned short *props;  // [256]
  const u8 *state_behind;       // [256]
  const unsigned short *accepts; // [ST_COUNT]
};
// entry t of the three tables (t < 256; the accepts table is shorter): what thread t of a workgroup writes
SJ_HD void rule_table_entry(u32 t, unsigned short *props, u8 *state_behind, unsigned short *accepts) {
  props[t] = (unsigned short)byte_props_of(t);
  state_behind[t] = u8(state_behind_byte(t));
  if (t < ST_COUNT) { accepts[t] = (unsigned short)state_accepts(t); }
}
SJ_HD u32 token_rule_tables(const rule_tables &T, bool first, u32 c, u32 prev, u32 prev2, u32 next, u32 ctx_prev, u32 ctx_prev2, int depth, u32 max_depth, u32 *rank) {
  *rank = 0;
  const u32 pc = T.props[c & 0xFFu], p2 = T.props[prev2 & 0xFFu];
  u32 st = T.state_behind[prev & 0xFFu];
  const u32 behind_key = ((p2 & P_LBRACE) != 0u) | (((p2 & P_COMMA) != 0u) & (ctx_prev2 == CTX_OBJECT));
  st = st == ST_BEHIND_COMMA ? ST_BEHIND_COMMA + (ctx_prev <= CTX_ARRAY ? ctx_prev : 0u) : st;
  st = st == ST_BEHIND_QUOTE ? (behind_key ? u32(ST_BEHIND_KEY) : u32(ST_BEHIND_VALUE)) : st;
  st = depth <= 0 ? u32(ST_DONE) : st;
  st = first ? u32(ST_ROOT) : st;
  if ((pc & T.accepts[st]) == 0u) { return SJ_TAPE_ERROR; }
  // a non-empty container one level too deep ('{' + 2 = '}', '[' + 2 = ']')
  if ((pc & (P_LBRACE | P_LBRACKET)) != 0u && next != c + 2u && max_depth <= 0x7FFFFFFFu && depth >= int(max_depth) - 1) {
    *rank = 1;
    return SJ_DEPTH_ERROR;
  }
  return 0;
}

// ---- the rule without a ctx array and without a depth array (round 4) ------------------------------------------------------------------------
// Two of the rule's inputs used to be arrays of their own: ctx (the kind of the container a ',' sits in: one byte per token, cleared per call,
// scattered by k_tape_match one byte per comma -- a sector of HBM traffic each -- and read back by k_tape_rules) and depth (4 B per token, written
// by k_tok_apply for k_tape_rules to read).  Neither needs to exist:
//   * a container kind matters to two tokens only: the one BEHIND a ',' and the one behind a '"' that stands behind a ','.  Both are the comma's
//     followers, so the COMMA judges them (comma_followers_rule) when k_tape_match visits it with its container's kind in hand; every other token
//     judges itself from the two bytes in front of it (token_rule_self).  A comma k_tape_match finds no container for judges nobody -- and needs
//     not: it stands at depth <= 0 and is itself the smaller error (depth_rule), or beyond the nesting limit, where the bracket that went too deep is.
//   * the depth decides two things -- "the root value has ended" (depth <= 0) and the nesting limit -- and both are said where the depth is
//     computed (k_tok_apply: depth_rule).
// The verdicts are keys; their minimum is the one token_rule / token_rule_tables give (tests/host/test_tape_rules.cpp: every combination; the
// model of tests/host/test_tape_model.cpp and the kernels run THIS form).
SJ_HD bool judged_by_comma(u32 prev, u32 prev2) { return prev == ',' || (prev == '"' && prev2 == ','); }
// what the depth in front of token i says: 0, SJ_TAPE_ERROR (rank 0: the root value has ended) or SJ_DEPTH_ERROR (rank 1)
SJ_HD u32 depth_rule(bool first, u32 c, u32 next, int depth, u32 max_depth, u32 *rank) {
  *rank = 0;
  if (!first && depth <= 0) { return SJ_TAPE_ERROR; }
  if (is_open_char(c) && next != c + 2u && max_depth <= 0x7FFFFFFFu && depth >= int(max_depth) - 1) { // ('{' + 2 = '}', '[' + 2 = ']')
    *rank = 1;
    return SJ_DEPTH_ERROR;
  }
  return 0;
}
// what a token can say about itself (depth > 0 assumed unless first; the followers of a comma are not judged here): 0, SJ_TAPE_ERROR (rank 0)
// or SJ_NUMBER_ERROR (rank 2: a ',' where '[' or ':' wants a value)
SJ_HD u32 token_rule_self(const rule_tables &T, bool first, u32 c, u32 prev, u32 prev2, u32 *rank) {
  *rank = 0;
  if (!first && judged_by_comma(prev, prev2)) { return 0; }
  const u32 pc = T.props[c & 0xFFu];
  u32 st = T.state_behind[prev & 0xFFu];
  st = st == ST_BEHIND_QUOTE ? ((prev2 == '{') ? u32(ST_BEHIND_KEY) : u32(ST_BEHIND_VALUE)) : st;
  st = first ? u32(ST_ROOT) : st;
  if ((pc & T.accepts[st]) == 0u) { return SJ_TAPE_ERROR; }
  if (!first && c == ',' && (prev == '[' || prev == ':')) { *rank = 2; return SJ_NUMBER_ERROR; }
  return 0;
}
// a ',' at list index j inside a container of kind ctx (CTX_OBJECT / CTX_ARRAY) judges the tokens behind it: c1 = token j + 1, c2 = token j + 2
// (have1 / have2: they exist).  Up to three keys: k[0] token j + 1 is not what the container wants (TAPE_ERROR), k[1] it is a ',' in an array's value
// position (NUMBER_ERROR), k[2] token j + 2 does not follow the string at j + 1 the way a key / a value is followed.  NO_ERROR_KEY where there is none.
struct follower_keys { u64 k[3]; };
SJ_HD follower_keys comma_followers_rule(u64 j, u32 ctx, u32 c1, bool have1, u32 c2, bool have2) {
  follower_keys f{{NO_ERROR_KEY, NO_ERROR_KEY, NO_ERROR_KEY}};
  if (!have1 || (ctx != CTX_OBJECT && ctx != CTX_ARRAY)) { return f; }
  const bool object = ctx == CTX_OBJECT;
  const bool ok1 = object ? c1 == '"' : starts_value(c1, false);       // object_continue wants a key, array_continue a value
  if (!ok1) { f.k[0] = error_key(j + 1, 0, SJ_TAPE_ERROR); }
  if (!object && c1 == ',') { f.k[1] = error_key(j + 1, 2, SJ_NUMBER_ERROR); } // comma_in_value_position
  if (have2 && c1 == '"') {
    const bool ok2 = object ? c2 == ':' : (c2 == ',' || is_close_char(c2)); // behind a key / behind a value
    if (!ok2) { f.k[2] = error_key(j + 2, 0, SJ_TAPE_ERROR); }
  }
  return f;
}

// what the comma at list index j knows about its followers without its container: COMMA_FINE_IN_OBJECT / COMMA_FINE_IN_ARRAY = a container of that
// kind would raise nothing (derived from comma_followers_rule itself: the two cannot drift apart)
SJ_HD u32 comma_fine_bits(u64 j, u32 c1, bool have1, u32 c2, bool have2) {
  const follower_keys o = comma_followers_rule(j, CTX_OBJECT, c1, have1, c2, have2), a = comma_followers_rule(j, CTX_ARRAY, c1, have1, c2, have2);
  const bool fine_o = o.k[0] == NO_ERROR_KEY && o.k[1] == NO_ERROR_KEY && o.k[2] == NO_ERROR_KEY;
  const bool fine_a = a.k[0] == NO_ERROR_KEY && a.k[1] == NO_ERROR_KEY && a.k[2] == NO_ERROR_KEY;
  return (fine_o ? COMMA_FINE_IN_OBJECT : 0u) | (fine_a ? COMMA_FINE_IN_ARRAY : 0u);
}
// the same, spelled out (what k_tok_apply evaluates for every comma; tests/host/test_tape_rules.cpp: equal to comma_fine_bits on every combination)
SJ_HD u32 comma_fine_bits_direct(u32 c1, bool have1, u32 c2, bool have2) {
  if (!have1) { return COMMA_FINE_IN_OBJECT | COMMA_FINE_IN_ARRAY; }
  const bool q1 = c1 == '"';
  const bool fine_o = q1 && (!have2 || c2 == ':');
  const bool fine_a = starts_value(c1, false) && c1 != ',' && !(q1 && have2 && !(c2 == ',' || is_close_char(c2)));
  return (fine_o ? COMMA_FINE_IN_OBJECT : 0u) | (fine_a ? COMMA_FINE_IN_ARRAY : 0u);
}
// ---- a token's byte, looked up -------------------------------------------------------------------------------------------------------------
// Everything k_tok_apply asks about a token's byte -- its contribution to the six counters, which list it goes to, what it is to the comma in front
// of it -- as bits of ONE table entry (a workgroup builds the 256 entries in LDS from the predicates above, so they cannot drift apart; tests/host/
// test_tape_rules.cpp compares every use on every byte).  Spelled out, the predicates cost a wave ~20 compares and as many scalar mask operations
// per token; the kernel issued 2 200 instructions per 1024 tokens and was bound by them.  Not covered: the ROOT token (list index 0), whose number
// path differs (takes_number_path) -- one token per document, handled by the caller with tok_contribution(c, true).
enum : u32 { TP_SLOTS = 3u, TP_SORT = 4u, TP_STRING = 8u, TP_OPEN = 16u, TP_CLOSE = 32u, TP_NUMBER = 64u, TP_ATOM = 128u, TP_COMMA = 256u, TP_COLON = 512u,
              TP_VALUE = 1024u /* starts_value(c, false) */, TP_COMMA_OR_CLOSE = 2048u, TP_KIND_SHIFT = 12u /* sort_kind of a bracket, 0 otherwise: 3 bits */ };
SJ_HD u32 token_props_of(u32 c) {
  const tok_packed p = tok_contribution(c, false);
  const bool bracket = is_open_char(c) || is_close_char(c);
  return (p.a & 3u) | ((p.a >> 16) ? TP_SORT : 0u) | ((p.b & 0xFFFFu) ? TP_STRING : 0u) | ((p.b >> 16) ? TP_OPEN : 0u) | ((p.c & 0xFFFFu) ? TP_CLOSE : 0u) |
         ((p.c >> 16) ? TP_NUMBER : 0u) | ((c == 't' || c == 'f' || c == 'n') ? TP_ATOM : 0u) | (c == ',' ? TP_COMMA : 0u) | (c == ':' ? TP_COLON : 0u) |
         (starts_value(c, false) ? TP_VALUE : 0u) | ((c == ',' || is_close_char(c)) ? TP_COMMA_OR_CLOSE : 0u) | ((bracket ? sort_kind(c, 0) : 0u) << TP_KIND_SHIFT);
}
SJ_HD tok_packed tok_contribution_of_props(u32 x) {
  tok_packed p;
  p.a = (x & TP_SLOTS) | ((x & TP_SORT) << 14);
  p.b = ((x >> 3) & 1u) | ((x & TP_OPEN) << 12);
  p.c = ((x >> 5) & 1u) | ((x & TP_NUMBER) << 10);
  return p;
}
SJ_HD u32 value_list_of_props(u32 x) {
  return (x & TP_NUMBER) ? u32(LIST_NUMBERS) : ((x & TP_STRING) ? u32(LIST_STRINGS) : (((x & TP_SLOTS) == 1u && !(x & TP_SORT)) ? u32(LIST_REST) : u32(LIST_NONE)));
}
SJ_HD u32 comma_fine_bits_of_props(u32 x1, bool have1, u32 x2, bool have2) {
  if (!have1) { return COMMA_FINE_IN_OBJECT | COMMA_FINE_IN_ARRAY; }
  const bool q1 = (x1 & TP_STRING) != 0u;
  const bool fine_o = q1 && (!have2 || (x2 & TP_COLON) != 0u);
  const bool fine_a = (x1 & TP_VALUE) != 0u && (x1 & TP_COMMA) == 0u && !(q1 && have2 && (x2 & TP_COMMA_OR_CLOSE) == 0u);
  return (fine_o ? COMMA_FINE_IN_OBJECT : 0u) | (fine_a ? COMMA_FINE_IN_ARRAY : 0u);
}
// the sort key of a bracket or comma from its entry (level: clamped by the caller)
SJ_HD u32 sort_key_of_props(u32 level, u32 x, u32 comma_bits) { return level | (((x & TP_COMMA) ? KIND_COMMA + comma_bits : (x >> TP_KIND_SHIFT) & 7u) << KIND_SHIFT); }

// ---- the token's own verdict from ONE table entry per byte (round 6) ------------------------------------------------------------------------------
// token_rule_self asks three tables (what a byte is, the state behind a byte, what a state accepts) in a kernel of its own (k_tape_rules, rounds 3-5: 44
// VALU instructions per token, the token bytes read once more).  k_tok_apply already holds every token's table entry in a register; with the state behind
// a byte and "the states that accept this byte" (the accept table turned around: bit s of the entry <=> byte_props_of(v) & state_accepts(s)) in the entry's
// upper half, the verdict is a dozen instructions on three registers and k_tape_rules is gone.
// entry of byte v: [15:0] token_props_of(v), [19:16] state_behind_byte(v), [20 + s] a byte v is accepted in state s (s < ST_COUNT <= 12)
constexpr u32 TE_STATE_SHIFT = 16, TE_ACCEPT_SHIFT = 20;
static_assert(ST_COUNT <= 12, "the accept bits of a token entry");
SJ_HD u32 token_entry_of(u32 v) {
  u32 e = token_props_of(v) | (state_behind_byte(v) << TE_STATE_SHIFT);
  const u32 pv = byte_props_of(v);
  for (u32 st = 0; st < ST_COUNT; st++) { if (pv & state_accepts(st)) { e |= 1u << (TE_ACCEPT_SHIFT + st); } }
  return e;
}
// token_rule_self from the entries of the token (xc), of the token in front (xprev) and of the one in front of that (xprev2); entries of bytes that are no
// token (in front of the list: the entry of byte 0) behave like the bytes themselves
SJ_HD u32 token_rule_self_entries(bool first, u32 xc, u32 xprev, u32 xprev2, u32 *rank) {
  // (written without a branch: every token of a wave takes the same dozen instructions)
  const u32 prev_comma = (xprev / TP_COMMA) & 1u, prev_quote = (xprev / TP_STRING) & 1u, prev2_comma = (xprev2 / TP_COMMA) & 1u;
  const u32 judged = first ? 0u : (prev_comma | (prev_quote & prev2_comma)); // judged_by_comma: not here
  const u32 kind2 = (xprev2 >> TP_KIND_SHIFT) & 7u, kind1 = (xprev >> TP_KIND_SHIFT) & 7u;
  u32 st = (xprev >> TE_STATE_SHIFT) & 15u;
  st = st == ST_BEHIND_QUOTE ? (kind2 == KIND_OPEN_OBJECT ? u32(ST_BEHIND_KEY) : u32(ST_BEHIND_VALUE)) : st;
  st = first ? u32(ST_ROOT) : st;
  const u32 refused = ((xc >> (TE_ACCEPT_SHIFT + st)) & 1u) ^ 1u;
  const u32 comma_for_value = first ? 0u : (((xc / TP_COMMA) & 1u) & ((kind1 == KIND_OPEN_ARRAY ? 1u : 0u) | ((xprev / TP_COLON) & 1u)));
  *rank = (judged | refused) ? 0u : (comma_for_value ? 2u : 0u);
  return judged ? 0u : (refused ? u32(SJ_TAPE_ERROR) : (comma_for_value ? u32(SJ_NUMBER_ERROR) : 0u));
}

// the list index of the token that writes the tape word at position p (a bracket: one word): tape positions do not decrease along the list and
// tokens without a word (':' ',') share theirs with the token behind them, so it is the LAST index whose position is p.  tpos: n + 1 entries.
template <class TPOS> SJ_HD u32 token_at_tape_position(const TPOS &tpos, u32 n, u32 p) {
  u32 lo = 0, hi = n; // the first index in [0, n] whose position exceeds p (tpos[n] = the total > p)
  while (lo < hi) {
    const u32 mid = lo + (hi - lo) / 2;
    if (u32(tpos[mid]) > p) { hi = mid; } else { lo = mid + 1; }
  }
  return lo ? lo - 1 : 0u;
}

// ---- the kinds of the open containers as a BIT STACK (round 4) --------------------------------------------------------------------------
// What a comma needs to know -- is the innermost open container an object or an array? -- the sort by nesting level answers for every
// comma (k_tape_match), at the price of a one-byte scatter per comma and a pass over every token behind it (k_tape_rules).  For documents
// nested less than 64 deep it is also a PREFIX SCAN: the walk's stack of container kinds is a word with one bit per depth (bit d: the
// container opened at depth d is an object); an opening bracket at depth d overwrites bit d, nothing else writes, and "the later one
// wins" composes associatively.  A closed container leaves its bit stale, which nobody reads before the next opening bracket at that depth
// rewrites it -- up to the document's first error, which is all that counts (the smallest offending index decides).
struct kind_stack {
  u64 mask, value; // bits written by the tokens summarised; their values
};
SJ_HD kind_stack kinds_then(const kind_stack &a, const kind_stack &b) { return kind_stack{a.mask | b.mask, (a.value & ~b.mask) | b.value}; } // first a, then b
SJ_HD kind_stack kinds_of_token(u32 c, int depth_in_front) {
  if ((c != '{' && c != '[') || depth_in_front < 0 || depth_in_front >= 64) { return kind_stack{0, 0}; }
  const u64 bit = u64(1) << depth_in_front;
  return kind_stack{bit, c == '{' ? bit : u64(0)};
}
// the kind of the container a ',' at this depth (in front of it) separates the members of; *deep is set when the stack word cannot say
SJ_HD u32 kinds_ctx(const kind_stack &in_front, int depth_in_front, bool *deep) {
  const int d = depth_in_front - 1;
  if (d < 0) { return CTX_NONE; }
  if (d >= 64) { *deep = true; return CTX_NONE; }
  if (!((in_front.mask >> d) & 1u)) { return CTX_NONE; }
  return ((in_front.value >> d) & 1u) ? u32(CTX_OBJECT) : u32(CTX_ARRAY);
}

// A ',' where the walk expects a VALUE (behind '[', ':' or an array's ',') is not a separator: visit_primitive hands it to
// parse_number like every byte below ':' and the document ends with NUMBER_ERROR (content rank), e.g. "[ ,1]" or {"a":,}.
SJ_HD bool comma_in_value_position(u64 i, u32 prev, u32 ctx_prev) {
  return i > 0 && (prev == '[' || prev == ':' || (prev == ',' && ctx_prev == CTX_ARRAY));
}

// tape words (/root/reference/doc/tape.md, tape_writer.h)
SJ_HD u64 tape_word(u32 type, u64 payload) { return (u64(type) << 56) | payload; }
// the same for a payload of 32 bits, put together from two dwords (a 64-bit shift costs a quarter-rate instruction on the device)
SJ_HD u64 tape_word32(u32 type, u32 payload) { return (u64(type << 24) << 32) | payload; }

} // namespace sjgpu
#endif
