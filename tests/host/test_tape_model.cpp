// tests/host/test_tape_model.cpp -- the data-parallel tape construction of simdjson_amd/csrc/sjgpu_tape.hip, run STEP BY STEP on
// the CPU with the very same per-token functions (sj_tape_rules.h, sj_number.h) and the same intermediate arrays -- the packed
// per-token counters and their prefix sums, the lists of value tokens with their tape positions, the brackets and commas (level and kind
// in one key) sorted stably by nesting level, "which container am I in" from a count of the opening brackets in front, the walk's rule
// in its split form (the depth's verdict, the token's own from the tables, the commas' verdicts on their followers), all reduced to the smallest error key -- and compared
// with the oracle's serial walk (oracle/sj_oracle_stage2.c, itself pinned against the reference): error code always, every tape
// word when the document is valid.  What this cannot cover is the GPU plumbing (scans, radix sort, atomics); that is what
// tests/test_gpu_parity.py::test_tape_* is for.
// Input on stdin: [u32 length][bytes] records.  Built with g++ by tests/test_tape_model.py.
#include "sj_oracle.h"
#include "sj_tape_rules.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace sjgpu;

struct doc_bytes {
  const uint8_t *buf;
  u32 len;
  u32 byte(u32 pos) const { return pos < len ? buf[pos] : 0x20u; }
};

// returns the error code; tape filled when it is SUCCESS.  The steps and the arrays are those of sjgpu_tape.hip (kernel names in the comments).
static u32 model(const uint8_t *buf, u32 len, const uint32_t *idx, u32 n, u32 max_depth, const uint32_t *str_offsets, u32 first_bad_string,
                 std::vector<u64> &tape) {
  if (n == 0) { return SJ_EMPTY; }
  const doc_bytes src{buf, len};
  // k_tok_classify: the byte of every token (two zero bytes in front and behind)
  std::vector<u32> tokc(n + 4, 0);
  auto C = [&](long long i) -> u32 { return tokc[size_t(i + 2)]; };
  for (u32 i = 0; i < n; i++) { tokc[i + 2] = src.byte(idx[i]); }
  // k_tok_apply: prefix sums of the six counters; tape position and depth per token; the lists of value tokens; the sort's input
  std::vector<long long> tpos(n + 1), depth(n + 1);
  std::vector<u64> strings_list, numbers_list, rest_list; // the device keeps strings and rest in one array (front / back)
  const u32 kmax = max_depth < 4095u ? max_depth : 4095u;
  std::vector<std::pair<u32, u32>> sort_in; // (key = level | kind << 12, token)
  long long words = 0, strs = 0, opens = 0, closes = 0, numbers = 0;
  u32 block_a = 0, block_b = 0, block_c = 0, top = 0;
  for (u32 i = 0; i <= n; i++) {
    if ((i & 4095u) == 0) { block_a = 0; block_b = 0; block_c = 0; } // the packed fields must hold a block's sums
    tpos[i] = words;
    depth[i] = opens - closes;
    if (i == n) { break; }
    const tok_packed p = tok_contribution(C(i), i == 0);
    const long long rest_before = one_word_rest(int(words), int(strs), int(numbers), int(opens), int(closes));
    if (rest_before != (long long)rest_list.size()) { fprintf(stderr, "rest list: %lld in front by the sums, %zu listed\n", rest_before, rest_list.size()); return 99; }
    const u64 entry = list_entry(u32(words), i);
    switch (value_list_of(p)) {
    case LIST_NUMBERS: numbers_list.push_back(entry); break;
    case LIST_STRINGS: strings_list.push_back(entry); break;
    case LIST_REST: rest_list.push_back(entry); break;
    default: break;
    }
    if (p.a >> 16) { // a bracket or a comma
      long long k = (p.b >> 16) ? depth[i] : depth[i] - 1;
      if (k < 0) { k = 0; }
      if (k > kmax) { k = kmax; }
      // a bracket travels with its tape position, a comma with its list index and with what it knows about its followers
      const bool comma = C(i) == ',';
      const u32 fine = comma ? comma_fine_bits(i, C((long long)i + 1), i + 1 < n, C((long long)i + 2), i + 2 < n) : 0u;
      sort_in.push_back({sort_key(u32(k), C(i), fine), comma ? i : u32(words)});
      if (u32(k) > top) { top = u32(k); }
    }
    words += p.a & 0xFFFFu; strs += p.b & 0xFFFFu; opens += p.b >> 16; closes += p.c & 0xFFFFu; numbers += p.c >> 16;
    block_a += p.a; block_b += p.b; block_c += p.c;
    if ((block_a & 0xFFFFu) > 0xFFFFu - 2 || (block_a >> 16) > 0xFFFEu || (block_b & 0xFFFFu) > 0xFFFEu) { return 98; } // (cannot happen: 4096 tokens x 2)
  }
  const u64 tape_words = u64(tpos[n]) + 2;
  tape.assign(tape_words, 0);
  // the sort: stable by level (the low 12 bits of the key); one pass of the device's radix sort when top < 64, two otherwise -- same result
  std::vector<std::pair<u32, u32>> sorted = sort_in;
  std::stable_sort(sorted.begin(), sorted.end(), [](const std::pair<u32, u32> &x, const std::pair<u32, u32> &y) { return (x.first & 0xFFFu) < (y.first & 0xFFFu); });
  (void)top;
  const u32 m = u32(sorted.size());
  // k_tape_opens + scan + k_tape_openpos
  std::vector<u32> cid(m), openpos(m);
  u32 opens_seen = 0;
  for (u32 j = 0; j < m; j++) {
    if (kind_is_open(sorted[j].first >> KIND_SHIFT)) { openpos[opens_seen] = j; opens_seen++; }
    cid[j] = opens_seen; // opens at sorted positions <= j
  }
  u64 errkey = NO_ERROR_KEY;
  auto report = [&](u64 k) { if (k < errkey) { errkey = k; } };
  // k_tape_match
  for (u32 j = 0; j < m; j++) {
    const u32 kj = sorted[j].first, kind = kj >> KIND_SHIFT, mine = sorted[j].second;
    if (kind_is_open(kind) || cid[j] == 0) { continue; }
    const u32 jo = openpos[cid[j] - 1], ko = sorted[jo].first;
    if (((ko ^ kj) & 0xFFFu) != 0) { continue; } // no container of my level in front of me: an error elsewhere says so
    const bool object = (ko >> KIND_SHIFT) == KIND_OPEN_OBJECT;
    if (kind_is_comma(kind)) { // the comma judges the two tokens behind it: nobody else knows what its container wants there
      if ((kind - KIND_COMMA) & (object ? COMMA_FINE_IN_OBJECT : COMMA_FINE_IN_ARRAY)) { continue; } // it said so when it was sent into the sort
      const u32 i = mine;
      const follower_keys f = comma_followers_rule(i, object ? CTX_OBJECT : CTX_ARRAY, C((long long)i + 1), i + 1 < n, C((long long)i + 2), i + 2 < n);
      if (f.k[0] == NO_ERROR_KEY && f.k[1] == NO_ERROR_KEY && f.k[2] == NO_ERROR_KEY) { fprintf(stderr, "a comma that is not fine raises nothing\n"); return 97; }
      for (u64 k : f.k) { report(k); }
      continue;
    }
    const u32 open_tp = sorted[jo].second, close_tp = mine; // tape positions travel with the brackets
    if ((kind == KIND_CLOSE_OBJECT) != object) { report(error_key(token_at_tape_position(tpos, n, close_tp), 0, SJ_TAPE_ERROR)); }
    const u64 open_at = 1 + u64(open_tp), close_at = 1 + u64(close_tp);
    const u64 count = (close_tp == open_tp + 1) ? 0 : (j - jo > 0xFFFFFFu ? 0xFFFFFFu : j - jo);
    tape[close_at] = tape_word(kind == KIND_CLOSE_OBJECT ? u32('}') : u32(']'), open_at);
    tape[open_at] = tape_word(object ? u32('{') : u32('['), (count << 32) | (close_at + 1));
  }
  // k_tape_rules: the rule from the tables a workgroup builds
  unsigned short props[256], accepts[ST_COUNT];
  u8 state_behind[256];
  for (u32 t = 0; t < 256; t++) { rule_table_entry(t, props, state_behind, accepts); }
  const rule_tables T{props, state_behind, accepts};
  {
    const u32 c0 = C(0), last = C((long long)n - 1);
    if ((c0 == '{' && last != '}') || (c0 == '[' && last != ']')) { report(error_key(0, 0, SJ_TAPE_ERROR)); } // json_iterator.h:138-143: in front of everything else
  }
  for (u32 i = 0; i < n; i++) {
    const int d = depth[i] > 0x7FFFFFFFll ? 0x7FFFFFFF : (depth[i] < -0x7FFFFFFFll ? -0x7FFFFFFF : int(depth[i]));
    u32 rank = 0;
    u32 g = depth_rule(i == 0, C(i), C((long long)i + 1), d, max_depth, &rank); // k_tok_apply: where the depth is computed
    if (g) { report(error_key(i, rank, g)); }
    g = token_rule_self(T, i == 0, C(i), C((long long)i - 1), C((long long)i - 2), &rank); // k_tape_rules: no ctx, no depth
    if (g) { report(error_key(i, rank, g)); }
  }
  if (depth[n] != 0) { report(error_key(n, 0, SJ_TAPE_ERROR)); } // the walk runs into the sentinel inside a container
  // k_tape_strings: entry k is the k-th string of the string buffer (the oracle's offsets of the string tokens, in order)
  for (size_t k = 0; k < strings_list.size(); k++) {
    const u32 i = u32(strings_list[k]);
    tape[1 + (strings_list[k] >> 32)] = tape_word32('"', str_offsets[i]);
  }
  // k_tape_atoms
  for (u64 entry : rest_list) {
    const u32 i = u32(entry), c = C(i);
    if (c != 't' && c != 'f' && c != 'n') { continue; } // no token at all: the rule has said so
    const bool ok = c == 't' ? atom_matches(src, idx[i], 't', 'r', 'u', 'e', 0)
                             : (c == 'f' ? atom_matches(src, idx[i], 'f', 'a', 'l', 's', 'e') : atom_matches(src, idx[i], 'n', 'u', 'l', 'l', 0));
    if (!ok) { report(error_key(i, 2, c == 't' ? SJ_T_ATOM_ERROR : (c == 'f' ? SJ_F_ATOM_ERROR : SJ_N_ATOM_ERROR))); }
    tape[1 + (entry >> 32)] = tape_word32(c, 0);
  }
  // k_tape_numbers (+ k_tape_slow_numbers: here in one go)
  static bigint big[2];
  for (u64 entry : numbers_list) {
    const u32 i = u32(entry);
    const number_value v = parse_number_token(src, idx[i], big);
    if (v.error) { report(error_key(i, 2, v.error)); continue; }
    tape[1 + (entry >> 32)] = tape_word(v.type, 0);
    tape[2 + (entry >> 32)] = v.bits;
  }
  if (first_bad_string != 0xFFFFFFFFu) { report(error_key(first_bad_string, 2, SJ_STRING_ERROR)); }
  tape[0] = tape_word('r', tape_words);
  tape[tape_words - 1] = tape_word('r', 0);
  return error_code_of(errkey);
}

int main(int argc, char **argv) {
  const u32 max_depth = argc > 1 ? u32(atoi(argv[1])) : 1024u;
  std::vector<uint8_t> doc;
  unsigned long docs = 0, valid = 0, codes[16] = {0};
  for (;;) {
    uint32_t len;
    if (fread(&len, 4, 1, stdin) != 1) { break; }
    doc.assign(len + 64, 0x20);
    if (len && fread(doc.data(), 1, len, stdin) != len) { return 2; }
    docs++;
    std::vector<uint32_t> idx(len + 8);
    uint32_t n = 0;
    const int e1 = sjo_stage1(doc.data(), len, SJO_REGULAR, len ? len : 1, idx.data(), &n);
    if (e1) { continue; }
    std::vector<uint64_t> want(len + 8);
    std::vector<uint8_t> sbuf(5 * (size_t(len) / 3) + 128);
    uint64_t tw = 0, sb = 0;
    const int e_want = sjo_stage2(doc.data(), len, idx.data(), n, max_depth, want.data(), want.size(), sbuf.data(), sbuf.size(), &tw, &sb);
    std::vector<uint32_t> off(n + 1);
    uint64_t bytes = 0;
    uint32_t strings = 0, first_bad = 0xFFFFFFFFu;
    std::vector<uint8_t> sbuf2(sbuf.size());
    (void)sjo_string_buffer(doc.data(), len, idx.data(), n, 0, sbuf2.data(), sbuf2.size(), off.data(), &bytes, &strings, &first_bad);
    std::vector<u64> got;
    const u32 e_got = model(doc.data(), len, idx.data(), n, max_depth, off.data(), first_bad, got);
    codes[e_got & 15]++;
    if (int(e_got) != e_want) {
      fprintf(stderr, "error code %u, the oracle says %d: %.*s\n", e_got, e_want, int(len > 300 ? 300 : len), (const char *)doc.data());
      return 1;
    }
    if (e_want == 0) {
      valid++;
      if (got.size() != tw || memcmp(got.data(), want.data(), tw * 8) != 0) {
        fprintf(stderr, "tape differs (%zu words, the oracle %llu): %.*s\n", got.size(), (unsigned long long)tw, int(len > 300 ? 300 : len), (const char *)doc.data());
        for (size_t k = 0; k < got.size() && k < tw; k++) {
          if (got[k] != want[k]) { fprintf(stderr, "  word %zu: %016llx, the oracle %016llx\n", k, (unsigned long long)got[k], (unsigned long long)want[k]); break; }
        }
        return 1;
      }
    }
  }
  printf("%lu documents, %lu valid;", docs, valid);
  for (int k = 0; k < 16; k++) { if (codes[k]) { printf(" code %d: %lu", k, codes[k]); } }
  printf("\n");
  return 0;
}
