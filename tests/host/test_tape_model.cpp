// tests/host/test_tape_model.cpp -- the data-parallel tape construction of simdjson_amd/csrc/sjgpu_tape.hip, run STEP BY STEP on
// the CPU with the very same per-token functions (sj_tape_rules.h, sj_number.h) and the same intermediate arrays -- per-token
// slot counts and bracket deltas, their prefix sums, the brackets and commas sorted (stably) by nesting level, "which container
// am I in" from a count of the opening brackets in front, per-token verdicts reduced to the smallest error key -- and compared
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

// returns the error code; tape filled when it is SUCCESS
static u32 model(const uint8_t *buf, u32 len, const uint32_t *idx, u32 n, u32 max_depth, const uint32_t *str_offsets, u32 first_bad_string,
                 std::vector<u64> &tape) {
  if (n == 0) { return SJ_EMPTY; }
  const doc_bytes src{buf, len};
  // pass A: per token
  std::vector<u32> tokc(n + 4, 0); // tokc[i + 2] = byte of token i; zeros in front and behind
  auto C = [&](long long i) -> u32 { return tokc[size_t(i + 2)]; };
  std::vector<long long> slots(n + 1, 0), delta(n + 1, 0), sel(n + 1, 0);
  for (u32 i = 0; i < n; i++) {
    const u32 c = src.byte(idx[i]);
    tokc[i + 2] = c;
    slots[i] = tape_slots(c, i == 0);
    delta[i] = is_open_char(c) ? 1 : (is_close_char(c) ? -1 : 0);
    sel[i] = (is_open_char(c) || is_close_char(c) || c == ',') ? 1 : 0;
  }
  // pass B: exclusive prefix sums
  std::vector<long long> tpos(n + 1), depth(n + 1), selpos(n + 1);
  long long a = 0, b = 0, s = 0;
  for (u32 i = 0; i <= n; i++) { tpos[i] = a; depth[i] = b; selpos[i] = s; a += slots[i]; b += delta[i]; s += sel[i]; }
  const u64 tape_words = u64(tpos[n]) + 2;
  tape.assign(tape_words, 0);
  // pass C: brackets and commas with their level
  const u32 m = u32(selpos[n]);
  const u32 kmax = max_depth < 4095u ? max_depth : 4095u;
  std::vector<std::pair<u32, u32>> sorted(m); // (level, token)
  for (u32 i = 0; i < n; i++) {
    if (!sel[i]) { continue; }
    long long k = is_open_char(C(i)) ? depth[i] : depth[i] - 1;
    if (k < 0) { k = 0; }
    if (k > kmax) { k = kmax; }
    sorted[size_t(selpos[i])] = {u32(k), i};
  }
  // pass D: stable sort by level
  std::stable_sort(sorted.begin(), sorted.end(), [](const std::pair<u32, u32> &x, const std::pair<u32, u32> &y) { return x.first < y.first; });
  // pass E: container ordinal = opening brackets at sorted positions <= j; where each container opens
  std::vector<u32> cid(m), openpos(m);
  u32 opens = 0;
  for (u32 j = 0; j < m; j++) {
    if (is_open_char(C(sorted[j].second))) { openpos[opens] = j; opens++; }
    cid[j] = opens;
  }
  u64 errkey = NO_ERROR_KEY;
  auto report = [&](u64 k) { if (k < errkey) { errkey = k; } };
  // pass F: per sorted element
  std::vector<uint8_t> ctx(n + 1, CTX_NONE);
  for (u32 j = 0; j < m; j++) {
    const u32 i = sorted[j].second, c = C(i);
    if (is_open_char(c) || cid[j] == 0) { continue; }
    const u32 jo = openpos[cid[j] - 1];
    if (sorted[jo].first != sorted[j].first) { continue; } // no container of my level in front of me: an error elsewhere says so
    const u32 io = sorted[jo].second, co = C(io);
    if (c == ',') {
      ctx[i] = co == '{' ? CTX_OBJECT : CTX_ARRAY;
    } else { // a closing bracket and its partner
      if ((c == '}') != (co == '{')) { report(error_key(i, 0, SJ_TAPE_ERROR)); }
      const u64 open_at = 1 + u64(tpos[io]), close_at = 1 + u64(tpos[i]);
      const u64 count = (i == io + 1) ? 0 : (j - jo > 0xFFFFFFu ? 0xFFFFFFu : j - jo);
      tape[close_at] = tape_word(c, open_at);
      tape[open_at] = tape_word(co, (count << 32) | (close_at + 1));
    }
  }
  // pass G: per token
  static bigint big[2];
  {
    const u32 c0 = C(0), last = C((long long)n - 1);
    if ((c0 == '{' && last != '}') || (c0 == '[' && last != ']')) { report(error_key(0, 0, SJ_TAPE_ERROR)); } // json_iterator.h:138-143: in front of everything else
  }
  for (u32 i = 0; i < n; i++) {
    const u32 c = C(i);
    u32 rank = 0;
    const u32 g = token_grammar_error(i, c, C((long long)i - 1), C((long long)i - 2), C((long long)i + 1), i >= 1 ? ctx[i - 1] : 0u, i >= 2 ? ctx[i - 2] : 0u,
                                      depth[i], max_depth, &rank);
    if (g) { report(error_key(i, rank, g)); }
    const u64 at = 1 + u64(tpos[i]);
    if (c == '"') {
      tape[at] = tape_word('"', str_offsets[i]);
    } else if (c == ',') {
      if (comma_in_value_position(i, C((long long)i - 1), i >= 1 ? ctx[i - 1] : 0u)) { report(error_key(i, 2, SJ_NUMBER_ERROR)); }
    } else if (is_open_char(c) || is_close_char(c) || c == ':') {
    } else if (takes_number_path(c, i == 0)) {
      const number_value v = parse_number_token(src, idx[i], big);
      if (v.error) { report(error_key(i, 2, v.error)); }
      else { tape[at] = tape_word(v.type, 0); tape[at + 1] = v.bits; }
    } else if (c == 't') {
      if (!atom_matches(src, idx[i], 't', 'r', 'u', 'e', 0)) { report(error_key(i, 2, SJ_T_ATOM_ERROR)); }
      tape[at] = tape_word('t', 0);
    } else if (c == 'f') {
      if (!atom_matches(src, idx[i], 'f', 'a', 'l', 's', 'e')) { report(error_key(i, 2, SJ_F_ATOM_ERROR)); }
      tape[at] = tape_word('f', 0);
    } else if (c == 'n') {
      if (!atom_matches(src, idx[i], 'n', 'u', 'l', 'l', 0)) { report(error_key(i, 2, SJ_N_ATOM_ERROR)); }
      tape[at] = tape_word('n', 0);
    }
  }
  if (depth[n] != 0) { report(error_key(n, 0, SJ_TAPE_ERROR)); } // the walk runs into the sentinel inside a container
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
