// tests/host/test_string_stream.cpp -- the string buffer as a stream compaction (simdjson_amd/csrc/sj_string_stream.h), block by
// block on the CPU with the very functions the kernels of sjgpu_string_stream.hip call -- bit planes, stage 1's escape and quote
// algebra, the escape classes, the \u bookkeeping with its 10-byte look-back -- against the oracle's string-at-a-time walk
// (oracle/sj_oracle.c: sjo_string_buffer, itself pinned against the reference's dom::parser).  A document whose strings are all
// valid must come out byte for byte; one with a string the reference rejects must be flagged (the kernels then fall back to the
// per-string path).  What this cannot cover is the GPU plumbing (wave scans, the LDS window, the fallback switch): that is what
// tests/test_gpu_parity.py::test_string_* is for.
// Input on stdin: [u32 length][bytes] records; argv[1] = allow_replacement (0 / 1).  Built with g++ by tests/test_string_stream.py.
#include "sj_oracle.h"
#include "sj_string_stream.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace sjgpu;

struct doc_bytes {
  const uint8_t *buf;
  u32 len;
  u32 byte(u32 pos) const { return pos < len ? buf[pos] : 0x20u; }
};
// the hand-over between consecutive blocks: what left the block in front at the top (the kernels: the lane in front, or the wave's carry)
struct block_chain {
  u_tops prev{0, 0, 0, 0}, mine{0, 0, 0, 0};
  u32 operator()(u32 stage, u32 top) {
    u32 *m = stage == 0 ? &mine.a : (stage == 1 ? &mine.b : (stage == 2 ? &mine.c : &mine.d));
    const u32 *p = stage == 0 ? &prev.a : (stage == 1 ? &prev.b : (stage == 2 ? &prev.c : &prev.d));
    *m = top;
    return *p;
  }
};

// returns false if a string is rejected (or the document ends inside one); out = the string buffer with the lengths filled in
static bool model(const uint8_t *buf, u32 len, bool allow, std::vector<uint8_t> &out, u32 &strings, bool *flagged_anywhere) {
  (void)allow; // (a lone surrogate sends the document down the per-string road whatever the option says: that road knows the replacement character)
  const doc_bytes src{buf, len};
  out.clear();
  strings = 0;
  u64 e_carry = 0;
  u32 s_carry = 0;
  block_chain chain;
  bool bad = false, bad_anywhere = false;
  size_t open_at = 0;
  for (u32 pos = 0; pos < len; pos += 64) {
    u32 w[16];
    for (u32 j = 0; j < 16; j++) {
      w[j] = src.byte(pos + 4 * j) | (src.byte(pos + 4 * j + 1) << 8) | (src.byte(pos + 4 * j + 2) << 16) | (src.byte(pos + 4 * j + 3) << 24);
    }
    const planes P = transpose64(w);
    const classes c = classify(P);
    u64 e_next = 0;
    const u64 escaped = escaped_mask(c.backslash, e_carry, e_next);
    e_carry = e_next;
    const u64 quote = andn(c.quote, escaped);
    const u64 in_string = prefix_xor(quote) ^ (0 - u64(s_carry));
    s_carry ^= u32(popc64(quote)) & 1u;
    string_block b = no_escapes(quote);
    u_masks um{0, 0, 0, 0, 0};
    chain.prev = chain.mine;
    chain.mine = u_tops{0, 0, 0, 0};
    if (c.backslash || escaped || chain.prev.any()) { // (escaped alone misses a backslash in the last byte of the block)
      const escape_classes ec = classify_escapes(P);
      b = simple_escapes(c.backslash, escaped, quote, ec);
      um = unicode_masks(escaped & ec.u, classify_hex(P), chain);
      apply_unicode(b, um);
    }
    if (b.bad & in_string & ~quote) { bad = true; }
    if (b.bad_u) { bad_anywhere = true; }
    const size_t before = out.size();
    for (u32 i = 0; i < 64 && pos + i < len; i++) {
      const u64 bit = u64(1) << i;
      if (quote & bit) {
        if (in_string & bit) { // opening
          open_at = out.size();
          out.insert(out.end(), 4, 0);
        } else {
          out.push_back(0);
          const u32 l = u32(out.size() - open_at - 5);
          memcpy(out.data() + open_at, &l, 4);
          strings++;
        }
      } else if (b.keep & in_string & bit) {
        u32 v = src.byte(pos + i);
        if (um.k2 & bit) { v = u_escape_byte(src, pos + i, 2); }
        else if (um.k3 & bit) { v = u_escape_byte(src, pos + i, 3); }
        else if (um.k4 & bit) { v = u_escape_byte(src, pos + i, 4); }
        else if (b.remap & bit) { v = simple_escape_value(v); }
        out.push_back(uint8_t(v));
      }
    }
    // the count the kernels use must agree with what was produced (bytes beyond len are spaces: they produce nothing outside
    // strings, and inside a string the document is rejected anyway)
    if (pos + 64 <= len && out.size() - before != block_output_bytes(b, quote, in_string)) {
      fprintf(stderr, "block at %u: %zu bytes produced, block_output_bytes says %u\n", pos, out.size() - before, block_output_bytes(b, quote, in_string));
      return false;
    }
  }
  if (s_carry) { bad = true; }
  *flagged_anywhere = bad_anywhere;
  return !bad && !bad_anywhere;
}

// the digit classes from the bit planes are the byte predicates the kernels apply to their look-back bytes
static bool hex_classes_agree() {
  for (u32 c = 0; c < 256; c++) {
    u32 w[16];
    for (u32 j = 0; j < 16; j++) { w[j] = c * 0x01010101u; }
    const hex_classes h = classify_hex(transpose64(w));
    const u64 all = ~u64(0);
    if (h.hex != (byte_is_hex(c) ? all : 0) || h.zero != (c == '0' ? all : 0) || h.oct != (byte_is_octal(c) ? all : 0) || h.d != (byte_is_d(c) ? all : 0) ||
        h.s8b != (byte_is_89ab(c) ? all : 0) || h.scf != (byte_is_cdef(c) ? all : 0)) {
      fprintf(stderr, "digit classes of byte %02x differ\n", c);
      return false;
    }
    if (byte_is_hex(c)) {
      const u32 want = c <= '9' ? c - '0' : (c | 0x20u) - 'a' + 10;
      if (hex_digit_value(c) != want) { fprintf(stderr, "hex_digit_value(%02x)\n", c); return false; }
    }
  }
  return true;
}

int main(int argc, char **argv) {
  const int allow = argc > 1 ? atoi(argv[1]) : 0;
  if (!hex_classes_agree()) { return 1; }
  std::vector<uint8_t> doc, got;
  unsigned long docs = 0, valid = 0, rejected = 0, unlisted = 0, declined = 0;
  for (;;) {
    uint32_t len;
    if (fread(&len, 4, 1, stdin) != 1) { break; }
    doc.assign(len + 64, 0x20);
    if (len && fread(doc.data(), 1, len, stdin) != len) { return 2; }
    docs++;
    std::vector<uint32_t> idx(len + 8);
    uint32_t n = 0;
    const int e1 = sjo_stage1(doc.data(), len, SJO_REGULAR, len ? len : 1, idx.data(), &n);
    if (e1 == SJO_UTF8_ERROR || e1 == SJO_CAPACITY) { continue; } // (an unclosed string is a case: the model must flag it)
    std::vector<uint8_t> want(5 * (size_t(len) / 3) + 128);
    uint64_t bytes = 0;
    uint32_t strings = 0, first_bad = 0xFFFFFFFFu;
    (void)sjo_string_buffer(doc.data(), len, idx.data(), n, allow, want.data(), want.size(), nullptr, &bytes, &strings, &first_bad);
    uint32_t quote_tokens = 0;
    for (uint32_t i = 0; i < n; i++) { quote_tokens += doc[idx[i]] == '"'; }
    u32 got_strings = 0;
    bool flagged_anywhere = false;
    const bool ok = model(doc.data(), len, allow != 0, got, got_strings, &flagged_anywhere);
    // quotes that open a string without being in the list (glued to a scalar: a"b") -- the kernels compare the two counts and
    // fall back; here: count the opening quotes the model saw
    uint32_t opened = got_strings; // closed strings; an unclosed one makes ok false
    if (ok && opened != quote_tokens) { unlisted++; continue; }
    if (first_bad != 0xFFFFFFFFu) {
      if (ok) { fprintf(stderr, "a rejected string went unnoticed: %.*s\n", int(len > 300 ? 300 : len), (const char *)doc.data()); return 1; }
      rejected++;
      continue;
    }
    if (!ok) {
      if (opened != quote_tokens || e1 != 0) { unlisted++; continue; }
      if (flagged_anywhere) {
        // a \u pattern the reference rejects was seen: fine if the option accepts lone surrogates (the per-string road substitutes them) or
        // if a backslash stands outside every string (the flag does not ask where: such a document is invalid for stage 2 anyway)
        bool outside = false, in = false;
        for (uint32_t k = 0; k < len; k++) {
          if (doc[k] == '\\') { if (!in) { outside = true; } k++; }
          else if (doc[k] == '"') { in = !in; }
        }
        if (allow || outside) { declined++; continue; }
      } // a rejected escape in a string that is not in the list, or an unclosed string
      fprintf(stderr, "flagged, but the oracle accepts every string: %.*s\n", int(len > 300 ? 300 : len), (const char *)doc.data());
      return 1;
    }
    if (got.size() != bytes || memcmp(got.data(), want.data(), bytes) != 0) {
      fprintf(stderr, "buffer differs (%zu bytes, the oracle %llu): %.*s\n", got.size(), (unsigned long long)bytes, int(len > 300 ? 300 : len), (const char *)doc.data());
      for (size_t k = 0; k < got.size() && k < bytes; k++) {
        if (got[k] != want[k]) { fprintf(stderr, "  byte %zu: %02x, the oracle %02x\n", k, got[k], want[k]); break; }
      }
      return 1;
    }
    valid++;
  }
  printf("%lu documents, %lu byte for byte, %lu rejected, %lu with unlisted quotes, %lu declined\n", docs, valid, rejected, unlisted, declined);
  return 0;
}
