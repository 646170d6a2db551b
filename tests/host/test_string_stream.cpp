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
struct patch_list {
  int value[64];
  void escape(int rel, u32 len, u32 packed) { for_each_escape_byte(rel, len, packed, [&](u32 p, u32 v) { value[p] = int(v); }); }
};

// returns false if a string is rejected (or the document ends inside one); out = the string buffer with the lengths filled in
static bool model(const uint8_t *buf, u32 len, bool allow, std::vector<uint8_t> &out, u32 &strings) {
  const doc_bytes src{buf, len};
  out.clear();
  strings = 0;
  u64 e_carry = 0;
  u32 s_carry = 0, u_prev = 0;
  bool bad = false;
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
    patch_list patches;
    for (int k = 0; k < 64; k++) { patches.value[k] = -1; }
    u64 U = 0;
    if (c.backslash || escaped || u_prev) { // (escaped alone misses a backslash in the last byte of the block)
      const escape_classes ec = classify_escapes(P);
      b = simple_escapes(c.backslash, escaped, quote, ec);
      U = escaped & ec.u;
      unicode_escapes(src, pos, U, u_prev, allow, b, patches);
    }
    u_prev = u32(U >> 54);
    if (b.bad & in_string & ~quote) { bad = true; }
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
        if (patches.value[i] >= 0) { v = u32(patches.value[i]); }
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
  return !bad;
}

int main(int argc, char **argv) {
  const int allow = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<uint8_t> doc, got;
  unsigned long docs = 0, valid = 0, rejected = 0, unlisted = 0;
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
    const bool ok = model(doc.data(), len, allow != 0, got, got_strings);
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
      if (opened != quote_tokens || e1 != 0) { unlisted++; continue; } // a rejected escape in a string that is not in the list, or an unclosed string
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
  printf("%lu documents, %lu byte for byte, %lu rejected, %lu with unlisted quotes\n", docs, valid, rejected, unlisted);
  return 0;
}
