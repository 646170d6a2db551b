// tests/host/test_block_math.cpp -- CPU check of the per-lane math in simdjson_amd/csrc/sj_block.h
// (the kernels' building blocks) against byte-at-a-time definitions and the C oracle.
// Built and run by tests/test_block_math.py with g++; exits non-zero on the first mismatch.
#include "sj_block.h"
#include "sj_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace sjgpu;

static uint64_t rng_state = 0x5eed1234abcdefull;
static uint64_t rnd() {
  uint64_t x = rng_state;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  rng_state = x;
  return x * 0x2545F4914F6CDD1Dull;
}
static const uint8_t ALPHA[] = { '"', '\\', '\\', '{', '}', '[', ']', ':', ',', ' ', '\n', '\t', '\r', 'a', '1', 't',
                                 0x01, 0x0C, 0x1A, 0x1E, 0x1F, 0x7F, 0x80, 0xBF, 0xC0, 0xC2, 0xE0, 0xED, 0xA0, 0x9F,
                                 0xF0, 0xF4, 0x90, 0x8F, 0xF5, 0xFF, 0xE2, 0x82, 0xAC, 0x00, 0x20, 0x22, 0x5C };
static void fill(uint8_t *p, size_t n, int mode) {
  for (size_t i = 0; i < n; i++) {
    if (mode == 0) { p[i] = uint8_t(rnd() >> 56); }
    else if (mode == 1) { p[i] = ALPHA[(rnd() >> 40) % sizeof ALPHA]; }
    else { p[i] = (rnd() >> 60) < 10 ? '\\' : ALPHA[(rnd() >> 40) % sizeof ALPHA]; }
  }
}
#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

static int is_ws(uint8_t b) { return b == 0x20 || b == 0x09 || b == 0x0A || b == 0x0D; }
static int is_op(uint8_t b) { if (b >= 0x80) return 0; uint8_t c = b | 0x20; return c == 0x2C || c == 0x3A || c == 0x7B || c == 0x7D; }

int main() {
  // 1. transposition + classes on single blocks
  for (int it = 0; it < 200000; it++) {
    uint8_t blk[64];
    fill(blk, 64, it % 3);
    u32 w[16];
    std::memcpy(w, blk, 64);
    planes P = transpose64(w);
    for (int k = 0; k < 8; k++) {
      u64 want = 0;
      for (int i = 0; i < 64; i++) { want |= u64((blk[i] >> k) & 1) << i; }
      CHECK(P.b[k] == want, "plane %d it %d", k, it);
    }
    classes c = classify(P);
    u64 bs = 0, q = 0, ws = 0, op = 0, ct = 0;
    for (int i = 0; i < 64; i++) {
      bs |= u64(blk[i] == '\\') << i; q |= u64(blk[i] == '"') << i; ws |= u64(is_ws(blk[i])) << i;
      op |= u64(is_op(blk[i])) << i; ct |= u64(blk[i] <= 0x1F) << i;
    }
    CHECK(c.backslash == bs && c.quote == q && c.ws == ws && c.op == op && c.ctrl == ct, "classes it %d", it);
  }
  // 1b. every byte value at every position of a block (the classes are functions of one byte: exhaustive)
  for (int v = 0; v < 256; v++) {
    for (int at = 0; at < 64; at++) {
      uint8_t blk[64];
      for (int i = 0; i < 64; i++) { blk[i] = uint8_t((v * 7 + i * 13) & 0xFF); }
      blk[at] = uint8_t(v);
      u32 w[16];
      std::memcpy(w, blk, 64);
      const classes c = classify(transpose64(w));
      for (int i = 0; i < 64; i++) {
        const uint8_t b = blk[i];
        CHECK(((c.backslash >> i) & 1) == u64(b == '\\') && ((c.quote >> i) & 1) == u64(b == '"') && ((c.ws >> i) & 1) == u64(is_ws(b)) &&
              ((c.op >> i) & 1) == u64(is_op(b)) && ((c.ctrl >> i) & 1) == u64(b <= 0x1F), "class of byte 0x%02X at %d", b, i);
      }
    }
  }
  // 2. escape / in-string / structural chain over multi-block buffers vs the oracle scan, and UTF-8
  for (int it = 0; it < 60000; it++) {
    size_t nblk = 1 + (rnd() >> 60) % 5;
    size_t len = nblk * 64 - ((rnd() >> 50) % 64);
    std::vector<uint8_t> buf(nblk * 64, 0x20);
    fill(buf.data(), len, 1 + it % 2);
    if (it % 7 == 0) { // long backslash runs crossing blocks
      size_t a = (rnd() >> 40) % len, b = a + (rnd() >> 40) % 140;
      for (size_t i = a; i < b && i < len; i++) { buf[i] = '\\'; }
    }
    if (it % 5 == 0) { for (size_t i = 0; i < len; i++) { if (buf[i] >= 0x80) buf[i] = 'x'; } } // ASCII-only variants
    std::vector<uint32_t> want_idx(len + 3), got_idx;
    uint32_t want_flags = 0;
    uint32_t want_n = sjo_scan(buf.data(), len, want_idx.data(), &want_flags);
    u64 e = 0; u32 s = 0, p = 0, carry = 0; u64 ctrl_err = 0, utf_err = 0;
    for (size_t bk = 0; bk < nblk; bk++) {
      u32 w[16];
      std::memcpy(w, buf.data() + 64 * bk, 64);
      planes P = transpose64(w);
      classes c = classify(P);
      u64 e_out;
      u64 escaped = escaped_mask(c.backslash, e, e_out);
      quote_scalar qs = quotes_and_scalars(c, escaped);
      u32 msb = u32(qs.nonquote_scalar >> 63);
      block_masks m = finish_block(c, qs, s, p);
      u64 structural = m.cand & ~m.string_tail;
      for (int i = 0; i < 64; i++) { if ((structural >> i) & 1) { got_idx.push_back(uint32_t(64 * bk + i)); } }
      ctrl_err |= c.ctrl & m.in_string;
      utf8_leads L = utf8_classify(P);
      utf_err |= utf8_errors(P, L, carry);
      carry = utf8_carry_out(L);
      e = e_out; s = u32(m.in_string >> 63); p = msb;
    }
    if (len % 64 == 0 && (carry & UTF8_CARRY_OPEN)) { utf_err |= 1; }
    uint32_t got_flags = (s ? 1u : 0u) | (ctrl_err ? 2u : 0u) | (utf_err ? 4u : 0u);
    CHECK(got_idx.size() == want_n, "n mismatch it %d: %zu vs %u (len %zu)", it, got_idx.size(), want_n, len);
    CHECK(std::memcmp(got_idx.data(), want_idx.data(), 4 * size_t(want_n)) == 0, "idx mismatch it %d", it);
    CHECK(got_flags == want_flags, "flags mismatch it %d: %u vs %u (len %zu)", it, got_flags, want_flags, len);
  }
  // 3. utf8_carry_from_bytes == carry produced by a block ending in those bytes
  for (int it = 0; it < 100000; it++) {
    uint8_t blk[64];
    fill(blk, 64, 1);
    u32 w[16];
    std::memcpy(w, blk, 64);
    planes P = transpose64(w);
    u32 co = utf8_carry_out(utf8_classify(P));
    CHECK(co == utf8_carry_from_bytes(blk[61], blk[62], blk[63]), "carry bytes it %d", it);
  }
  std::printf("block math OK\n");
  return 0;
}
