// tests/host/test_number.cpp -- CPU driver for simdjson_amd/csrc/sj_number.h (the code the tape kernels run per number token).
// Reads one token text per line from stdin and prints "error type bits" for it, computed twice: with the big-integer decision
// in line (the way a host caller would) and the way the device does it -- bracket first, big integers only for the tokens that
// report `slow` -- and fails if the two disagree.  Built with g++ and driven by tests/test_number.py, which compares the output
// with the reference's tape for "[<token>]" and with the oracle.
#include "sj_number.h"

#include <cstdio>
#include <iostream>
#include <string>

using namespace sjgpu;

struct text_bytes {
  const std::string *s;
  u32 byte(u32 pos) const { return pos < s->size() ? u32((unsigned char)(*s)[pos]) : 0x20u; }
};

int main() {
  for (u32 c = 0; c < 256; c++) { // the terminator test from masks against the spelled-out one
    if (not_structural_or_whitespace(c) != not_structural_or_whitespace_spelled(c)) { fprintf(stderr, "not_structural_or_whitespace(%02x): the two forms disagree\n", c); return 1; }
  }
  std::string line;
  static bigint big[2];
  unsigned long slow_tokens = 0;
  while (std::getline(std::cin, line)) {
    const text_bytes src{&line};
    const number_value a = parse_number_token(src, 0, big);
    number_shape shape;
    number_value b = parse_number_token(src, 0, nullptr, &shape);
    if (b.slow) {
      slow_tokens++;
      u64 bits = 0;
      if (!decide_long_decimal(src, shape, big, bits)) { b.error = SJ_NUMBER_ERROR; }
      else { b.bits |= bits; }
      b.slow = false;
    }
    if (a.error != b.error || (a.error == 0 && (a.type != b.type || a.bits != b.bits))) {
      std::fprintf(stderr, "in-line and deferred decisions disagree on %s\n", line.c_str());
      return 1;
    }
    std::printf("%u %c %016llx\n", a.error, a.error ? '-' : char(a.type), (unsigned long long)(a.error ? 0 : a.bits));
  }
  std::fprintf(stderr, "slow tokens: %lu\n", slow_tokens);
  return 0;
}
