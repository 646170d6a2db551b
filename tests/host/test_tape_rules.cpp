// tests/host/test_tape_rules.cpp -- the two forms of "what may stand here" in simdjson_amd/csrc/sj_tape_rules.h must agree: token_rule
// (the walk's rule spelled out, which the oracle-pinned model of tests/host/test_tape_model.cpp runs) and token_rule_tables (the form the
// kernel k_tape_rules applies: byte properties, a state per byte in front, an accept mask per state).  Compared on every byte value for the
// token itself, on every byte that any rule distinguishes (and a few it does not) for the two tokens in front and the one behind, on every
// container kind, and on depths around 0 and around the nesting limit.
#include "sj_tape_rules.h"

#include <cstdio>
#include <vector>

using namespace sjgpu;

int main() {
  unsigned short props[256], accepts[ST_COUNT];
  u8 state_behind[256];
  for (u32 t = 0; t < 256; t++) { rule_table_entry(t, props, state_behind, accepts); }
  const rule_tables T{props, state_behind, accepts};
  const std::vector<u32> around = {0, '{', '[', '}', ']', ':', ',', '"', 't', 'f', 'n', '1', '-', '!', ' ', 'x', 0x7F, 0xC3, '9', '0'};
  const int depths[] = {-3, -1, 0, 1, 2, 15, 16, 17, 1023, 1024, 1025};
  const u32 limits[] = {1, 2, 16, 17, 1024};
  unsigned long checked = 0;
  for (u32 c = 0; c < 256; c++) {
    for (u32 prev : around) {
      for (u32 prev2 : around) {
        for (u32 next : around) {
          for (u32 ctx_prev = 0; ctx_prev < 3; ctx_prev++) {
            for (u32 ctx_prev2 = 0; ctx_prev2 < 3; ctx_prev2++) {
              for (int depth : depths) {
                for (u32 max_depth : limits) {
                  for (int first = 0; first < 2; first++) {
                    u32 r1 = 9, r2 = 9;
                    const u32 e1 = token_rule(first != 0, c, prev, prev2, next, ctx_prev, ctx_prev2, depth, max_depth, &r1);
                    const u32 e2 = token_rule_tables(T, first != 0, c, prev, prev2, next, ctx_prev, ctx_prev2, depth, max_depth, &r2);
                    checked++;
                    if (e1 != e2 || r1 != r2) {
                      fprintf(stderr, "c %02x prev %02x prev2 %02x next %02x ctx %u %u depth %d limit %u first %d: rule says %u (rank %u), tables say %u (rank %u)\n", c, prev, prev2,
                              next, ctx_prev, ctx_prev2, depth, max_depth, first, e1, r1, e2, r2);
                      return 1;
                    }
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  printf("%lu combinations agree\n", checked);
  return 0;
}
