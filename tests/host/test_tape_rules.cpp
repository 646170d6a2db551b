// tests/host/test_tape_rules.cpp -- the two forms of "what may stand here" in simdjson_amd/csrc/sj_tape_rules.h must agree: token_rule
// (the walk's rule spelled out, which the oracle-pinned model of tests/host/test_tape_model.cpp runs) and token_rule_tables (the form the
// kernel k_tape_rules applies: byte properties, a state per byte in front, an accept mask per state).  Compared on every byte value for the
// token itself, on every byte that any rule distinguishes (and a few it does not) for the two tokens in front and the one behind, on every
// container kind, and on depths around 0 and around the nesting limit.
// Round 4: the form the kernels run has neither a ctx nor a depth array -- the depth's verdict (depth_rule), the token's own (token_rule_self) and
// the verdict of the comma one or two tokens in front (comma_followers_rule) -- and the MINIMUM of their keys must be the key token_rule gives,
// for every combination in which the comma has a container (a comma without one judges nobody: see sj_tape_rules.h for why that is enough).
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
                    {
                      const u64 i = first ? 0 : 7; // (any index: the keys only compare at equal index)
                      u64 want = e1 ? error_key(i, r1, e1) : NO_ERROR_KEY;
                      if (c == ',' && comma_in_value_position(i, prev, ctx_prev)) { const u64 k = error_key(i, 2, SJ_NUMBER_ERROR); want = k < want ? k : want; }
                      const bool by_comma = !first && judged_by_comma(prev, prev2);
                      const u32 kind = prev == ',' ? ctx_prev : ctx_prev2; // of the comma that judges this token
                      if (!(by_comma && depth > 0 && kind == CTX_NONE)) {
                        u64 got = NO_ERROR_KEY;
                        u32 r = 9;
                        u32 e = depth_rule(first != 0, c, next, depth, max_depth, &r);
                        if (e) { const u64 k = error_key(i, r, e); got = k < got ? k : got; }
                        e = token_rule_self(T, first != 0, c, prev, prev2, &r);
                        if (e) { const u64 k = error_key(i, r, e); got = k < got ? k : got; }
                        if (by_comma) {
                          // the comma sits at i - 1 (this token is its first follower) or at i - 2 (this token follows the string behind it)
                          const follower_keys f = prev == ',' ? comma_followers_rule(i - 1, kind, c, true, next, true) : comma_followers_rule(i - 2, kind, '"', true, c, true);
                          const u64 mine[2] = {prev == ',' ? f.k[0] : f.k[2], prev == ',' ? f.k[1] : NO_ERROR_KEY};
                          for (u64 k : mine) { got = k < got ? k : got; }
                        }
                        if (got != want) {
                          fprintf(stderr, "c %02x prev %02x prev2 %02x next %02x ctx %u %u depth %d limit %u first %d: rule key %llx, the split form %llx\n", c, prev, prev2, next,
                                  ctx_prev, ctx_prev2, depth, max_depth, first, (unsigned long long)want, (unsigned long long)got);
                          return 1;
                        }
                      }
                    }
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
  for (u32 c1 = 0; c1 < 256; c1++) {
    for (u32 c2 = 0; c2 < 256; c2++) {
      for (int have = 0; have < 4; have++) {
        const bool have1 = (have & 1) != 0, have2 = have1 && (have & 2) != 0; // (a second follower without a first does not occur)
        if (comma_fine_bits(5, c1, have1, c2, have2) != comma_fine_bits_direct(c1, have1, c2, have2)) {
          fprintf(stderr, "comma_fine_bits: followers %02x %02x (%d %d): derived %u, direct %u\n", c1, c2, have1, have2, comma_fine_bits(5, c1, have1, c2, have2),
                  comma_fine_bits_direct(c1, have1, c2, have2));
          return 1;
        }
        checked++;
      }
    }
  }
  for (u32 c = 0; c < 256; c++) { // the table form of what k_tok_apply asks about a byte
    const u32 x = token_props_of(c);
    const tok_packed p = tok_contribution(c, false), q = tok_contribution_of_props(x);
    const bool same = p.a == q.a && p.b == q.b && p.c == q.c && value_list_of(p) == value_list_of_props(x) && ((x & TP_OPEN) != 0u) == is_open_char(c) &&
                      ((x & TP_ATOM) != 0u) == (c == 't' || c == 'f' || c == 'n') && ((x & TP_COMMA) != 0u) == (c == ',');
    bool keys = true;
    for (u32 level : {0u, 5u, 4095u}) {
      for (u32 fine = 0; fine < 4; fine++) { if ((p.a >> 16) && sort_key(level, c, c == ',' ? fine : 0u) != sort_key_of_props(level, x, c == ',' ? fine : 0u)) { keys = false; } }
    }
    if (!same || !keys) { fprintf(stderr, "byte %02x: the table entry %x disagrees with the predicates\n", c, x); return 1; }
    for (u32 c2 = 0; c2 < 256; c2++) {
      for (int have = 0; have < 4; have++) {
        const bool have1 = (have & 1) != 0, have2 = have1 && (have & 2) != 0;
        if (comma_fine_bits_direct(c, have1, c2, have2) != comma_fine_bits_of_props(x, have1, token_props_of(c2), have2)) {
          fprintf(stderr, "comma_fine_bits from table entries: followers %02x %02x (%d %d) disagree\n", c, c2, have1, have2);
          return 1;
        }
        checked++;
      }
    }
  }
  { // round 6: the token's own verdict from one table entry per byte (what k_tok_apply evaluates) against token_rule_self, on EVERY triple of bytes
    u32 entry[256];
    for (u32 v = 0; v < 256; v++) { entry[v] = token_entry_of(v); if ((entry[v] & 0xFFFFu) != token_props_of(v)) { fprintf(stderr, "entry %02x: lower half\n", v); return 1; } }
    for (u32 c = 0; c < 256; c++) {
      for (u32 prev = 0; prev < 256; prev++) {
        for (u32 prev2 = 0; prev2 < 256; prev2++) {
          for (int first = 0; first < 2; first++) {
            u32 r1 = 9, r2 = 9;
            const u32 e1 = token_rule_self(T, first != 0, c, prev, prev2, &r1), e2 = token_rule_self_entries(first != 0, entry[c], entry[prev], entry[prev2], &r2);
            if (e1 != e2 || r1 != r2) {
              fprintf(stderr, "c %02x prev %02x prev2 %02x first %d: token_rule_self says %u (rank %u), the entries say %u (rank %u)\n", c, prev, prev2, first, e1, r1, e2, r2);
              return 1;
            }
            checked++;
          }
        }
      }
    }
  }
  printf("%lu combinations agree\n", checked);
  return 0;
}
