// tests/host/test_escape_carry_model.cpp -- GROUNDWORK for carrying the escape state in the scan (DESIGN.md section 4,
// worst cases): a model of the scheme that would replace k_escape_table, checked against a plain sequential scan on adversarial input.
//
// Today a segment whose look-back is all backslashes learns its escape carry-in from a table that a separate kernel fills by walking
// back over the run -- a second read of the input on documents that are mostly long backslash runs.  The scheme modelled here needs no
// walk and no table:
//   * a segment whose look-back (LB bytes) is nothing but backslashes ASSUMES "first byte not escaped" (case B); one whose previous byte
//     is a quote behind LB - 1 backslashes assumes "that quote is real" (case C); every other segment knows its carries exactly (case A);
//   * every segment publishes, from its own bytes: its escape element {set 0, set 1, pass} (trailing run parity; pass = all
//     backslashes, SEG is even), whether its last byte is a quote and whether that quote is real / escaped / "escaped iff the
//     segment's carry-in makes it so" (everything in front of it is backslashes), and -- case B only -- the position L of the first
//     byte behind its leading run and whether that byte is a quote;
//   * a serial pass over the segments (the resolve kernels' job) composes the elements, finds which assumptions were wrong, and
//     repairs: a wrong escape assumption with a quote at L flips the quote's realness = the in-string state of everything behind it
//     (pick the other hypothesis, toggle the parity the segment hands on) and the candidate bit of byte L + 1 if that byte is a scalar
//     (a scalar behind a real closing quote starts a token, behind an escaped one it does not); a wrong quote assumption (case C)
//     toggles the candidate bit of byte 0 if that byte is a scalar.  L + 1 beyond the segment IS the next segment's case C.
// The model works on bytes, not bit planes, with segment and look-back sizes as parameters (small ones make every corner frequent);
// "structural" follows the reference's definitions (json_scanner.h:44-90, json_string_scanner.h:62-85, json_escape_scanner.h:50-71).
// Usage: test_escape_carry_model <seed> <documents> [sabotage 1-4: leave one repair out -- the run must fail]
#include "sj_oracle.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef std::vector<uint8_t> bytes;

static bool is_ws(uint8_t b) { return b == 0x20 || b == 0x09 || b == 0x0A || b == 0x0D; }
static bool is_op(uint8_t b) { const uint8_t l = b | 0x20; return b < 0x80 && (l == 0x2C || l == 0x3A || l == 0x7B || l == 0x7D); }
static bool is_scalar(uint8_t b) { return !is_ws(b) && !is_op(b); }

struct scan_out {
  std::vector<uint8_t> cand, tail; // per byte: structural candidate (strings ignored); in_string ^ quote, relative to "starts outside"
  int parity;                      // real quotes, modulo 2
};
// the scan of bytes [from, to) with the carries given: first byte escaped?, previous byte a non-quote scalar?
static scan_out scan_bytes(const bytes &b, size_t from, size_t to, int e_in, int p_in) {
  scan_out o;
  o.cand.assign(to - from, 0);
  o.tail.assign(to - from, 0);
  int e = e_in, s = 0, prev_nqs = p_in;
  for (size_t i = from; i < to; i++) {
    const uint8_t c = b[i];
    const int escaped = e;
    e = escaped ? 0 : (c == '\\');
    const int quote = c == '"' && !escaped;
    s ^= quote;
    const int scalar = is_scalar(c);
    o.cand[i - from] = uint8_t(is_op(c) || (scalar && !prev_nqs));
    o.tail[i - from] = uint8_t(s ^ quote);
    prev_nqs = scalar && !quote;
  }
  o.parity = s;
  return o;
}

enum { E_SET0 = 0, E_SET1 = 1, E_PASS = 2 };
enum { LQ_NONE = 0, LQ_REAL = 1, LQ_ESCAPED = 2, LQ_DEPENDS = 3 };
struct segment {
  scan_out scan;
  int kind;      // 'A' exact, 'B' escape carry assumed 0, 'C' previous quote assumed real
  int eo, lq;
  size_t L;      // case B: length of the leading backslash run
  bool quote_at_L, scalar_behind_L, scalar_first;
};

static unsigned long n_case_b = 0, n_case_c = 0, n_flip_quote = 0, n_patch_behind = 0, n_patch_first = 0, n_depends = 0;
static int sabotage = 0; // 1 ... 4: leave one of the repairs out (the test must then fail: it has teeth)
static std::vector<uint32_t> segmented(const bytes &doc, size_t len, size_t SEG, size_t LB) {
  bytes b = doc;
  while (b.size() % SEG) { b.push_back(0x20); } // the kernels read bytes beyond len as spaces
  const size_t nseg = b.size() / SEG;
  std::vector<segment> seg(nseg);
  for (size_t k = 0; k < nseg; k++) { // ---- "summarize": every segment by itself
    const size_t S = k * SEG;
    segment &g = seg[k];
    int e_in = 0, p_in = 0;
    g.kind = 'A';
    if (S > 0) {
      const size_t lb0 = S >= LB ? S - LB : 0;
      size_t run = 0; // backslashes that end at S - 1, as far as the look-back shows
      while (run < S - lb0 && b[S - 1 - run] == '\\') { run++; }
      if (S >= LB && run == LB) { g.kind = 'B'; e_in = 0; p_in = 1; }
      else if (b[S - 1] == '"') {
        size_t r2 = 0; // backslashes in front of the quote
        while (r2 < S - 1 - lb0 && b[S - 2 - r2] == '\\') { r2++; }
        if (S >= LB && r2 == LB - 1) { g.kind = 'C'; e_in = 0; p_in = 0; }
        else { e_in = 0; p_in = int(r2 & 1); } // an escaped quote is a non-quote scalar
      } else {
        e_in = int(run & 1);
        p_in = is_scalar(b[S - 1]);
      }
    }
    g.scan = scan_bytes(b, S, S + SEG, e_in, p_in);
    size_t t = 0; // trailing backslashes
    while (t < SEG && b[S + SEG - 1 - t] == '\\') { t++; }
    g.eo = t == SEG ? E_PASS : int(t & 1);
    g.lq = LQ_NONE;
    if (b[S + SEG - 1] == '"') {
      size_t q = 0;
      while (q < SEG - 1 && b[S + SEG - 2 - q] == '\\') { q++; }
      g.lq = q == SEG - 1 ? LQ_DEPENDS : ((q & 1) ? LQ_ESCAPED : LQ_REAL);
    }
    g.L = 0;
    while (g.L < SEG && b[S + g.L] == '\\') { g.L++; }
    g.quote_at_L = g.L < SEG && b[S + g.L] == '"';
    g.scalar_behind_L = g.L + 1 < SEG && is_scalar(b[S + g.L + 1]);
    g.scalar_first = is_scalar(b[S]);
  }
  std::vector<uint32_t> out; // ---- "resolve" + "emit": one serial pass
  int s = 0, e_prev_out = 0, lq_prev = LQ_NONE;
  for (size_t k = 0; k < nseg; k++) {
    segment &g = seg[k];
    const size_t S = k * SEG;
    // the true escape carry-in (cases A and C know it; the model recomputes it for A from the elements as a cross-check)
    const int e_true = k == 0 ? 0 : e_prev_out;
    bool flip_quote = false;
    std::vector<uint8_t> cand = g.scan.cand;
    n_case_b += g.kind == 'B';
    n_case_c += g.kind == 'C';
    if (g.kind == 'B' && e_true == 1) {
      if (g.quote_at_L) {
        flip_quote = sabotage != 1;
        n_flip_quote++;
        if (g.scalar_behind_L && sabotage != 2) { cand[g.L + 1] ^= 1; n_patch_behind++; }
      }
    }
    if (g.kind == 'C') {
      const bool escaped = lq_prev == LQ_ESCAPED; // (lq_prev is resolved below, never DEPENDS here)
      if (escaped && g.scalar_first && sabotage != 3) { cand[0] ^= 1; n_patch_first++; }
    }
    const int S_eff = s ^ (flip_quote ? 1 : 0);
    const size_t before = out.size();
    for (size_t i = 0; i < SEG; i++) {
      if (cand[i] && !(g.scan.tail[i] ^ S_eff) && S + i < len) { out.push_back(uint32_t(S + i)); }
    }
    // what the resolve step would have to know WITHOUT the bytes: the segment's count under both in-string hypotheses (as today) plus
    // the +-1 of each patch under both -- its output base for the next segment must come out of the summaries alone
    {
      int count[2] = {0, 0}, d_behind[2] = {0, 0}, d_first[2] = {0, 0};
      for (int h = 0; h < 2; h++) {
        for (size_t i = 0; i < SEG; i++) { count[h] += g.scan.cand[i] && !(g.scan.tail[i] ^ h); }
        if (g.scalar_behind_L && !(g.scan.tail[g.L + 1] ^ h)) { d_behind[h] = g.scan.cand[g.L + 1] ? -1 : 1; }
        if (g.scalar_first && !(g.scan.tail[0] ^ h)) { d_first[h] = g.scan.cand[0] ? -1 : 1; }
      }
      int expect = count[S_eff];
      if (g.kind == 'B' && e_true == 1 && g.quote_at_L && g.scalar_behind_L) { expect += d_behind[S_eff]; }
      if (g.kind == 'C' && lq_prev == LQ_ESCAPED && g.scalar_first) { expect += d_first[S_eff]; }
      if (sabotage == 0 && size_t(expect) != out.size() - before) {
        fprintf(stderr, "segment %zu: the summaries say %d structurals, the masks hold %zu\n", k, expect, out.size() - before);
        exit(3);
      }
    }
    s ^= g.scan.parity ^ (flip_quote ? 1 : 0);
    n_depends += g.lq == LQ_DEPENDS;
    lq_prev = g.lq == LQ_DEPENDS ? (((SEG - 1 + size_t(sabotage == 4 ? 0 : e_true)) & 1) ? LQ_ESCAPED : LQ_REAL) : g.lq;
    e_prev_out = g.eo == E_PASS ? e_true : g.eo;
  }
  return out;
}

int main(int argc, char **argv) {
  uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
  const long docs = argc > 2 ? atol(argv[2]) : 200000;
  sabotage = argc > 3 ? atoi(argv[3]) : 0;
  auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(seed >> 33); };
  const char alphabet[] = "\\\\\\\\\\\\\\\"\"a1 ,{}[]:\n";
  unsigned long checked = 0;
  for (long d = 0; d < docs; d++) {
    const size_t SEG = size_t(8) << (rnd() % 4); // 8, 16, 32, 64
    size_t LB = size_t(2) << (rnd() % 3);        // 2, 4, 8
    if (LB > SEG) { LB = SEG; }
    const size_t len = 1 + rnd() % (SEG * 12);
    bytes doc(len);
    for (size_t i = 0; i < len;) { // runs of backslashes of all lengths, single characters in between
      if (rnd() % 3 == 0) {
        size_t run = 1 + rnd() % (2 * SEG + 3);
        while (run-- && i < len) { doc[i++] = '\\'; }
      } else {
        doc[i++] = uint8_t(alphabet[rnd() % (sizeof alphabet - 1)]);
      }
    }
    // the sequential scan: one "segment" that is the whole document -- anchored against the oracle's scan
    const scan_out whole = scan_bytes(doc, 0, len, 0, 0);
    std::vector<uint32_t> want;
    for (size_t i = 0; i < len; i++) { if (whole.cand[i] && !whole.tail[i]) { want.push_back(uint32_t(i)); } }
    std::vector<uint32_t> oidx(len + 8);
    uint32_t oflags = 0;
    const uint32_t on = sjo_scan(doc.data(), len, oidx.data(), &oflags);
    if (on != want.size() || !std::equal(want.begin(), want.end(), oidx.begin())) {
      fprintf(stderr, "the model's sequential scan and the oracle disagree (%zu vs %u structurals)\n", want.size(), on);
      return 2;
    }
    const std::vector<uint32_t> got = segmented(doc, len, SEG, LB);
    checked++;
    if (got != want) {
      fprintf(stderr, "SEG %zu LB %zu len %zu: %zu structurals, the sequential scan has %zu\n", SEG, LB, len, got.size(), want.size());
      for (size_t i = 0; i < len; i++) { fputc(doc[i] == '\n' ? '/' : doc[i], stderr); }
      fputc('\n', stderr);
      return 1;
    }
  }
  printf("%lu documents: the segmented scan without an escape table equals the sequential scan (case B %lu, case C %lu segments; quote flips %lu, "
         "patches behind the quote %lu, patches of the first byte %lu, carry-dependent last quotes %lu)\n", checked, n_case_b, n_case_c, n_flip_quote, n_patch_behind,
         n_patch_first, n_depends);
  return 0;
}
