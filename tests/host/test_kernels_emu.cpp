// tests/host/test_kernels_emu.cpp -- CPU tier: the stage-1 / minify / validate_utf8 KERNEL SOURCES (sjgpu_kernels.hip, sjgpu_fused.hip,
// sjgpu_small.hip, compiled as C++ against tests/host/emu) run on documents and are compared with the oracle, launcher by launcher:
//   split      launch_stage1 / launch_minify          (summarize -> resolve -> emit)
//   fused      launch_stage1_fused / launch_minify_fused: 16 KiB tiles below the small-input limit, the pipelined 64 KiB-tile kernels
//              and k_minify_onchip above it (the limit is lowered for the test so that a few hundred KiB take the large-input kernels)
//   docs       launch_docs (one workgroup per document)
//   ranges     a document scanned as consecutive ranges of one buffer (the overlapped host path's protocol)
//   parity     launch_string_parity
// The documents are adversarial for the carries between spans: backslash runs of every length across 64-byte, 4 KiB, 16 KiB, 64 KiB
// and 1 MiB boundaries, quotes behind them, control characters, multi-byte UTF-8 (valid and not), dense and empty output.
// Usage: test_kernels_emu <seed> <documents> [max KiB per document] [what: all|split|fused|docs|ranges]
#include "sjgpu.h"
#include "sjgpu_internal.h"
#include "sj_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sjgpu { extern uint64_t debug_fused_small_below; }
using namespace sjgpu;

static uint64_t rng_state = 1;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return uint32_t(rng_state >> 33);
}
static uint32_t rnd_below(uint32_t n) { return n ? rnd() % n : 0; }

typedef std::vector<uint8_t> bytes;
static void put(bytes &d, const char *s) { d.insert(d.end(), s, s + strlen(s)); }

// ---- documents ----------------------------------------------------------------------------------------------------------------
static const uint32_t BOUNDARIES[] = {64, 4096, 16384, 65536, 32768, 1u << 20};
// pad with `fill` so that the next byte lands at (a multiple of one of the boundaries) + delta
static void pad_to_boundary(bytes &d, uint8_t fill) {
  const uint32_t B = BOUNDARIES[rnd_below(4)];
  static const int deltas[] = {-66, -65, -64, -63, -3, -2, -1, 0, 1, 2, 3, 62, 63, 64, 65};
  const int delta = deltas[rnd_below(sizeof deltas / sizeof deltas[0])];
  const size_t at = d.size();
  size_t target = (at / B + 1) * B + size_t(int(B) + delta) % B;
  if (target > at + 70000) { target = at + rnd_below(200); }
  d.insert(d.end(), target - at, fill);
}
static void soup(bytes &d, size_t n) { // token soup: every class of byte the scanner distinguishes
  static const char alphabet[] = "\\\\\"\"\"  \n\t\r,,::[]{}aZ09-.e+\x0c\x1a\x01\x1f\x7f";
  for (size_t i = 0; i < n; i++) {
    const uint32_t r = rnd_below(40);
    if (r == 0) { // a multi-byte character, sometimes broken
      static const char *const utf[] = {"\xc3\xa9", "\xe2\x82\xac", "\xf0\x9f\x98\x80", "\xe0\xa0\x80", "\xed\x9f\xbf", "\xc0\xaf", "\xed\xa0\x80", "\xf4\x90\x80\x80", "\x80", "\xe2\x82", "\xf0\x9f"};
      put(d, utf[rnd_below(rnd_below(4) ? 5 : 11)]);
    } else {
      d.push_back(uint8_t(alphabet[rnd_below(sizeof alphabet - 1 - (rnd_below(6) ? 5 : 0))])); // control characters are rarer
    }
  }
}
static void value(bytes &d, int depth) { // a valid JSON value
  const uint32_t r = rnd_below(depth > 5 ? 4 : 7);
  if (r == 0) { put(d, "12.5e3"); }
  else if (r == 1) { put(d, "true"); }
  else if (r == 2 || r == 3) {
    d.push_back('"');
    const uint32_t n = rnd_below(rnd_below(8) ? 20 : 300);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t k = rnd_below(30);
      if (k == 0) { put(d, "\\\""); }
      else if (k == 1) { put(d, "\\\\"); }
      else if (k == 2) { put(d, "\\u00e9"); }
      else if (k == 3) { put(d, "\xe2\x82\xac"); }
      else if (k == 4) { put(d, " , : [ { "); }
      else { d.push_back(uint8_t('a' + rnd_below(26))); }
    }
    d.push_back('"');
  } else if (r == 4) {
    d.push_back('[');
    const uint32_t n = rnd_below(6);
    for (uint32_t i = 0; i < n; i++) { if (i) { put(d, rnd_below(2) ? "," : ", "); } value(d, depth + 1); }
    d.push_back(']');
  } else {
    d.push_back('{');
    const uint32_t n = rnd_below(5);
    for (uint32_t i = 0; i < n; i++) { if (i) { put(d, ",\n  "); } put(d, "\"k"); d.push_back(uint8_t('0' + i)); put(d, "\": "); value(d, depth + 1); }
    d.push_back('}');
  }
}
static bool g_sparse_only = false;  // the "sparse" road: every document is of kind 11
static bool g_mostly_plain = false; // the "direct" road: three documents in four are NDJSON-like (plain: every 16 KiB segment meets a newline in its first chunk)
static bytes make_document(size_t target) {
  bytes d;
  uint32_t kind = rnd_below(12);
  if (g_mostly_plain && rnd_below(4)) { kind = 1; }
  if (g_sparse_only) { kind = 11; }
  while (d.size() < target) {
    switch (kind) {
    case 0: soup(d, 1 + rnd_below(400)); break;
    case 1: // valid JSON, one value after the other (NDJSON-like)
      value(d, 0);
      d.push_back('\n');
      break;
    case 2: case 3: { // backslash runs placed around boundaries, in and out of strings, quotes glued to their ends
      pad_to_boundary(d, rnd_below(3) ? ' ' : 'x');
      if (rnd_below(4)) { d.push_back('"'); }
      static const uint32_t lens[] = {0, 1, 2, 3, 62, 63, 64, 65, 66, 127, 128, 129, 4095, 4096, 4097, 16383, 16384, 16385, 32768, 32769, 65535, 65536, 65537, 20000, 20001};
      uint32_t run = lens[rnd_below(sizeof lens / sizeof lens[0])];
      if (rnd_below(3) == 0) { run = rnd_below(70000); }
      d.insert(d.end(), run, '\\');
      const uint32_t t = rnd_below(6);
      if (t == 0) { put(d, "\"a"); } else if (t == 1) { put(d, "\"\""); } else if (t == 2) { put(d, "\" ,"); } else if (t == 3) { put(d, "n\""); } else if (t == 4) { put(d, "\"\\\"1"); }
      soup(d, rnd_below(40));
      break;
    }
    case 8: case 9: case 10: { // the corners of the escape carry (sj_xcarry.h): a run that begins at least 64 bytes in front of a span boundary
      // (4, 8 and 16 KiB spans exist) and ends a few bytes behind it, or one byte short of the next boundary, or covers whole spans; behind
      // it a quote -- real or escaped, that is the question -- and behind the quote a scalar, an operator, whitespace or another quote
      static const uint32_t spans[] = {4096, 8192, 16384, 65536};
      const uint32_t N = spans[rnd_below(4)];
      const size_t at = d.size();
      const size_t boundary = (at / N + 1) * N;
      const uint32_t before = 64 + rnd_below(4) * 1 + (rnd_below(3) ? 0 : rnd_below(200)) - (rnd_below(5) == 0 ? 2 : 0); // 62 ... : sometimes just too short
      size_t start = boundary > before ? boundary - before : 0;
      if (start < at) { start = at; }
      const uint8_t fill = kind == 8 ? ' ' : (kind == 9 ? 'x' : '\n');
      if (rnd_below(2)) { d.insert(d.end(), start - at, fill); }
      else { if (start > at) { d.push_back('"'); } d.insert(d.end(), start > at ? start - at - 1 : 0, 'y'); } // the run sits inside a string
      static const int ends[] = {0, 1, 2, 3, 4, 5, 63, 64, 65, 66, -1, -2, -3, -64, -65};
      const int e = ends[rnd_below(sizeof ends / sizeof ends[0])];
      const uint32_t whole = rnd_below(3) == 0 ? rnd_below(4) : 0; // whole spans of backslashes in between
      const size_t stop = size_t(int64_t(boundary + size_t(whole) * N + ((e < 0) ? N : 0)) + e);
      if (stop > d.size()) { d.insert(d.end(), stop - d.size(), '\\'); }
      static const char *const tails[] = {"\"a", "\"\"", "\" ,", "\"x\"", "\",1", "\"\n", "n\"", "\"1", "\"\\\"1", "\"", "a", ",", " "};
      put(d, tails[rnd_below(sizeof tails / sizeof tails[0])]);
      if (rnd_below(2)) { // (no control characters: one inside a string and nobody compares the document's list)
        static const char plain[] = "\\\"\"  ,,::[]{}aZ09-.e+";
        for (uint32_t i = 0, n = rnd_below(30); i < n; i++) { d.push_back(uint8_t(plain[rnd_below(sizeof plain - 1)])); }
      }
      break;
    }
    case 4: { // long strings (two-hypothesis segments), sometimes with a control character inside
      d.push_back('"');
      const uint32_t n = rnd_below(50000);
      for (uint32_t i = 0; i < n; i++) { d.push_back(uint8_t("abc ,:[]{}\\n"[rnd_below(12)])); if (d.back() == '\\') { d.push_back('n'); } }
      if (rnd_below(4) == 0) { d.push_back('\n'); }
      put(d, "\", ");
      break;
    }
    case 5: // dense output
      for (uint32_t i = 0, n = 1 + rnd_below(6000); i < n; i++) { d.push_back(uint8_t("[]{},:1"[rnd_below(7)])); }
      break;
    case 6: { // non-ASCII text
      d.push_back('"');
      for (uint32_t i = 0, n = rnd_below(3000); i < n; i++) { put(d, rnd_below(2) ? "\xe6\x97\xa5" : "\xd0\x96"); if (rnd_below(300) == 0) { d.push_back(0xE6); } }
      put(d, "\"\n");
      break;
    }
    case 11: { // round 6: a HANDFUL of candidates per 16 KiB segment -- k_stage1_summarize ships such segments as a list of <= 32 words, not as planes.  Segments
      // are filled with filler (spaces: nothing; letters: one scalar run; the inside of a string; backslashes in a string) and exactly k candidates are
      // dropped at random places, k around the limit: 0, 1, 2, 31, 32, 33, 40, 64; some segments get a newline in their first chunk (resolved), some none
      // (two hypotheses: the list carries the string_tail bits), quotes open and close across segments so that both hypotheses are selected
      static const uint32_t counts[] = {0, 1, 2, 3, 16, 31, 32, 32, 33, 34, 40, 64, 200};
      const size_t seg_end = (d.size() / SEG_BYTES + 1) * SEG_BYTES;
      const uint32_t k = counts[rnd_below(sizeof counts / sizeof counts[0])];
      const uint32_t filler = rnd_below(4);
      const uint8_t fill = filler == 0 ? ' ' : (filler == 1 ? 'x' : (filler == 2 ? 'y' : '\\'));
      if (filler >= 2 && rnd_below(2)) { d.push_back('"'); } // the rest of the segment lies inside a string (or closes one that was open)
      const size_t begin = d.size();
      d.insert(d.end(), seg_end > begin ? seg_end - begin : 0, fill);
      if (fill == '\\' && ((seg_end - begin) & 1u)) { d.back() = 'z'; } // (an even run: what follows is not escaped)
      if (rnd_below(2) && seg_end - begin > 64) { d[begin + rnd_below(uint32_t(std::min<size_t>(seg_end - begin, 4000)))] = '\n'; } // resolved -- unless inside a string: then the document is an error and nobody compares
      for (uint32_t i = 0; i < k && seg_end - begin > 8; i++) {
        const size_t at = begin + 2 + rnd_below(uint32_t(seg_end - begin - 4));
        static const char cands[] = ",:[]{}\"\"17";
        d[at] = uint8_t(cands[rnd_below(sizeof cands - 1)]);
        if (rnd_below(3) == 0) { d[at - 1] = ' '; } // a scalar behind whitespace starts a token
      }
      break;
    }
    default: // one enormous backslash run
      put(d, "[\"");
      d.insert(d.end(), target + rnd_below(3), '\\');
      put(d, rnd_below(2) ? "\", 1]" : "\"\", 1]");
      break;
    }
  }
  if (rnd_below(3) == 0) { d.resize(rnd_below(uint32_t(d.size())) + 1); } // cut anywhere: ends inside strings, escapes, characters
  return d;
}

// ---- one "context" ---------------------------------------------------------------------------------------------------------------
struct workspace {
  size_t cap = 0;
  std::vector<uint4> masks;
  std::vector<seg_summary> summ;
  std::vector<seg_prefix> pref;
  std::vector<uint64_t> result_and_desc; // [result (2 words)][descriptors][ticket]
  std::vector<uint8_t> esc;
  std::vector<uint8_t> tokstage, tok; // launch_stage1 with a token stream: the staging area between its two kernels, the stream itself
  std::vector<uint32_t> idx;
  std::vector<uint8_t> out;
  uint8_t *in = nullptr; // 16-byte aligned copy of the document, nothing readable... (the kernels must not depend on what lies behind len)
  std::vector<uint8_t> in_store;
  scan_result_dev *result() { return reinterpret_cast<scan_result_dev *>(result_and_desc.data()); }
  uint64_t *desc() { return result_and_desc.data() + 2; }
  void fit(const bytes &doc) {
    const size_t len = doc.size();
    if (len > cap) {
      cap = len * 2 + 65536;
      const size_t nseg = num_segments(cap);
      masks.assign(nseg * (SEG_BYTES / BLOCK_BYTES), uint4{0, 0, 0, 0});
      summ.assign(nseg + num_groups(cap), seg_summary{0, 0, 0, 0});
      pref.assign(nseg, seg_prefix{0, 0});
      result_and_desc.assign(2 + num_fused_tiles(cap) * 4 + 8 + FUSED_WORKSPACE_EXTRA_WORDS, 0);
      esc.assign(SEGMENT_BYTES_TABLE, 0); // launch_string_parity: one byte per segment
      idx.assign(cap + 16, 0);
      tokstage.assign(size_t(nseg) * SEG_BYTES + 64, 0);
      tok.assign(cap + 64, 0);
      out.assign(cap + 64, 0);
      in_store.assign(cap + 64, 0);
    }
    in = in_store.data() + ((16 - (reinterpret_cast<uintptr_t>(in_store.data()) & 15)) & 15);
    memcpy(in, doc.data(), len);
    memset(in + len, 0x5C, 32); // garbage behind the end: backslashes, to catch a read past len
  }
};

static unsigned long n_checked = 0, n_failed = 0;
static void report(const char *what, const bytes &doc, const char *detail);
// the single-pass kernels are launched with clean = true (nothing is cleared in front of them) and must leave the workspace as they found it:
// descriptors, ticket, the count of workgroups gone and the flag word all zero again (sjgpu_fused.hip: leave_and_clean)
static void check_workspace_clean(workspace &w, const bytes &doc, const char *what) {
  n_checked++;
  for (size_t i = 2; i < w.result_and_desc.size(); i++) {
    if (w.result_and_desc[i] != 0) {
      char detail[96];
      snprintf(detail, sizeof detail, "workspace word %zu is %llx behind the kernel", i - 2, (unsigned long long)w.result_and_desc[i]);
      report(what, doc, detail);
      std::fill(w.result_and_desc.begin() + 2, w.result_and_desc.end(), 0);
      return;
    }
  }
}
static void report(const char *what, const bytes &doc, const char *detail) {
  n_failed++;
  fprintf(stderr, "MISMATCH %s: len %zu: %s\n", what, doc.size(), detail);
  if (n_failed <= 3) {
    static int dumped = 0;
    char name[64];
    snprintf(name, sizeof name, "/tmp/emu_fail_%d.bin", dumped++);
    if (FILE *f = fopen(name, "wb")) { fwrite(doc.data(), 1, doc.size(), f); fclose(f); fprintf(stderr, "  document saved as %s\n", name); }
  }
}

struct expected {
  std::vector<uint32_t> idx;
  uint32_t n = 0, flags = 0;
  bytes mini;
  int mini_err = 0;
  int utf8_ok = 1;
};
static expected oracle(const bytes &doc) {
  expected e;
  e.idx.resize(doc.size() + 8);
  e.n = sjo_scan(doc.data(), doc.size(), e.idx.data(), &e.flags);
  e.mini.resize(doc.size() + 8);
  size_t ml = 0;
  e.mini_err = sjo_minify(doc.data(), doc.size(), e.mini.data(), &ml);
  e.mini.resize(ml);
  e.utf8_ok = sjo_validate_utf8(doc.data(), doc.size());
  return e;
}

static long n_direct_done = 0, n_direct_gave_up = 0;
static void check_stage1(const char *what, const bytes &doc, const expected &e, workspace &w) {
  const scan_result_dev r = *w.result();
  n_checked++;
  char detail[256];
  const uint32_t got_flags = r.flags & 7u;
  if (r.flags & ~7u) { snprintf(detail, sizeof detail, "flags %#x", r.flags); report(what, doc, detail); return; }
  if (got_flags != e.flags) { snprintf(detail, sizeof detail, "flags %#x, oracle %#x (n %u / %u)", got_flags, e.flags, r.n, e.n); report(what, doc, detail); return; }
  if (e.flags & SJGPU_F_UNESCAPED_CTRL) { return; } // the reference returns before anybody looks at the list
  if (r.n != e.n) {
    snprintf(detail, sizeof detail, "n %u, oracle %u", r.n, e.n);
    report(what, doc, detail);
    if (getenv("SJ_EMU_DOC")) { for (uint32_t i = 0; i < r.n && i < e.n + 1; i++) { if (i >= e.n || w.idx[i] != e.idx[i]) { fprintf(stderr, "  first difference: idx[%u] = %u, oracle %u\n", i, w.idx[i], i < e.n ? e.idx[i] : 0u); break; } } }
    return;
  }
  for (uint32_t i = 0; i < e.n; i++) {
    if (w.idx[i] != e.idx[i]) { snprintf(detail, sizeof detail, "idx[%u] = %u, oracle %u (n %u)", i, w.idx[i], e.idx[i], e.n); report(what, doc, detail); return; }
  }
  const uint32_t L = uint32_t(doc.size());
  if (w.idx[e.n] != L || w.idx[e.n + 1] != L || w.idx[e.n + 2] != 0) { report(what, doc, "sentinels"); }
}
// the token stream beside the list: tok[i] = the byte at idx[i]
static void check_tokens(const char *what, const bytes &doc, const expected &e, workspace &w) {
  const scan_result_dev r = *w.result();
  n_checked++;
  if ((r.flags & 7u) != e.flags || (e.flags & SJGPU_F_UNESCAPED_CTRL) || r.n != e.n) { return; } // check_stage1 has reported it / nobody looks at the list
  for (uint32_t i = 0; i < e.n; i++) {
    const uint8_t want = e.idx[i] < doc.size() ? doc[e.idx[i]] : uint8_t(0x20);
    if (w.tok[i] != want) {
      char detail[128];
      snprintf(detail, sizeof detail, "tok[%u] = %#x, the byte at idx[%u] = %u is %#x (n %u)", i, w.tok[i], i, e.idx[i], want, e.n);
      report(what, doc, detail);
      return;
    }
  }
}
static void check_minify(const char *what, const bytes &doc, const expected &e, workspace &w) {
  const scan_result_dev r = *w.result();
  n_checked++;
  char detail[256];
  if (r.flags & ~1u) { snprintf(detail, sizeof detail, "flags %#x", r.flags); report(what, doc, detail); return; }
  const bool unclosed = (r.flags & 1u) != 0;
  if (unclosed != (e.mini_err != 0)) { snprintf(detail, sizeof detail, "unclosed %d, oracle error %d", int(unclosed), e.mini_err); report(what, doc, detail); return; }
  if (unclosed) { return; }
  if (r.out_len != e.mini.size()) { snprintf(detail, sizeof detail, "out_len %llu, oracle %zu", (unsigned long long)r.out_len, e.mini.size()); report(what, doc, detail); return; }
  if (memcmp(w.out.data(), e.mini.data(), e.mini.size()) != 0) {
    size_t i = 0;
    while (w.out[i] == e.mini[i]) { i++; }
    snprintf(detail, sizeof detail, "byte %zu differs", i);
    report(what, doc, detail);
  }
}

int main(int argc, char **argv) {
  rng_state = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
  const long docs = argc > 2 ? atol(argv[2]) : 100;
  const size_t max_kib = argc > 3 ? size_t(atol(argv[3])) : 200;
  const std::string what = argc > 4 ? argv[4] : "all";
  const bool all = what == "all";
  g_mostly_plain = what == "direct";
  g_sparse_only = what == "sparse";
  workspace w;
  size_t total_bytes = 0;
  for (long k = 0; k < docs; k++) {
    // mostly a few segments; some documents cross the 1 MiB groups of the resolve step
    size_t target = 1 + rnd_below(uint32_t(max_kib * 1024));
    if (rnd_below(4)) { target = 1 + target % (96 * 1024); }
    if (rnd_below(10) == 0) { target = rnd_below(300); }
    bytes doc = make_document(target);
    if (g_mostly_plain && rnd_below(3) == 0 && doc.size() > 100) { // ... some of them broken: a raw control character inside a string, a bad UTF-8 byte, an unclosed string
      const uint32_t how = rnd_below(3);
      const size_t at = doc.size() / 2 + rnd_below(uint32_t(doc.size() / 3));
      if (how == 0) { static const char bad[] = "\"a\x01z\" "; doc.insert(doc.begin() + long(at), bad, bad + 6); }
      else if (how == 1) { doc[at] = 0xFF; }
      else { put(doc, "\"open , "); }
    }
    if (const char *path = getenv("SJ_EMU_DOC")) { // one document from a file (debugging a reported mismatch)
      doc.clear();
      if (FILE *f = fopen(path, "rb")) { int c; while ((c = fgetc(f)) != EOF) { doc.push_back(uint8_t(c)); } fclose(f); }
    }
    const size_t len = doc.size();
    total_bytes += len;
    const expected e = oracle(doc);
    w.fit(doc);
    const scan_origin whole{0, 0, 0};
    if (all || what == "split" || what == "sparse") {
      scan_origin org = whole;
      std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu);
      launch_stage1(w.in, len, w.masks.data(), w.summ.data(), w.pref.data(), w.idx.data(), len + 3, w.result(), org, nullptr, nullptr);
      check_stage1("split stage 1", doc, e, w);
      std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu); // the same scan with the token stream beside the offsets
      std::fill(w.tok.begin(), w.tok.begin() + len + 8, uint8_t(0xEE));
      launch_stage1(w.in, len, w.masks.data(), w.summ.data(), w.pref.data(), w.idx.data(), len + 3, w.result(), org, nullptr, nullptr, w.tokstage.data(), w.tok.data());
      check_stage1("split stage 1 with tokens", doc, e, w);
      check_tokens("split stage 1 with tokens", doc, e, w);
      launch_minify(w.in, len, w.summ.data(), w.pref.data(), w.out.data(), w.result(), org, nullptr, nullptr);
      check_minify("split minify", doc, e, w);
      launch_validate_utf8(w.in, len, w.result(), nullptr, nullptr);
      n_checked++;
      if (((w.result()->flags & SJGPU_F_UTF8_ERROR) == 0) != (e.utf8_ok != 0)) { report("validate_utf8", doc, "verdict"); }
    }
    if (all || what == "fused") {
      for (int large = 0; large < 2; large++) {
        debug_fused_small_below = large ? 1 : FUSED_SMALL_BELOW; // 1: every document takes the pipelined 64 KiB-tile kernels
        scan_origin org = whole;
        std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu);
        *w.result() = scan_result_dev{0xDEADBEEFu, 0xFFFFFFFFu, ~0ull}; // nothing clears the result either: every field is written
        launch_stage1_fused(w.in, len, w.desc(), w.idx.data(), len + 3, w.result(), org, 6, nullptr, nullptr, true);
        check_stage1(large ? "pipelined stage 1" : "fused stage 1 (16 KiB tiles)", doc, e, w);
        check_workspace_clean(w, doc, large ? "pipelined stage 1" : "fused stage 1 (16 KiB tiles)");
        std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu); // the same scan with the token stream beside the offsets (round 6: gathered at emission)
        std::fill(w.tok.begin(), w.tok.begin() + len + 8, uint8_t(0xEE));
        *w.result() = scan_result_dev{0xDEADBEEFu, 0xFFFFFFFFu, ~0ull};
        launch_stage1_fused(w.in, len, w.desc(), w.idx.data(), len + 3, w.result(), org, 6, nullptr, nullptr, true, w.tok.data());
        check_stage1(large ? "pipelined stage 1 with tokens" : "fused stage 1 with tokens", doc, e, w);
        check_tokens(large ? "pipelined stage 1 with tokens" : "fused stage 1 with tokens", doc, e, w);
        check_workspace_clean(w, doc, large ? "pipelined stage 1 with tokens" : "fused stage 1 with tokens");
        *w.result() = scan_result_dev{0xDEADBEEFu, 0xFFFFFFFFu, ~0ull};
        launch_minify_fused(w.in, len, w.desc(), w.out.data(), w.result(), org, 6, nullptr, nullptr, true);
        check_minify(large ? "on-chip minify" : "fused minify (16 KiB tiles)", doc, e, w);
        check_workspace_clean(w, doc, large ? "on-chip minify" : "fused minify (16 KiB tiles)");
      }
      debug_fused_small_below = FUSED_SMALL_BELOW;
    }
    if ((all || what == "direct") && len >= 4096) { // round 6: the one-pass kernel for plain input -- it answers exactly, or it gives up (SJGPU_F_INTERNAL) and leaves the workspace clean
      scan_origin org = whole;
      std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu);
      *w.result() = scan_result_dev{0xDEADBEEFu, 0xFFFFFFFFu, ~0ull};
      launch_stage1_direct(w.in, len, w.desc(), w.idx.data(), len + 3, w.result(), org, 6, nullptr, nullptr, true);
      if (w.result()->flags & SJGPU_F_INTERNAL) { n_direct_gave_up++; n_checked++; }
      else { n_direct_done++; check_stage1("direct stage 1", doc, e, w); }
      check_workspace_clean(w, doc, "direct stage 1");
    }
    if ((all || what == "docs") && len <= (size_t(1) << 20)) {
      scan_result_dev r{0, 0, 0};
      std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu);
      launch_docs(0, w.in, nullptr, doc_desc{0, 0, uint32_t(len), 0}, 1, w.idx.data(), &r, nullptr);
      *w.result() = r;
      check_stage1("one workgroup per document, stage 1", doc, e, w);
      launch_docs(1, w.in, nullptr, doc_desc{0, 0, uint32_t(len), 0}, 1, w.out.data(), &r, nullptr);
      *w.result() = r;
      check_minify("one workgroup per document, minify", doc, e, w);
    }
    if ((all || what == "ranges") && len > RANGE_ALIGN) { // consecutive ranges of one buffer, the state between them as run_streamed carries it
      for (int fused = 0; fused < 2; fused++) {
        uint32_t cursor = 0, in_string = 0, x_carry = 0, flags = 0;
        std::fill(w.idx.begin(), w.idx.begin() + len + 8, 0xDEADBEEFu);
        for (size_t b = 0; b < len; b += RANGE_ALIGN) {
          const size_t end = b + RANGE_ALIGN < len ? b + RANGE_ALIGN : len;
          const bool last = end == len;
          scan_origin org{uint64_t(b), cursor, (in_string ? CARRY_IN_STRING : 0u) | (x_carry ? CARRY_X : 0u) | CARRY_SHARD | (last ? 0u : CARRY_MORE)};
          if (fused) { launch_stage1_fused(w.in, end, w.desc(), w.idx.data(), len + 3, w.result(), org, 6, nullptr, nullptr, true); }
          else { launch_stage1(w.in, end, w.masks.data(), w.summ.data(), w.pref.data(), w.idx.data(), len + 3, w.result(), org, nullptr, nullptr); }
          flags |= w.result()->flags & ~(1u | SJGPU_F_RANGE_CARRY);
          cursor = w.result()->n;
          in_string = w.result()->flags & 1u;
          x_carry = w.result()->flags & SJGPU_F_RANGE_CARRY;
        }
        w.result()->flags = flags | in_string;
        check_stage1(fused ? "ranges, single pass" : "ranges, split", doc, e, w);
      }
    }
    if (all || what == "parity") {
      launch_string_parity(w.in, len, w.result(), w.esc.data(), nullptr);
      n_checked++;
      if ((w.result()->n & 1u) != (e.flags & 1u)) { report("string parity", doc, "parity"); }
    }
    if (n_failed > 20) { break; }
  }
  if (n_direct_done + n_direct_gave_up) { printf("direct: %ld completed, %ld gave up\n", n_direct_done, n_direct_gave_up); }
  printf("%ld documents, %zu bytes: %lu comparisons with the oracle, %lu mismatches\n", docs, total_bytes, n_checked, n_failed);
  return n_failed ? 1 : 0;
}
