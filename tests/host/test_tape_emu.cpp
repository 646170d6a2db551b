// tests/host/test_tape_emu.cpp -- CPU tier: the stage-2 KERNEL SOURCES (sjgpu_tape.hip, sjgpu_string_stream.hip, sjgpu_strings.hip and the scans of
// sjgpu_finish.hip, compiled as C++ against tests/host/emu: a workgroup = an OS thread, a lane = a fiber) run the three launches of
// sjgpu_stage2_device -- launch_tape_front, launch_parse_strings, launch_tape -- on whole documents and are compared with the oracle's serial walk
// (oracle/sj_oracle_stage2.c, pinned against the reference's dom::parser::parse): the error code always, every tape word and every byte of the
// string buffer when the document is valid.  tests/host/test_tape_model.cpp checks the construction; this checks the kernels as written (block and
// tile boundaries, the scans, the sort, look-backs, LDS hand-overs).  What is left to the GPU tier is what hipcc and the hardware make of them.
// Input on stdin: [u32 length][bytes] records.  Usage: test_tape_emu [max_depth] [string road: 0 = as decided, 1 = per-string kernels forced]
#include "sjgpu.h"
#include "sjgpu_internal.h"
#include "sj_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace sjgpu;

static void *aligned(std::vector<uint8_t> &store, size_t bytes, uint8_t fill) {
  store.assign(bytes + 512, fill);
  uintptr_t p = reinterpret_cast<uintptr_t>(store.data());
  p = (p + 255) & ~uintptr_t(255);
  return reinterpret_cast<void *>(p);
}

int main(int argc, char **argv) {
  const uint32_t max_depth = argc > 1 ? uint32_t(atoi(argv[1])) : 1024u;
  const int force_walk = argc > 2 ? atoi(argv[2]) : 0;
  sj_emu::max_concurrent_workgroups = 4;
  if (force_walk) { setenv("SJGPU_STRING_STREAM", "0", 1); } // launch_parse_strings reads the switch per call
  unsigned long docs = 0, valid = 0, codes[32] = {0}, roads[3] = {0, 0, 0}, reruns = 0;
  std::vector<uint8_t> doc_store, ws_store, tape_store, sbuf_store, idx_store;
  for (;;) {
    uint32_t len;
    if (fread(&len, 4, 1, stdin) != 1) { break; }
    // the document sits in a 256-byte aligned buffer; what lies behind it is NOT padding the kernels may rely on (the product's buffers hold the
    // previous document there): quotes and backslashes
    uint8_t *doc = static_cast<uint8_t *>(aligned(doc_store, size_t(len) + 4096, '"'));
    for (size_t k = len; k < size_t(len) + 4096; k += 2) { doc[k] = '\\'; }
    if (len && fread(doc, 1, len, stdin) != len) { return 2; }
    docs++;
    uint32_t *idx = static_cast<uint32_t *>(aligned(idx_store, (size_t(len) + 16) * 4, 0xEE));
    uint32_t n = 0;
    {
      std::vector<uint8_t> padded(doc, doc + len); // the oracle reads a space-padded copy
      padded.resize(size_t(len) + 64, 0x20);
      const int e1 = sjo_stage1(padded.data(), len, SJO_REGULAR, len ? len : 1, idx, &n);
      if (e1 || n == 0) { continue; }
      // ---- the list passes of sjgpu_finish.hip: the one-pass depth scan and the boundary search of the streaming modes, against a serial count --------
      {
        std::vector<uint8_t> scratch_store, depth_store;
        uint8_t *scratch = static_cast<uint8_t *>(aligned(scratch_store, depth_scan_scratch_bytes(n), 0xA5));
        int32_t *depth = static_cast<int32_t *>(aligned(depth_store, (size_t(n) + 1) * 4, 0x77));
        launch_depth_scan(doc, idx, n, depth, scratch, nullptr);
        int run = 0;
        uint32_t boundary = 0; // the last list index whose token directly follows a value (0: none)
        for (uint32_t i = 0; i <= n; i++) {
          if (depth[i] != run) { fprintf(stderr, "MISMATCH: depth[%u] = %d, a serial count says %d (document %lu, %u structurals)\n", i, depth[i], run, docs, n); return 1; }
          if (i < n) {
            const uint8_t c = doc[idx[i]];
            run += (c == '{' || c == '[') - (c == '}' || c == ']');
            if (i >= 1) {
              const uint8_t b = doc[idx[i - 1]];
              if (c != ':' && c != ',' && c != '}' && c != ']' && b != '{' && b != '[' && b != ':' && b != ',') { boundary = i; }
            }
          }
        }
        int ob = 0, ab = 0;
        for (uint32_t i = boundary; i < n; i++) {
          const uint8_t c = doc[idx[i]];
          ob += (c == '{') - (c == '}');
          ab += (c == '[') - (c == ']');
        }
        const uint32_t keep = (ob == 0 && ab == 0) ? n : boundary; // complete_prefix of stage1_finish.cpp (find_next_document_index.h:39-98)
        std::vector<uint8_t> fin_store;
        uint8_t *fin = static_cast<uint8_t *>(aligned(fin_store, finish_workspace_bytes(n), 0xA5));
        launch_finish(SJGPU_STREAMING_FINAL, doc, len, idx, n, fin, nullptr);
        const finish_state fs = *reinterpret_cast<const finish_state *>(fin);
        if (fs.keep != keep || fs.boundary_plus1 != (boundary ? boundary + 1 : 0)) {
          fprintf(stderr, "MISMATCH: the boundary search keeps %u of %u structurals (boundary + 1 = %u), a serial walk %u (boundary %u) (document %lu)\n", fs.keep, n,
                  fs.boundary_plus1, keep, boundary, docs);
          return 1;
        }
      }
      std::vector<uint64_t> want(size_t(len) + 8);
      std::vector<uint8_t> want_s(5 * (size_t(len) / 3) + 128);
      uint64_t tw = 0, sb = 0;
      const int e_want = sjo_stage2(padded.data(), len, idx, n, max_depth, want.data(), want.size(), want_s.data(), want_s.size(), &tw, &sb);

      // ---- the device road, as sjgpu_stage2_device lays it out -------------------------------------------------------------------------
      const size_t scratch_at = 256, scratch = strings_scratch_bytes(n, len), offs_at = scratch_at + scratch;
      const size_t tape_at = (offs_at + (size_t(n) + 1) * 4 + 255) & ~size_t(255);
      uint8_t *ws = static_cast<uint8_t *>(aligned(ws_store, tape_at + tape_workspace_bytes(n, len), 0xA5)); // the workspace is NOT zeroed by the caller
      const size_t tape_cap = size_t(len) + 8, str_cap = 5 * (size_t(len) / 3) + 256;
      uint64_t *tape = static_cast<uint64_t *>(aligned(tape_store, tape_cap * 8, 0x5A));
      uint8_t *sbuf = static_cast<uint8_t *>(aligned(sbuf_store, str_cap, 0x5A));
      strings_result_dev *sres = reinterpret_cast<strings_result_dev *>(ws);
      uint32_t *offsets = reinterpret_cast<uint32_t *>(ws + offs_at);
      // (sjgpu_stage2_device's optimistic loop: the stream alone and the sort in one pass first; what a document's results ask for is enqueued again)
      strings_result_dev hs;
      tape_result_dev ht;
      int string_roads = STRINGS_STREAM_ONLY;
      bool deep = false;
      for (;;) {
        const int *string_tokens = launch_tape_front(doc, len, idx, n, max_depth, ws + tape_at, nullptr);
        const strings_handoff strs = launch_parse_strings(doc, len, idx, n, false, sbuf, str_cap, offsets, sres, ws + scratch_at, nullptr, string_tokens, string_roads);
        launch_tape(doc, len, idx, n, max_depth, offsets, strs, sbuf, tape, tape_cap, ws + tape_at, nullptr, deep);
        hs = *sres;
        ht = *reinterpret_cast<const tape_result_dev *>(ws + tape_at);
        bool again = false;
        if (string_roads == STRINGS_STREAM_ONLY && hs.path == 2 && !hs.overflow) { string_roads = STRINGS_WALK_ONLY; again = true; }
        if (!deep && ht.max_level >= TAPE_ONE_PASS_LEVELS) { deep = true; again = true; }
        if (!again) { break; }
        reruns++;
      }
      roads[hs.path < 3 ? hs.path : 0]++;
      uint64_t key = ht.error_key;
      if (hs.first_bad != 0xFFFFFFFFu) {
        const uint64_t sk = (uint64_t(hs.first_bad) << 8) | (2u << 4) | 5u;
        if (sk < key) { key = sk; }
      }
      int e_got = 0;
      if (key != ~uint64_t(0)) { e_got = int(key & 0xFu); }
      else if (hs.overflow || ht.overflow) { e_got = 99; }
      codes[e_got & 31]++;
      if (e_got != e_want) {
        fprintf(stderr, "MISMATCH: error code %d, the oracle says %d (document %lu, %u bytes, %u structurals, error key %llx): %.*s\n", e_got, e_want, docs, len, n,
                (unsigned long long)key, int(len > 300 ? 300 : len), (const char *)doc);
        return 1;
      }
      // ---- the string pass on its own (sjgpu_parse_strings_device: both roads enqueued, its own count of the string tokens, offsets per structural) ----
      {
        std::vector<uint8_t> sbuf2_store, ws2_store;
        uint8_t *ws2 = static_cast<uint8_t *>(aligned(ws2_store, offs_at + (size_t(n) + 1) * 4 + 256, 0xA5));
        uint8_t *sbuf2 = static_cast<uint8_t *>(aligned(sbuf2_store, str_cap, 0x5A));
        strings_result_dev *sres2 = reinterpret_cast<strings_result_dev *>(ws2);
        uint32_t *offsets2 = reinterpret_cast<uint32_t *>(ws2 + offs_at);
        (void)launch_parse_strings(doc, len, idx, n, false, sbuf2, str_cap, offsets2, sres2, ws2 + scratch_at, nullptr);
        const strings_result_dev h2 = *sres2;
        std::vector<uint8_t> want_b(str_cap + 64);
        std::vector<uint32_t> want_off(size_t(n) + 1);
        uint64_t wb = 0;
        uint32_t wstrings = 0, wbad = 0xFFFFFFFFu;
        (void)sjo_string_buffer(padded.data(), len, idx, n, 0, want_b.data(), want_b.size(), want_off.data(), &wb, &wstrings, &wbad);
        bool same = h2.first_bad == wbad;
        if (wbad == 0xFFFFFFFFu) { // all strings valid: the whole buffer and every offset (CSR: a structural that is no string has the next record's)
          same = same && h2.bytes == wb && h2.strings == wstrings && memcmp(sbuf2, want_b.data(), wb) == 0;
          if (!same) { size_t k = 0; while (k < wb && sbuf2[k] == want_b[k]) { k++; } fprintf(stderr, "  (the buffer differs at byte %zu: %02x, the oracle %02x)\n", k, sbuf2[k], want_b[k]); }
          uint32_t next = uint32_t(wb);
          for (uint32_t i = n; same && i-- > 0;) {
            if (want_off[i] != 0xFFFFFFFFu) { next = want_off[i]; }
            same = offsets2[i] == next;
            if (!same) { fprintf(stderr, "  (offsets[%u] = %u, the oracle %u)\n", i, offsets2[i], next); }
          }
          same = same && offsets2[n] == uint32_t(wb);
        }
        if (!same) {
          fprintf(stderr, "MISMATCH: the string pass on its own (road %u): first bad %u (the oracle %u), %llu bytes (%llu), %u strings (%u) (document %lu, %u bytes): %.*s\n", h2.path,
                  h2.first_bad, wbad, (unsigned long long)h2.bytes, (unsigned long long)wb, h2.strings, wstrings, docs, len, int(len > 300 ? 300 : len), (const char *)doc);
          return 1;
        }
      }
      if (e_want == 0) {
        valid++;
        if (ht.tape_words != tw || memcmp(tape, want.data(), tw * 8) != 0) {
          fprintf(stderr, "MISMATCH: tape differs (%llu words, the oracle %llu; document %lu, %u bytes, %u structurals): %.*s\n", (unsigned long long)ht.tape_words,
                  (unsigned long long)tw, docs, len, n, int(len > 300 ? 300 : len), (const char *)doc);
          for (size_t k = 0; k < ht.tape_words && k < tw; k++) {
            if (tape[k] != want[k]) { fprintf(stderr, "  word %zu: %016llx, the oracle %016llx\n", k, (unsigned long long)tape[k], (unsigned long long)want[k]); break; }
          }
          return 1;
        }
        if (hs.bytes != sb || memcmp(sbuf, want_s.data(), sb) != 0) {
          size_t k = 0;
          while (k < sb && k < hs.bytes && sbuf[k] == want_s[k]) { k++; }
          fprintf(stderr, "MISMATCH: string buffer differs (%llu bytes, the oracle %llu, first difference at %zu, road %u; document %lu, %u bytes): %.*s\n",
                  (unsigned long long)hs.bytes, (unsigned long long)sb, k, hs.path, docs, len, int(len > 300 ? 300 : len), (const char *)doc);
          return 1;
        }
      }
    }
  }
  printf("%lu documents, %lu valid, 0 mismatches;", docs, valid);
  for (int k = 0; k < 32; k++) { if (codes[k]) { printf(" code %d: %lu", k, codes[k]); } }
  printf(" (string roads: stream %lu, per-string %lu; second rounds: %lu)\n", roads[1], roads[2], reruns);
  return 0;
}
