/* simdjson_amd/csrc/corpus.c -- deterministic synthetic JSON corpora (host-side tooling for the
 * tests and bench.py; not on the hot path).
 *
 * Shapes follow the reference's own benchmark generators (restated, not copied):
 *   large_random : /root/reference/benchmark/large_random/large_random.h:43-60
 *                  "[\n" { "x":<r>,  "y":<r>, "z":<r>} records joined by ",\n", trailer "\n]\n";
 *                  <r> = a double in [0,1) printed with 6 significant digits (operator<< default).
 *   amazon_ndjson: /root/reference/benchmark/large_amazon_cellphones/large_amazon_cellphones.h:66-81
 *                  one JSON array per line, 9 columns shaped like jsonexamples/amazon_cellphones.ndjson
 *                  (ASIN, brand, title, two URLs, rating, review URL, review count, price).
 *   twitter_like : pretty-printed nested objects with escapes and 2/3/4-byte UTF-8, in the spirit of
 *                  jsonexamples/twitter.json (which does not travel to the GPU box).
 * The RNG is our own xorshift64* so every buffer is a pure function of (kind, seed, size): units
 * (records / lines / statuses) come in blocks of UNITS_PER_BLOCK, block b drawing from a stream seeded
 * by (kind, seed, b), which lets the blocks be generated on all host cores (OpenMP) without changing
 * the bytes.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t rnd(rng_t *r) {
  uint64_t x = r->s;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  r->s = x;
  return x * 0x2545F4914F6CDD1Dull;
}
static uint32_t rnd_below(rng_t *r, uint32_t n) { return (uint32_t)((rnd(r) >> 32) * (uint64_t)n >> 32); }
static double rnd_unit(rng_t *r) { return (double)(rnd(r) >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { uint8_t *p; size_t len, cap; } out_t;
static void put(out_t *o, const char *s, size_t n) {
  if (o->len + n <= o->cap) { memcpy(o->p + o->len, s, n); }
  o->len += n;
}
static void puts_(out_t *o, const char *s) { put(o, s, strlen(s)); }
static void putc_(out_t *o, char c) { put(o, &c, 1); }

/* ---- large_random ------------------------------------------------------------------------- */
static void put_record(out_t *o, rng_t *r) {
  char tmp[96];
  int k = snprintf(tmp, sizeof tmp, "{ \"x\":%g,  \"y\":%g, \"z\":%g}", rnd_unit(r), rnd_unit(r), rnd_unit(r));
  put(o, tmp, (size_t)k);
}
static void unit_large_random(out_t *o, rng_t *r, uint64_t index) {
  if (index == 0) { put_record(o, r); puts_(o, "\n"); }
  else { puts_(o, ",\n"); put_record(o, r); }
}

/* ---- amazon-style NDJSON -------------------------------------------------------------------- */
static const char *const BRANDS[] = { "Nokia", "Motorola", "Samsung", "Apple", "Sony", "LG", "HTC", "Google",
                                      "BlackBerry", "Huawei", "Xiaomi", "OnePlus", "ASUS", "ZTE" };
static const char *const WORDS[] = { "Phone", "Unlocked", "Dual-Band", "Tri-Mode", "w/", "Voice", "Activated",
  "Dialing", "&", "Bright", "White", "Backlit", "Screen", "GSM", "4G", "LTE", "Smartphone", "64GB", "Black",
  "(Renewed)", "5.8\\\"", "Display", "AT&T", "Verizon", "T-Mobile", "Prepaid", "Carrier", "Locked", "-", "Gray",
  "International", "Version", "No", "Warranty", "Caf\xC3\xA9", "Edition", "Pro", "Max", "Mini", "\xE2\x84\xA2" };
static void put_asin(out_t *o, const char *a) { put(o, a, 10); }
static void put_line(out_t *o, rng_t *r) {
  static const char AL[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";
  char asin[11], tmp[64];
  asin[0] = 'B'; asin[1] = '0';
  for (int i = 2; i < 10; i++) { asin[i] = AL[rnd_below(r, 36)]; }
  asin[10] = 0;
  puts_(o, "[\""); put_asin(o, asin); puts_(o, "\",\"");
  puts_(o, BRANDS[rnd_below(r, sizeof BRANDS / sizeof *BRANDS)]);
  puts_(o, "\",\"");
  uint32_t nw = 3 + rnd_below(r, 14);
  for (uint32_t i = 0; i < nw; i++) {
    if (i) { putc_(o, ' '); }
    puts_(o, WORDS[rnd_below(r, sizeof WORDS / sizeof *WORDS)]);
  }
  puts_(o, "\",\"https://www.amazon.com/");
  uint32_t ns = 2 + rnd_below(r, 5);
  for (uint32_t i = 0; i < ns; i++) {
    if (i) { putc_(o, '-'); }
    const char *w = WORDS[rnd_below(r, 20)]; /* ASCII-only prefix of the word table */
    for (const char *c = w; *c; c++) { if ((*c >= 'A' && *c <= 'Z') || (*c >= 'a' && *c <= 'z') || (*c >= '0' && *c <= '9')) { putc_(o, *c); } }
  }
  puts_(o, "/dp/"); put_asin(o, asin);
  puts_(o, "\",\"https://m.media-amazon.com/images/I/");
  for (int i = 0; i < 11; i++) { putc_(o, AL[rnd_below(r, 36)]); }
  puts_(o, "._AC_UY218_SEARCH213888_FMwebp_QL75_.jpg\",");
  int k = snprintf(tmp, sizeof tmp, "%.2g", 1.0 + 4.0 * rnd_unit(r)); put(o, tmp, (size_t)k);
  puts_(o, ",\"https://www.amazon.com/product-reviews/"); put_asin(o, asin);
  k = snprintf(tmp, sizeof tmp, "\",%u,\"", rnd_below(r, 2000)); put(o, tmp, (size_t)k);
  if (rnd_below(r, 4)) { k = snprintf(tmp, sizeof tmp, "$%u.%02u", 5 + rnd_below(r, 900), rnd_below(r, 100)); put(o, tmp, (size_t)k); }
  puts_(o, "\"]\n");
}
static void unit_amazon(out_t *o, rng_t *r, uint64_t index) { (void)index; put_line(o, r); }

/* ---- twitter-like ---------------------------------------------------------------------------- */
static const char *const TEXT[] = { "RT", "@user", "hello", "world", "\\n", "\\\"quoted\\\"", "http:\\/\\/t.co\\/AbC123",
  "\xE3\x81\x93\xE3\x82\x93\xE3\x81\xAB\xE3\x81\xA1\xE3\x81\xAF", "\xE6\x97\xA5\xE6\x9C\xAC\xE8\xAA\x9E", "\xF0\x9F\x98\x80",
  "caf\xC3\xA9", "\\u3042\\u3044", "\\\\", "#tag", "lol", "the", "quick", "brown", "\xD0\xBF\xD1\x80\xD0\xB8\xD0\xB2\xD0\xB5\xD1\x82",
  "\xF0\x9F\x87\xAF\xF0\x9F\x87\xB5", "a\\tb", "...", "!!", "\xEF\xBC\x81" };
static void indent(out_t *o, int d) { for (int i = 0; i < d; i++) { puts_(o, "  "); } }
static void put_text(out_t *o, rng_t *r) {
  putc_(o, '"');
  uint32_t nw = 1 + rnd_below(r, 18);
  for (uint32_t i = 0; i < nw; i++) {
    if (i) { putc_(o, ' '); }
    puts_(o, TEXT[rnd_below(r, sizeof TEXT / sizeof *TEXT)]);
  }
  putc_(o, '"');
}
static void put_status(out_t *o, rng_t *r, int d) {
  char tmp[96];
  indent(o, d); puts_(o, "{\n");
  indent(o, d + 1); puts_(o, "\"metadata\": {\n");
  indent(o, d + 2); puts_(o, "\"result_type\": \"recent\",\n");
  indent(o, d + 2); puts_(o, "\"iso_language_code\": \"ja\"\n");
  indent(o, d + 1); puts_(o, "},\n");
  indent(o, d + 1); puts_(o, "\"created_at\": \"Sun Aug 31 00:29:15 +0000 2014\",\n");
  int k = snprintf(tmp, sizeof tmp, "\"id\": %llu,\n", (unsigned long long)(rnd(r) >> 5)); indent(o, d + 1); put(o, tmp, (size_t)k);
  indent(o, d + 1); puts_(o, "\"text\": "); put_text(o, r); puts_(o, ",\n");
  indent(o, d + 1); puts_(o, "\"truncated\": "); puts_(o, rnd_below(r, 2) ? "true" : "false"); puts_(o, ",\n");
  indent(o, d + 1); puts_(o, "\"in_reply_to_status_id\": null,\n");
  indent(o, d + 1); puts_(o, "\"user\": {\n");
  k = snprintf(tmp, sizeof tmp, "\"id\": %u,\n", (uint32_t)rnd(r)); indent(o, d + 2); put(o, tmp, (size_t)k);
  indent(o, d + 2); puts_(o, "\"name\": "); put_text(o, r); puts_(o, ",\n");
  indent(o, d + 2); puts_(o, "\"description\": "); put_text(o, r); puts_(o, ",\n");
  k = snprintf(tmp, sizeof tmp, "\"followers_count\": %u,\n", rnd_below(r, 100000)); indent(o, d + 2); put(o, tmp, (size_t)k);
  indent(o, d + 2); puts_(o, "\"entities\": { \"urls\": [ ], \"hashtags\": [");
  uint32_t nh = rnd_below(r, 4);
  for (uint32_t i = 0; i < nh; i++) {
    k = snprintf(tmp, sizeof tmp, "%s{ \"indices\": [ %u, %u ] }", i ? ", " : " ", rnd_below(r, 140), rnd_below(r, 140));
    put(o, tmp, (size_t)k);
  }
  puts_(o, " ] }\n");
  indent(o, d + 1); puts_(o, "},\n");
  k = snprintf(tmp, sizeof tmp, "\"retweet_count\": %u,\n", rnd_below(r, 5000)); indent(o, d + 1); put(o, tmp, (size_t)k);
  k = snprintf(tmp, sizeof tmp, "\"coordinates\": [ %.6f, %.6f ]\n", 180.0 * rnd_unit(r) - 90.0, 360.0 * rnd_unit(r) - 180.0);
  indent(o, d + 1); put(o, tmp, (size_t)k);
  indent(o, d); puts_(o, "}");
}
static void unit_twitter(out_t *o, rng_t *r, uint64_t index) {
  if (index) { puts_(o, ",\n"); }
  put_status(o, r, 2);
}

/* ---- block-parallel driver ----------------------------------------------------------------------------
 * text = header + unit(0) + unit(1) + ... + unit(n-1) + trailer with the smallest n >= 1 such that
 * the total is >= target. */
#define UNITS_PER_BLOCK 16384u
typedef void (*unit_fn)(out_t *, rng_t *, uint64_t);

static rng_t block_rng(uint64_t salt, uint64_t seed, uint64_t block) {
  rng_t r = { (seed + 1) * 0x9E3779B97F4A7C15ull ^ (block + 1) * 0xD1B54A32D192ED03ull ^ salt };
  if (r.s == 0) { r.s = 0x1234567ull; }
  (void)rnd(&r); (void)rnd(&r);
  return r;
}
/* generates units [first, first+count) of block `block` into o (o may be a counting-only sink) */
static void gen_block(out_t *o, unit_fn fn, uint64_t salt, uint64_t seed, uint64_t block, uint32_t count) {
  rng_t r = block_rng(salt, seed, block);
  for (uint32_t i = 0; i < count; i++) { fn(o, &r, block * UNITS_PER_BLOCK + i); }
}

static size_t generate(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t salt, const char *header,
                       const char *trailer, unit_fn fn, size_t approx_unit, uint64_t *n_units) {
  const size_t hl = strlen(header), tl = strlen(trailer);
  size_t nblk_cap = target / (approx_unit * UNITS_PER_BLOCK / 2) + 2, nblk = 0;
  size_t *blen = (size_t *)calloc(nblk_cap, sizeof(size_t));
  uint8_t **bbuf = (uint8_t **)calloc(nblk_cap, sizeof(uint8_t *));
  if (!blen || !bbuf) { free(blen); free(bbuf); return 0; }
  size_t total = hl + tl;
  int oom = 0;
  /* generate waves of blocks until header+units+trailer reaches the target */
  while (total < target && nblk < nblk_cap && !oom) {
    size_t want = (target - total) / (approx_unit * UNITS_PER_BLOCK) + 1;
    if (want > nblk_cap - nblk) { want = nblk_cap - nblk; }
    const size_t bcap = (size_t)UNITS_PER_BLOCK * approx_unit * 4 + 4096;
#pragma omp parallel for schedule(dynamic, 1)
    for (long b = (long)nblk; b < (long)(nblk + want); b++) {
      uint8_t *p = (uint8_t *)malloc(bcap);
      out_t o = { p, 0, p ? bcap : 0 };
      gen_block(&o, fn, salt, seed, (uint64_t)b, UNITS_PER_BLOCK);
      if (!p || o.len > bcap) { free(p); p = NULL; }
      bbuf[b] = p;
      blen[b] = o.len;
    }
    for (size_t b = nblk; b < nblk + want; b++) {
      if (!bbuf[b]) { oom = 1; }
      total += blen[b];
    }
    nblk += want;
  }
  size_t result = 0;
  if (!oom && total >= target && nblk > 0) {
    /* drop whole surplus blocks, then cut the last block at unit granularity */
    while (nblk > 1 && total - blen[nblk - 1] >= target) { total -= blen[--nblk]; }
    const size_t before_last = total - blen[nblk - 1];
    rng_t r = block_rng(salt, seed, nblk - 1);
    out_t cnt = { NULL, 0, 0 };
    uint32_t units_last = 0;
    while (units_last < UNITS_PER_BLOCK && (units_last == 0 || before_last + cnt.len < target)) {
      fn(&cnt, &r, (uint64_t)(nblk - 1) * UNITS_PER_BLOCK + units_last);
      units_last++;
    }
    const size_t last_len = cnt.len;
    total = before_last + last_len;
    if (total <= cap) {
      size_t *boff = (size_t *)malloc(nblk * sizeof(size_t));
      if (boff) {
        size_t o = hl;
        for (size_t b = 0; b < nblk; b++) { boff[b] = o; o += (b + 1 == nblk) ? last_len : blen[b]; }
        memcpy(dst, header, hl);
#pragma omp parallel for schedule(dynamic, 1)
        for (long b = 0; b < (long)nblk; b++) { memcpy(dst + boff[b], bbuf[b], ((size_t)b + 1 == nblk) ? last_len : blen[b]); }
        memcpy(dst + o, trailer, tl);
        result = total;
        if (n_units) { *n_units = (uint64_t)(nblk - 1) * UNITS_PER_BLOCK + units_last; }
        free(boff);
      }
    }
  }
  for (size_t b = 0; b < nblk_cap; b++) { free(bbuf[b]); }
  free(bbuf);
  free(blen);
  return result;
}

/* Each fills dst (capacity cap) with the smallest unit count whose text is >= target bytes; returns
 * the byte length, or 0 if cap is too small / out of memory. */
size_t sjc_large_random(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_records) {
  return generate(dst, cap, target, seed, 0x1234567ull, "[\n", "\n]\n", unit_large_random, 48, n_records);
}
size_t sjc_amazon_ndjson(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_lines) {
  size_t n = generate(dst, cap, target, seed, 0x7654321ull,
                      "[\"asin\",\"brand\",\"title\",\"url\",\"image\",\"rating\",\"reviewUrl\",\"totalReviews\",\"prices\"]\n", "",
                      unit_amazon, 400, n_lines);
  if (n && n_lines) { *n_lines += 1; }
  return n;
}
size_t sjc_twitter_like(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_statuses) {
  return generate(dst, cap, target, seed, 0xABCDEFull, "{\n  \"statuses\": [\n",
                  "\n  ],\n  \"search_metadata\": { \"count\": 100, \"since_id\": 0 }\n}\n", unit_twitter, 900, n_statuses);
}
