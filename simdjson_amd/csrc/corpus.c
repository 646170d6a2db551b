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
 * The RNG is our own xorshift64* so every buffer is a pure function of (kind, seed, size).
 */
#include <stdint.h>
#include <stdio.h>
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
/* Fills dst with the smallest record count whose text is >= target bytes (and <= cap). Returns
 * the byte length, or 0 if cap is too small. */
size_t sjc_large_random(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_records) {
  rng_t r = { seed * 0x9E3779B97F4A7C15ull + 0x1234567ull };
  out_t o = { dst, 0, cap };
  uint64_t n = 0;
  puts_(&o, "[\n");
  put_record(&o, &r); n++;
  puts_(&o, "\n");
  while (o.len + 3 < target) {
    puts_(&o, ",\n");
    put_record(&o, &r); n++;
  }
  puts_(&o, "\n]\n");
  if (n_records) { *n_records = n; }
  return o.len <= cap ? o.len : 0;
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
size_t sjc_amazon_ndjson(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_lines) {
  rng_t r = { seed * 0x9E3779B97F4A7C15ull + 0x7654321ull };
  out_t o = { dst, 0, cap };
  uint64_t n = 1;
  puts_(&o, "[\"asin\",\"brand\",\"title\",\"url\",\"image\",\"rating\",\"reviewUrl\",\"totalReviews\",\"prices\"]\n");
  while (o.len < target) { put_line(&o, &r); n++; }
  if (n_lines) { *n_lines = n; }
  return o.len <= cap ? o.len : 0;
}

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
size_t sjc_twitter_like(uint8_t *dst, size_t cap, size_t target, uint64_t seed, uint64_t *n_statuses) {
  rng_t r = { seed * 0x9E3779B97F4A7C15ull + 0xABCDEFull };
  out_t o = { dst, 0, cap };
  uint64_t n = 0;
  puts_(&o, "{\n  \"statuses\": [\n");
  do {
    if (n) { puts_(&o, ",\n"); }
    put_status(&o, &r, 2); n++;
  } while (o.len + 64 < target);
  puts_(&o, "\n  ],\n  \"search_metadata\": { \"count\": 100, \"since_id\": 0 }\n}\n");
  if (n_statuses) { *n_statuses = n; }
  return o.len <= cap ? o.len : 0;
}
