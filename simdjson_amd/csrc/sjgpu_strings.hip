// simdjson_amd/csrc/sjgpu_strings.hip -- SURVEY 8(f3): the strings of a document, unescaped on the device.
//
// The reference's stage 2 walks the structural list and, at every quote, calls stringparsing::parse_string
// (/root/reference/src/generic/stage2/stringparsing.h:150-193) to copy the string into document::string_buf as
// [u32 length][unescaped bytes][0] (tape_builder.h:415-433, visit_string :187-205).  That work is independent per token:
// the list says where every string begins, the bytes say where it ends.  Here one lane owns one structural:
//   1. k_strings<false> : the string behind every quote of the list is walked once and reports 5 + unescaped length (0 for
//                      the other structurals and for strings the reference rejects: bad escape, bad \u hex, unpaired
//                      surrogate -- the first such structural is kept with atomicMin);
//   2. exclusive scan of the sizes (the three scan kernels of sjgpu_finish.hip): offsets[i] = where structural i's record
//      begins -- for a string exactly the payload of the reference's tape entry for it -- offsets[n] = bytes used;
//   3. k_strings<true>  : the records are written.
// Output is byte for byte document::string_buf of the reference's dom parse (tests/test_gpu_parity.py::test_string_buffer_*).
// Bytes at or beyond len read as 0x20, like the padding of a padded_string.
// Since round 3 these kernels are the SECOND road: launch_parse_strings first lets sjgpu_string_stream.hip try the buffer as a stream
// compaction of the document, which takes every document whose strings are all valid and all listed; what it declines -- a string the
// reference rejects, an unclosed string, a quote glued to a scalar -- is done here, with the results these kernels always gave (the first
// offender, the records of the valid strings).  The switch is a word in device memory: the kernels of the road not taken return at once.
#include "sjgpu_device.h"

#include <cstdlib>

namespace sjgpu {
namespace {

constexpr u32 STR_THREADS = 256;
constexpr u32 NO_STRING = 0xFFFFFFFFu;

// positions are 32-bit in this file: the C API refuses inputs beyond 2.4e9 bytes (record offsets are 32 bits, too), and the
// walkers spend most of their instructions on position arithmetic
__device__ __forceinline__ u32 str_byte(const u8 *__restrict__ buf, u32 len, u32 pos) { return pos < len ? u32(buf[pos]) : 0x20u; }

// four bytes at any position of a 4-byte aligned buffer (two aligned loads + v_alignbyte); the caller guarantees pos + 8 <= len
__device__ __forceinline__ u32 load4_unaligned(const u8 *__restrict__ buf, u32 pos) {
  const u32 *w = reinterpret_cast<const u32 *>(buf + (pos & ~3u));
  return __builtin_amdgcn_alignbyte(w[1], w[0], pos & 3u);
}
typedef u32 __attribute__((aligned(1))) u32_unaligned; // gfx950 stores a dword at any byte address
// 0x80 in every byte of x that is zero; exact for the LOWEST flagged byte, which is all the callers use
__device__ __forceinline__ u32 zero_bytes(u32 x) { return (x - 0x01010101u) & ~x & 0x80808080u; }

// ---- where a string's bytes come from ------------------------------------------------------------------------------------------------
// straight from the input ...
struct global_bytes {
  const u8 *buf;
  u32 len;
  __device__ __forceinline__ bool can4(u32 pos) const { return pos + 8u <= len; }
  __device__ __forceinline__ u32 load4(u32 pos) const { return load4_unaligned(buf, pos); }
  __device__ __forceinline__ u32 byte(u32 pos) const { return str_byte(buf, len, pos); }
};
// ... or from the lane's row of LDS, where the STAGE_DWORDS dwords around a SHORT string were parked by independent loads: the walk
// over the string then costs LDS latencies instead of one global round trip per step (positions outside the row -- the look-ahead of
// a malformed escape at the end of a string -- fall back to the input)
constexpr u32 STAGE_DWORDS = 10;
struct staged_bytes {
  const u32 *row; // STAGE_DWORDS dwords holding the input bytes [lo, lo + 4 STAGE_DWORDS)
  u32 lo;
  const u8 *buf;
  u32 len;
  __device__ __forceinline__ bool can4(u32 pos) const { return pos - lo <= 4u * STAGE_DWORDS - 8u; } // wraps for pos < lo: false
  __device__ __forceinline__ u32 load4(u32 pos) const {
    const u32 off = pos - lo;
    return __builtin_amdgcn_alignbyte(row[(off >> 2) + 1], row[off >> 2], off & 3u);
  }
  __device__ __forceinline__ u32 byte(u32 pos) const {
    const u32 off = pos - lo; // wraps for pos < lo: out of range either way
    return off < 4u * STAGE_DWORDS ? (row[off >> 2] >> (8u * (off & 3u))) & 0xFFu : str_byte(buf, len, pos);
  }
};

// jsoncharutils::hex_to_u32_nocheck (/root/reference/include/simdjson/generic/jsoncharutils.h:31-38)
template <class SRC>
__device__ __forceinline__ u32 hex4(const SRC &src, u32 pos) {
  u32 v = 0;
#pragma unroll
  for (u32 k = 0; k < 4; k++) {
    const u32 c = src.byte(pos + k);
    u32 d;
    if (c - u32('0') <= 9u) { d = c - u32('0'); }
    else if ((c | 0x20u) - u32('a') <= 5u) { d = (c | 0x20u) - u32('a') + 10u; }
    else { return 0xFFFFFFFFu; }
    v = (v << 4) | d;
  }
  return v;
}
__device__ __forceinline__ u32 escape_value(u32 c) { // escape_map, stringparsing.h:22-43 (0 = not an escape); a chain of selects, not a switch (which compiles into divergent branches)
  u32 v = 0u;
  v = c == u32('"') ? 0x22u : v;
  v = c == u32('/') ? 0x2fu : v;
  v = c == u32('\\') ? 0x5cu : v;
  v = c == u32('b') ? 0x08u : v;
  v = c == u32('f') ? 0x0cu : v;
  v = c == u32('n') ? 0x0au : v;
  v = c == u32('r') ? 0x0du : v;
  v = c == u32('t') ? 0x09u : v;
  return v;
}

// parse_string for the string whose first byte sits at pos; WRITE: the unescaped bytes go to dst.  Returns the unescaped
// length or -1 (the reference's nullptr).
// `plain` bytes of v (0 ... 4), low byte first, to dst: at most three stores, no loop
__device__ __forceinline__ void store_low_bytes(u8 *__restrict__ dst, u32 v, u32 plain) {
  typedef unsigned short __attribute__((aligned(1))) u16_unaligned;
  if (plain == 4u) { *reinterpret_cast<u32_unaligned *>(dst) = v; return; }
  if (plain & 2u) { *reinterpret_cast<u16_unaligned *>(dst) = (unsigned short)v; }
  if (plain & 1u) { dst[plain & 2u] = u8(v >> (8u * (plain & 2u))); }
}

template <bool WRITE, class SRC>
__device__ __forceinline__ int unescape(const SRC &src, u32 len, u32 pos, u8 *__restrict__ dst, bool allow_replacement) {
  u32 o = 0;
  for (;;) {
    u32 c;
    if (src.can4(pos)) { // four bytes at a time while nothing special shows up
      const u32 v = src.load4(pos);
      const u32 special = zero_bytes(v ^ 0x22222222u) | zero_bytes(v ^ 0x5C5C5C5Cu);
      const u32 plain = special ? (u32(__builtin_ctz(special)) >> 3) : 4u; // bytes in front of the first quote / backslash
      if (WRITE) { store_low_bytes(dst + o, v, plain); }
      o += plain;
      pos += plain;
      if (!special) { continue; }
      c = (v >> (8u * plain)) & 0xFFu;
    } else {
      if (pos >= len) { return -1; } // no closing quote (stage 1 said UNCLOSED_STRING)
      c = src.byte(pos);
      if (c != '"' && c != '\\') {
        if (WRITE) { dst[o] = u8(c); }
        o++;
        pos++;
        continue;
      }
    }
    if (c == '"') { return int(o); }
    const u32 e = src.byte(pos + 1);
    if (e != 'u') {
      const u32 m = escape_value(e);
      if (!m) { return -1; }
      if (WRITE) { dst[o] = u8(m); }
      o++;
      pos += 2;
      continue;
    }
    // handle_unicode_codepoint (stringparsing.h:50-96)
    u32 cp = hex4(src, pos + 2);
    pos += 6;
    if (cp >= 0xd800u && cp < 0xdc00u) {
      if (src.byte(pos) != '\\' || src.byte(pos + 1) != 'u') {
        if (!allow_replacement) { return -1; }
        cp = 0xfffdu;
      } else {
        const u32 low = hex4(src, pos + 2) - 0xdc00u;
        if (low >> 10) {
          if (!allow_replacement) { return -1; }
          cp = 0xfffdu; // the second escape is not consumed: it is looked at again on its own
        } else {
          cp = (((cp - 0xd800u) << 10) | low) + 0x10000u;
          pos += 6;
        }
      }
    } else if (cp >= 0xdc00u && cp <= 0xdfffu) {
      if (!allow_replacement) { return -1; }
      cp = 0xfffdu;
    }
    // jsoncharutils::codepoint_to_utf8 (:52-80)
    if (cp <= 0x7Fu) {
      if (WRITE) { dst[o] = u8(cp); }
      o += 1;
    } else if (cp <= 0x7FFu) {
      if (WRITE) { dst[o] = u8((cp >> 6) + 192u); dst[o + 1] = u8((cp & 63u) + 128u); }
      o += 2;
    } else if (cp <= 0xFFFFu) {
      if (WRITE) { dst[o] = u8((cp >> 12) + 224u); dst[o + 1] = u8(((cp >> 6) & 63u) + 128u); dst[o + 2] = u8((cp & 63u) + 128u); }
      o += 3;
    } else if (cp <= 0x10FFFFu) {
      if (WRITE) {
        dst[o] = u8((cp >> 18) + 240u); dst[o + 1] = u8(((cp >> 12) & 63u) + 128u);
        dst[o + 2] = u8(((cp >> 6) & 63u) + 128u); dst[o + 3] = u8((cp & 63u) + 128u);
      }
      o += 4;
    } else {
      return -1; // not hex
    }
  }
}

// ---- one long string, all 64 lanes of a wave -------------------------------------------------------------------------------------
// Lanes look at 4 bytes each (256 per step).  Everything up to the first quote or backslash of the step is plain text and is
// copied by the lanes that hold it; a quote ends the string, an escape is decoded by uniform code (all lanes compute the same
// thing, lane 0 stores) and the walk resumes behind it.  Same results as unescape<>, which the short strings use.
template <bool WRITE>
__device__ __forceinline__ int unescape_wave(const u8 *__restrict__ buf, u32 len, u32 pos, u8 *__restrict__ dst, bool allow_replacement, u32 lane) {
  u32 o = 0; // wave-uniform
  for (;;) {
    const u32 mine = pos + 4u * lane;
    u32 v;
    if (pos + 264u <= len) { v = load4_unaligned(buf, mine); }
    else { v = str_byte(buf, len, mine) | (str_byte(buf, len, mine + 1) << 8) | (str_byte(buf, len, mine + 2) << 16) | (str_byte(buf, len, mine + 3) << 24); }
    const u32 special = zero_bytes(v ^ 0x22222222u) | zero_bytes(v ^ 0x5C5C5C5Cu);
    const u32 first = special ? (u32(__builtin_ctz(special)) >> 3) : 4u;
    const u64 m = __ballot(special != 0);
    if (m == 0) { // 256 plain bytes
      if (pos + 256 > len) { return -1; } // ran off the input: no closing quote
      if (WRITE) { *reinterpret_cast<u32_unaligned *>(dst + o + 4u * lane) = v; }
      o += 256;
      pos += 256;
      continue;
    }
    const u32 l0 = ctz64(m);
    const u32 k = readlane_dyn(first, l0);
    const u32 plain = 4u * l0 + k;
    if (pos + plain >= len) { return -1; }
    if (WRITE) {
      if (lane < l0) { *reinterpret_cast<u32_unaligned *>(dst + o + 4u * lane) = v; }
      if (lane == l0) {
        for (u32 b = 0; b < k; b++) { dst[o + 4u * l0 + b] = u8(v >> (8u * b)); }
      }
    }
    const u32 c = (readlane_dyn(v, l0) >> (8u * k)) & 0xFFu;
    o += plain;
    pos += plain;
    if (c == '"') { return int(o); }
    // an escape: decoded by every lane alike (uniform), stored by lane 0
    const u32 e = str_byte(buf, len, pos + 1);
    if (e != 'u') {
      const u32 mapped = escape_value(e);
      if (!mapped) { return -1; }
      if (WRITE && lane == 0) { dst[o] = u8(mapped); }
      o++;
      pos += 2;
      continue;
    }
    const global_bytes src{buf, len};
    u32 cp = hex4(src, pos + 2);
    pos += 6;
    if (cp >= 0xd800u && cp < 0xdc00u) {
      if (str_byte(buf, len, pos) != '\\' || str_byte(buf, len, pos + 1) != 'u') {
        if (!allow_replacement) { return -1; }
        cp = 0xfffdu;
      } else {
        const u32 low = hex4(src, pos + 2) - 0xdc00u;
        if (low >> 10) {
          if (!allow_replacement) { return -1; }
          cp = 0xfffdu;
        } else {
          cp = (((cp - 0xd800u) << 10) | low) + 0x10000u;
          pos += 6;
        }
      }
    } else if (cp >= 0xdc00u && cp <= 0xdfffu) {
      if (!allow_replacement) { return -1; }
      cp = 0xfffdu;
    }
    u32 bytes, packed; // UTF-8, first byte in the low bits
    if (cp <= 0x7Fu) { bytes = 1; packed = cp; }
    else if (cp <= 0x7FFu) { bytes = 2; packed = ((cp >> 6) + 192u) | (((cp & 63u) + 128u) << 8); }
    else if (cp <= 0xFFFFu) { bytes = 3; packed = ((cp >> 12) + 224u) | ((((cp >> 6) & 63u) + 128u) << 8) | (((cp & 63u) + 128u) << 16); }
    else if (cp <= 0x10FFFFu) {
      bytes = 4;
      packed = ((cp >> 18) + 240u) | ((((cp >> 12) & 63u) + 128u) << 8) | ((((cp >> 6) & 63u) + 128u) << 16) | (((cp & 63u) + 128u) << 24);
    } else {
      return -1; // not hex
    }
    if (WRITE && lane == 0) {
      for (u32 b = 0; b < bytes; b++) { dst[o + b] = u8(packed >> (8u * b)); }
    }
    o += bytes;
  }
}

// ---- the two passes ------------------------------------------------------------------------------------------------------------------
// A workgroup takes STR_TILE consecutive structurals, gathers the ones that are quotes into three LDS lists by the distance to the
// next structural -- short (<= STR_SHORT bytes), medium (<= STR_MEDIUM) and long -- and works them off with every lane busy: one
// lane per short string (from its LDS row), one lane per medium string (from the input), one wave per long string.  Only ~27 % of
// twitter-like structurals are strings, and their lengths span 0 ... 140 bytes: without the lists three lanes in four idle and
// the rest wait for the longest string of their wave.  The medium class exists because of escapes: a URL with eight "\/" costs a
// wave eight uniform iterations of ~80 instructions, a lane eight of ~30 with 63 other strings in flight beside it (counters,
// profiles/r02_pmc_strings.txt: 6 000 VALU instructions per wave with tweets and URLs on the wave path, 57 % of all issue slots).
constexpr u32 STR_TILE = 2048, STR_SHORT = 28, STR_MEDIUM = 192;

// WRITE = false: sizes[i] = 5 + unescaped length for a valid string, else 0; sizes[n] = 0 (the scan turns it into the total).
// WRITE = true : offsets[] = exclusive scan of the sizes (n + 1 entries, CSR style: offsets[i + 1] - offsets[i] = size of
//                structural i's record, 0 = it has none; offsets[n] = bytes used); the records are written.
template <bool WRITE>
__global__ __launch_bounds__(STR_THREADS) void k_strings(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, u32 n, u32 allow_replacement,
                                                       u32 *__restrict__ sizes_or_offsets, u8 *__restrict__ out, u64 out_cap, strings_result_dev *__restrict__ res,
                                                       const u32 *__restrict__ go) {
  if (*go == 0) { return; } // the stream (sjgpu_string_stream.hip) has written the buffer
  constexpr u32 PER_THREAD = STR_TILE / STR_THREADS;
  __shared__ unsigned short sh_short[STR_TILE], sh_medium[STR_TILE], sh_long[STR_TILE]; // positions inside the tile
  __shared__ u32 sh_stage[STR_THREADS][STAGE_DWORDS + 1];          // a lane's row; the odd stride keeps the rows on different banks
  __shared__ u32 sh_n_short, sh_n_medium, sh_n_long;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool allow = allow_replacement != 0;
  const u32 len32 = u32(len); // <= 2.4e9 (sjgpu_parse_strings_device)
  if (tid == 0) { sh_n_short = 0; sh_n_medium = 0; sh_n_long = 0; }
  __syncthreads();
  const u64 tile0 = u64(blockIdx.x) * STR_TILE;
  // ---- which structurals are strings: all loads of the tile's share in flight together (list, then the bytes it points at)
  u32 at[PER_THREAD], reach[PER_THREAD], first_byte[PER_THREAD];
#pragma unroll
  for (u32 k = 0; k < PER_THREAD; k++) {
    const u64 i = tile0 + k * STR_THREADS + tid;
    at[k] = 0xFFFFFFFFu;
    reach[k] = 0;
    if (i < n) {
      at[k] = idx[i];
      reach[k] = idx[i + 1] - at[k]; // the sentinel behind the list (= len) for the last one
    }
  }
#pragma unroll
  for (u32 k = 0; k < PER_THREAD; k++) { first_byte[k] = at[k] < len32 ? u32(buf[at[k]]) : 0u; }
#pragma unroll
  for (u32 k = 0; k < PER_THREAD; k++) {
    const u64 i = tile0 + k * STR_THREADS + tid;
    const bool is_string = first_byte[k] == '"';
    if (!WRITE && i <= n && !is_string) { sizes_or_offsets[i] = 0; }
    const bool is_short = is_string && reach[k] <= STR_SHORT + 1u;
    const bool is_medium = is_string && !is_short && reach[k] <= STR_MEDIUM + 1u;
    const u64 ms = __ballot(is_short), mm = __ballot(is_medium), ml = __ballot(is_string && !is_short && !is_medium);
    u32 base_s = 0, base_m = 0, base_l = 0;
    if (lane == 0) {
      if (ms) { base_s = atomicAdd(&sh_n_short, u32(popc64(ms))); }
      if (mm) { base_m = atomicAdd(&sh_n_medium, u32(popc64(mm))); }
      if (ml) { base_l = atomicAdd(&sh_n_long, u32(popc64(ml))); }
    }
    base_s = readlane(base_s, 0);
    base_m = readlane(base_m, 0);
    base_l = readlane(base_l, 0);
    const u64 below = lanemask_lt(lane);
    if (is_short) { sh_short[base_s + u32(popc64(ms & below))] = (unsigned short)(i - tile0); }
    else if (is_medium) { sh_medium[base_m + u32(popc64(mm & below))] = (unsigned short)(i - tile0); }
    else if (is_string) { sh_long[base_l + u32(popc64(ml & below))] = (unsigned short)(i - tile0); }
  }
  __syncthreads();
  const u32 n_short = sh_n_short, n_medium = sh_n_medium, n_long = sh_n_long;
  u32 valid = 0;
  // ---- one lane per short string: its bytes are parked in the lane's LDS row first (independent loads), the walk runs from there
  for (u32 j0 = 0; j0 < n_short; j0 += STR_THREADS) { // workgroup-uniform trip count
    const u32 j = j0 + tid;
    const bool mine = j < n_short;
    u64 i = 0;
    u32 first = 0, lo = 0, off = 0, size = 0;
    bool go = mine;
    if (mine) {
      i = tile0 + sh_short[j];
      first = idx[i] + 1u;
      lo = first & ~3u;
      if (WRITE) {
        off = sizes_or_offsets[i];
        size = sizes_or_offsets[i + 1] - off;
        go = size != 0;
        if (go && u64(off) + size > out_cap) { res->overflow = 1; go = false; }
      }
      if (go) {
        u32 *row = sh_stage[tid];
        if (lo + 4u * STAGE_DWORDS <= len32) {
          const u32 *w = reinterpret_cast<const u32 *>(buf + lo);
#pragma unroll
          for (u32 d = 0; d < STAGE_DWORDS; d++) { row[d] = w[d]; }
        } else { // the end of the input: byte by byte, 0x20 beyond it
          for (u32 d = 0; d < STAGE_DWORDS; d++) {
            const u32 p = lo + 4u * d;
            row[d] = str_byte(buf, len32, p) | (str_byte(buf, len32, p + 1) << 8) | (str_byte(buf, len32, p + 2) << 16) | (str_byte(buf, len32, p + 3) << 24);
          }
        }
      }
    }
    wave_lds_fence(); // a lane reads only its own row
    if (go) {
      const staged_bytes src{sh_stage[tid], lo, buf, len32};
      if (!WRITE) {
        const int l = unescape<false>(src, len32, first, nullptr, allow);
        if (l < 0) { atomicMin(&res->first_bad, u32(i)); sizes_or_offsets[i] = 0; }
        else { sizes_or_offsets[i] = 5u + u32(l); valid++; }
      } else {
        u8 *rec = out + off;
        const u32 l = size - 5u;
        *reinterpret_cast<u32_unaligned *>(rec) = l;
        (void)unescape<true>(src, len32, first, rec + 4, allow);
        rec[4 + l] = 0;
      }
    }
    wave_lds_fence();
  }
  // ---- one lane per medium string, straight from the input
  for (u32 j = tid; j < n_medium; j += STR_THREADS) {
    const u64 i = tile0 + sh_medium[j];
    const u32 first = idx[i] + 1u;
    const global_bytes src{buf, len32};
    if (!WRITE) {
      const int l = unescape<false>(src, len32, first, nullptr, allow);
      if (l < 0) { atomicMin(&res->first_bad, u32(i)); sizes_or_offsets[i] = 0; }
      else { sizes_or_offsets[i] = 5u + u32(l); valid++; }
    } else {
      const u32 off = sizes_or_offsets[i], size = sizes_or_offsets[i + 1] - off;
      if (size == 0) { continue; }
      if (u64(off) + size > out_cap) { res->overflow = 1; continue; }
      u8 *rec = out + off;
      const u32 l = size - 5u;
      *reinterpret_cast<u32_unaligned *>(rec) = l;
      (void)unescape<true>(src, len32, first, rec + 4, allow);
      rec[4 + l] = 0;
    }
  }
  // ---- one wave per long string
  for (u32 j = wave; j < n_long; j += STR_THREADS / 64) {
    const u64 i = tile0 + sh_long[j];
    const u32 first = idx[i] + 1u;
    if (!WRITE) {
      const int l = unescape_wave<false>(buf, len32, first, nullptr, allow, lane);
      if (lane == 0) {
        if (l < 0) { atomicMin(&res->first_bad, u32(i)); sizes_or_offsets[i] = 0; }
        else { sizes_or_offsets[i] = 5u + u32(l); valid++; }
      }
    } else {
      const u32 off = sizes_or_offsets[i], size = sizes_or_offsets[i + 1] - off;
      if (size == 0) { continue; } // wave-uniform
      if (u64(off) + size > out_cap) { res->overflow = 1; continue; }
      u8 *rec = out + off;
      const u32 l = size - 5u;
      (void)unescape_wave<true>(buf, len32, first, rec + 4, allow, lane);
      if (lane == 0) {
        *reinterpret_cast<u32_unaligned *>(rec) = l;
        rec[4 + l] = 0;
      }
    }
  }
  if (!WRITE) {
    const u32 total = wave_sum(valid);
    if (lane == 0 && total) { atomicAdd(&res->strings, total); }
  } else if (blockIdx.x == 0 && tid == 0) {
    res->bytes = sizes_or_offsets[n];
  }
}

// ---- On-Demand's raw key comparison for every key of the list at once ------------------------------------------------------------------
// value_iterator::find_field_raw (/root/reference/include/simdjson/generic/ondemand/value_iterator-inl.h:132, :229) walks the fields of an
// object and compares each key's RAW bytes with the wanted name: raw_json_string::unsafe_is_equal(length, target)
// (raw_json_string-inl.h:66-69) = room for the target, its bytes, a quote behind them.  Keys are independent: one lane per structural
// decides "am I a key (a string whose next structural is ':') and which of the K wanted names am I".  Names sit in one small block
// (lens, then bytes back to back) that every lane reads through the scalar / L1 path.
constexpr u32 KEY_NONE = 0xFFFFFFFFu;
__global__ __launch_bounds__(STR_THREADS) void k_match_keys(const u8 *__restrict__ buf, u64 len, const u32 *__restrict__ idx, u32 n, const u32 *__restrict__ lens,
                                                          const u8 *__restrict__ names, u32 K, u32 *__restrict__ out, u32 *__restrict__ matches) {
  const u64 i = u64(blockIdx.x) * STR_THREADS + threadIdx.x;
  u32 found = KEY_NONE;
  if (i < n) {
    const u32 at = idx[i], next = idx[i + 1]; // the sentinel behind the list bounds the last one
    if (at < len && buf[at] == '"' && i + 1 < n && buf[next] == ':' && next - at >= 2u) {
      const u32 room = next - at - 2u; // what separates this structural from the next, minus the two quotes
      u32 off = 0;
      for (u32 k = 0; k < K && found == KEY_NONE; k++) {
        const u32 m = lens[k];
        if (room >= m && u64(at) + 1 + m < len && buf[at + 1 + m] == '"') {
          u32 j = 0;
          while (j < m && buf[at + 1 + j] == names[off + j]) { j++; }
          if (j == m) { found = k; }
        }
        off += m;
      }
    }
    out[i] = found;
  }
  const u64 hit = __ballot(found != KEY_NONE);
  if (hit && (threadIdx.x & 63u) == 0) { atomicAdd(matches, u32(popc64(hit))); }
}

} // namespace

// names_block (device): [u32 lens[K]][bytes of the K names back to back]; matches (device, one u32, zeroed here)
void launch_match_keys(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, const uint8_t *names_block, uint32_t K, uint32_t *out, uint32_t *matches,
                       hipStream_t s) {
  (void)hipMemsetAsync(matches, 0, sizeof(uint32_t), s);
  if (n == 0) { return; }
  hipLaunchKernelGGL(k_match_keys, dim3(u32((u64(n) + STR_THREADS - 1) / STR_THREADS)), dim3(STR_THREADS), 0, s, buf, len, idx, n,
                     reinterpret_cast<const u32 *>(names_block), names_block + size_t(K) * sizeof(u32), K, out, matches);
}

static inline size_t up256(size_t x) { return (x + 255) & ~size_t(255); }
strings_scratch carve_strings_scratch(void *base, uint32_t n, uint64_t len) {
  uint8_t *b = static_cast<uint8_t *>(base);
  const size_t nseg = num_segments(len);
  strings_scratch w;
  size_t at = 0;
  w.ctrl = b + at; at += 256;
  w.partial = reinterpret_cast<int *>(b + at); at += up256((size_t(n) / 4096 + 72) * 4);
  // (the stream's token pass: one int per tile of 4096 structurals + one, padded to 64 ints, then 64 x 8 bytes per tile -- sjgpu_string_stream.hip)
  const size_t tok_tiles = (size_t(n) + 1 + 4095) / 4096, tok_bytes = ((tok_tiles + 1 + 63) & ~size_t(63)) * 4 + tok_tiles * 512;
  w.kord = reinterpret_cast<int *>(b + at); at += up256(tok_bytes > (size_t(n) + 2) * 4 ? tok_bytes : (size_t(n) + 2) * 4);
  w.outq = reinterpret_cast<uint32_t *>(b + at); at += up256((size_t(n) + 2) * 4);
  w.seg_summary = b + at; at += up256(nseg * STRS_SUMMARY_BYTES);
  w.seg_base = b + at; at += up256(nseg * STRS_BASE_BYTES);
  w.bytes = at;
  return w;
}
size_t strings_scratch_bytes(uint32_t n, uint64_t len) { return carve_strings_scratch(nullptr, n, len).bytes; }

// scratch: strings_scratch_bytes(n, len), 256-byte aligned; offsets: n + 1 words; everything asynchronous on `s`.
// Two roads to the same buffer: the stream compaction of sjgpu_string_stream.hip for documents whose strings are all valid and all
// listed, the per-string kernels above for the rest (the decision is taken on the device: the kernels of the road not taken return
// at once, its scan runs over zero entries).
strings_handoff launch_parse_strings(const uint8_t *buf, uint64_t len, const uint32_t *idx, uint32_t n, bool allow_replacement, uint8_t *out, uint64_t out_cap,
                                     uint32_t *offsets, strings_result_dev *res, void *scratch, hipStream_t s, const int *listed, int roads) {
  const strings_scratch w = carve_strings_scratch(scratch, n, len);
  const u32 n1 = n + 1;
  // STRINGS_WALK_ONLY is the retry behind a stream that DECLINED the document: its count / resolve / token / scan / write / finalize launches would
  // all run again only to have their verdict overwritten below (round 4 did that: a declined document paid for the stream twice -- ADVICE r4)
  if (roads != STRINGS_WALK_ONLY) { enqueue_string_stream(buf, len, idx, n, allow_replacement, out, out_cap, offsets, res, w, s, listed); }
  const u32 *ctrl = static_cast<const u32 *>(w.ctrl); // strs_ctrl: [1] = entries of this path's scan, [3] = it runs
  const char *sw = std::getenv("SJGPU_STRING_STREAM"); // A/B switch, read per call (the tests flip it)
  if (roads == STRINGS_STREAM_ONLY && !(sw && sw[0] == '0')) { return strings_handoff{w.outq, ctrl + 2}; } // a declined document comes back with path == 2
  if ((sw && sw[0] == '0') || roads == STRINGS_WALK_ONLY) { // force the per-string kernels: overwrite the verdict
    u32 *c = static_cast<u32 *>(w.ctrl);
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c), int(n1), 2, s);
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c + 2), 0, 1, s);
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c + 3), 1, 1, s);
    (void)hipMemsetAsync(res, 0, sizeof(strings_result_dev), s);
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&res->first_bad), int(NO_STRING), 1, s);
    (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&res->path), 2, 1, s);
  }
  const u32 grid = u32((u64(n1) + STR_TILE - 1) / STR_TILE);
  hipLaunchKernelGGL(k_strings<false>, dim3(grid), dim3(STR_THREADS), 0, s, buf, len, idx, n, allow_replacement ? 1u : 0u, offsets, out, out_cap, res, ctrl + 3);
  enqueue_scan(reinterpret_cast<int *>(offsets), n1, ctrl + 1, w.partial, s);
  hipLaunchKernelGGL(k_strings<true>, dim3(grid), dim3(STR_THREADS), 0, s, buf, len, idx, n, allow_replacement ? 1u : 0u, offsets, out, out_cap, res, ctrl + 3);
  return strings_handoff{w.outq, ctrl + 2};
}

} // namespace sjgpu
