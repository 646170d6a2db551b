// simdjson_amd/csrc/sjgpu_mgpu.hip -- ONE host buffer scanned by SEVERAL GPUs of the node, behind the C-ABI (C++ callers,
// the simdjson plug-in): what simdjson_amd/sharded.py does with one process per GPU and torch.distributed, done here in
// one process with one host thread per device and no collective at all.
//
// Why no RCCL on this path: the caller's document and index array live in HOST memory.  Every GPU has its own PCIe link,
// so each device uploads its own shard and downloads its own offsets straight into their final place in the caller's
// array -- G links in parallel instead of one (the host-buffer path of one GPU is PCIe-bound: 25-50 GB/s).
// The only thing that crosses between shards is ONE BIT per shard, exchanged through host memory between two phases:
//   phase A  upload shard g to device g and scan it as if it began OUTSIDE a string (sjgpu_stage1_shard_device /
//            sjgpu_minify_shard_device with carry 0).  Whether it ENDS inside one is its quote parity, whatever its true carry-in.
//   -------  in_string(g) = XOR of the parities of the shards in front (host)
//   phase B  only the shards whose true carry-in is 1 are scanned again, from HBM (a clean cut sits behind whitespace or an
//            operator, so that is the rare case: one read of every byte per device instead of the two a parity pre-pass cost)
//   -------  output offset(g) = sum of the counts in front (host)
//   phase C  stage 1: add the shard's byte offset to its offsets on the device; download into idx_out + offset(g)
// Shards are cut with sjgpu_clean_cut (the byte in front of a cut is ASCII whitespace or , : [ ] { }), so escapes, the
// previous-scalar bit and UTF-8 state are zero at every cut -- the reference's serial carries
// (/root/reference/src/generic/stage1/json_escape_scanner.h:50-71, json_string_scanner.h:62-85, json_scanner.h:128-157)
// reduce to that one bit.  The concatenation is bit-identical to the one-GPU scan (tests/test_gpu_parity.py::test_mgpu_*;
// a device may be listed more than once, which is how a one-GPU box tests the logic).
// Device-resident multi-GPU pipelines (shards that STAY in HBM) are the business of one process per GPU:
// simdjson_amd/sharded.py over torch.distributed (RCCL), bench.py --gpus N.
#include "sjgpu.h"

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

__global__ __launch_bounds__(256) void k_add_base(uint32_t *idx, uint32_t n, uint32_t base) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) { idx[i] += base; }
}

struct rendezvous { // reusable barrier for the shard threads of one call
  std::mutex m;
  std::condition_variable cv;
  unsigned waiting = 0, generation = 0, parties = 0;
  void arrive() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned gen = generation;
    if (++waiting == parties) {
      waiting = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct start_gate { // shard threads wait here until ALL of them exist (or are told that they never will)
  std::mutex m;
  std::condition_variable cv;
  int state = 0; // 0 closed, 1 go, 2 cancelled
  void open(bool go) {
    { std::lock_guard<std::mutex> lk(m); state = go ? 1 : 2; }
    cv.notify_all();
  }
  bool pass() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return state != 0; });
    return state == 1;
  }
};

struct shard_dev {
  int device = 0;
  sjgpu_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  uint8_t *d_in = nullptr;
  size_t d_in_bytes = 0;
  uint8_t *d_out = nullptr; // stage 1: u32 offsets; minify: bytes
  size_t d_out_bytes = 0;
};

int grow(void **p, size_t *have, size_t want) {
  if (*have >= want) { return 0; }
  if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
  size_t g = size_t(1) << 20;
  while (g < want) { g <<= 1; }
  if (hipMalloc(p, g) != hipSuccess) { return SJGPU_E_NOMEM; }
  *have = g;
  return 0;
}

} // namespace

struct sjgpu_mgpu {
  std::vector<shard_dev> devs;
};

namespace {

struct shard_job {
  size_t lo = 0, hi = 0;       // bytes [lo, hi) of the caller's buffer
  uint32_t parity = 0, carry = 0;
  sjgpu_scan_result res{0, 0, 0};
  uint64_t out_at = 0;         // first output unit of this shard in the caller's output
  int rc = 0;
};

// op 0 stage 1, 1 minify
int run_shards(sjgpu_mgpu *m, int op, const uint8_t *buf, size_t len, void *out_host, size_t out_cap, std::vector<shard_job> &jobs) {
  const size_t G = m->devs.size();
  jobs.assign(G, shard_job{});
  size_t at = 0;
  for (size_t g = 0; g < G; g++) { // cuts: first clean position at or after (g + 1) * len / G
    jobs[g].lo = at;
    const size_t target = (g + 1 == G) ? len : ((g + 1) * len) / G;
    size_t cut = (g + 1 == G) ? len : sjgpu_clean_cut(buf, len, target < at ? at : target);
    if (cut < at) { cut = at; }
    jobs[g].hi = cut;
    at = cut;
  }
  rendezvous meet;
  meet.parties = unsigned(G);
  start_gate gate;
  auto work = [&](size_t g) {
    shard_dev &d = m->devs[g];
    shard_job &j = jobs[g];
    const size_t n = j.hi - j.lo;
    auto fail = [&](int rc) { j.rc = rc; };
#ifndef SJGPU_SELFTEST_MGPU_NO_SETDEVICE // tests/test_mgpu_emu.py: without this line the several-device emulation must find calls made under the wrong device
    bool ok = hipSetDevice(d.device) == hipSuccess;
#else
    bool ok = true;
#endif
    if (!ok) { fail(SJGPU_E_HIP); }
    if (!gate.pass()) { return; } // a thread could not be started: nobody works, nobody waits
    auto scan = [&](int carry) {
      int rc = op == 0 ? sjgpu_stage1_shard_device(d.ctx, d.d_in, n, carry, d.d_out, n + 3, d.stream)
                       : sjgpu_minify_shard_device(d.ctx, d.d_in, n, carry, d.d_out, d.stream);
      if (!rc) { rc = sjgpu_result(d.ctx, d.stream, &j.res); }
      if (!rc && (j.res.flags & (SJGPU_F_INTERNAL | SJGPU_F_IDX_OVERFLOW))) { rc = 24; }
      return rc;
    };
    // ---- phase A: upload, scan as if the shard began outside a string
    if (ok && n) {
      int rc = grow(reinterpret_cast<void **>(&d.d_in), &d.d_in_bytes, n + 64);
      if (!rc) { rc = grow(reinterpret_cast<void **>(&d.d_out), &d.d_out_bytes, op == 0 ? (n + 16) * sizeof(uint32_t) : n + 64); }
      if (!rc && sjgpu_capacity(d.ctx) < n) { rc = sjgpu_set_capacity(d.ctx, n); }
      if (!rc && hipMemcpyAsync(d.d_in, buf + j.lo, n, hipMemcpyHostToDevice, d.stream) != hipSuccess) { rc = SJGPU_E_HIP; }
      if (!rc) { rc = scan(0); }
      if (rc) { fail(rc); ok = false; }
      j.parity = (j.res.flags & SJGPU_F_UNCLOSED_STRING) ? 1u : 0u; // begins outside, ends inside <=> an odd number of quotes
    }
    meet.arrive();
    if (g == 0) { // one thread folds the bits (the others wait at the next rendezvous)
      uint32_t s = 0;
      for (size_t k = 0; k < G; k++) {
        jobs[k].carry = s;
        s ^= jobs[k].parity;
      }
    }
    meet.arrive();
    // ---- phase B: a shard that really begins inside a string is scanned again (resident: no second upload)
    if (ok && n && j.rc == 0 && j.carry) {
      const int rc = scan(1);
      if (rc) { fail(rc); ok = false; }
    } else if (n == 0) {
      j.res.flags = j.carry ? SJGPU_F_UNCLOSED_STRING : 0u; // an empty shard passes the state through
    }
    meet.arrive();
    if (g == 0) {
      uint64_t cursor = 0;
      for (size_t k = 0; k < G; k++) {
        jobs[k].out_at = cursor;
        cursor += op == 0 ? uint64_t(jobs[k].res.n) : jobs[k].res.out_len;
      }
    }
    meet.arrive();
    // ---- phase C: into the caller's array, each over its own link
    if (ok && n && j.rc == 0) {
      const uint64_t units = op == 0 ? uint64_t(j.res.n) : j.res.out_len;
      if (j.out_at + units > out_cap) { fail(SJGPU_E_OVERFLOW); return; }
      if (units) {
        if (op == 0 && j.lo) {
          hipLaunchKernelGGL(k_add_base, dim3(unsigned((units + 255) / 256)), dim3(256), 0, d.stream, reinterpret_cast<uint32_t *>(d.d_out), uint32_t(units),
                             uint32_t(j.lo));
        }
        const size_t unit = op == 0 ? sizeof(uint32_t) : 1;
        if (hipMemcpyAsync(static_cast<uint8_t *>(out_host) + j.out_at * unit, d.d_out, units * unit, hipMemcpyDeviceToHost, d.stream) != hipSuccess ||
            hipStreamSynchronize(d.stream) != hipSuccess) {
          fail(SJGPU_E_HIP);
        }
      }
    }
  };
  std::vector<std::thread> threads;
  bool started = true;
  try {
    threads.reserve(G);
    for (size_t g = 1; g < G; g++) { threads.emplace_back(work, g); }
  } catch (...) { // out of threads: the ones that exist would wait for ever at the first rendezvous -- send them home instead
    started = false;
  }
  gate.open(started);
  if (started) { work(0); }
  for (std::thread &t : threads) { t.join(); }
  if (!started) { return SJGPU_E_NOMEM; }
  for (const shard_job &j : jobs) {
    if (j.rc) { return j.rc; }
  }
  return 0;
}

} // namespace

extern "C" {

int sjgpu_mgpu_create(const int *devices, int count, sjgpu_mgpu **out) {
  if (!out || !devices || count < 1 || count > 64) { return SJGPU_E_BADARG; }
  *out = nullptr;
  sjgpu_mgpu *m = new (std::nothrow) sjgpu_mgpu();
  if (!m) { return SJGPU_E_NOMEM; }
  m->devs.resize(size_t(count));
  for (int g = 0; g < count; g++) {
    shard_dev &d = m->devs[size_t(g)];
    d.device = devices[g];
    int rc = sjgpu_ctx_create(d.device, 0xFFFFFFFFu, &d.ctx);
    if (!rc && (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess)) { rc = SJGPU_E_HIP; }
    if (rc) {
      sjgpu_mgpu_destroy(m);
      return rc;
    }
  }
  *out = m;
  return 0;
}

void sjgpu_mgpu_destroy(sjgpu_mgpu *m) {
  if (!m) { return; }
  for (shard_dev &d : m->devs) {
    (void)hipSetDevice(d.device);
    if (d.stream) { (void)hipStreamSynchronize(d.stream); (void)hipStreamDestroy(d.stream); }
    if (d.d_in) { (void)hipFree(d.d_in); }
    if (d.d_out) { (void)hipFree(d.d_out); }
    sjgpu_ctx_destroy(d.ctx);
  }
  delete m;
}

int sjgpu_mgpu_count(const sjgpu_mgpu *m) { return m ? int(m->devs.size()) : 0; }

int sjgpu_mgpu_stage1(sjgpu_mgpu *m, const uint8_t *buf, size_t len, int mode, uint32_t *idx_out, size_t idx_words, uint32_t *n_io,
                      uint32_t *next_io) {
  if (!m || !n_io || mode < SJGPU_REGULAR || mode > SJGPU_COMMA_DELIMITED_FINAL) { return SJGPU_E_BADARG; }
  if (len > 0xFFFFFFFFull) { return 1; } // CAPACITY: offsets are 32-bit (SIMDJSON_MAXSIZE_BYTES)
  if (len == 0) { return 13; }           // EMPTY (json_structural_indexer.h:197)
  if (!buf || !idx_out) { return SJGPU_E_BADARG; }
  if (mode != SJGPU_REGULAR) {
    len = sjgpu_trim_partial_utf8(buf, len);
    if (len == 0) { return 11; }
  }
  if (idx_words < 3) { return SJGPU_E_OVERFLOW; }
  std::vector<shard_job> jobs;
  const int rc = run_shards(m, 0, buf, len, idx_out, idx_words - 3, jobs);
  if (rc) { return rc; }
  uint64_t n = 0;
  uint32_t flags = 0;
  for (const shard_job &j : jobs) {
    n += j.res.n;
    flags |= j.res.flags & ~uint32_t(SJGPU_F_UNCLOSED_STRING); // a shard that merely ENDS inside a string is no error ...
  }
  flags |= jobs.back().res.flags & SJGPU_F_UNCLOSED_STRING;     // ... unless it is the last one
  if (n + 3 > idx_words) { return SJGPU_E_OVERFLOW; }
  return sjgpu_stage1_finish_host(buf, len, mode, idx_out, uint32_t(n), flags, n_io, next_io); // precedence, sentinels, streaming cuts
}

int sjgpu_mgpu_minify(sjgpu_mgpu *m, const uint8_t *buf, size_t len, uint8_t *dst, size_t *dst_len) {
  if (!m || !dst_len) { return SJGPU_E_BADARG; }
  *dst_len = 0;
  if (len == 0) { return 0; }
  if (!buf || !dst) { return SJGPU_E_BADARG; }
  if (len / m->devs.size() > 0xFFFFFFF0ull) { return 1; }
  std::vector<shard_job> jobs;
  const int rc = run_shards(m, 1, buf, len, dst, len, jobs);
  if (rc) { return rc; }
  if (jobs.back().res.flags & SJGPU_F_UNCLOSED_STRING) { return 15; } // json_minifier.h:42-47: dst_len stays 0
  uint64_t total = 0;
  for (const shard_job &j : jobs) { total += j.res.out_len; }
  *dst_len = size_t(total);
  return 0;
}

int sjgpu_mgpu_validate_utf8(sjgpu_mgpu *m, const uint8_t *buf, size_t len, int *ok) {
  if (!m || !ok) { return SJGPU_E_BADARG; }
  *ok = 1;
  if (len == 0) { return 0; }
  if (!buf) { return SJGPU_E_BADARG; }
  const size_t G = m->devs.size();
  std::vector<size_t> cuts(G + 1, 0);
  cuts[G] = len;
  for (size_t g = 1; g < G; g++) { // in front of a character's first byte
    size_t cut = (g * len) / G;
    if (cut < cuts[g - 1]) { cut = cuts[g - 1]; }
    int back = 0;
    while (cut > cuts[g - 1] && back < 4 && (buf[cut] & 0xC0u) == 0x80u) { cut--; back++; }
    if (back == 4) { *ok = 0; return 0; }
    cuts[g] = cut;
  }
  std::vector<int> verdict(G, 1), rcs(G, 0);
  auto work = [&](size_t g) {
    const size_t n = cuts[g + 1] - cuts[g];
    if (n) { rcs[g] = sjgpu_validate_utf8(m->devs[g].ctx, buf + cuts[g], n, &verdict[g]); } // pieces / overlapped upload inside
  };
  std::vector<std::thread> threads;
  size_t launched = 1;
  try {
    threads.reserve(G);
    for (size_t g = 1; g < G; g++) { threads.emplace_back(work, g); launched = g + 1; }
  } catch (...) {
  }
  work(0);
  for (size_t g = launched; g < G; g++) { work(g); } // shards whose thread could not be started: on this one (no rendezvous here)
  for (std::thread &t : threads) { t.join(); }
  for (size_t g = 0; g < G; g++) {
    if (rcs[g]) { return rcs[g]; }
    if (!verdict[g]) { *ok = 0; }
  }
  return 0;
}

} // extern "C"
