// tests/host/test_comm_emu.cpp -- CPU tier: sjgpu_comm_gather_indices' OWN code (simdjson_amd/csrc/sjgpu_comm.hip compiled as C++ against
// tests/host/emu) with a world of two and three.  The ranks are threads of this process, RCCL is tests/stubs/rccl_loopback.cpp opened
// through SJGPU_RCCL_LIB exactly as librccl is on a GPU box, "device" memory is host memory, k_widen_all runs under the emulator.
// What this exercises that a world of one cannot: the exact-count ncclSend / ncclRecv group, ranks without offsets, the root on every
// rank, the growth round of the root's staging array (and the steady state after it), a root that cannot allocate (NOMEM there,
// SJGPU_E_PEER elsewhere, nothing in flight), a root whose output array is too small (refused AFTER the exchange has drained), and
// k_widen_all with several bases beyond 2^32 (global position = base of the sending rank + offset:
// /root/reference/include/simdjson/dom/document_stream-inl.h:250).
// Usage: test_comm_emu <world> [seed]
#include "sjgpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <thread>
#include <vector>

static uint64_t rng_state = 1;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return uint32_t(rng_state >> 33);
}

struct rank_io {
  std::vector<uint32_t> idx; // shard-relative ascending offsets
  uint64_t base = 0;
  std::vector<uint64_t> out; // root only
  size_t out_cap = 0;
  int rc = 0;
  uint64_t total = 0;
  std::vector<uint64_t> counts;
  int ranks_seen = 0;
};

static int failures = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      std::fprintf(stderr, "MISMATCH %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                   \
      std::fprintf(stderr, "\n");                          \
      failures++;                                          \
    }                                                      \
  } while (0)

// one collective call on every rank; comms[r] belongs to thread r for the duration
static void collective(std::vector<sjgpu_comm *> &comms, std::vector<rank_io> &io, int root) {
  const int world = int(comms.size());
  std::vector<std::thread> ts;
  for (int r = 0; r < world; r++) {
    ts.emplace_back([&, r] {
      rank_io &me = io[size_t(r)];
      me.counts.assign(size_t(world), 0);
      me.rc = sjgpu_comm_gather_indices(comms[size_t(r)], me.idx.empty() ? nullptr : me.idx.data(), uint32_t(me.idx.size()), me.base, root,
                                        r == root ? me.out.data() : nullptr, r == root ? me.out_cap : 0, &me.total, me.counts.data(), nullptr);
      me.ranks_seen = sjgpu_comm_ranks(comms[size_t(r)]);
    });
  }
  for (auto &t : ts) { t.join(); }
}

static void fill_shards(std::vector<rank_io> &io, uint32_t scale, int empty_rank) {
  uint64_t base = 0;
  for (size_t r = 0; r < io.size(); r++) {
    rank_io &me = io[r];
    me.idx.clear();
    const uint32_t n = int(r) == empty_rank ? 0u : (scale / 2 + rnd() % scale);
    uint32_t at = 0;
    for (uint32_t i = 0; i < n; i++) {
      at += 1 + rnd() % 29;
      me.idx.push_back(at);
    }
    me.base = base;
    base += (5ull << 30) + rnd(); // shards of more than 4 GiB: the bases leave 32 bits at the second rank
  }
}

static void check_gather(const std::vector<rank_io> &io, int root, const char *what) {
  uint64_t total = 0;
  for (const rank_io &r : io) { total += r.idx.size(); }
  for (size_t r = 0; r < io.size(); r++) {
    CHECK(io[r].rc == 0, "%s: rank %zu returned %d", what, r, io[r].rc);
    CHECK(io[r].total == total, "%s: rank %zu saw total %llu, not %llu", what, r, (unsigned long long)io[r].total, (unsigned long long)total);
    CHECK(io[r].ranks_seen == int(io.size()), "%s: rank %zu: the communicator spans %d ranks", what, r, io[r].ranks_seen);
    for (size_t k = 0; k < io.size(); k++) { CHECK(io[r].counts[k] == io[k].idx.size(), "%s: rank %zu has count[%zu] = %llu", what, r, k, (unsigned long long)io[r].counts[k]); }
  }
  const rank_io &rt = io[size_t(root)];
  size_t at = 0;
  for (size_t r = 0; r < io.size(); r++) {
    for (uint32_t v : io[r].idx) {
      if (at < rt.out.size() && rt.out[at] != io[r].base + v) {
        CHECK(false, "%s: global position %zu is %llu, not %llu (rank %zu)", what, at, (unsigned long long)rt.out[at], (unsigned long long)(io[r].base + v), r);
        return;
      }
      at++;
    }
  }
}

int main(int argc, char **argv) {
  const int world = argc > 1 ? std::atoi(argv[1]) : 2;
  rng_state = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
  if (!std::getenv("SJGPU_RCCL_LIB")) {
    std::fprintf(stderr, "SJGPU_RCCL_LIB must name the loop-back library\n");
    return 2;
  }
  unsigned char id[128];
  if (sjgpu_comm_unique_id(id, sizeof id) != 0) {
    std::fprintf(stderr, "sjgpu_comm_unique_id failed: %s\n", sjgpu_comm_last_error(nullptr));
    return 2;
  }
  std::vector<sjgpu_comm *> comms(size_t(world), nullptr);
  {
    std::vector<std::thread> ts; // collective: every rank joins
    std::vector<int> rcs(size_t(world), 0);
    for (int r = 0; r < world; r++) { ts.emplace_back([&, r] { rcs[size_t(r)] = sjgpu_comm_create(r, world, id, sizeof id, 0, &comms[size_t(r)]); }); }
    for (auto &t : ts) { t.join(); }
    for (int r = 0; r < world; r++) {
      if (rcs[size_t(r)] != 0 || !comms[size_t(r)]) {
        std::fprintf(stderr, "sjgpu_comm_create(rank %d) = %d\n", r, rcs[size_t(r)]);
        return 2;
      }
    }
  }
  std::vector<rank_io> io{size_t(world)};
  int calls = 0;
  // 1. every rank as the root; growth round on the first call of a root, steady state on the second, growth again when the job grows
  for (int root = 0; root < world; root++) {
    for (uint32_t scale : {2000u, 2000u, 9000u}) {
      fill_shards(io, scale, -1);
      uint64_t total = 0;
      for (rank_io &r : io) { total += r.idx.size(); }
      io[size_t(root)].out.assign(size_t(total) + 5, ~0ull);
      io[size_t(root)].out_cap = io[size_t(root)].out.size();
      collective(comms, io, root);
      calls++;
      check_gather(io, root, "gather");
    }
  }
  // 2. a rank without a single offset (its send is not posted, its receive not expected), as a sender and as the root
  for (int empty = 0; empty < world; empty++) {
    const int root = (empty + 1) % world;
    for (int rt : {root, empty}) {
      fill_shards(io, 700, empty);
      uint64_t total = 0;
      for (rank_io &r : io) { total += r.idx.size(); }
      io[size_t(rt)].out.assign(size_t(total) + 1, ~0ull);
      io[size_t(rt)].out_cap = io[size_t(rt)].out.size();
      collective(comms, io, rt);
      calls++;
      check_gather(io, rt, "gather with an empty shard");
    }
  }
  // 3. nobody has anything
  for (rank_io &r : io) { r.idx.clear(); }
  io[0].out.assign(4, ~0ull);
  io[0].out_cap = 4;
  collective(comms, io, 0);
  calls++;
  check_gather(io, 0, "empty gather");
  // 4. the root's output array is too small: refused AFTER the exchange has drained, the senders are not left waiting and return 0
  fill_shards(io, 500, -1);
  io[0].out.assign(8, ~0ull);
  io[0].out_cap = 8;
  collective(comms, io, 0);
  calls++;
  CHECK(io[0].rc == SJGPU_E_OVERFLOW, "a root with a small array returned %d", io[0].rc);
  for (int r = 1; r < world; r++) { CHECK(io[size_t(r)].rc == 0, "sender %d of a refused gather returned %d", r, io[size_t(r)].rc); }
  // 5. a root that cannot allocate its staging array: NOMEM there, PEER elsewhere, nothing in flight -- and the communicator works afterwards
  fill_shards(io, 60000, -1); // more than the staging array has ever held: the growth round runs
  {
    uint64_t total = 0;
    for (rank_io &r : io) { total += r.idx.size(); }
    const int root = world - 1;
    io[size_t(root)].out.assign(size_t(total), ~0ull);
    io[size_t(root)].out_cap = io[size_t(root)].out.size();
    setenv("SJGPU_DEBUG_COMM_FAIL_STAGING", "1", 1);
    collective(comms, io, root);
    calls++;
    unsetenv("SJGPU_DEBUG_COMM_FAIL_STAGING");
    for (int r = 0; r < world; r++) {
      CHECK(io[size_t(r)].rc == (r == root ? SJGPU_E_NOMEM : SJGPU_E_PEER), "rank %d of a gather whose root cannot allocate returned %d", r, io[size_t(r)].rc);
    }
    collective(comms, io, root);
    calls++;
    check_gather(io, root, "gather after a failed one");
  }
  // what the stub saw: one transfer per non-empty sender and successful call -- the sends really went through ncclSend / ncclRecv
  uint64_t sends = 0, collectives = 0;
  if (void *h = dlopen(std::getenv("SJGPU_RCCL_LIB"), RTLD_NOW | RTLD_NOLOAD)) {
    typedef void (*stats_fn)(uint64_t *, uint64_t *);
    if (stats_fn f = reinterpret_cast<stats_fn>(dlsym(h, "sjgpu_loopback_stats"))) { f(&sends, &collectives); }
  }
  CHECK((sends > 0 || world == 1) && collectives >= uint64_t(calls), "the loop-back library saw %llu transfers and %llu collectives in %d calls", (unsigned long long)sends,
        (unsigned long long)collectives, calls);
  for (sjgpu_comm *c : comms) { sjgpu_comm_destroy(c); }
  std::printf("world %d: %d collective calls, %llu transfers through ncclSend / ncclRecv, %llu all-gathers, %d mismatches\n", world, calls, (unsigned long long)sends,
              (unsigned long long)collectives, failures);
  return failures ? 1 : 0;
}
