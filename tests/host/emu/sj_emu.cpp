// tests/host/emu/sj_emu.cpp -- TEST INFRASTRUCTURE: the scheduler behind tests/host/emu/hip/hip_runtime.h.
// One OS thread per workgroup in flight, one fiber per lane; fibers switch only at wave / workgroup operations
// (cooperative, round robin), so a launch is deterministic per workgroup and needs no locks inside one.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sys/mman.h>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "the fiber switch below is written for x86-64 (the CPU tier's machines)"
#endif

extern "C" void sj_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl sj_emu_switch
.type sj_emu_switch,@function
sj_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size sj_emu_switch,.-sj_emu_switch
)");

namespace sj_emu {

unsigned max_concurrent_workgroups = 4;
size_t fail_allocations_above = ~size_t(0);

namespace {

constexpr size_t STACK_BYTES = size_t(512) << 10;
constexpr unsigned MAX_THREADS = 1024;

struct wave_state {
  uint64_t slot[2][64];
  unsigned arrived = 0, live = 0;
  uint64_t completed = 0; // wave-level operations finished so far
  uint64_t exited = 0;    // lanes that have left the kernel: they read as zero in every later operation
  // the operation `completed` has everybody's deposit: lanes that have left contribute zero (their slots are only touched
  // here, when no live lane can still be reading the operation two back, which used the same slots)
  void complete() {
    const unsigned p = unsigned(completed & 1u);
    for (uint64_t m = exited; m; m &= m - 1) { slot[p][__builtin_ctzll(m)] = 0; }
    arrived = 0;
    completed++;
  }
};
struct fiber {
  fiber_ids id;
  void *sp = nullptr;
  bool done = true;
  uint64_t ops = 0; // wave-level operations this lane has entered
};
struct workgroup {
  dim3 bid, grid;
  unsigned nthreads = 0, live = 0;
  unsigned bar_arrived = 0;
  uint64_t bar_completed = 0;
  int bar_or[2] = {0, 0}; // __syncthreads_or: the predicates of the barrier in flight, by its parity
  std::vector<fiber> fibers;
  std::vector<wave_state> waves;
  std::vector<char *> stacks; // kept between workgroups of this OS thread
  unsigned cur = 0;
  void *sched_sp = nullptr;
  const std::function<void()> *body = nullptr;
  ~workgroup() {
    for (char *p : stacks) { munmap(p, STACK_BYTES); }
  }
};
thread_local workgroup *wg = nullptr;

void leave_lane();
void trampoline() {
  (*wg->body)();
  leave_lane();
  for (;;) { yield_all(); } // never resumed once every lane has left
}

// switch to the next live fiber (round robin); returns when this fiber is resumed.  With no other live fiber: returns at once.
void switch_away() {
  workgroup &g = *wg;
  const unsigned me = g.cur;
  for (unsigned k = 1; k <= g.nthreads; k++) {
    const unsigned t = (me + k) % g.nthreads;
    if (t == me) { break; }
    if (!g.fibers[t].done) {
      g.cur = t;
      sj_emu_switch(&g.fibers[me].sp, g.fibers[t].sp);
      return;
    }
  }
}

void leave_lane() {
  workgroup &g = *wg;
  fiber &f = g.fibers[g.cur];
  wave_state &w = g.waves[f.id.wave];
  f.done = true;
  w.exited |= 1ull << f.id.lane;
  w.live--;
  g.live--;
  if (w.live && w.arrived == w.live) { w.complete(); } // the others were waiting for this lane only
  if (g.live && g.bar_arrived == g.live) {
    g.bar_arrived = 0;
    g.bar_completed++;
  }
  if (g.live == 0) { // back to the OS thread's own stack
    void *dummy;
    sj_emu_switch(&dummy, g.sched_sp);
  }
}

void run_workgroup(workgroup &g, unsigned b, dim3 grid, dim3 block, const std::function<void()> &body) {
  g.bid = dim3(b);
  g.grid = grid;
  g.nthreads = block.x;
  if (g.nthreads == 0 || g.nthreads > MAX_THREADS) // (a last wave with fewer than 64 lanes: the missing lanes read as zero, like lanes that have left)
    { std::fprintf(stderr, "sj_emu: block size %u\n", g.nthreads); std::abort(); }
  g.live = g.nthreads;
  g.bar_arrived = 0;
  g.bar_completed = 0;
  g.body = &body;
  g.fibers.assign(g.nthreads, fiber());
  g.waves.assign((g.nthreads + 63) / 64, wave_state());
  while (g.stacks.size() < g.nthreads) {
    void *p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::perror("sj_emu: mmap"); std::abort(); }
    g.stacks.push_back(static_cast<char *>(p));
  }
  for (unsigned t = 0; t < g.nthreads; t++) {
    fiber &f = g.fibers[t];
    f.id.tid = dim3(t);
    f.id.lane = t & 63u;
    f.id.wave = t >> 6;
    f.done = false;
    f.ops = 0;
    g.waves[t >> 6].live++;
    // [r15 r14 r13 r12 rbx rbp][return address = trampoline][pad]: rsp is 8 modulo 16 when the trampoline is entered
    uint64_t *top = reinterpret_cast<uint64_t *>(g.stacks[t] + STACK_BYTES);
    top[-1] = 0;
    top[-2] = reinterpret_cast<uint64_t>(&trampoline);
    for (int k = 3; k <= 8; k++) { top[-k] = 0; }
    f.sp = top - 8;
  }
  for (wave_state &w : g.waves) { std::memset(w.slot, 0, sizeof w.slot); }
  g.cur = 0;
  wg = &g;
  sj_emu_switch(&g.sched_sp, g.fibers[0].sp);
  wg = nullptr;
}

} // namespace

const fiber_ids &ids() { return wg->fibers[wg->cur].id; }
const dim3 &block_idx() { return wg->bid; }
const dim3 &grid_dim() { return wg->grid; }

void exchange(uint64_t mine, uint64_t (&all)[64]) {
  workgroup &g = *wg;
  fiber &f = g.fibers[g.cur];
  wave_state &w = g.waves[f.id.wave];
  if (f.ops != w.completed) {
    std::fprintf(stderr, "sj_emu: lane %u of wave %u enters wave operation %llu while the wave is at %llu -- a wave-level operation inside divergent control flow\n",
                 f.id.lane, f.id.wave, (unsigned long long)f.ops, (unsigned long long)w.completed);
    std::abort();
  }
  const unsigned p = unsigned(f.ops & 1u);
  const uint64_t op = f.ops++;
  w.slot[p][f.id.lane] = mine;
  if (++w.arrived == w.live) { w.complete(); }
  while (w.completed <= op) { switch_away(); }
  std::memcpy(all, w.slot[p], sizeof all);
}
void wave_sync() {
  uint64_t all[64];
  exchange(0, all);
}
void block_sync() {
  workgroup &g = *wg;
  const uint64_t gen = g.bar_completed;
  if (++g.bar_arrived == g.live) {
    g.bar_arrived = 0;
    g.bar_completed++;
  }
  while (g.bar_completed == gen) { switch_away(); }
}
int block_or(int pred) { // __syncthreads_or: a barrier that also ORs a predicate over the workgroup
  workgroup &g = *wg;
  const uint64_t gen = g.bar_completed;
  const unsigned p = unsigned(gen & 1u);
  if (g.bar_arrived == 0) { g.bar_or[p] = 0; } // the first to arrive clears the slot (the barrier two back has been read by everybody)
  if (pred) { g.bar_or[p] = 1; }
  if (++g.bar_arrived == g.live) {
    g.bar_arrived = 0;
    g.bar_completed++;
  }
  while (g.bar_completed == gen) { switch_away(); }
  return g.bar_or[p];
}
void yield_all() {
  switch_away();
  std::this_thread::yield();
}

// A launch is handed to a pool of OS threads that lives as long as the process (one workgroup in flight per thread).
namespace {
struct pool_t {
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> threads;
  uint64_t job_id = 0;
  unsigned want = 0, running = 0; // threads that should take part in the current job / have not finished it yet
  std::atomic<unsigned> next{0};
  unsigned nblocks = 0;
  dim3 grid, block;
  const std::function<void()> *body = nullptr;
  bool stop = false;

  void work(unsigned index) {
    workgroup g; // keeps its stacks for the life of the thread
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || (job_id != seen && index < want); });
        if (stop) { return; }
        seen = job_id;
      }
      for (;;) {
        const unsigned b = next.fetch_add(1);
        if (b >= nblocks) { break; }
        run_workgroup(g, b, grid, block, *body);
      }
      std::lock_guard<std::mutex> lk(m);
      if (--running == 0) { cv_done.notify_all(); }
    }
  }
  void run(dim3 grid_, dim3 block_, const std::function<void()> &body_, unsigned nthreads) {
    std::unique_lock<std::mutex> lk(m);
    while (threads.size() < nthreads) {
      const unsigned index = unsigned(threads.size());
      threads.emplace_back([this, index] { work(index); });
    }
    grid = grid_;
    block = block_;
    body = &body_;
    nblocks = grid_.x;
    next.store(0);
    want = running = nthreads;
    job_id++;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return running == 0; });
    want = 0;
  }
  ~pool_t() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      cv_job.notify_all();
    }
    for (std::thread &t : threads) { t.join(); }
  }
};
pool_t pool;
} // namespace

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  if (grid.x == 0) { return; }
  // one launch at a time: the pool is one set of workgroup threads (host threads that launch concurrently -- sjgpu_mgpu.hip's shard threads under
  // tests/host/test_mgpu_emu.cpp -- take turns, as kernels of different streams may on a device)
  static std::mutex one_launch;
  std::lock_guard<std::mutex> lk(one_launch);
  pool.run(grid, block, body, std::min(grid.x, std::max(1u, max_concurrent_workgroups)));
}

} // namespace sj_emu
