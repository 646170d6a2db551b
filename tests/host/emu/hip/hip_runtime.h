// tests/host/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: runs the product's gfx950 kernel SOURCES on the CPU.
//
// The .hip files under simdjson_amd/csrc are compiled as plain C++ with this directory first on the include path, so
// that `#include <hip/hip_runtime.h>` finds this file instead of ROCm's.  A workgroup is one OS thread, every lane of it
// a FIBER on that thread (sj_emu.cpp); wave-level intrinsics (ballot, readlane, DPP, shuffles, mbcnt) are exchanges
// between the 64 fibers of a wave, __syncthreads() a rendezvous of the workgroup's fibers, __shared__ memory a
// thread_local array (one OS thread = one workgroup), global memory the host's.  Several workgroups run at once (one
// OS thread each), so kernels that wait for each other through memory (the look-back of sjgpu_fused.hip) run as they
// do on the device -- minus the timing.  What this checks: the LOGIC of the kernels as written (carries, summaries,
// look-back protocol, emission), document by document against the oracle, in the CPU tier.  What it cannot check: what
// hipcc makes of them (tests/test_host_logic.py has the barrier / spill checkers for that) and memory-model races.
// Lanes do NOT run in lockstep here: a lane runs until its next wave-level operation.  Code that hands data from lane
// to lane through LDS must therefore separate the write from the read by wave_lds_fence() (which it must on the
// device, too, to stop the compiler).
#ifndef SJ_EMU_HIP_RUNTIME_H
#define SJ_EMU_HIP_RUNTIME_H

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define SJ_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct int4 { int32_t x, y, z, w; };
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { return int4{x, y, z, w}; }

typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
typedef void *hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t count, hipStream_t) {
  for (size_t i = 0; i < count; i++) { static_cast<int32_t *>(p)[i] = v; }
  return hipSuccess;
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
// the memory and stream calls of the host-side code that is run here too (sjgpu_comm.hip: tests/host/test_comm_emu.cpp): device memory is
// the host's, a stream is the calling thread (everything "enqueued" has happened when the call returns), there is one device
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipHostMallocDefault = 0;
namespace sj_emu { extern size_t fail_allocations_above; } // test hook: hipMalloc of more bytes than this fails (default: never)
#ifdef SJ_EMU_DEVICES
namespace sj_emu { int &current_device(); void note_alloc(const void *p, size_t n, int device); }
#endif
static inline hipError_t hipMalloc(void **p, size_t n) {
  if (n > sj_emu::fail_allocations_above) { *p = nullptr; return hipErrorOutOfMemory; }
  *p = std::malloc(n ? n : 1);
#ifdef SJ_EMU_DEVICES
  if (*p) { sj_emu::note_alloc(*p, n ? n : 1, sj_emu::current_device()); }
#endif
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
#ifndef SJ_EMU_DEVICES
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
#endif
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
#ifndef SJ_EMU_DEVICES
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(dst, src, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
#else
// ---- SEVERAL fake devices (-DSJ_EMU_DEVICES=N; tests/host/test_mgpu_emu.cpp: sjgpu_mgpu.hip with distinct devices on a box that has none) ----
// What a one-GPU box cannot show: code that allocates, copies or launches while ANOTHER device is current.  Here every thread has a current device
// (hipSetDevice), every allocation and every stream belongs to the device that was current when it was made, and every call that names a stream or a
// device pointer checks that it is used under its own device -- a violation is counted (sj_emu::device_violations) and described on stderr, the call
// still happens.  Memory is the host's, a stream is the calling thread.
namespace sj_emu {
int &current_device();                                  // of the calling thread; starts at 0, like HIP's
void note_alloc(const void *p, size_t n, int device);
void drop_alloc(const void *p);
int device_of(const void *p);                           // -1: not device memory (host memory)
void device_violation(const char *what, int expected, int current);
extern int device_violations;
struct stream_rec { int device; };
} // namespace sj_emu
constexpr unsigned hipStreamNonBlocking = 1;
static inline hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= SJ_EMU_DEVICES) { return hipErrorUnknown; }
  sj_emu::current_device() = d;
  return hipSuccess;
}
static inline hipError_t hipGetDevice(int *d) { *d = sj_emu::current_device(); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = SJ_EMU_DEVICES; return hipSuccess; }
static inline hipError_t hipFree(void *p) {
  if (p) {
    const int d = sj_emu::device_of(p);
    if (d != sj_emu::current_device()) { sj_emu::device_violation("hipFree of another device's memory", d, sj_emu::current_device()); }
    sj_emu::drop_alloc(p);
  }
  std::free(p);
  return hipSuccess;
}
static inline void sj_emu_check_stream(hipStream_t s, const char *what) {
  const int d = s ? static_cast<sj_emu::stream_rec *>(s)->device : 0; // the null stream belongs to device 0's thread in these tests
  if (d != sj_emu::current_device()) { sj_emu::device_violation(what, d, sj_emu::current_device()); }
}
static inline void sj_emu_check_ptr(const void *p, const char *what) {
  const int d = sj_emu::device_of(p);
  if (d >= 0 && d != sj_emu::current_device()) { sj_emu::device_violation(what, d, sj_emu::current_device()); }
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new sj_emu::stream_rec{sj_emu::current_device()}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { sj_emu_check_stream(s, "hipStreamDestroy under another device"); delete static_cast<sj_emu::stream_rec *>(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { sj_emu_check_stream(s, "hipStreamSynchronize under another device"); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  sj_emu_check_stream(s, "hipMemcpyAsync on another device's stream");
  sj_emu_check_ptr(dst, "hipMemcpyAsync into another device's memory");
  sj_emu_check_ptr(src, "hipMemcpyAsync from another device's memory");
  if (kind == hipMemcpyHostToDevice && (sj_emu::device_of(dst) < 0 || sj_emu::device_of(src) >= 0)) { sj_emu::device_violation("hipMemcpyHostToDevice: wrong kinds of memory", -1, -1); }
  if (kind == hipMemcpyDeviceToHost && (sj_emu::device_of(src) < 0 || sj_emu::device_of(dst) >= 0)) { sj_emu::device_violation("hipMemcpyDeviceToHost: wrong kinds of memory", -1, -1); }
  std::memcpy(dst, src, n);
  return hipSuccess;
}
#endif

namespace sj_emu {
struct fiber_ids { dim3 tid; unsigned lane, wave; };
const fiber_ids &ids();    // of the running fiber
const dim3 &block_idx();   // of the running workgroup
const dim3 &grid_dim();
// every live lane of the wave deposits `mine`; returns when all have, with everybody's values (0 for lanes that have left)
void exchange(uint64_t mine, uint64_t (&all)[64]);
void wave_sync();
void block_sync();
void yield_all(); // let the other fibers of this workgroup and the other workgroups run
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
extern unsigned max_concurrent_workgroups; // OS threads per launch (default 4)
} // namespace sj_emu

#define threadIdx (::sj_emu::ids().tid)
#define blockIdx (::sj_emu::block_idx())
#define gridDim (::sj_emu::grid_dim())
#ifndef SJ_EMU_DEVICES
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) ::sj_emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })
#else
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) (sj_emu_check_stream((stream), "kernel launch on another device's stream"), ::sj_emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); }))
#endif

using std::max;
using std::min;

// ---- per-lane bit operations -------------------------------------------------------------------------
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz(unsigned(x)) : 32; }
static inline unsigned long long __brevll(unsigned long long x) {
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) { r |= ((x >> i) & 1ull) << (63 - i); }
  return r;
}
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return unsigned(((uint64_t(hi) << 32) | lo) >> (8u * (sh & 3u))); }
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {
  const uint64_t v = (uint64_t(hi) << 32) | lo;
  unsigned r = 0;
  for (int j = 0; j < 4; j++) {
    const unsigned s = (sel >> (8 * j)) & 0xFFu;
    unsigned b = 0;
    if (s < 8) { b = unsigned(v >> (8 * s)) & 0xFFu; }
    else if (s == 0x0C) { b = 0; }
    else if (s >= 0x0D) { b = 0xFFu; }
    else { b = (unsigned(v >> (8 * (2 * (s - 8) + 1) + 7)) & 1u) ? 0xFFu : 0u; } // 8..11: sign of byte 1/3/5/7
    r |= b << (8 * j);
  }
  return r;
}

// ---- wave-level operations ---------------------------------------------------------------------------
static inline uint64_t __ballot(int pred) {
  uint64_t all[64];
  sj_emu::exchange(pred ? 1u : 0u, all);
  uint64_t m = 0;
  for (int i = 0; i < 64; i++) { m |= (all[i] & 1u) << i; }
  return m;
}
static inline int __builtin_amdgcn_readlane(int v, int l) {
  uint64_t all[64];
  sj_emu::exchange(uint32_t(v), all);
  return int(uint32_t(all[l & 63]));
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return __builtin_amdgcn_readlane(v, 0); } // callers are wave-uniform, lane 0 is live
template <class T> static inline T __shfl_up(T v, unsigned d) {
  static_assert(sizeof(T) <= 8, "");
  uint64_t all[64], mine = 0;
  std::memcpy(&mine, &v, sizeof(T));
  sj_emu::exchange(mine, all);
  const unsigned lane = sj_emu::ids().lane;
  const uint64_t r = lane >= d ? all[lane - d] : mine;
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> static inline T __shfl_xor(T v, int m) {
  uint64_t all[64], mine = 0;
  std::memcpy(&mine, &v, sizeof(T));
  sj_emu::exchange(mine, all);
  const uint64_t r = all[(sj_emu::ids().lane ^ unsigned(m)) & 63u];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> static inline T __shfl(T v, int l) {
  uint64_t all[64], mine = 0;
  std::memcpy(&mine, &v, sizeof(T));
  sj_emu::exchange(mine, all);
  const uint64_t r = all[unsigned(l) & 63u];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
// v_mbcnt_lo/hi_u32_b32: add + number of set bits of mask below this lane (low / high half of the wave)
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
  const unsigned lane = sj_emu::ids().lane;
  const unsigned lt = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
  return add + unsigned(__builtin_popcount(mask & lt));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
  const unsigned lane = sj_emu::ids().lane;
  const unsigned lt = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
  return add + unsigned(__builtin_popcount(mask & lt));
}
// DPP (the controls the kernels use): row_shr:n = 0x110 + n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143; a lane whose
// row is not in row_mask, or whose source does not exist (bound_ctrl = false), keeps `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  uint64_t all[64];
  sj_emu::exchange(uint32_t(src), all);
  const unsigned lane = sj_emu::ids().lane, row = lane >> 4, in_row = lane & 15u;
  if (!((unsigned(row_mask) >> row) & 1u) || !((unsigned(bank_mask) >> (in_row >> 2)) & 1u)) { return old; }
  if (ctrl >= 0x111 && ctrl <= 0x11F) {
    const unsigned n = unsigned(ctrl) & 15u;
    if (in_row >= n) { return int(uint32_t(all[lane - n])); }
    return bound_ctrl ? 0 : old;
  }
  if (ctrl == 0x138) { return lane ? int(uint32_t(all[lane - 1])) : (bound_ctrl ? 0 : old); } // wave_shr:1
  if (ctrl == 0x142) { return row ? int(uint32_t(all[row * 16 - 1])) : old; }
  if (ctrl == 0x143) { return row >= 2 ? int(uint32_t(all[31])) : old; }
  __builtin_trap(); // a control this emulation does not know
}
static inline void __builtin_amdgcn_wave_barrier() { sj_emu::wave_sync(); }
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) { sj_emu::yield_all(); }
static inline void __syncthreads() { sj_emu::block_sync(); }
namespace sj_emu { int block_or(int pred); }
static inline int __syncthreads_or(int pred) { return sj_emu::block_or(pred); }
// 100 MHz on the device; here wall time at 1 MHz, so that the kernels' one-second give-up limits become 100 s
static inline uint64_t wall_clock64() {
  return uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}

// ---- atomics -------------------------------------------------------------------------------------------
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, T(v), __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, T(v), __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicXor(T *p, U v) { return __atomic_fetch_xor(p, T(v), __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicMin(T *p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (T(v) < old && !__atomic_compare_exchange_n(p, &old, T(v), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <class T, class U> static inline T atomicMax(T *p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (T(v) > old && !__atomic_compare_exchange_n(p, &old, T(v), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}

#endif
