// compat.h -- one source tree, two builds.
//
//   default            : hipcc --offload-arch=gfx950  -> libocnhip.so (THE product; needs an MI355X)
//   -DOCN_HOST_EMU     : g++ -x c++                   -> tests/hostemu/libocnhip_hostemu.so
//
// The host-emulation build exists only so that the *same kernel source* can be executed on the
// build container (which has no GPU) by `pytest -m "not gpu"` to catch indexing / arithmetic bugs
// before GPU minutes are spent.  It is test infrastructure: the Python package never loads it on
// its own (it loads libocnhip.so or raises), and nothing in bench.py / smoke() may use it.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef OCN_HOST_EMU
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#define OCN_DEVFN __device__ __forceinline__

// g_ocn_dry: a time step is being REPLAYED from a hipGraph (api.hip): the host-side bookkeeping of the step runs again
// (clock, buffer rotations), every launch and stream operation is skipped
extern thread_local int g_ocn_dry;

// A failed launch (bad configuration, LDS over the limit, a lost device) is reported by the NEXT runtime call, not by the
// launch itself, and used to surface only at ocn_sync -- as a step that "succeeded" and an error that named no kernel.
// Every launch and every unchecked asynchronous call now asks hipGetLastError() right away (host-side, no synchronisation)
// and the first failure since the last report is kept with the text of the call; the C-ABI entry points that launch work
// (ocn_time_step, ocn_update_state, ...) return OCN_EHIP with that text (api.hip api_ret).
struct ocn_launch_error {
  hipError_t err;
  char what[200];
};
extern thread_local ocn_launch_error g_ocn_launch_err;
static inline void ocn_note_error(hipError_t e, const char* what) {
  if (e == hipSuccess || g_ocn_launch_err.err != hipSuccess) return;
  g_ocn_launch_err.err = e;
  snprintf(g_ocn_launch_err.what, sizeof g_ocn_launch_err.what, "%s", what);
}
template <class K, class... A>
static inline void ocn_launch_impl(const char* what, K kern, dim3 grid, dim3 block, hipStream_t s, A... args) {
  if (g_ocn_dry) return;
  hipLaunchKernelGGL(kern, grid, block, 0, s, args...);
  ocn_note_error(hipGetLastError(), what);
}
// the whole argument list is the message: "k_tend4<...>, f.grd, f.blk, s, m->gd, a"
#define ocn_launch(...) ocn_launch_impl(#__VA_ARGS__, __VA_ARGS__)
// kernels that use __shared__ / __syncthreads go through the same call on the GPU
#define ocn_launch_sync(...) ocn_launch_impl(#__VA_ARGS__, __VA_ARGS__)
// an asynchronous runtime call whose result the caller has no way to return
#define OCN_ASYNC(call) ocn_note_error((call), #call)
#define OCN_SHARED __shared__
#define OCN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// an opaque use of three loaded values: the compiler must have them in registers at this point of every path (it places
// the s_waitcnt of their loads here instead of wherever their registers are next written)
#define OCN_TOUCH3(a, b, c) asm volatile("" ::"v"(a), "v"(b), "v"(c))

// ---- wave-level primitives of the tiled kernels ----------------------------------------------------------------------
// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no VGPR round trip): lane l of the
// wave lands at dst_wave_base + 16 l; EXEC-masked lanes neither load nor write (tools/micro_checks.hip).
typedef __attribute__((address_space(3))) void ocn_lds_void;
typedef const __attribute__((address_space(1))) void ocn_glb_void;
__device__ __forceinline__ void ocn_glds16(const void* src_lane, void* dst_wave_base, int /*lane*/) {
  __builtin_amdgcn_global_load_lds((ocn_glb_void*)src_lane, (ocn_lds_void*)dst_wave_base, 16, 0, 0);
}
// value held by the next lane of the wave (lane + 1); lane 63 receives 0.  v_mov_b32_dpp wave_shl:1, two per double.
__device__ __forceinline__ double ocn_shfl_next(double x) {
  union { double d; int i[2]; } a, b;
  a.d = x;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130, 0xf, 0xf, false);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xf, 0xf, false);
  return b.d;
}
// value held by lane (lane ^ 16) of the wave (ds_bpermute_b32: LDS crossbar, no LDS storage, no barrier)
__device__ __forceinline__ double ocn_shfl_xor16(double x) {
  union { double d; int i[2]; } a, b;
  a.d = x;
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int addr = (lane ^ 16) << 2;
  b.i[0] = __builtin_amdgcn_ds_bpermute(addr, a.i[0]);
  b.i[1] = __builtin_amdgcn_ds_bpermute(addr, a.i[1]);
  return b.d;
}
#define OCN_WAVE 64
// a value that is the same in every lane of the wave, moved to a scalar register: branches on it are scalar branches
// (s_setprio and the LDS-DMA base in M0 are scalar state -- behind a "divergent" branch they would execute regardless of EXEC)
#define OCN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#else
// ------------------------------------------------------------------------------------------------
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
#define g_ocn_dry 0   /* no graphs in the host emulation */
extern std::recursive_mutex g_emu_launch_mutex;   // one emulated kernel at a time (static "LDS" arrays are shared)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define OCN_DEVFN inline
#define OCN_SHARED static
#define OCN_SCHED_FENCE() ((void)0)
#define OCN_ASYNC(call) ((void)(call))
#define OCN_TOUCH3(a, b, c) ((void)0)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
typedef struct emu_stream* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event* hipEvent_t;
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

static inline const char* hipGetErrorString(hipError_t) { return "host-emu error"; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }   // every emulated stream is synchronous
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return 0;
}

// barrier-free kernels: plain nested loops
template <class K, class... A>
static inline void ocn_launch(K kern, dim3 grid, dim3 block, hipStream_t, A... args) {
  std::lock_guard<std::recursive_mutex> lk(g_emu_launch_mutex);
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx) {
              threadIdx = dim3(tx, ty, tz);
              kern(args...);
            }
      }
}

// kernels with __syncthreads: one OS thread per GPU thread of a block, blocks run one at a time
struct emu_barrier {
  std::mutex m;
  std::condition_variable cv;
  unsigned count = 0, n = 0, gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    unsigned g = gen;
    if (++count == n) { count = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return g != gen; });
  }
};
extern emu_barrier g_emu_barrier;
static inline void __syncthreads() { g_emu_barrier.wait(); }

// wave-level primitives, emulated with one OS thread per GPU thread (every thread of the block must make the call)
#define OCN_WAVE 64
#define OCN_UNIFORM(x) (x)
static inline void ocn_glds16(const void* src_lane, void* dst_wave_base, int lane) {
  memcpy((char*)dst_wave_base + 16 * lane, src_lane, 16);
}
extern double g_emu_shfl[4096];
static inline double ocn_shfl_xor16(double x) {
  const unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  g_emu_shfl[tid] = x;
  g_emu_barrier.wait();
  const double y = g_emu_shfl[tid ^ 16];
  g_emu_barrier.wait();
  return y;
}
static inline double ocn_shfl_next(double x) {
  const unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const unsigned n = blockDim.x * blockDim.y * blockDim.z;
  g_emu_shfl[tid] = x;
  g_emu_barrier.wait();
  const double y = ((tid & 63) != 63 && tid + 1 < n) ? g_emu_shfl[tid + 1] : 0.0;
  g_emu_barrier.wait();
  return y;
}

template <class K, class... A>
static inline void ocn_launch_sync(K kern, dim3 grid, dim3 block, hipStream_t, A... args) {
  std::lock_guard<std::recursive_mutex> lk(g_emu_launch_mutex);
  unsigned nt = block.x * block.y * block.z;
  g_emu_barrier.n = nt;
  g_emu_barrier.count = 0;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        std::vector<std::thread> th;
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
          th.emplace_back([=]() {
            gridDim = grid;
            blockDim = block;
            blockIdx = dim3(bx, by, bz);
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            kern(args...);
          });
        for (auto& x : th) x.join();
      }
}
#endif

#define OCN_LDS_BYTES 163840   /* LDS per workgroup on gfx950 (160 KiB) */

#define OCN_HIP_CHECK(ctx, call)                                                            \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      ocn_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return OCN_EHIP;                                                                      \
    }                                                                                       \
  } while (0)
