// tools/emu/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product (consul_amd/lib.py refuses the library built with it).
//
// A wave64 lock-step emulator of exactly the slice of HIP that consul_amd/csrc/*.hip uses, so that the UNMODIFIED kernel source can be
// compiled by g++ for the build container's host cores (which have no GPU) and run beside the checker: every lane of a workgroup is a
// fiber; a fiber runs until it meets a wave collective (__ballot / __any / __shfl* / wave_barrier), a workgroup barrier or its end;
// when every live lane of a wave is parked the lanes parked at the same call site are resolved together — which is what CDNA's
// reconvergence gives for collectives in wave-uniform control flow (the only kind the kernels use; lanes parked at DIFFERENT source
// positions are counted, EMU_TRACE=1 prints them as line:column, and the earliest position goes first).  Workgroups run one after the other, waves of a workgroup
// interleave only at barriers: one legal schedule among the many the device may take, which is enough for results that are
// schedule-independent by construction (bit-identical to the checker) and for ASan / UBSan to see every load and store a kernel makes.
// What it is NOT: a model of the memory system (no races, no LDS bank conflicts, no timing).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

// ---------------------------------------------------------------- language
#define __host__
#define __device__
#define __global__
#define __constant__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (x)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct int2 { int32_t x, y; };
struct alignas(16) int4 { int32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int32_t x, int32_t y) { return int2{x, y}; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace emu {
enum State : uint8_t { RUN, AT_COLL, AT_BAR, DONE };
enum Op : uint8_t { BALLOT, ANY, ALL, SHFL, SHFL_DOWN, SHFL_UP, SHFL_XOR, WAVE_BARRIER };
struct Lane {
  void* sp;                 // the parked fiber's stack pointer
  dim3 tid;
  uint32_t lane;            // within the wave
  State state;
  Op op;
  const void* site;
  uint64_t val; int arg;    // deposited operand
  uint64_t res;             // result handed back
  unsigned char* stack;
  uint32_t cep;             // wave collectives this lane has been released from (the race detector's lane-level epoch)
};
extern Lane* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern uint64_t n_diverged, n_shfl_inactive, n_launches, n_blocks;
alignas(64) extern unsigned char dyn_lds[160 * 1024];
// (marked convergent all the way down to the context switch — which does NOT stop the host compiler from specialising a collective's call
// per predecessor, hence the source-position call sites below)
__attribute__((convergent)) uint64_t park_collective(Op op, uint64_t val, int arg, const void* site);
__attribute__((convergent)) int park_barrier(int pred);
}  // namespace emu
#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define warpSize 64

// ---------------------------------------------------------------- wave and workgroup collectives
// A call site is named by its SOURCE position (line << 12 | column of the call in its immediate caller — for a helper such as
// wave_append_one that is the helper's own line, whatever it was inlined into), not by its return address: the host compiler is free to
// duplicate a call into both arms of a branch (it does: __ballot(want) becomes __ballot(0) and __ballot(1) in send_state), and the lanes of
// ONE source-level collective would then park at two machine addresses.
#define EMU_HERE int line_ = __builtin_LINE(), int col_ = __builtin_COLUMN()
#define EMU_SITE ((const void*)(uintptr_t)((unsigned)line_ << 12 | ((unsigned)col_ & 4095u)))
static inline uint64_t __ballot(int pred, EMU_HERE) { return emu::park_collective(emu::BALLOT, pred != 0, 0, EMU_SITE); }
static inline int __any(int pred, EMU_HERE) { return (int)emu::park_collective(emu::ANY, pred != 0, 0, EMU_SITE); }
static inline int __all(int pred, EMU_HERE) { return (int)emu::park_collective(emu::ALL, pred != 0, 0, EMU_SITE); }
template <class T> static inline T emu_shfl(emu::Op op, T v, int arg, const void* site) {
  static_assert(sizeof(T) <= 8, "shuffle operand");
  uint64_t b = 0; memcpy(&b, &v, sizeof(T));
  b = emu::park_collective(op, b, arg, site);
  T r; memcpy(&r, &b, sizeof(T)); return r;
}
template <class T> static inline T __shfl(T v, int src, int = 64, EMU_HERE) { return emu_shfl(emu::SHFL, v, src, EMU_SITE); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64, EMU_HERE) { return emu_shfl(emu::SHFL_DOWN, v, (int)d, EMU_SITE); }
template <class T> static inline T __shfl_up(T v, unsigned d, int = 64, EMU_HERE) { return emu_shfl(emu::SHFL_UP, v, (int)d, EMU_SITE); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64, EMU_HERE) { return emu_shfl(emu::SHFL_XOR, v, m, EMU_SITE); }
static inline void __builtin_amdgcn_wave_barrier_emu(EMU_HERE) { emu::park_collective(emu::WAVE_BARRIER, 0, 0, EMU_SITE); }
#define __builtin_amdgcn_wave_barrier() __builtin_amdgcn_wave_barrier_emu()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
static inline unsigned long long __builtin_amdgcn_s_memtime_emu() { return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(); }
#define __builtin_amdgcn_s_memtime() __builtin_amdgcn_s_memtime_emu()
#define __builtin_amdgcn_readfirstlane(x) (x)   /* (only ever applied to wave-uniform values: a hint that keeps them in scalar registers) */
__attribute__((convergent)) static inline void __syncthreads() { emu::park_barrier(0); }
__attribute__((convergent)) static inline int __syncthreads_or(int p) { return emu::park_barrier(p != 0); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
static inline uint32_t __lane_id() { return emu::cur->lane; }
#define __hip_atomic_load(p, order, scope) (*(volatile std::remove_reference_t<decltype(*(p))>*)(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(volatile std::remove_reference_t<decltype(*(p))>*)(p) = (v)))

// ---------------------------------------------------------------- integer / float intrinsics
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
using std::isfinite; using std::isnan; using std::isinf;
static inline unsigned long long wall_clock64() { return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() / 10; }   // 100 MHz, like the device's
static inline double __dadd_rn(double a, double b) { return a + b; }   // (build with -ffp-contract=off, as the product is)
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline double __dsqrt_rn(double a) { return sqrt(a); }
static inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
template <class A, class B, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
static inline std::common_type_t<A, B> min(A a, B b) { using T = std::common_type_t<A, B>; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
static inline std::common_type_t<A, B> max(A a, B b) { using T = std::common_type_t<A, B>; return (T)a > (T)b ? (T)a : (T)b; }

// ---------------------------------------------------------------- atomics (one host thread: plain read-modify-write; the race build marks them)
#ifdef EMU_RACE
namespace emu { void atomic_begin(); void atomic_end(); }
#define EMU_AB emu::atomic_begin();
#define EMU_AE emu::atomic_end();
#else
#define EMU_AB
#define EMU_AE
#endif
template <class T, class U> static inline T atomicAdd(T* p, U v) { EMU_AB T o = *p; *p = (T)(o + (T)v); EMU_AE return o; }
template <class T, class U> static inline T atomicSub(T* p, U v) { EMU_AB T o = *p; *p = (T)(o - (T)v); EMU_AE return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { EMU_AB T o = *p; *p = (T)(o | (T)v); EMU_AE return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { EMU_AB T o = *p; *p = (T)(o & (T)v); EMU_AE return o; }
template <class T, class U> static inline T atomicXor(T* p, U v) { EMU_AB T o = *p; *p = (T)(o ^ (T)v); EMU_AE return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { EMU_AB T o = *p; if ((T)v < o) *p = (T)v; else *p = o; EMU_AE return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { EMU_AB T o = *p; if ((T)v > o) *p = (T)v; else *p = o; EMU_AE return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { EMU_AB T o = *p; *p = (T)v; EMU_AE return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U c, V v) { EMU_AB T o = *p; if (o == (T)c) *p = (T)v; else *p = o; EMU_AE return o; }

// ---------------------------------------------------------------- runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipStreamCaptureModeThreadLocal = 1, hipIpcMemLazyEnablePeerAccess = 1, hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3 };
namespace emu {
struct Graph { std::vector<std::function<void()>> ops; };
struct Stream { Graph* capturing = nullptr; };
struct Event { std::chrono::steady_clock::time_point t; };
void launch_now(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void submit(Stream* st, std::function<void()> op);
}  // namespace emu
typedef emu::Stream* hipStream_t;
typedef emu::Event* hipEvent_t;
typedef emu::Graph* hipGraph_t;
typedef emu::Graph* hipGraphExec_t;
typedef void* hipDeviceptr_t;
struct hipIpcMemHandle_t { char reserved[64]; };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (emulated)" : "error (emulated)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) {
  void* q = nullptr; if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
  const char* poison = getenv("EMU_POISON"); memset(q, poison ? (int)strtol(poison, nullptr, 0) : 0, n);   // (device memory comes uninitialised: EMU_POISON=0xA5 shows who relies on zeros)
  *p = (T*)q; return hipSuccess;
}
template <class T> static inline hipError_t hipExtMallocWithFlags(T** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { emu::submit(st, [=] { memmove(d, s, n); }); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { emu::submit(st, [=] { memset(d, v, n); }); return hipSuccess; }
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t st) {
  emu::submit(st, [=] { uint32_t* p = (uint32_t*)d; for (size_t i = 0; i < count; i++) p[i] = (uint32_t)v; }); return hipSuccess;
}
template <class S> static inline hipError_t hipMemcpyFromSymbol(void* d, const S& sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
  memcpy(d, (const char*)&sym + off, n); return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu::Stream; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t s, int) { s->capturing = new emu::Graph; return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) { *g = s->capturing; s->capturing = nullptr; return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void*, void*, size_t) { *x = new emu::Graph(*g); return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t) { for (auto& op : g->ops) op(); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu::Event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof *h); memcpy(h->reserved, &p, sizeof p); return hipSuccess; }
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof *p); return hipSuccess; }
static inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }

template <class... P, class... A>
static inline void emu_launch(void (*k)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t st, A... a) {
  std::tuple<P...> args{static_cast<P>(a)...};
  emu::submit(st, [=] { emu::launch_now(grid, block, shmem, [&] { std::apply(k, args); }); });
}
#define hipLaunchKernelGGL(k, ...) emu_launch(k, __VA_ARGS__)
