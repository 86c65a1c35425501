// HIP runtime shim for the product build (gfx950).  Everything the engine needs from the runtime goes
// through these few names so that kernel *bodies* (plain per-thread functions, MP_HD) stay free of
// runtime calls.  tools/hostemu/ shadows this header with a CPU loop runner for kernel debugging on a
// machine without a GPU; that emulator is a development aid, is never built into libmpshuffle.so and is
// never loaded by the package.
#ifndef MP_RT_HPP
#define MP_RT_HPP
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#define MP_HD __host__ __device__ __forceinline__
#define MP_HD_NOINLINE __host__ __device__ __noinline__
#define MP_GLOBAL __global__
#define MP_RT_NAME "hip-gfx950"

namespace mp {
namespace rt {

typedef hipStream_t Stream;
typedef hipEvent_t Event;

inline void check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + what + ": " + hipGetErrorString(e));
}
inline int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
inline void set_device(int d) { check(hipSetDevice(d), "hipSetDevice"); }
inline void* dmalloc(size_t bytes) {
  void* p = nullptr;
  check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc");
  return p;
}
inline void dfree(void* p) {
  if (p) (void)hipFree(p);
}
inline void h2d(void* d, const void* h, size_t n, Stream s) { check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "h2d"); }
inline void d2h(void* h, const void* d, size_t n, Stream s) { check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "d2h"); }
inline void d2d(void* d, const void* s_, size_t n, Stream s) { check(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s), "d2d"); }
inline void dzero(void* d, size_t n, Stream s) { check(hipMemsetAsync(d, 0, n, s), "memset"); }
inline Stream stream_create() {
  Stream s;
  check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream create");
  return s;
}
inline void stream_destroy(Stream s) { (void)hipStreamDestroy(s); }
inline void stream_sync(Stream s) { check(hipStreamSynchronize(s), "stream sync"); }
inline Event event_create() {
  Event e;
  check(hipEventCreate(&e), "event create");
  return e;
}
inline void event_destroy(Event e) { (void)hipEventDestroy(e); }
inline void event_record(Event e, Stream s) { check(hipEventRecord(e, s), "event record"); }
inline float event_ms(Event a, Event b) {
  check(hipEventSynchronize(b), "event sync");
  float ms = 0;
  check(hipEventElapsedTime(&ms, a, b), "event elapsed");
  return ms;
}
inline void launch_check(const char* name) { check(hipGetLastError(), name); }
inline void clear_error() { (void)hipGetLastError(); }      // after a failed allocation that is going to be retried smaller
inline void stream_wait(Stream s, Event e) { check(hipStreamWaitEvent(s, e, 0), "stream wait event"); }
inline void event_sync(Event e) { check(hipEventSynchronize(e), "event sync"); }
// page-locked host memory: DMA at PCIe speed and truly asynchronous copies (pageable buffers go through the runtime's
// bounce buffers at ~9 GB/s and block the calling thread)
inline void mem_info(size_t* free_b, size_t* total_b) { check(hipMemGetInfo(free_b, total_b), "hipMemGetInfo"); }
inline uint32_t cu_count() {
  int dev = 0, n = 0;
  check(hipGetDevice(&dev), "hipGetDevice");
  check(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute(CUs)");
  return n > 0 ? (uint32_t)n : 1u;
}
inline void* host_alloc(size_t bytes) {
  void* p = nullptr;
  check(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
  return p;
}
inline void host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
// a page-locked host word the DEVICE can write directly (zero-copy): a kernel leaves a flag there and the host reads it after an
// event, without a copy call in between.  *dev_ptr = the address kernels use
inline void* host_alloc_mapped(size_t bytes, void** dev_ptr) {
  void* p = nullptr;
  check(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(mapped)");
  check(hipHostGetDevicePointer(dev_ptr, p, 0), "hipHostGetDevicePointer");
  return p;
}

}  // namespace rt
}  // namespace mp
// A kernel is a body `void body(const Args&, uint32_t x, uint32_t y)` run once per (x, y) of a
// (nx, ny) index space; x is the fast (lane) axis.  256-thread workgroups = 4 wave64.
#define MP_KERNEL(NAME, ARGS, BODY)                                               \
  template <class C>                                                              \
  MP_GLOBAL void __launch_bounds__(256) NAME(ARGS a, uint32_t nx) {               \
    uint32_t x = blockIdx.x * 256u + threadIdx.x;                                 \
    if (x < nx) BODY<C>(a, x, blockIdx.y);                                        \
  }

// same, with a register budget for >= WAVES waves per SIMD (the group-arithmetic kernels: 4 waves = 128 VGPRs)
#define MP_KERNEL_OCC(NAME, ARGS, BODY, WAVES)                                    \
  template <class C>                                                              \
  MP_GLOBAL void __launch_bounds__(256, WAVES) NAME(ARGS a, uint32_t nx) {        \
    uint32_t x = blockIdx.x * 256u + threadIdx.x;                                 \
    if (x < nx) BODY<C>(a, x, blockIdx.y);                                        \
  }

// ---- wave-cooperative kernels: one 64-lane wave = one work item (kernels_bucket.hpp) ---------------------------------------
// The body is written against a tiny execution interface so that the SAME source runs on the GPU (each lane executes the
// lambda once, lanes talk through LDS and cross-lane shuffles) and in the development emulator (tools/hostemu: the lambda
// runs for lane = 0..63 in turn between two sync points):
//   wv.lanes([&](uint32_t lane) { ... })    per-lane code; state that lives across sections is a PerLane<T> indexed by lane
//   wv.sync()                               LDS written before is visible to every lane of the wave afterwards
//   wv.lds                                  this wave's slice of LDS (32-bit words)
//   wv.atomic_add(p, v)                     LDS atomic, returns the old value
//   wv.excl_scan(x) / wv.max(x)             wave-level exclusive prefix sum / maximum over a PerLane<uint32_t>
//   wv.quad_rot<K>(x) / wv.quad_bcast<K>(x, out)   exchanges inside aligned groups of four lanes;  wv.sync_global(): as sync(), for HBM
namespace mp {
template <class T>
struct PerLane {
  T v;
  __device__ __forceinline__ T& operator[](uint32_t) { return v; }
  __device__ __forceinline__ const T& operator[](uint32_t) const { return v; }
};
struct WaveCtx {
  static constexpr uint32_t NL = 64, LOG_NL = 6;      // lanes of one work item
  template <class T>
  using PL = PerLane<T>;
  uint32_t lane;
  uint32_t* lds;
  template <class Fn>
  __device__ __forceinline__ void lanes(Fn f) {
    f(lane);
  }
  __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ uint32_t atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
  // exclusive prefix sum across the 64 lanes (Hillis-Steele over DPP / ds_bpermute shuffles)
  __device__ __forceinline__ void excl_scan(PerLane<uint32_t>& x) {
    uint32_t incl = x.v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t up = __shfl_up(incl, s, 64);
      if (lane >= (uint32_t)s) incl += up;
    }
    x.v = incl - x.v;
  }
  __device__ __forceinline__ uint32_t max(const PerLane<uint32_t>& x) {
    uint32_t m = x.v;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const uint32_t o = __shfl_xor(m, s, 64);
      m = o > m ? o : m;
    }
    return m;
  }
  // does any lane of the wave hold a non-zero flag?  (wavefront ballot)
  __device__ __forceinline__ bool any(const PerLane<uint32_t>& x) { return __ballot(x.v != 0) != 0; }
  // (every DPP move below ends in an empty asm that makes its result opaque: LLVM's DPP combiner (GCNDPPCombine, ROCm 7.2) folds such a
  // move into a following v_subrev_u32 as `v_subrev_u32_dpp`, and the folded form returned lane-dependent differences on gfx950 --
  // tools/quadcheck; with the move kept as an instruction of its own the results are right)
  static __device__ __forceinline__ uint32_t dpp_keep(uint32_t x) {
    asm("" : "+v"(x));
    return x;
  }
  // QUADS = aligned groups of four lanes (the transcript kernels put one BLAKE2s state on a quad, hash.hpp).  One DPP move each:
  // quad_rot<K>: lane j takes the value of lane (j + K) mod 4 of its quad; quad_bcast<K>: every lane takes lane K's
  template <int K>
  __device__ __forceinline__ void quad_rot(PerLane<uint32_t>& x) {
    constexpr int ctrl = ((0 + K) & 3) | (((1 + K) & 3) << 2) | (((2 + K) & 3) << 4) | (((3 + K) & 3) << 6);
    x.v = dpp_keep((uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.v, ctrl, 0xF, 0xF, true));
  }
  template <int K>
  __device__ __forceinline__ void quad_bcast(const PerLane<uint32_t>& x, PerLane<uint32_t>& out) {
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    out.v = dpp_keep((uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.v, ctrl, 0xF, 0xF, true));
  }
  // lane K's copy of a per-lane value (any struct of 32-bit words), read from inside a lanes() section -- only of values written
  // in an EARLIER section
  template <int K, class T>
  __device__ __forceinline__ T quad_read(const PerLane<T>& x, uint32_t) const {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    T r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&x.v);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) d[i] = dpp_keep((uint32_t)__builtin_amdgcn_update_dpp(0, (int)s[i], ctrl, 0xF, 0xF, true));
    return r;
  }
  // the next work item of a persistent wave: one atomic on a global counter, the same value in every lane
  __device__ __forceinline__ uint32_t next_item(uint32_t* counter) {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  }
  // the XCD this wave runs on (0 .. 7; each XCD has an L2 of its own): hwreg(HW_REG_XCC_ID, 0, 4)
  __device__ __forceinline__ uint32_t xcd() const { return (uint32_t)__builtin_amdgcn_s_getreg(6164) & 7u; }
  // Asynchronous gather into LDS: N words (a multiple of 4) per lane from global memory straight into this wave's staging area, no
  // registers in between (global_load_lds_dwordx4: the 16-byte chunk c of lane l lands at area[c * 256 + 4 l]); take() reads a lane's
  // words back.  What was read from the area before must have arrived (LGKM) before the next stage() may overwrite it.
  template <int N>
  __device__ __forceinline__ void stage(uint32_t* area, const uint32_t* g, uint32_t) {
    static_assert(N % 4 == 0, "whole 16-byte chunks");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the previous take() has its data
#pragma unroll
    for (int c = 0; c < N / 4; ++c)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 4 * c),
                                       (__attribute__((address_space(3))) void*)(area + c * 256), 16, 0, 0);
  }
  template <int N>
  __device__ __forceinline__ void take(const uint32_t* area, uint32_t* w, uint32_t) const {
#pragma unroll
    for (int c = 0; c < N / 4; ++c) {
      const uint4 t = *reinterpret_cast<const uint4*>(area + c * 256 + 4 * lane);
      w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w;
    }
  }
  // global-memory words written by any lane of the wave before are visible to every lane of the wave afterwards
  __device__ __forceinline__ void sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};
// ---- workgroup-cooperative kernels: one 256-lane workgroup (4 waves) = one work item (kernels_bucket.hpp, windows of 12 bits and more:
// 2^(c-1) buckets dealt to 256 lanes instead of 64).  The same interface as WaveCtx with NL = 256: sync() is the workgroup barrier, the
// scans and maxima go wave-wide through DPP and across the four waves through a few words of LDS (`scratch`, in front of `lds`).
// Every sync / scan / max / next_item must be reached by all four waves (workgroup-uniform control flow).
struct BlockCtx {
  static constexpr uint32_t NL = 256, LOG_NL = 8;
  static constexpr uint32_t SCRATCH_WORDS = 16;
  template <class T>
  using PL = PerLane<T>;
  uint32_t lane;
  uint32_t* lds;
  uint32_t* scratch;
  template <class Fn>
  __device__ __forceinline__ void lanes(Fn f) {
    f(lane);
  }
  __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ void sync_global() { sync(); }
  __device__ __forceinline__ uint32_t atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
  __device__ __forceinline__ void excl_scan(PerLane<uint32_t>& x) {
    uint32_t incl = x.v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t up = __shfl_up(incl, s, 64);
      if ((lane & 63u) >= (uint32_t)s) incl += up;
    }
    if ((lane & 63u) == 63u) scratch[lane >> 6] = incl;
    sync();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (lane >> 6); ++w) base += scratch[w];
    sync();
    x.v = base + incl - x.v;
  }
  __device__ __forceinline__ uint32_t max(const PerLane<uint32_t>& x) {
    uint32_t m = x.v;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const uint32_t o = __shfl_xor(m, s, 64);
      m = o > m ? o : m;
    }
    if ((lane & 63u) == 0u) scratch[4 + (lane >> 6)] = m;
    sync();
    m = scratch[4];
#pragma unroll
    for (uint32_t w = 1; w < 4; ++w) m = scratch[4 + w] > m ? scratch[4 + w] : m;
    sync();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
  }
  __device__ __forceinline__ uint32_t next_item(uint32_t* counter) {
    if (lane == 0) scratch[8] = atomicAdd(counter, 1u);
    sync();
    const uint32_t v = scratch[8];
    sync();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  }
  __device__ __forceinline__ uint32_t xcd() const { return (uint32_t)__builtin_amdgcn_s_getreg(6164) & 7u; }
  // as WaveCtx::stage / take, with one staging area of N x 64 words per wave of the workgroup
  template <int N>
  __device__ __forceinline__ void stage(uint32_t* area, const uint32_t* g, uint32_t) {
    static_assert(N % 4 == 0, "whole 16-byte chunks");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint32_t* mine = area + (lane >> 6) * (N * 64);
#pragma unroll
    for (int c = 0; c < N / 4; ++c)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 4 * c),
                                       (__attribute__((address_space(3))) void*)(mine + c * 256), 16, 0, 0);
  }
  template <int N>
  __device__ __forceinline__ void take(const uint32_t* area, uint32_t* w, uint32_t) const {
    const uint32_t* mine = area + (lane >> 6) * (N * 64);
#pragma unroll
    for (int c = 0; c < N / 4; ++c) {
      const uint4 t = *reinterpret_cast<const uint4*>(mine + c * 256 + 4 * (lane & 63u));
      w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w;
    }
  }
};
}  // namespace mp
// one 256-lane workgroup per work item, `lds_words` words of dynamic LDS for the body (+ BlockCtx::SCRATCH_WORDS for the context)
#define MP_BLOCK_KERNEL_OCC(NAME, ARGS, BODY, WAVES)                                           \
  template <class C>                                                                           \
  MP_GLOBAL void __launch_bounds__(256, WAVES) NAME(ARGS a, uint32_t nblocks, uint32_t lds_words) { \
    extern __shared__ uint32_t mp_dyn_lds[];                                                   \
    mp::BlockCtx wv{threadIdx.x, mp_dyn_lds + mp::BlockCtx::SCRATCH_WORDS, mp_dyn_lds};        \
    BODY<C>(a, blockIdx.x, wv);                                                                \
  }
#define MP_BLOCK_LAUNCH(NAME, C, stream, nblocks, lds_words, args)                                                            \
  do {                                                                                                                        \
    if ((nblocks) > 0) {                                                                                                      \
      const size_t bytes_ = ((size_t)(lds_words) + mp::BlockCtx::SCRATCH_WORDS) * 4, cap_ = mp::rt::lds_per_workgroup();      \
      if (bytes_ > cap_)                                                                                                      \
        throw std::runtime_error(#NAME ": one work item needs " + std::to_string(bytes_) + " bytes of LDS, the device offers " + \
                                 std::to_string(cap_) + " per workgroup (narrower windows)");                                 \
      if (bytes_ > 64u * 1024u)                                                                                               \
        mp::rt::check(hipFuncSetAttribute(reinterpret_cast<const void*>(&NAME<C>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)bytes_), "LDS size attribute");                                                \
      hipLaunchKernelGGL((NAME<C>), dim3((nblocks)), dim3(256), bytes_, (stream), (args), (uint32_t)(nblocks), (uint32_t)(lds_words)); \
      mp::rt::launch_check(#NAME);                                                                                            \
    }                                                                                                                         \
  } while (0)
// a workgroup holds up to 4 waves (= 4 independent work items); `lds_words` 32-bit words of dynamic LDS per wave
#define MP_WAVE_KERNEL(NAME, ARGS, BODY) MP_WAVE_KERNEL_OCC(NAME, ARGS, BODY, 2)
#define MP_WAVE_KERNEL_OCC(NAME, ARGS, BODY, WAVES)                                            \
  template <class C>                                                                           \
  MP_GLOBAL void __launch_bounds__(256, WAVES) NAME(ARGS a, uint32_t nwaves, uint32_t lds_words) { \
    extern __shared__ uint32_t mp_dyn_lds[];                                                   \
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                  \
    if (wid >= nwaves) return;                                                                 \
    mp::WaveCtx wv{threadIdx.x & 63u, mp_dyn_lds + (size_t)(threadIdx.x >> 6) * lds_words};    \
    BODY<C>(a, wid, wv);                                                                       \
  }
#define MP_WAVE_KERNEL_INST(X, NAME, ARGS, C) X MP_GLOBAL void NAME<C>(ARGS, uint32_t, uint32_t);
// LDS a workgroup may use on the current device (gfx950: 160 KB per CU; queried, so that a part with less fails with a message)
namespace mp {
namespace rt {
inline size_t lds_per_workgroup() {
  static const size_t v = [] {      // (initialised once, whichever host thread comes first)
    int dev = 0, b = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || b <= 0)
      b = 64 * 1024;
    return (size_t)b;
  }();
  return v;
}
}  // namespace rt
}  // namespace mp
// waves per workgroup: as many (<= 4) as fit twice into the CU's LDS
namespace mp {
namespace rt {
inline uint32_t waves_per_block(size_t lds_words) {
  const size_t bytes = lds_words * 4, cap = lds_per_workgroup();
  uint32_t wpb = 4;
  while (wpb > 1 && wpb * bytes > cap / 2) wpb >>= 1;
  return wpb;
}
}  // namespace rt
}  // namespace mp
#define MP_WAVE_LAUNCH(NAME, C, stream, nwaves, lds_words, args)                                                              \
  do {                                                                                                                        \
    if ((nwaves) > 0) {                                                                                                       \
      const size_t bytes_ = (size_t)(lds_words) * 4, cap_ = mp::rt::lds_per_workgroup();                                      \
      const uint32_t wpb_ = mp::rt::waves_per_block(lds_words);                                                               \
      if (wpb_ * bytes_ > cap_)                                                                                               \
        throw std::runtime_error(#NAME ": one work item needs " + std::to_string(bytes_) + " bytes of LDS, the device offers " + \
                                 std::to_string(cap_) + " per workgroup (fewer terms per MSM / links per chain equation)");      \
      if (wpb_ * bytes_ > 64u * 1024u)                                                                                        \
        mp::rt::check(hipFuncSetAttribute(reinterpret_cast<const void*>(&NAME<C>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)(wpb_ * bytes_)), "LDS size attribute");                                       \
      hipLaunchKernelGGL((NAME<C>), dim3(((nwaves) + wpb_ - 1) / wpb_), dim3(64 * wpb_), wpb_ * bytes_, (stream), (args),      \
                         (uint32_t)(nwaves), (uint32_t)(lds_words));                                                          \
      mp::rt::launch_check(#NAME);                                                                                            \
    }                                                                                                                         \
  } while (0)

// explicit instantiation / extern declaration of kernel NAME for curve C (X = `template` or `extern template`)
#define MP_KERNEL_INST(X, NAME, ARGS, C) X MP_GLOBAL void NAME<C>(ARGS, uint32_t);
#define MP_LAUNCH(NAME, C, stream, nx, ny, args)                                  \
  do {                                                                            \
    if ((nx) > 0 && (ny) > 0) {                                                   \
      hipLaunchKernelGGL((NAME<C>), dim3(((nx) + 255u) / 256u, (ny)), dim3(256), 0, (stream), (args), (uint32_t)(nx)); \
      mp::rt::launch_check(#NAME);                                                \
    }                                                                             \
  } while (0)
#endif  // MP_RT_HPP
