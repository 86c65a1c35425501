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
inline void stream_wait(Stream s, Event e) { check(hipStreamWaitEvent(s, e, 0), "stream wait event"); }
inline void event_sync(Event e) { check(hipEventSynchronize(e), "event sync"); }
// page-locked host memory: DMA at PCIe speed and truly asynchronous copies (pageable buffers go through the runtime's
// bounce buffers at ~9 GB/s and block the calling thread)
inline void* host_alloc(size_t bytes) {
  void* p = nullptr;
  check(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
  return p;
}
inline void host_free(void* p) {
  if (p) (void)hipHostFree(p);
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
