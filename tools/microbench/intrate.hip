// Microbenchmark: issue rate of the integer / fp64 instructions a 256-bit Montgomery
// multiplier can be built from on gfx950.  Prints ops/s per instruction kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

template<int KIND>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u + seed;
  uint64_t acc[4] = {a, b, (uint64_t)a * b, 7};
  double d[4] = {(double)a, (double)b, 1.5, 2.5};
  double dm = (double)(seed | 1) * 1e-9;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (KIND == 0) {        // v_mad_u64_u32 : 32x32+64
        acc[u & 3] = (uint64_t)(uint32_t)acc[u & 3] * (uint32_t)b + acc[u & 3];
      } else if constexpr (KIND == 1) { // v_mul_lo_u32
        acc[u & 3] = (uint32_t)acc[u & 3] * (uint32_t)(b + u);
      } else if constexpr (KIND == 2) { // v_mul_hi_u32
        acc[u & 3] = __umulhi((uint32_t)acc[u & 3], b + u) + 1u;
      } else if constexpr (KIND == 3) { // v_fma_f64
        d[u & 3] = __builtin_fma(d[u & 3], dm, d[(u + 1) & 3]);
      } else if constexpr (KIND == 4) { // v_add_co_u32 + v_addc (64-bit add)
        acc[u & 3] = acc[u & 3] + acc[(u + 1) & 3];
      } else if constexpr (KIND == 5) { // v_mad_u32_u24
        acc[u & 3] = __umul24((uint32_t)acc[u & 3], b) + (uint32_t)acc[(u + 1) & 3];
      } else if constexpr (KIND == 6) { // v_add_u32 (32-bit add baseline)
        acc[u & 3] = (uint32_t)acc[u & 3] + (uint32_t)acc[(u + 1) & 3] ;
      } else if constexpr (KIND == 7) { // v_mad_u64_u32 with 8 independent chains (ILP)
        acc[u & 3] = (uint64_t)(uint32_t)(acc[u & 3] >> 7) * (uint32_t)b + acc[(u+2) & 3];
      }
    }
  }
  uint64_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  r ^= (uint64_t)(d[0] + d[1] + d[2] + d[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

template<int KIND> int run(const char* name, uint32_t* dout, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<KIND><<<blocks, 256>>>(dout, 1); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<KIND><<<blocks, 256>>>(dout, 2);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * ITERS * UNROLL;
  printf("%-28s blocks=%5d  %8.3f ms  %8.2f Gop/s  (%.2f lane-ops/clk/SIMD @2.4GHz)\n", name, blocks, ms, ops / ms * 1e-6,
         ops / (ms * 1e-3) / (256.0 * 4 * 2.4e9));
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  uint32_t* dout; CK(hipMalloc(&dout, 8192 * 256 * 4));
  for (int blocks : {2048, 8192}) {
    run<0>("v_mad_u64_u32 (dep chain x4)", dout, blocks);
    run<7>("v_mad_u64_u32 (shifted)", dout, blocks);
    run<1>("v_mul_lo_u32", dout, blocks);
    run<2>("v_mul_hi_u32 (+add)", dout, blocks);
    run<3>("v_fma_f64", dout, blocks);
    run<4>("u64 add (add_co+addc)", dout, blocks);
    run<5>("v_mad_u32_u24", dout, blocks);
    run<6>("v_add_u32", dout, blocks);
  }
  return 0;
}
