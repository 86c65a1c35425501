// Microbenchmark: throughput of candidate 256-bit Montgomery multipliers for the STARK base field on gfx950.
//   cios      : mental-poker_amd/csrc/field.hpp fe_mul (8x32 limbs, CIOS in plain C)
//   comba_asm : 8x32 limbs, product scanning, v_mad_u64_u32 with carry-out + v_addc into a third word (inline asm)
//   comba_grp : same, one asm statement per column
//   l29       : 9x29-bit limbs, product scanning with lazy 64-bit column sums (plain C, no carries in the inner loop)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../mental-poker_amd/csrc/field.hpp"
using namespace mp;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__device__ constexpr uint32_t MOD[8] = {1,0,0,0,0,0,0x11,0x08000000};

__device__ __forceinline__ void condsub(uint32_t r[8], const uint32_t t[8], uint32_t t8) {
  uint32_t d[8]; uint64_t br = 0;
  #pragma unroll
  for (int i = 0; i < 8; ++i) { uint64_t x = (uint64_t)t[i] - MOD[i] - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
  bool ge = (br == 0) || t8;
  #pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = ge ? d[i] : t[i];
}
__device__ __forceinline__ void mac(uint64_t& acc, uint32_t& acc2, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mul_comba_asm(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  uint64_t acc = 0; uint32_t acc2 = 0; uint32_t t[8]; uint32_t m[8];
  #pragma unroll
  for (int k = 0; k < 15; ++k) {
    #pragma unroll
    for (int i = 0; i < 8; ++i) { int j = k - i; if (j < 0 || j > 7) continue; mac(acc, acc2, a[i], b[j]); }
    if (k < 8) {
      #pragma unroll
      for (int i = 0; i < k; ++i) { if (MOD[k-i] == 0) continue; mac(acc, acc2, m[i], MOD[k-i]); }
      m[k] = 0u - (uint32_t)acc;
      uint32_t c = (uint32_t)acc != 0;
      acc = (uint64_t)(uint32_t)(acc >> 32) + c + ((uint64_t)acc2 << 32); acc2 = 0;
    } else {
      #pragma unroll
      for (int i = k - 7; i < 8; ++i) { if (MOD[k-i] == 0) continue; mac(acc, acc2, m[i], MOD[k-i]); }
      t[k-8] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)acc2 << 32); acc2 = 0;
    }
  }
  t[7] = (uint32_t)acc;
  condsub(r, t, (uint32_t)(acc >> 32));
}
// grouped: one asm statement per column, products listed explicitly
#define MAC1 "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC2 MAC1 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC3 MAC2 "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC4 MAC3 "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC5 MAC4 "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC6 MAC5 "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC7 MAC6 "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC8 MAC7 "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC9 MAC8 "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define MAC10 MAC9 "v_mad_u64_u32 %0, vcc, %20, %21, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define P(i,j) "v"(x[i]), "v"(y[j])
__device__ __forceinline__ void mul_comba_grp(uint32_t r[8], const uint32_t x[8], const uint32_t y[8]) {
  uint64_t acc = 0; uint32_t acc2 = 0; uint32_t t[8]; uint32_t m[8];
  const uint32_t P6 = 0x11u, P7 = 0x08000000u;
#define NEXTLOW(k) { m[k] = 0u - (uint32_t)acc; uint32_t c = (uint32_t)acc != 0; acc = (uint64_t)(uint32_t)(acc >> 32) + c + ((uint64_t)acc2 << 32); acc2 = 0; }
#define NEXTHIGH(k) { t[k-8] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)acc2 << 32); acc2 = 0; }
  asm(MAC1 : "+v"(acc), "+v"(acc2) : P(0,0) : "vcc"); NEXTLOW(0)
  asm(MAC2 : "+v"(acc), "+v"(acc2) : P(0,1), P(1,0) : "vcc"); NEXTLOW(1)
  asm(MAC3 : "+v"(acc), "+v"(acc2) : P(0,2), P(1,1), P(2,0) : "vcc"); NEXTLOW(2)
  asm(MAC4 : "+v"(acc), "+v"(acc2) : P(0,3), P(1,2), P(2,1), P(3,0) : "vcc"); NEXTLOW(3)
  asm(MAC5 : "+v"(acc), "+v"(acc2) : P(0,4), P(1,3), P(2,2), P(3,1), P(4,0) : "vcc"); NEXTLOW(4)
  asm(MAC6 : "+v"(acc), "+v"(acc2) : P(0,5), P(1,4), P(2,3), P(3,2), P(4,1), P(5,0) : "vcc"); NEXTLOW(5)
  asm(MAC8 : "+v"(acc), "+v"(acc2) : P(0,6), P(1,5), P(2,4), P(3,3), P(4,2), P(5,1), P(6,0), "v"(m[0]), "v"(P6) : "vcc"); NEXTLOW(6)
  asm(MAC10 : "+v"(acc), "+v"(acc2) : P(0,7), P(1,6), P(2,5), P(3,4), P(4,3), P(5,2), P(6,1), P(7,0), "v"(m[1]), "v"(P6), "v"(m[0]), "v"(P7) : "vcc"); NEXTLOW(7)
  asm(MAC9 : "+v"(acc), "+v"(acc2) : P(1,7), P(2,6), P(3,5), P(4,4), P(5,3), P(6,2), P(7,1), "v"(m[2]), "v"(P6), "v"(m[1]), "v"(P7) : "vcc"); NEXTHIGH(8)
  asm(MAC8 : "+v"(acc), "+v"(acc2) : P(2,7), P(3,6), P(4,5), P(5,4), P(6,3), P(7,2), "v"(m[3]), "v"(P6), "v"(m[2]), "v"(P7) : "vcc"); NEXTHIGH(9)
  asm(MAC7 : "+v"(acc), "+v"(acc2) : P(3,7), P(4,6), P(5,5), P(6,4), P(7,3), "v"(m[4]), "v"(P6), "v"(m[3]), "v"(P7) : "vcc"); NEXTHIGH(10)
  asm(MAC6 : "+v"(acc), "+v"(acc2) : P(4,7), P(5,6), P(6,5), P(7,4), "v"(m[5]), "v"(P6), "v"(m[4]), "v"(P7) : "vcc"); NEXTHIGH(11)
  asm(MAC5 : "+v"(acc), "+v"(acc2) : P(5,7), P(6,6), P(7,5), "v"(m[6]), "v"(P6), "v"(m[5]), "v"(P7) : "vcc"); NEXTHIGH(12)
  asm(MAC4 : "+v"(acc), "+v"(acc2) : P(6,7), P(7,6), "v"(m[7]), "v"(P6), "v"(m[6]), "v"(P7) : "vcc"); NEXTHIGH(13)
  asm(MAC2 : "+v"(acc), "+v"(acc2) : P(7,7), "v"(m[7]), "v"(P7) : "vcc"); NEXTHIGH(14)
  t[7] = (uint32_t)acc;
  condsub(r, t, (uint32_t)(acc >> 32));
}
// 9 x 29-bit limbs
__device__ __forceinline__ void mul_l29(uint32_t r[9], const uint32_t a[9], const uint32_t b[9]) {
  const uint32_t MASK = (1u << 29) - 1, P6 = 17u << 18, P8 = 1u << 19;
  uint64_t c[18];
  #pragma unroll
  for (int k = 0; k < 18; ++k) c[k] = 0;
  #pragma unroll
  for (int i = 0; i < 9; ++i)
    #pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a[i] * b[j];
  #pragma unroll
  for (int k = 0; k < 9; ++k) {
    uint32_t m = (0u - (uint32_t)c[k]) & MASK;
    c[k] += m;
    c[k + 6] += (uint64_t)m * P6;
    c[k + 8] += (uint64_t)m * P8;
    c[k + 1] += c[k] >> 29;
  }
  #pragma unroll
  for (int k = 9; k < 17; ++k) { r[k - 9] = (uint32_t)c[k] & MASK; c[k + 1] += c[k] >> 29; }
  r[8] = (uint32_t)c[17];
}

// l29 with the sparse-modulus reduction products done as shifts (no mads): m*P6 = (m<<22)+(m<<18), m*P8 = m<<19
__device__ __forceinline__ void mul_l29s(uint32_t r[9], const uint32_t a[9], const uint32_t b[9]) {
  const uint32_t MASK = (1u << 29) - 1;
  uint64_t c[18];
  #pragma unroll
  for (int k = 0; k < 18; ++k) c[k] = 0;
  #pragma unroll
  for (int i = 0; i < 9; ++i)
    #pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a[i] * b[j];
  #pragma unroll
  for (int k = 0; k < 9; ++k) {
    uint32_t m = (0u - (uint32_t)c[k]) & MASK;
    uint64_t m64 = m;
    asm volatile("" : "+v"(m64));     // keep the compiler from re-folding the shifts into a multiply
    c[k] += m64;
    c[k + 6] += (m64 << 22) + (m64 << 18);
    c[k + 8] += m64 << 19;
    c[k + 1] += c[k] >> 29;
  }
  #pragma unroll
  for (int k = 9; k < 17; ++k) { r[k - 9] = (uint32_t)c[k] & MASK; c[k + 1] += c[k] >> 29; }
  r[8] = (uint32_t)c[17];
}
// sqr with 45 products
__device__ __forceinline__ void sqr_l29(uint32_t r[9], const uint32_t a[9]) {
  const uint32_t MASK = (1u << 29) - 1, P6 = 17u << 18, P8 = 1u << 19;
  uint64_t c[18]; uint32_t a2[9];
  #pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a[i] << 1;
  #pragma unroll
  for (int k = 0; k < 18; ++k) c[k] = 0;
  #pragma unroll
  for (int i = 0; i < 9; ++i) { c[2*i] += (uint64_t)a[i]*a[i];
    #pragma unroll
    for (int j = i + 1; j < 9; ++j) c[i + j] += (uint64_t)a2[i] * a[j]; }
  #pragma unroll
  for (int k = 0; k < 9; ++k) {
    uint32_t m = (0u - (uint32_t)c[k]) & MASK;
    c[k] += m; c[k + 6] += (uint64_t)m * P6; c[k + 8] += (uint64_t)m * P8; c[k + 1] += c[k] >> 29;
  }
  #pragma unroll
  for (int k = 9; k < 17; ++k) { r[k - 9] = (uint32_t)c[k] & MASK; c[k + 1] += c[k] >> 29; }
  r[8] = (uint32_t)c[17];
}

template<int V> __global__ void __launch_bounds__(256) k(const uint32_t* in, uint32_t* out, int iters) {
  uint32_t a[9], b[9];
  size_t tid = blockIdx.x * 256 + threadIdx.x;
  for (int i=0;i<9;i++){a[i]=in[(tid%1024)*18+i]; b[i]=in[(tid%1024)*18+9+i];}
  if (V >= 3) { for (int i=0;i<9;i++){a[i]&=(1u<<29)-1; b[i]&=(1u<<29)-1;} } else { a[7]&=0x07ffffff; b[7]&=0x07ffffff; }
  #pragma unroll 1
  for (int it=0; it<iters; ++it) {
    uint32_t r[9];
    if (V==0) { Fe<StarkFq> x,y; for(int i=0;i<8;i++){x.v[i]=a[i];y.v[i]=b[i];} Fe<StarkFq> z=fe_mul<StarkFq>(x,y); for(int i=0;i<8;i++) r[i]=z.v[i]; }
    else if (V==1) mul_comba_asm(r,a,b);
    else if (V==2) mul_comba_grp(r,a,b);
    else if (V==3) mul_l29(r,a,b);
    else if (V==4) mul_l29s(r,a,b);
    else { sqr_l29(r,a); for(int i=0;i<9;i++) r[i]^=b[i]&1; }
    for(int i=0;i<9;i++){a[i]=b[i]; b[i]=r[i];}
  }
  for (int i=0;i<9;i++) out[tid*9+i]=b[i];
}
template<int V> int run(const char* name, uint32_t* din, uint32_t* dout, int blocks, int iters, uint32_t* hout) {
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V><<<blocks,256>>>(din,dout,iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<V><<<blocks,256>>>(din,dout,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  CK(hipMemcpy(hout, dout, 9*4*4, hipMemcpyDeviceToHost));
  printf("%-12s blocks=%5d %8.3f ms  %8.2f Gmul/s   out0=%08x %08x %08x\n", name, blocks, ms, (double)blocks*256*iters/ms*1e-6, hout[0], hout[1], hout[7]);
  return 0;
}
int main(){
  uint32_t *din,*dout; CK(hipMalloc(&din,1024*18*4)); CK(hipMalloc(&dout,(size_t)8192*256*9*4));
  uint32_t h[1024*18]; uint32_t s=12345; for(int i=0;i<1024*18;i++){s=s*1664525u+1013904223u; h[i]=s;}
  CK(hipMemcpy(din,h,sizeof(h),hipMemcpyHostToDevice));
  uint32_t hout[36];
  for (int blocks : {1024, 4096}) {
    run<0>("cios",din,dout,blocks,2000,hout); run<1>("comba_asm",din,dout,blocks,2000,hout); run<2>("comba_grp",din,dout,blocks,2000,hout); run<3>("l29",din,dout,blocks,2000,hout); run<4>("l29_shift",din,dout,blocks,2000,hout); run<5>("l29_sqr",din,dout,blocks,2000,hout);
  }
  return 0;
}
