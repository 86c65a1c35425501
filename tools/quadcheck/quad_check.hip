// Development check: the four-lane group operations of kernels_quad.hpp against the one-lane ones of curve.hpp, on the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mental-poker_amd/csrc tools/quadcheck/quad_check.hip -o tools/quadcheck/quad_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mental-poker_amd/csrc/kernels_bucket.hpp"
using namespace mp;

template <class C>
__device__ bool same_point(const Xyzz<C>& a, const Xyzz<C>& b) {
  typedef typename C::FqP F;
  const bool ia = fe_is_zero(a.ZZ), ib = fe_is_zero(b.ZZ);
  if (ia || ib) return ia == ib;
  return fe_is_zero(fe_sub<F>(fe_mul<F>(a.X, b.ZZ), fe_mul<F>(b.X, a.ZZ))) && fe_is_zero(fe_sub<F>(fe_mul<F>(a.Y, b.ZZZ), fe_mul<F>(b.Y, a.ZZZ)));
}
template <class C>
__global__ void __launch_bounds__(64) k_check(uint32_t* out) {
  typedef typename C::FqP F;
  WaveCtx wv{threadIdx.x & 63u, nullptr};
  const uint32_t l = threadIdx.x, quad = l >> 2;
  Aff<C> g;
  g.x = fe_unpack<F>(C::GX_MONT);
  g.y = fe_unpack<F>(C::GY_MONT);
  // P = (quad + 2) G, Q = (3 quad + 5) G (XYZZ), qa = 2G + ... affine base g
  Xyzz<C> P = xyzz_inf<C>(), Q = xyzz_inf<C>();
  for (uint32_t i = 0; i < quad + 2; ++i) xyzz_madd_ip<C>(P, g);
  for (uint32_t i = 0; i < 3 * quad + 5; ++i) xyzz_madd_ip<C>(Q, g);
  uint32_t fails = 0;
  PerLane<uint32_t> on;
  // doubling
  {
    Xyzz<C> r = P;
    xyzz_dbl_ip<C>(r);
    PerLane<Xyzz<C>> p;
    p.v = P;
    on.v = (quad & 1u) ? 1u : 0u;          // odd quads only: divergence between quads
    xyzz_dbl_quad<C>(wv, p, on);
    if (!same_point<C>(p.v, (quad & 1u) ? r : P)) fails |= 1;
    on.v = 1;
    p.v = P;
    xyzz_dbl_quad<C>(wv, p, on);
    if (!same_point<C>(p.v, r)) fails |= 2;
  }
  // mixed addition / subtraction
  {
    Xyzz<C> r = P, rs = P;
    xyzz_madd_signed_ip<C>(r, g, false);
    xyzz_madd_signed_ip<C>(rs, g, true);
    PerLane<Xyzz<C>> p;
    PerLane<Aff<C>> q;
    q.v = g;
    p.v = P;
    on.v = 1;
    xyzz_madd_quad<C>(wv, p, q, on);
    if (!same_point<C>(p.v, r)) fails |= 4;
    p.v = P;
    on.v = (quad % 3 == 0) ? 0u : ((quad % 3 == 1) ? 1u : 2u);
    xyzz_madd_quad<C>(wv, p, q, on);
    if (!same_point<C>(p.v, quad % 3 == 0 ? P : (quad % 3 == 1 ? r : rs))) fails |= 8;
  }
  // full addition
  {
    Xyzz<C> r = P;
    xyzz_add_ip<C>(r, Q);
    PerLane<Xyzz<C>> p, q;
    p.v = P;
    q.v = Q;
    on.v = 1;
    xyzz_add_quad<C>(wv, p, q, on);
    if (!same_point<C>(p.v, r)) fails |= 16;
    // P + P through the addition
    Xyzz<C> d = P;
    xyzz_dbl_ip<C>(d);
    p.v = P;
    q.v = P;
    xyzz_add_quad<C>(wv, p, q, on);
    if (!same_point<C>(p.v, d)) fails |= 32;
  }
  out[l] = fails;
}
template <class C>
void run(const char* name) {
  uint32_t* d;
  if (hipMalloc(&d, 64 * 4) != hipSuccess) {
    printf("%-10s hipMalloc failed\n", name);
    return;
  }
  hipLaunchKernelGGL((k_check<C>), dim3(1), dim3(64), 0, 0, d);
  uint32_t h[64];
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) {
    printf("%-10s kernel failed\n", name);
    return;
  }
  uint32_t any = 0;
  for (int i = 0; i < 64; ++i) any |= h[i];
  printf("%-10s fails mask 0x%x  per lane:", name, any);
  for (int i = 0; i < 64; ++i) printf(" %x", h[i]);
  printf("\n");
  (void)hipFree(d);
}
int main() {
  run<Stark>("stark");
  run<Bn254>("bn254");
  run<Secp256k1>("secp256k1");
  run<Bls12_377>("bls12_377");
  return 0;
}
