// Probe kernels for tools/gen_mad_counts.py: one field operation each, straight-line, so that the multiply-add
// instructions (v_mad_u64_u32 / v_mad_i64_i32) of the compiled product / square / fused product pair can be counted in
// the gfx950 assembly of the SAME sources and flags the engine is built from.  Never linked into libmpshuffle.so.
#include "../../mental-poker_amd/csrc/field.hpp"

namespace mp {
template <class F>
__device__ __forceinline__ Fe<F> probe_ld(const uint32_t* p) {
  Fe<F> a;
  for (int i = 0; i < (F::L29 ? F::NL29 : F::NW); ++i) a.v[i] = p[i];
  return a;
}
template <class F>
__device__ __forceinline__ void probe_st(uint32_t* p, const Fe<F>& a) {
  for (int i = 0; i < (F::L29 ? F::NL29 : F::NW); ++i) p[i] = a.v[i];
}
template <class F>
__global__ void probe_mul(uint32_t* io) {
  uint32_t* p = io + threadIdx.x * 64;
  probe_st<F>(p, fe_mul<F>(probe_ld<F>(p), probe_ld<F>(p + 16)));
}
template <class F>
__global__ void probe_sqr(uint32_t* io) {
  uint32_t* p = io + threadIdx.x * 64;
  probe_st<F>(p, fe_sqr<F>(probe_ld<F>(p)));
}
template <class F>
__global__ void probe_mulsub(uint32_t* io) {
  uint32_t* p = io + threadIdx.x * 64;
  probe_st<F>(p, fe_mulsub<F>(probe_ld<F>(p), probe_ld<F>(p + 16), probe_ld<F>(p + 32), probe_ld<F>(p + 48)));
}
#define PROBE(F)                                  \
  template __global__ void probe_mul<F>(uint32_t*); \
  template __global__ void probe_sqr<F>(uint32_t*); \
  template __global__ void probe_mulsub<F>(uint32_t*);
PROBE(StarkFq) PROBE(StarkFr) PROBE(Bn254Fq) PROBE(Bn254Fr) PROBE(Secp256k1Fq) PROBE(Secp256k1Fr) PROBE(Bls12_377Fq) PROBE(Bls12_377Fr)
}  // namespace mp

// ---- group operations (bench.py's VALU-issue roofline): the main path of one mixed addition / doubling as k_var_msm, k_fixed_msm,
// k_remask and k_bucket_msm execute it -- operand unpacked from the packed table format, conditional negation, in-place update.
// The nested doubling of the (never taken) P + P case is left out (curve.hpp, PROBE).
#include "../../mental-poker_amd/csrc/curve.hpp"
namespace mp {
template <class C>
__global__ void probe_madd(uint32_t* io, int neg) {
  typedef typename C::FqP F;
  uint32_t* p = io + threadIdx.x * 256;
  Xyzz<C> a;
  a.X = probe_ld<F>(p); a.Y = probe_ld<F>(p + 16); a.ZZ = probe_ld<F>(p + 32); a.ZZZ = probe_ld<F>(p + 48);
  Aff<C> q;
  q.x = fe_unpack<F>(p + 64);
  q.y = fe_unpack<F>(p + 64 + F::NW);
  xyzz_madd_signed_ip<C, true>(a, q, neg != 0);
  probe_st<F>(p, a.X); probe_st<F>(p + 16, a.Y); probe_st<F>(p + 32, a.ZZ); probe_st<F>(p + 48, a.ZZZ);
}
template <class C>
__global__ void probe_dbl(uint32_t* io) {
  typedef typename C::FqP F;
  uint32_t* p = io + threadIdx.x * 256;
  Xyzz<C> a;
  a.X = probe_ld<F>(p); a.Y = probe_ld<F>(p + 16); a.ZZ = probe_ld<F>(p + 32); a.ZZZ = probe_ld<F>(p + 48);
  xyzz_dbl_ip<C>(a);
  probe_st<F>(p, a.X); probe_st<F>(p + 16, a.Y); probe_st<F>(p + 32, a.ZZ); probe_st<F>(p + 48, a.ZZZ);
}
template <class C>
__global__ void probe_xadd(uint32_t* io) {
  typedef typename C::FqP F;
  uint32_t* p = io + threadIdx.x * 256;
  Xyzz<C> a, b;
  a.X = probe_ld<F>(p); a.Y = probe_ld<F>(p + 16); a.ZZ = probe_ld<F>(p + 32); a.ZZZ = probe_ld<F>(p + 48);
  b.X = probe_ld<F>(p + 64); b.Y = probe_ld<F>(p + 80); b.ZZ = probe_ld<F>(p + 96); b.ZZZ = probe_ld<F>(p + 112);
  xyzz_add_ip<C, true>(a, b);
  probe_st<F>(p, a.X); probe_st<F>(p + 16, a.Y); probe_st<F>(p + 32, a.ZZ); probe_st<F>(p + 48, a.ZZZ);
}
#define PROBE_GROUP(C)                                   \
  template __global__ void probe_madd<C>(uint32_t*, int); \
  template __global__ void probe_dbl<C>(uint32_t*);         \
  template __global__ void probe_xadd<C>(uint32_t*);
PROBE_GROUP(Stark) PROBE_GROUP(Bn254) PROBE_GROUP(Secp256k1) PROBE_GROUP(Bls12_377)
}  // namespace mp
