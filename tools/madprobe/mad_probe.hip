// Probe kernels for tools/gen_mad_counts.py: one field operation each, straight-line, so that the multiply-add
// instructions (v_mad_u64_u32 / v_mad_i64_i32) of the compiled product / square / fused product pair can be counted in
// the gfx950 assembly of the SAME sources and flags the engine is built from.  Never linked into libmpshuffle.so.
#include "../../mental-poker_amd/csrc/field.hpp"

namespace mp {
template <class F>
__device__ __forceinline__ Fe<F> probe_ld(const uint32_t* p) {
  Fe<F> a;
  for (int i = 0; i < (F::L29 ? 9 : F::NW); ++i) a.v[i] = p[i];
  return a;
}
template <class F>
__device__ __forceinline__ void probe_st(uint32_t* p, const Fe<F>& a) {
  for (int i = 0; i < (F::L29 ? 9 : F::NW); ++i) p[i] = a.v[i];
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
