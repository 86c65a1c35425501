// curve_bls12_377.hip -- instantiates the engine for one curve (separate TU: the curves compile in parallel)
#include "engine_core.hpp"
namespace mp {
MP_MSM_KERNELS(extern template, Bls12_377)
MP_BUCKET_KERNELS(extern template, Bls12_377)
MP_DECOMPRESS_KERNELS(extern template, Bls12_377)
}
MP_DEFINE_CURVE(Bls12_377)
