// curve_bn254.hip -- instantiates the engine for one curve (separate TU: the curves compile in parallel)
#include "engine_core.hpp"
namespace mp {
MP_MSM_KERNELS(extern template, Bn254)
MP_BUCKET_KERNELS(extern template, Bn254)
MP_DECOMPRESS_KERNELS(extern template, Bn254)
}
MP_DEFINE_CURVE(Bn254)
