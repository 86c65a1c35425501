// curve_secp256k1.hip -- instantiates the engine for one curve (separate TU: the curves compile in parallel)
#include "engine_core.hpp"
namespace mp {
MP_MSM_KERNELS(extern template, Secp256k1)
MP_BUCKET_KERNELS(extern template, Secp256k1)
MP_DECOMPRESS_KERNELS(extern template, Secp256k1)
}
MP_DEFINE_CURVE(Secp256k1)
