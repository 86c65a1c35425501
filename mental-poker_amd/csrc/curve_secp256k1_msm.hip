// curve_secp256k1_msm.hip -- the group-arithmetic kernels of one curve (explicit instantiations; see kernels_msm.hpp)
#include "kernels_bucket.hpp"
#include "kernels_decompress.hpp"
namespace mp {
MP_MSM_KERNELS(template, Secp256k1)
MP_BUCKET_KERNELS(template, Secp256k1)
MP_DECOMPRESS_KERNELS(template, Secp256k1)
}
