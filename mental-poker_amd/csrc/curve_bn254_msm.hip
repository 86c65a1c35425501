// curve_bn254_msm.hip -- the group-arithmetic kernels of one curve (explicit instantiations; see kernels_msm.hpp)
#include "kernels_bucket.hpp"
#include "kernels_decompress.hpp"
namespace mp {
MP_MSM_KERNELS(template, Bn254)
MP_BUCKET_KERNELS(template, Bn254)
MP_DECOMPRESS_KERNELS(template, Bn254)
}
