// curve_stark_msm.hip -- the group-arithmetic kernels of one curve (explicit instantiations; see kernels_msm.hpp)
#include "kernels_bucket.hpp"
#include "kernels_decompress.hpp"
namespace mp {
MP_MSM_KERNELS(template, Stark)
MP_BUCKET_KERNELS(template, Stark)
MP_DECOMPRESS_KERNELS(template, Stark)
}
