// curve_stark.hip -- instantiates the engine for one curve (separate TU: the curves compile in parallel)
#include "engine_core.hpp"
namespace mp {
MP_MSM_KERNELS(extern template, Stark)
MP_BUCKET_KERNELS(extern template, Stark)
MP_DECOMPRESS_KERNELS(extern template, Stark)
}
MP_DEFINE_CURVE(Stark)
