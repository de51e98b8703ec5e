// single-pass NTT kernels for n = 2^11 (see kernels_fast_impl.cuh / ntt_core.cuh)
#include "kernels_fast_impl.cuh"

namespace mi355ntt {
MI355NTT_DEFINE_SIZE(11)
}  // namespace mi355ntt
