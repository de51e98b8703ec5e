// single-pass NTT kernels for n = 2^12 (see kernels_fast_impl.cuh / ntt_core.cuh)
#include "kernels_fast_impl.cuh"

namespace mi355ntt {
MI355NTT_DEFINE_SIZE(12)
}  // namespace mi355ntt
