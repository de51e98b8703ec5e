// single-pass NTT kernels for n = 2^14 (see kernels_fast_impl.cuh / ntt_core.cuh)
#include "kernels_fast_impl.cuh"

namespace mi355ntt {
MI355NTT_DEFINE_SIZE(14)
}  // namespace mi355ntt
