// the n = 2^15 fused products with an element-wise epilogue in the store path (kernels_epi.cuh): a translation unit of its own so the
// instantiations compile in parallel with the rest
#include "kernels_fast_impl.cuh"
#include "kernels_epi.cuh"

namespace mi355ntt {

hipError_t fast_mul_epi_15(int kind, int hl, u64* d_a, const u64* d_b, const TwPair* twf, const TwPair* twi, const PrimeDev* pr, unsigned num,
                           unsigned division, hipStream_t s, const u64* other, const void* consts)
{
    const PolymulEpi epi{other, static_cast<const EpiPrime*>(consts)};
    if (kind == 1) return launch_mul_epi15<1>(hl, d_a, d_b, twf, twi, pr, num, division, s, epi);
    return hipErrorInvalidValue;
}

}  // namespace mi355ntt
