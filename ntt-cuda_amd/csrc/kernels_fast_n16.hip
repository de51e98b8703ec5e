// n = 2^16 on the n = 2^15 kernels: the launches whose first stage couples the two half-size transforms of a polynomial
// (k_forward15<.., SPLIT>, kernels_fast_impl.cuh); a translation unit of its own so the six extra instantiations compile in
// parallel with the rest
#include "kernels_fast_impl.cuh"

namespace mi355ntt {

// n = 2^16 forward / inverse on a 2^15 table set: one launch, each workgroup transforms both halves of its polynomials (coupling stage fused)
static bool fwd_split_ok(unsigned num) { return !use_latency_path<15>(2 * num, false); }
static hipError_t launch_fwd_split16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                                     hipStream_t s)
{
#ifndef MI355NTT_ONLY_HL4N
    dim3 g(persistent_grid<15>(num)), b(1024);
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        k_forward15<H, NR, 1><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num);
    });
#endif
    return hipGetLastError();
}

static hipError_t launch_inv_split16(int hl, u64* d_a, const u64* d_bhat, const TwPair* tw, const PrimeDev* pr, unsigned num,
                                     unsigned division, unsigned base, hipStream_t s)
{
#ifndef MI355NTT_ONLY_HL4N
    dim3 g(persistent_grid<15>(num)), b(1024);
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if (d_bhat) k_inverse15_split<H, NR, true><<<g, b, 0, s>>>(d_a, d_bhat, tw, pr, division, base, num);
        else k_inverse15_split<H, NR, false><<<g, b, 0, s>>>(d_a, nullptr, tw, pr, division, base, num);
    });
#endif
    return hipGetLastError();
}

hipError_t fast_fwd_pair_16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                            hipStream_t s, unsigned* d_flags)
{
#ifndef MI355NTT_ONLY_HL4N
    const unsigned cap = current_device_cus() / 2;       // pairs of co-resident workgroups, one workgroup per CU
    const unsigned pairs = num < cap ? num : cap;
    if (2 * pairs > kPairFlagWords) return hipErrorInvalidValue;
    dim3 g(2 * pairs), b(1024);
    bool launched = true;
    dispatch_class(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        // (the 61/62-bit classes -- exact quotients, a reduction in every stage -- do not fit 128 VGPRs in this form: they keep the
        // single-workgroup launch, fast_forward_split16 asks fast_fwd_pair_ok_16 first)
        if constexpr (H >= 4) k_forward15_pair<H, NR><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num, d_flags);
        else launched = false;
    });
    if (!launched) return hipErrorNotSupported;
#endif
    return hipGetLastError();
}

bool fast_fwd_split_ok_15(unsigned num) { return fwd_split_ok(num); }
bool fast_fwd_pair_ok_16(int hl) { return (hl & 15) >= 4; }
hipError_t fast_inv_split_16(int hl, u64* d_a, const u64* d_bhat, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division,
                             unsigned base, hipStream_t s)
{
    return launch_inv_split16(hl, d_a, d_bhat, tw, pr, num, division, base, s);
}
hipError_t fast_fwd_split_16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                             hipStream_t s)
{
    return launch_fwd_split16(hl, d_a, tw, pr, num, division, base, s);
}
}  // namespace mi355ntt
