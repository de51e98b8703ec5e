// n = 2^16 on the n = 2^15 kernels: the launches whose first stage couples the two half-size transforms of a polynomial
// (k_forward15<.., SPLIT>, kernels_fast_impl.cuh); a translation unit of its own so the six extra instantiations compile in
// parallel with the rest
#include "kernels_fast_impl.cuh"

namespace mi355ntt {

// n = 2^16 forward / inverse on a 2^15 table set: one launch, each workgroup transforms both halves of its polynomials (coupling stage fused)
// From how many polynomials on the one-launch forms win over the stage launch + small-batch kernels (tools/crossover16.py,
// profiles/r03_n65536_crossover.txt): the pair launch runs 2 num workgroups, flat 42 ... 52 us up to 128 polynomials; the
// single-workgroup forms transform both halves one after the other, flat 80 us up to 256.  No second window as at n = 2^15:
// up to 256 polynomials every workgroup makes one pass.  op: 0 forward (pair form available or not), 1 inverse, 2 fused product.
static bool split_ok(unsigned num, int op, bool pair)
{
    static const long forced = [] {
        const char* e = std::getenv("MI355NTT_LATENCY_PATH_MAX");      // (tuning / tests, as use_latency_path: counts half-size transforms)
        return e ? (long)std::strtoul(e, nullptr, 10) : -1L;
    }();
    if (forced >= 0) return 2ul * num > (unsigned long)forced;
    const unsigned from = op == 0 ? (pair ? 72u : 120u) : op == 1 ? 120u : 96u;
    return num >= from;
}
// Classes 5 (59-bit near-2^k) and 3 (61-bit near-2^k: the shape of the reference's gamma, demo.cu:93) have kernels of their own here
// since round 5 (before: folded into classes 4 and 2).  MI355NTT_N16_NO_CLASS35=1 folds them again (A/B measurements).
static bool fold_classes_35()
{
    static const bool on = std::getenv("MI355NTT_N16_NO_CLASS35") != nullptr;
    return on;
}
static hipError_t launch_fwd_split16(int hl, u64* d_a, const TwPair* tw, const PrimeDev* pr, unsigned num, unsigned division, unsigned base,
                                     hipStream_t s)
{
#ifndef MI355NTT_ONLY_HL4N
    dim3 g(persistent_grid<15>(num)), b(1024);
    auto go = [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        k_forward15<H, NR, 1><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num);
    };
    // (forward: classes 5 / 3 stay folded into 4 / 2 -- their own instantiations of this kernel need 16 / 24 bytes of scratch and measure
    // no faster, 0.232 against 0.228 ms per 512 polynomials at 61 bits; the inverse and fused forms below gain 10 % / 7 % from class 3)
    dispatch_class<false>(hl, go);
#endif
    return hipGetLastError();
}

static hipError_t launch_inv_split16(int hl, u64* d_a, const u64* d_bhat, const TwPair* tw, const PrimeDev* pr, unsigned num,
                                     unsigned division, unsigned base, hipStream_t s)
{
#ifndef MI355NTT_ONLY_HL4N
    dim3 g(persistent_grid<15>(num)), b(1024);
    auto go = [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        if (d_bhat) k_inverse15_split<H, NR, true><<<g, b, 0, s>>>(d_a, d_bhat, tw, pr, division, base, num);
        else k_inverse15_split<H, NR, false><<<g, b, 0, s>>>(d_a, nullptr, tw, pr, division, base, num);
    };
    if (fold_classes_35()) dispatch_class<false>(hl, go);
    else dispatch_class<true>(hl, go);
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
    dispatch_class<false>(hl, [&](auto hc, auto nc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool NR = decltype(nc)::value;
        // (the 61/62-bit classes -- exact quotients, a reduction in every stage -- fit 128 VGPRs in this form only with spills and
        // then run slower than the single-workgroup launch, 0.267 against 0.226 ms per 512 polynomials: they keep that one,
        // fast_forward_split16 asks fast_fwd_pair_ok_16 first)
        if constexpr (H >= 4) k_forward15_pair<H, NR><<<g, b, 0, s>>>(d_a, tw, pr, division, base, num, d_flags);
        else launched = false;
    });
    if (!launched) return hipErrorNotSupported;
#endif
    return hipGetLastError();
}

bool fast_split_ok_16(unsigned num, int op, bool pair) { return split_ok(num, op, pair); }
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
