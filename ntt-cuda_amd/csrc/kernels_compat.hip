// kernels_compat.hip -- literal, stage-per-launch kernels behind the raw-parameter entry points.
//
// These follow the reference's arithmetic step by step (Algorithm 7 with the caller's mu/bit_length,
// canonical residues after every stage, halving in every GS stage) so that a caller who brings its
// own tables and Barrett constants gets exactly what ntt_60bit.cuh would have produced.  For n <= 2^14
// the stages of one polynomial run in one launch out of LDS; larger (or other) n one launch per stage.  They are the
// device-side ground truth the fast kernels in kernels_fast.hip are checked against; they are not the
// throughput path.
//
// Reference counterparts (BFV_Scheme/): CTBasedNTTInner(_batch) ntt_60bit.cuh:192-223,527-561;
// GSBasedINTTInner(_batch) :225-265,563-606; barrett* poly_arithmetic.cuh:9-126.
#include <cstdint>

#include "kernels.hpp"

#include <algorithm>
#include "modarith.cuh"

namespace mi355ntt {

namespace {

constexpr int kBlock = 256;

// One CT stage over a batch.  The reference puts the polynomial in blockIdx.y (ntt_60bit.cuh:391-394); a grid's y extent
// stops at 65535, so here the grid is one-dimensional: block b works on polynomial b / bpp (bpp = blocks per polynomial,
// passed as a shift; n is a power of two; the grid is capped and strides over the `blocks` blocks), modulus index = y % division, data offset y*n (:404), table offset index*n (:422).
__global__ void __launch_bounds__(kBlock) ct_stage_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned n,
                                                          unsigned length, unsigned division, unsigned bpp_shift, unsigned blocks, ModSet m,
                                                          const unsigned* __restrict__ guard)
{
    if (guard && guard[0] != guard[1]) return;        // fallback leg of a checked raw call: the throughput kernel did the work
    for (unsigned blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
        unsigned y = blk >> bpp_shift;
        unsigned idx = y % division;
        u64 q = m.q[idx], mu = m.mu[idx];
        u32 k = m.k[idx];
        unsigned g = (blk & ((1u << bpp_shift) - 1u)) * kBlock + threadIdx.x;
        if (g >= n / 2) continue;
        unsigned step = (n / length) / 2;
        unsigned p = g / step;
        unsigned j = p * step * 2 + (g % step);
        u64* poly = a + (size_t)y * n;
        u64 psi = tabs[(size_t)idx * n + length + p];
        u64 U = poly[j];
        u64 V = barrett_mul(poly[j + step], psi, q, mu, k);
        poly[j] = add_mod(U, V, q);
        poly[j + step] = sub_mod(U, V, q);
    }
}

// One GS stage with the n^-1 halving (ntt_60bit.cuh:225-265)
__global__ void __launch_bounds__(kBlock) gs_stage_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned n,
                                                          unsigned length, unsigned division, unsigned bpp_shift, unsigned blocks, ModSet m,
                                                          const unsigned* __restrict__ guard)
{
    if (guard && guard[0] != guard[1]) return;
    for (unsigned blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
        unsigned y = blk >> bpp_shift;
        unsigned idx = y % division;
        u64 q = m.q[idx], mu = m.mu[idx];
        u32 k = m.k[idx];
        unsigned g = (blk & ((1u << bpp_shift) - 1u)) * kBlock + threadIdx.x;
        if (g >= n / 2) continue;
        unsigned step = (n / length) / 2;
        unsigned p = g / step;
        unsigned j = p * step * 2 + (g % step);
        u64* poly = a + (size_t)y * n;
        u64 psiinv = tabs[(size_t)idx * n + length + p];
        u64 q2 = (q + 1) >> 1;
        u64 U = poly[j];
        u64 V = poly[j + step];
        poly[j] = half_mod(add_mod(U, V, q), q2);
        u64 d = barrett_mul(sub_mod(U, V, q), psiinv, q, mu, k);
        poly[j + step] = half_mod(d, q2);
    }
}

// c[i] = a[i] * b[i] mod q[y % division]   (barrett / barrett_batch / barrett_batch_3param).  One block row per
// polynomial (bpp blocks, grid-stride over the polynomial); VEC: two coefficients (16 bytes) per lane and access.
template <bool VEC>
__global__ void __launch_bounds__(kBlock) pointwise_kernel(u64* __restrict__ c, const u64* __restrict__ a,
                                                           const u64* __restrict__ b, unsigned n, unsigned division, unsigned bpp, ModSet m,
                                                           bool shared_b, unsigned group)
{
    const unsigned y = blockIdx.x / bpp, bx = blockIdx.x % bpp;
    const unsigned idx = y % division;
    // one second operand per polynomial, or `division` of them per key group of `group` polynomials (0: one group)
    b += (size_t)(shared_b ? (group ? (y / group) * division + idx : idx) : y) * n;
    const u64 q = m.q[idx], mu = m.mu[idx];
    const u32 k = m.k[idx];
    const size_t base = (size_t)y * n;
    if constexpr (VEC) {
        const ulonglong2* a2 = reinterpret_cast<const ulonglong2*>(a + base);
        const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(b);
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(c + base);
        for (unsigned x = bx * kBlock + threadIdx.x; x < n / 2; x += bpp * kBlock) {
            const ulonglong2 u = a2[x], w = b2[x];
            ulonglong2 r;
            r.x = barrett_mul(u.x, w.x, q, mu, k);
            r.y = barrett_mul(u.y, w.y, q, mu, k);
            c2[x] = r;
        }
    } else {
        for (unsigned x = bx * kBlock + threadIdx.x; x < n; x += bpp * kBlock) c[base + x] = barrett_mul(a[base + x], b[x], q, mu, k);
    }
}

// 1 where two table sets differ in an entry the transforms read (entry 0 of every table is never read:
// stage `length` reads [length, 2 length)): the raw entry points use it once per (table, moduli) to decide whether a
// caller's table is the one a context derives from (q, psi).
__global__ void __launch_bounds__(kBlock) tables_differ_kernel(const u64* __restrict__ x, const u64* __restrict__ y, unsigned n,
                                                               unsigned count, unsigned* __restrict__ flag)
{
    bool diff = false;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)count * n; i += (size_t)gridDim.x * kBlock)
        if ((i & (n - 1)) != 0 && x[i] != y[i]) diff = true;
    if (diff) atomicOr(flag, 1u);
}

// Stream-ordered form of the same comparison (checked raw calls): guard[0] <- epoch; guard[1] <- epoch where they differ.
__global__ void __launch_bounds__(kBlock) tables_check_kernel(const u64* __restrict__ x, const u64* __restrict__ y, unsigned n,
                                                              unsigned count, unsigned* __restrict__ guard, unsigned epoch,
                                                              unsigned* __restrict__ host_word)
{
    bool diff = false;
    const ulonglong2* x2 = reinterpret_cast<const ulonglong2*>(x);
    const ulonglong2* y2 = reinterpret_cast<const ulonglong2*>(y);
    const size_t pairs = (size_t)count * n / 2;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += (size_t)gridDim.x * kBlock) {
        const ulonglong2 u = x2[i], v = y2[i];
        if (u.y != v.y || (u.x != v.x && ((2 * i) & (n - 1)) != 0)) diff = true;      // entry 0 of a table is never read
    }
    if (diff) {
        guard[1] = epoch;                // (every writer stores the same value; a plain store stays right when the host epoch wraps)
        // host-mapped word of the raw-cache entry: its next call learns that the table no longer holds what the context was derived
        // from and derives a new one (capi.cpp, raw_run) instead of taking the guarded literal leg for ever
        if (host_word) __hip_atomic_store(host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) guard[0] = epoch;
}

// a[i] = a[i] * b mod q   (barrett_int)
__global__ void __launch_bounds__(kBlock) pointwise_scalar_kernel(u64* __restrict__ a, u64 b, unsigned n, u64 q, u64 mu, u32 k)
{
    unsigned x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= n) return;
    a[x] = barrett_mul(a[x], b, q, mu, k);
}

// All stages that fit a slice of 2^LOGS coefficients in one launch, the slice resident in LDS (2^LOGS * 8 B <= 128 KiB):
// the same butterflies on the same indices as the stage kernels above (the reference's CTBasedNTTInnerSingle /
// GSBasedINTTInnerSingle do the same inside their block-private slices, ntt_60bit.cuh:63-190), so the same words.
// One 1024-thread workgroup per slice; a polynomial has n >> LOGS slices.  n <= 2^14: the whole transform.  Larger n: the
// stages with length < slices-per-polynomial couple the slices and run as stage launches around this kernel -- at
// n = 2^15 one stage launch + this kernel = 2 passes over memory (the reference: 3 + 1 forward, 1 + 4 inverse,
// ntt_60bit.cuh:318-324,354-359).
template <int LOGS, bool FWD>
__global__ void __launch_bounds__(1024) literal_lds_kernel(u64* __restrict__ a, const u64* __restrict__ tabs, unsigned n, unsigned division,
                                                           unsigned slices, ModSet m, const unsigned* __restrict__ guard)
{
    if (guard && guard[0] != guard[1]) return;
    constexpr unsigned ns = 1u << LOGS, T = ns / 2 < 1024 ? ns / 2 : 1024, PER = ns / 2 / T;
    __shared__ u64 sh[ns];
    const unsigned spp = n >> LOGS;                       // slices per polynomial (a power of two)
    const unsigned t = threadIdx.x;
    for (unsigned sl = blockIdx.x; sl < slices; sl += gridDim.x) {       // (uniform per workgroup: the barriers below are safe)
        const unsigned y = sl / spp, h = sl % spp, idx = y % division;
        const u64 q = m.q[idx], mu = m.mu[idx];
        const u32 k = m.k[idx];
        u64* slice = a + (size_t)y * n + (size_t)h * ns;
        const u64* tab = tabs + (size_t)idx * n;
        const unsigned g0 = h * (ns / 2);                // first butterfly index of this slice within the polynomial
        if (t < T)
            for (unsigned i = t; i < ns; i += T) sh[i] = slice[i];
        __syncthreads();
        if constexpr (FWD) {
            for (unsigned length = spp; length < n; length *= 2) {
                const unsigned step = (n / length) / 2;
                if (t < T)
                    for (unsigned it = 0; it < PER; it++) {
                        const unsigned g = g0 + t + it * T, p = g / step, j = p * step * 2 + (g % step) - h * ns;
                        const u64 U = sh[j];
                        const u64 V = barrett_mul(sh[j + step], tab[length + p], q, mu, k);
                        sh[j] = add_mod(U, V, q);
                        sh[j + step] = sub_mod(U, V, q);
                    }
                __syncthreads();
            }
        } else {
            const u64 q2 = (q + 1) >> 1;
            for (unsigned length = n / 2; length >= spp && length >= 1; length /= 2) {
                const unsigned step = (n / length) / 2;
                if (t < T)
                    for (unsigned it = 0; it < PER; it++) {
                        const unsigned g = g0 + t + it * T, p = g / step, j = p * step * 2 + (g % step) - h * ns;
                        const u64 U = sh[j], V = sh[j + step];
                        sh[j] = half_mod(add_mod(U, V, q), q2);
                        sh[j + step] = half_mod(barrett_mul(sub_mod(U, V, q), tab[length + p], q, mu, k), q2);
                    }
                __syncthreads();
            }
        }
        if (t < T)
            for (unsigned i = t; i < ns; i += T) slice[i] = sh[i];
        __syncthreads();                                  // the next slice reuses the image
    }
}

unsigned log2u(unsigned x)
{
    unsigned r = 0;
    while ((1u << r) < x) r++;
    return r;
}

// stage launches: 1-D grid of (n / 2 / kBlock) blocks per polynomial
template <bool FWD>
void launch_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m, hipStream_t s,
                  const unsigned* guard = nullptr)
{
    const unsigned bpp = (n / 2 + kBlock - 1) / kBlock, sh = log2u(bpp);
    const unsigned long long total = (unsigned long long)num << sh;
    const unsigned blocks = total > 0xffffffffull ? 0xffffffffu : (unsigned)total;
    // capped: a guarded launch that has nothing to do must cost ~nothing (its price is the dispatch of its workgroups: 64 of them when
    // the launch is a fallback leg that normally returns at once -- round 6: the raw-cache entry whose table has changed derives a new
    // context at its next call, so the leg does real work once or twice per rewritten table, not for ever -- 8192 when it is the transform)
    const unsigned cap = guard ? 64u : 8192u;
    const unsigned grid = blocks < cap ? blocks : cap;
    if (FWD) ct_stage_kernel<<<grid, kBlock, 0, s>>>(d_a, d_tabs, n, length, division, sh, blocks, m, guard);
    else gs_stage_kernel<<<grid, kBlock, 0, s>>>(d_a, d_tabs, n, length, division, sh, blocks, m, guard);
}

// workgroups of the LDS kernel: as many as can be resident (160 KiB of LDS per CU), each strides over the slices
unsigned lds_grid(unsigned slices, int logs, bool guarded = false)
{
    const unsigned per_cu = 163840u / (8u << logs) ? 163840u / (8u << logs) : 1u;
    // 1024-thread workgroups: at most two per CU (32 in all for a guarded fallback leg: it normally returns at once, see launch_stage)
    const unsigned cap = guarded ? 32u : current_device_cus() * (per_cu > 2 ? 2 : per_cu);
    return slices < cap ? slices : cap;
}

// the stages inside LDS-sized slices; returns the number of slices per polynomial it used (0: n not served)
template <bool FWD>
unsigned launch_literal_lds(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m, hipStream_t s,
                            const unsigned* guard)
{
    switch (n) {
    case 2048: literal_lds_kernel<11, FWD><<<lds_grid(num, 11, guard != nullptr), 1024, 0, s>>>(d_a, d_tabs, n, division, num, m, guard); return 1;
    case 4096: literal_lds_kernel<12, FWD><<<lds_grid(num, 12, guard != nullptr), 1024, 0, s>>>(d_a, d_tabs, n, division, num, m, guard); return 1;
    case 8192: literal_lds_kernel<13, FWD><<<lds_grid(num, 13, guard != nullptr), 1024, 0, s>>>(d_a, d_tabs, n, division, num, m, guard); return 1;
    case 16384: literal_lds_kernel<14, FWD><<<lds_grid(num, 14, guard != nullptr), 1024, 0, s>>>(d_a, d_tabs, n, division, num, m, guard); return 1;
    default: break;
    }
    if (n >= 32768 && n <= (1u << 20) && (n & (n - 1)) == 0) {
        const unsigned spp = n >> 14;
        if ((unsigned long long)num * spp > 0x7fffffffull) return 0;
        literal_lds_kernel<14, FWD><<<lds_grid(num * spp, 14, guard != nullptr), 1024, 0, s>>>(d_a, d_tabs, n, division, num * spp, m, guard);
        return spp;
    }
    return 0;
}

}  // namespace

hipError_t compat_forward_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s, const unsigned* guard)
{
    if (n >= 32768 && (n & (n - 1)) == 0 && n <= (1u << 20)) {
        // stages 1 .. spp/2 couple the LDS slices: stage launches first, then everything else out of LDS
        const unsigned spp = n >> 14;
        for (unsigned length = 1; length < spp; length *= 2) launch_stage<true>(d_a, n, d_tabs, length, num, division, m, s, guard);
        if (launch_literal_lds<true>(d_a, n, d_tabs, num, division, m, s, guard)) return hipGetLastError();
        for (unsigned length = spp; length < n; length *= 2) launch_stage<true>(d_a, n, d_tabs, length, num, division, m, s, guard);
        return hipGetLastError();
    }
    if (launch_literal_lds<true>(d_a, n, d_tabs, num, division, m, s, guard)) return hipGetLastError();
    for (unsigned length = 1; length < n; length *= 2) launch_stage<true>(d_a, n, d_tabs, length, num, division, m, s, guard);
    return hipGetLastError();
}

hipError_t compat_inverse_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s, const unsigned* guard)
{
    if (n >= 32768 && (n & (n - 1)) == 0 && n <= (1u << 20)) {
        const unsigned spp = n >> 14;
        if (launch_literal_lds<false>(d_a, n, d_tabs, num, division, m, s, guard)) {
            for (unsigned length = spp / 2; length >= 1; length /= 2) launch_stage<false>(d_a, n, d_tabs, length, num, division, m, s, guard);
            return hipGetLastError();
        }
    } else if (launch_literal_lds<false>(d_a, n, d_tabs, num, division, m, s, guard)) {
        return hipGetLastError();
    }
    for (unsigned length = n / 2; length >= 1; length /= 2) launch_stage<false>(d_a, n, d_tabs, length, num, division, m, s, guard);
    return hipGetLastError();
}

hipError_t compat_ct_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s)
{
    launch_stage<true>(d_a, n, d_tabs, length, num, division, m, s);
    return hipGetLastError();
}

hipError_t compat_gs_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s)
{
    launch_stage<false>(d_a, n, d_tabs, length, num, division, m, s);
    return hipGetLastError();
}

hipError_t compat_pointwise(u64* d_c, const u64* d_a, const u64* d_b, unsigned n, unsigned num, unsigned division,
                            const ModSet& m, hipStream_t s, bool shared_b, unsigned group)
{
    const bool vec = (n % 2 == 0) && ((((uintptr_t)d_c | (uintptr_t)d_a | (uintptr_t)d_b) & 15u) == 0);
    unsigned bpp = ((vec ? n / 2 : n) + kBlock - 1) / kBlock;
    if (bpp > 64) bpp = 64;                               // grid-stride inside the polynomial beyond that
    while ((unsigned long long)bpp * num > 0x7fffffffull && bpp > 1) bpp /= 2;
    if (vec) pointwise_kernel<true><<<bpp * num, kBlock, 0, s>>>(d_c, d_a, d_b, n, division, bpp, m, shared_b, group);
    else pointwise_kernel<false><<<bpp * num, kBlock, 0, s>>>(d_c, d_a, d_b, n, division, bpp, m, shared_b, group);
    return hipGetLastError();
}

hipError_t compat_pointwise_scalar(u64* d_a, u64 b, unsigned n, u64 q, u64 mu, unsigned k, hipStream_t s)
{
    pointwise_scalar_kernel<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(d_a, b, n, q, mu, k);
    return hipGetLastError();
}

// Synthetic inputs of the measurement recipe: polynomial y = splitmix64 stream of seed seed_base + y,
// value i = z_i mod q[y % division] (z_i = the i-th output; the state after i + 1 steps is seed + (i + 1) * golden ratio).
__global__ void __launch_bounds__(kBlock) synth_splitmix_kernel(u64* __restrict__ a, unsigned n, unsigned num, unsigned division, ModSet m,
                                                                u64 seed_base)
{
    const size_t total = (size_t)num * n;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < total; g += (size_t)gridDim.x * kBlock) {
        const unsigned y = (unsigned)(g / n), i = (unsigned)(g % n);
        u64 z = seed_base + y + (u64)(i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        a[g] = z % m.q[y % division];
    }
}

hipError_t compat_synth_splitmix(u64* d_a, unsigned n, unsigned num, unsigned division, const ModSet& m, u64 seed_base, hipStream_t s)
{
    if (num == 0) return hipSuccess;
    const size_t total = (size_t)num * n;
    const unsigned grid = (unsigned)std::min<size_t>((total + kBlock - 1) / kBlock, 65536);
    synth_splitmix_kernel<<<grid, kBlock, 0, s>>>(d_a, n, num, division, m, seed_base);
    return hipGetLastError();
}

hipError_t compat_tables_differ(const u64* d_x, const u64* d_y, unsigned n, unsigned count, unsigned* d_flag, hipStream_t s)
{
    tables_differ_kernel<<<256, kBlock, 0, s>>>(d_x, d_y, n, count, d_flag);
    return hipGetLastError();
}

hipError_t compat_tables_check(const u64* d_x, const u64* d_y, unsigned n, unsigned count, unsigned* d_guard, unsigned epoch, hipStream_t s,
                               unsigned* d_host_word)
{
    tables_check_kernel<<<128, kBlock, 0, s>>>(d_x, d_y, n, count, d_guard, epoch, d_host_word);
    return hipGetLastError();
}


// ---- the stand-alone element-wise wrappers of poly_arithmetic.cuh:312-352 -----------------------------------------------------
// poly_add (:144-154: `>`, not `>=` -- a sum equal to q stays q), poly_add_integer (:156-166, same comparison), poly_sub (:168-179:
// adds q where a < b and NEVER subtracts b -- mirrored literally), poly_negate (:334-338), mod_t (:128-142:
// the low 64 bits of a b, masked with t - 1 held in a 32-bit register).  One streaming kernel, 16 bytes per lane, grid-stride.
namespace {
// VEC: 16 bytes per lane (both pointers 16-byte aligned); otherwise one word per lane -- the reference's kernels take any pointer to
// unsigned long long, and its wrappers return void: a call on a + odd_offset must transform, not be refused.
template <int OP, bool VEC>
__global__ void __launch_bounds__(256) k_elementwise(u64* __restrict__ a, const u64* __restrict__ b, u64 scalar, u64 q, size_t count)
{
    auto f = [&](u64 x, u64 y) -> u64 {
        if constexpr (OP == kEwAdd || OP == kEwAddInteger) {
            u64 r = x + y;
            if (r > q) r -= q;
            return r;
        } else if constexpr (OP == kEwSub) {
            return x < y ? x + q : x;
        } else if constexpr (OP == kEwNegate) {
            const u64 r = q - x;
            return r * (u64)(r != q);
        } else {
            return (x * y) & (u64)(unsigned)(q - 1);                // (q carries t: `register unsigned mask = t - 1`)
        }
    };
    constexpr bool VEC_B = (OP == kEwAdd || OP == kEwSub);
    const size_t stride = (size_t)gridDim.x * 256;
    if constexpr (VEC) {
        const size_t pairs = count / 2;
        ulonglong2* a2 = reinterpret_cast<ulonglong2*>(a);
        const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(b);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride) {
            ulonglong2 x = a2[i];
            ulonglong2 y;
            if constexpr (VEC_B) y = b2[i];
            else y = make_ulonglong2(scalar, scalar);
            x.x = f(x.x, y.x);
            x.y = f(x.y, y.y);
            a2[i] = x;
        }
        if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) a[count - 1] = f(a[count - 1], VEC_B ? b[count - 1] : scalar);
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) a[i] = f(a[i], VEC_B ? b[i] : scalar);
    }
}
}  // namespace

hipError_t compat_elementwise(int op, u64* d_a, const u64* d_b, u64 scalar, u64 q, size_t count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    const bool vec = ((reinterpret_cast<uintptr_t>(d_a) | reinterpret_cast<uintptr_t>(d_b)) & 15u) == 0;      // (a null d_b counts as aligned)
    size_t blocks = ((vec ? count / 2 : count) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    const unsigned g = (unsigned)blocks;
#define MI355NTT_EW(OP, B, SC)                                                        \
    do {                                                                              \
        if (vec) k_elementwise<OP, true><<<g, 256, 0, s>>>(d_a, B, SC, q, count);     \
        else k_elementwise<OP, false><<<g, 256, 0, s>>>(d_a, B, SC, q, count);        \
    } while (0)
    switch (op) {
    case kEwAdd: MI355NTT_EW(kEwAdd, d_b, 0); break;
    case kEwAddInteger: MI355NTT_EW(kEwAddInteger, nullptr, scalar); break;
    case kEwSub: MI355NTT_EW(kEwSub, d_b, 0); break;
    case kEwNegate: MI355NTT_EW(kEwNegate, nullptr, 0); break;
    case kEwMulIntT: MI355NTT_EW(kEwMulIntT, nullptr, scalar); break;
    default: return hipErrorInvalidValue;
    }
#undef MI355NTT_EW
    return hipGetLastError();
}

}  // namespace mi355ntt
