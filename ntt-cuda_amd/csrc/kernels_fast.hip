// kernels_fast.hip -- throughput path of the NTT engine (context entry points).
//
// forward / inverse / fused polymul: one workgroup per polynomial, single pass over HBM, see ntt_core.cuh.
// Replaces CTBasedNTTInner*/GSBasedINTTInner* (ntt_60bit.cuh:63-265,388-606) and the
// forwardNTT_batch -> barrett_batch -> inverseNTT_batch triple of the BFV drivers
// (bfv_encryption.cuh:268-271, bfv_keygen.cuh:129-133, bfv_decryption.cuh:98-101).
#include "kernels.hpp"
#include "modarith.cuh"
#include "ntt_core.cuh"
#include "kernels_fast_impl.cuh"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace mi355ntt {

namespace {

// ---- pointwise c = a (.) b, 16 bytes per lane ----------------------------------------------------
__global__ void __launch_bounds__(256)
k_pointwise(u64* __restrict__ c, const u64* __restrict__ a, const u64* __restrict__ b, unsigned n, unsigned division,
            const PrimeDev* __restrict__ primes, unsigned bpp)
{
    const unsigned y = blockIdx.x / bpp, bx = blockIdx.x % bpp;      // (a one-dimensional grid: no 65535 limit on the batch)
    const PrimeDev p = primes[y % division];
    const size_t base = (size_t)y * n;
    const ulonglong2* a2 = reinterpret_cast<const ulonglong2*>(a + base);
    const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(b + base);
    ulonglong2* c2 = reinterpret_cast<ulonglong2*>(c + base);
    for (unsigned i = bx * 256 + threadIdx.x; i < n / 2; i += bpp * 256) {
        const ulonglong2 x = a2[i], w = b2[i];
        ulonglong2 r;
        r.x = barrett_mul(x.x, w.x, p.q, p.mu, p.k);
        r.y = barrett_mul(x.y, w.y, p.q, p.mu, p.k);
        c2[i] = r;
    }
}

// Clock probe (measurement helper): one wave reads the shader-cycle counter (s_memtime) and the 100 MHz constant clock
// (s_memrealtime), spins until 20 us of constant clock have passed and reads both again; the differences go into the guard record
// in front of the PrimeDev array (bytes 16..31; bytes 0..7 are the guard words of the checked raw calls).  Enqueued right behind
// a run of launches it reports the shader clock that run left the chip at (the power management moves the clock over
// milliseconds, the probe takes 20 us).  Both readings come from the same wave: the cycle counters of different XCDs are not
// synchronised, so two one-shot probes cannot be subtracted.  A kernel of its own because marks inside k_forward15 /
// k_inverse15 cost the general-prime instantiations their last free VGPRs (scratch again; measured in round 3).
__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long* __restrict__ rec)
{
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(8);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (r1 - r0 < 2000);
    if (threadIdx.x == 0) {
        rec[2] = t1 - t0;
        rec[3] = r1 - r0;
    }
}

// Foreign load for tests and measurements (mi355ntt_ctx_occupy): `workgroups` workgroups of 1024 threads holding 144 KiB of LDS each --
// one whole CU apiece, like the n = 2^15 kernels -- spin for `ticks` of the 100 MHz constant clock and exit.  Touches no memory.
__global__ void __launch_bounds__(1024) k_occupy(unsigned long long ticks, unsigned* sink)
{
    __shared__ unsigned long long hold[18432];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    hold[threadIdx.x] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    if (hold[(threadIdx.x + 1) & 1023] == 1 && sink) sink[0] = 1;      // (never true: keeps the LDS allocation alive)
}

ModSet shifted(const ModSet& m, unsigned base, unsigned division)
{
    ModSet r;
    std::memset(&r, 0, sizeof(r));
    for (unsigned i = 0; i < division && base + i < kMaxPrimes; i++) {
        r.q[i] = m.q[base + i];
        r.mu[i] = m.mu[base + i];
        r.k[i] = m.k[base + i];
    }
    return r;
}

}  // namespace

// ---- the device's pair flags (k_forward15_pair, k_ntt30x PAIR) ----------------------------------
// One zeroed flag buffer per device, owned by ONE stream at a time and shared by BOTH word sizes: a launch of a pair kernel -- 60-bit
// or 30-bit -- is followed by an event on its stream, and another stream gets the buffer only once that event has completed (until
// then it runs the single-workgroup / stage-launch form).  So the pair kernels in flight on a device are ordered on one stream: their
// flags never mix, and workgroups spinning for a partner can only be waiting for workgroups of their own launch that are next in
// line for a CU (two pair kernels co-running on different streams could fill complementary CUs with partner-less spinners).  A
// capturing stream never takes it (a graph may replay next to anything), nor does a stream restricted to part of the CUs (the grid
// must be resident as a whole).  Created with the first n = 2^16 context on the device (fast_tables_create) or at the first 30-bit
// n = 2^16 call (which allocates its per-stream scratch at that point anyway).
struct PairSlot {
    unsigned* d_flags = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t owner = nullptr;
    bool in_flight = false;
    unsigned long long faults = 0;           // sticky: launches of this device that gave up on a partner (mi355ntt_pair_fault_count)
    volatile unsigned* h_err = nullptr;      // host-mapped: written by a workgroup that gave up on its partner (kernels.hpp, watchdog)
    unsigned* d_err = nullptr;               // the same word as the device sees it
};
namespace {
constexpr int kMaxDevices = 64;
PairSlot g_pair[kMaxDevices];
std::mutex g_pair_mutex;

// every CU of the device available to stream s?  (hipExtStreamCreateWithCUMask / ROC_GLOBAL_CU_MASK restrict it)
bool stream_has_all_cus(hipStream_t s)
{
    uint32_t mask[16] = {};
    if (hipExtStreamGetCUMask(s, 16, mask) != hipSuccess) {
        (void)hipGetLastError();
        return true;                                               // (no mask information: the default is every CU)
    }
    unsigned bits = 0;
    for (uint32_t m : mask) bits += (unsigned)__builtin_popcount(m);
    return bits >= current_device_cus();
}
}  // namespace

namespace {
hipError_t pair_init_locked(PairSlot& p)
{
    if (p.d_flags) return hipSuccess;
    unsigned* f = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void**)&f, kPairFlagWords * sizeof(unsigned))) != hipSuccess) return e;
    void* h = nullptr;
    void* d = nullptr;
    if ((e = hipMemset(f, 0, kPairFlagWords * sizeof(unsigned))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&p.done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipHostMalloc(&h, 64, hipHostMallocMapped)) != hipSuccess || (e = hipHostGetDevicePointer(&d, h, 0)) != hipSuccess) {
        if (h) (void)hipHostFree(h);
        (void)hipFree(f);
        return e;
    }
    std::memset(h, 0, 64);
    const unsigned long long tail[2] = {pair_watchdog_ticks(), (unsigned long long)reinterpret_cast<uintptr_t>(d)};      // kPairTicksWord, kPairErrPtrWord
    static_assert(kPairErrPtrWord == kPairTicksWord + 2, "tail layout");
    if ((e = hipMemcpy(f + kPairTicksWord, tail, sizeof(tail), hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipHostFree(h);
        (void)hipFree(f);
        return e;
    }
    p.h_err = static_cast<volatile unsigned*>(h);
    p.d_err = static_cast<unsigned*>(d);
    p.d_flags = f;
    return hipSuccess;
}
}  // namespace

hipError_t pair_init_current_device()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipSuccess;           // (no slot: the single-workgroup form runs)
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    return pair_init_locked(g_pair[dev]);
}

// the slot when stream s may launch a pair kernel now, else null; pair_release records the launch
unsigned long long pair_watchdog_ticks()
{
    static const unsigned long long ticks = [] {
        const char* e = std::getenv("MI355NTT_PAIR_WATCHDOG_MS");
        const long ms = e ? std::atol(e) : 0;
        return ms > 0 ? (unsigned long long)ms * 100000ull : kPairWatchdogTicks;
    }();
    return ticks;
}

PairSlot* pair_acquire(hipStream_t s, hipError_t* status)
{
    if (status) *status = hipSuccess;
    static const bool off = std::getenv("MI355NTT_NO_PAIR16") != nullptr;      // (A/B measurements)
    if (off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    if (!stream_has_all_cus(s)) return nullptr;
    g_pair_mutex.lock();
    PairSlot& p = g_pair[dev];
    // (first use on this device without an n = 2^16 context -- the 30-bit path: allocated here, outside any capture)
    bool ok = pair_init_locked(p) == hipSuccess && p.d_flags != nullptr;
    if (ok && p.in_flight && p.owner != s) {
        if (hipEventQuery(p.done) == hipSuccess) p.in_flight = false;
        else ok = false;
    }
    if (!ok) {
        (void)hipGetLastError();                                    // (hipErrorNotReady of the query is not an error of this call)
        g_pair_mutex.unlock();
        return nullptr;
    }
    if (p.h_err && *p.h_err != 0) {
        // an earlier pair launch gave up on a partner: its data is invalid and the flags are in an arbitrary state.  Clean the slot
        // behind whatever is still queued on this stream (every queued pair launch finds the dead word and returns at once) and
        // report; the next call starts from a clean slot.
        *p.h_err = 0;
        p.faults++;
        (void)hipMemsetAsync(p.d_flags, 0, kPairLiveWords * sizeof(unsigned), s);
        (void)hipMemsetAsync(p.d_flags + kPairDeadWord, 0, sizeof(unsigned), s);
        p.owner = s;
        p.in_flight = (hipEventRecord(p.done, s) == hipSuccess);
        if (!p.in_flight) (void)hipStreamSynchronize(s);
        g_pair_mutex.unlock();
        if (status) *status = hipErrorLaunchFailure;
        return nullptr;
    }
    p.owner = s;
    return &p;                                                      // (mutex held until pair_release: launch and event stay together)
}
void pair_release(PairSlot* p, hipStream_t s)
{
    p->in_flight = (hipEventRecord(p->done, s) == hipSuccess);
    if (!p->in_flight) (void)hipStreamSynchronize(s);               // (no event: be sure instead)
    g_pair_mutex.unlock();
}
unsigned* pair_flags(PairSlot* p) { return p->d_flags; }
unsigned long long pair_fault_count(int device)
{
    if (device < 0 || device >= kMaxDevices) return 0;
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    // (an event nobody has asked about yet counts as well: the error word is still up)
    return g_pair[device].faults + ((g_pair[device].h_err && *g_pair[device].h_err) ? 1ull : 0ull);
}


// ------------------------------------------------------------------------------------------------
size_t fast_prime_record_bytes() { return sizeof(PrimeDev); }

hipError_t fast_tables_create(FastTables* t, unsigned n, unsigned num_primes, const PrimeParams* prime, const u64* h_psi,
                              const u64* h_psiinv, const u64* d_psi, const u64* d_psiinv, const u64* split_fwd, const u64* split_inv, bool literal)
{
    t->n = n;
    t->log_n = 0;
    while ((1u << t->log_n) < n) t->log_n++;
    t->num_primes = num_primes;
    std::memset(&t->mods, 0, sizeof(t->mods));
    t->d_psi = d_psi;
    t->d_psiinv = d_psiinv;
    t->hl = 6;
    bool all_near = true;
    std::vector<PrimeDev> pd(num_primes);
    for (unsigned i = 0; i < num_primes; i++) {
        const PrimeParams& pp = prime[i];
        t->prime[i] = pp;
        t->mods.q[i] = pp.q;
        t->mods.mu[i] = pp.mu;
        t->mods.k[i] = pp.k;
        int hl = 64 - (int)pp.k;
        if (hl > 6) hl = 6;
        if (hl < t->hl) t->hl = hl;
        PrimeDev& d = pd[i];
        d.q = pp.q;
        d.nq = 0ULL - pp.q;
        d.si = split_inv ? mulmod(split_inv[i], (pp.q >> 1) + 1, pp.q) : 0;       // (... and its inverse counterpart, times 2^-1)
        d.si_p = split_inv ? shoup(d.si, pp.q) : 0;
        d.sf = split_fwd ? split_fwd[i] : 0;                   // (n = 2^16 contexts: the stage that couples the two halves)
        d.sf_p = split_fwd ? shoup(split_fwd[i], pp.q) : 0;
        d.mu = pp.mu;
        d.k = pp.k;
        const unsigned g = pp.k - 1 < 16 ? pp.k - 1 : 16;
        d.red_sh1 = pp.k - 1 - g;
        d.red_sh2 = g;
        d.red_c = (u32)((((u128)1) << (31 + pp.k)) / pp.q);
        // near-2^k shape: q = 2^k - delta, k > 32, delta < 2^24 and 2^(64-k) * delta + 2 * delta < 2^k (reduce_2q_near)
        const u128 dl = (((u128)1) << pp.k) - pp.q;
        // ... and 2 delta^2 + 3 delta < 2^k (mul_fold_near: the fused products' fold multiplication)
        const bool near_ok = pp.k > 32 && dl < ((u128)1 << 24) && ((dl << (64 - pp.k)) + 2 * dl) < (((u128)1) << pp.k) &&
                             (2 * dl * dl + 3 * dl) < (((u128)1) << pp.k);
        d.delta = near_ok ? (u32)dl : 0;
        d.near_sh = pp.k > 32 ? pp.k - 32 : 0;
        d.near_mask = pp.k > 32 ? (u32)((1ull << (pp.k - 32)) - 1) : 0;
        d.lit = (literal && !pp.barrett_exact) ? 1u : 0u;
        d.twn[0] = TwPair{pp.ninv, shoup(pp.ninv, pp.q)};
        for (unsigned j = 1; j < 32; j++) {                   // (n >= 2048: the entries exist)
            const u64 w = mulmod(h_psiinv[(size_t)i * n + j], pp.ninv, pp.q);
            d.twn[j] = TwPair{w, shoup(w, pp.q)};
        }
        if (!near_ok) all_near = false;
    }
    if (all_near) t->hl |= 16;
    if (literal) t->hl = HL_LIT;                         // (the reference's own arithmetic: one class for every modulus size)
    if (split_fwd) {                                     // (n = 2^16 contexts: the device's pair flags exist before the first call)
        const hipError_t pe = pair_init_current_device();
        if (pe != hipSuccess) return pe;
    }
    const size_t words = (size_t)num_primes * n;
    // device layout: stage blocks transposed per round (ntt_core.cuh, tw_dev_index); entry 0 is never read
    std::vector<TwPair> hf(words), hi(words);
    const int logn = (int)t->log_n;
    const bool sp = logn >= 11 && logn <= 15;           // sizes served by the single-pass kernels
    for (unsigned i = 0; i < num_primes; i++) {
        const u64 q = prime[i].q;
        const size_t o = (size_t)i * n;
        for (unsigned j = 0; j < n; j++) {              // default: reference order (also the fallback for other n)
            hf[o + j] = TwPair{h_psi[o + j], shoup(h_psi[o + j], q)};
            hi[o + j] = TwPair{h_psiinv[o + j], shoup(h_psiinv[o + j], q)};
        }
        if (!sp) continue;
        const int nr = (logn + 4) / 5;
        for (int rho = 0; rho < nr; rho++) {
            // forward round rho: index bits top .. B
            {
                const int top = logn - 1 - 5 * rho, B = top - 4 > 0 ? top - 4 : 0;
                const unsigned nthi = (1u << (logn - 5)) >> B;
                for (int j = top - B; j >= 0; j--) {
                    const unsigned len = 1u << (logn - 1 - (B + j));
                    for (unsigned thi = 0; thi < nthi; thi++)
                        for (unsigned u = 0; u < (1u << (4 - j)) && ((thi << (4 - j)) + u) < len; u++) {
                            const size_t src = o + len + (thi << (4 - j)) + u;
                            hf[o + tw_dev_index(logn, B, j, len, thi, u)] = TwPair{h_psi[src], shoup(h_psi[src], q)};
                        }
                }
            }
            // inverse round rho: index bits low .. (B + 4)
            {
                const int low = 5 * rho, B = low < logn - 5 ? low : logn - 5;
                const unsigned nthi = (1u << (logn - 5)) >> B;
                for (int j = low - B; j <= 4; j++) {
                    const unsigned len = 1u << (logn - 1 - (B + j));
                    for (unsigned thi = 0; thi < nthi; thi++)
                        for (unsigned u = 0; u < (1u << (4 - j)) && ((thi << (4 - j)) + u) < len; u++) {
                            const size_t src = o + len + (thi << (4 - j)) + u;
                            hi[o + tw_dev_index(logn, B, j, len, thi, u)] = TwPair{h_psiinv[src], shoup(h_psiinv[src], q)};
                        }
                }
            }
        }
    }
    hipError_t e;
    if ((e = hipMalloc((void**)&t->d_fwd, words * sizeof(TwPair))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t->d_inv, words * sizeof(TwPair))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&t->d_primes_alloc, (num_primes + 1) * sizeof(PrimeDev))) != hipSuccess) return e;
    if ((e = hipMemset(t->d_primes_alloc, 0, sizeof(PrimeDev))) != hipSuccess) return e;          // guard words (kernels.hpp)
    t->d_primes = static_cast<PrimeDev*>(t->d_primes_alloc) + 1;
    if ((e = hipMemcpy(t->d_fwd, hf.data(), words * sizeof(TwPair), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(t->d_inv, hi.data(), words * sizeof(TwPair), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(t->d_primes, pd.data(), num_primes * sizeof(PrimeDev), hipMemcpyHostToDevice)) != hipSuccess) return e;
    return hipSuccess;
}

void fast_tables_destroy(FastTables* t)
{
    if (t->d_fwd) (void)hipFree(t->d_fwd);
    if (t->d_inv) (void)hipFree(t->d_inv);
    if (t->d_primes_alloc) (void)hipFree(t->d_primes_alloc);
    t->d_primes_alloc = nullptr;
    t->d_fwd = t->d_inv = nullptr;
    t->d_primes = nullptr;
}

hipError_t fast_clock_probe(const FastTables& t, hipStream_t s)
{
    if (!t.d_primes_alloc) return hipErrorInvalidValue;
    k_clock_probe<<<1, 64, 0, s>>>(static_cast<unsigned long long*>(t.d_primes_alloc));
    return hipGetLastError();
}

hipError_t fast_occupy(unsigned workgroups, unsigned microseconds, hipStream_t s)
{
    if (workgroups == 0) return hipSuccess;
    k_occupy<<<workgroups, 1024, 0, s>>>((unsigned long long)microseconds * 100ull, nullptr);
    return hipGetLastError();
}

hipError_t fast_probed_clock_mhz(const FastTables& t, double* mhz)
{
    *mhz = 0.0;
    if (!t.d_primes_alloc) return hipSuccess;
    unsigned long long w[2] = {0, 0};                // {shader cycles, 100 MHz ticks} of the last probe
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    if ((e = hipMemcpy(w, static_cast<const char*>(t.d_primes_alloc) + 16, sizeof(w), hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if (w[1] != 0) *mhz = (double)w[0] / (double)w[1] * 100.0;
    return hipSuccess;
}

// n = 2^15, a batch of k full rounds of the persistent grid plus a short tail (round 5): a persistent launch pays a whole extra
// iteration for the tail -- every workgroup that has no polynomial left idles while the others transform theirs -- so the tail runs on
// the small-batch kernels instead (kernels_lat.cuh: a polynomial spread over 64 waves), in a launch sequence of its own behind the
// head's.  Returns the number of polynomials of the head (a multiple of `division`: polynomial y' of the tail keeps prime
// y' % division), 0 = no split.  Pays for tails of up to ~100 polynomials (kernels.hpp, kTailSplitMax*): beyond that the extra
// persistent iteration costs no more than the small-batch launches.  MI355NTT_NO_TAIL_SPLIT=1 (A/B) and MI355NTT_LATENCY_PATH_MAX
// (forced paths: tests) switch it off.
static unsigned tail_split_head(const FastTables& t, unsigned num, unsigned division, bool fused)
{
    if (t.log_n != 15 || division == 0) return 0;
    static const bool off = std::getenv("MI355NTT_NO_TAIL_SPLIT") != nullptr || std::getenv("MI355NTT_LATENCY_PATH_MAX") != nullptr;
    if (off) return 0;
    const unsigned cap = current_device_cus();
    if (num <= cap || num % cap == 0) return 0;
    unsigned head = (num / cap) * cap;
    head -= head % division;
    const unsigned tail = num - head;
    return (head == 0 || tail > (fused ? kTailSplitMaxFused : kTailSplitMax)) ? 0 : head;
}

hipError_t fast_forward_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s)
{
    if (const unsigned head = tail_split_head(t, num, division, false)) {
        const hipError_t e = fast_forward_batch(t, d_a, head, division, prime_base, s);
        return e != hipSuccess ? e : fast_forward_batch(t, d_a + (size_t)head * t.n, num - head, division, prime_base, s);
    }
    const TwPair* tw = reinterpret_cast<const TwPair*>(t.d_fwd);
    const PrimeDev* pr = reinterpret_cast<const PrimeDev*>(t.d_primes);
    switch (t.log_n) {
    case 11: return fast_fwd_11(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 12: return fast_fwd_12(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 13: return fast_fwd_13(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 14: return fast_fwd_14(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 15: return fast_fwd_15(t.hl, d_a, tw, pr, num, division, prime_base, s);
    default:
        return compat_forward_batch(d_a, t.n, t.d_psi + (size_t)prime_base * t.n, num, division,
                                    shifted(t.mods, prime_base, division), s);
    }
}

// n = 2^16 as two half-size transforms per polynomial with the coupling stage fused into the lower half's loads (t holds the
// 2 P "virtual primes" of n/2 = 2^15; num, division, prime_base count full-size polynomials / real primes)
// (class 0 has no fused forms: its coupling stage is the literal stage kernel in memory)
bool fast_forward_split16_ok(const FastTables& t, unsigned num)
{
    return t.log_n == 15 && (t.hl & 15) != HL_LIT && fast_split_ok_16(num, 0, fast_fwd_pair_ok_16(t.hl));
}
bool fast_inverse_split16_ok(const FastTables& t, unsigned num, bool product)
{
    return t.log_n == 15 && (t.hl & 15) != HL_LIT && fast_split_ok_16(num, product ? 2 : 1, false);
}
hipError_t fast_forward_split16(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s)
{
    // two workgroups per polynomial (k_forward15_pair: 1 x / 1 x traffic) when this stream may own the device's pair flags
    hipError_t st = hipSuccess;
    if (PairSlot* slot = fast_fwd_pair_ok_16(t.hl) ? pair_acquire(s, &st) : nullptr) {
        const hipError_t e = fast_fwd_pair_16(t.hl, d_a, reinterpret_cast<const TwPair*>(t.d_fwd), reinterpret_cast<const PrimeDev*>(t.d_primes),
                                              num, division, prime_base, s, slot->d_flags);
        pair_release(slot, s);
        return e;
    }
    if (st != hipSuccess) return st;                     // (an earlier pair launch gave up on a partner: reported here, nothing launched)
    return fast_fwd_split_16(t.hl, d_a, reinterpret_cast<const TwPair*>(t.d_fwd), reinterpret_cast<const PrimeDev*>(t.d_primes), num,
                             division, prime_base, s);
}

hipError_t fast_inverse_split16(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s,
                                const u64* d_bhat)
{
    return fast_inv_split_16(t.hl, d_a, d_bhat, reinterpret_cast<const TwPair*>(t.d_inv), reinterpret_cast<const PrimeDev*>(t.d_primes), num,
                             division, prime_base, s);
}

hipError_t fast_inverse_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s)
{
    if (const unsigned head = tail_split_head(t, num, division, false)) {
        const hipError_t e = fast_inverse_batch(t, d_a, head, division, prime_base, s);
        return e != hipSuccess ? e : fast_inverse_batch(t, d_a + (size_t)head * t.n, num - head, division, prime_base, s);
    }
    const TwPair* tw = reinterpret_cast<const TwPair*>(t.d_inv);
    const PrimeDev* pr = reinterpret_cast<const PrimeDev*>(t.d_primes);
    switch (t.log_n) {
    case 11: return fast_inv_11(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 12: return fast_inv_12(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 13: return fast_inv_13(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 14: return fast_inv_14(t.hl, d_a, tw, pr, num, division, prime_base, s);
    case 15: return fast_inv_15(t.hl, d_a, tw, pr, num, division, prime_base, s);
    default:
        return compat_inverse_batch(d_a, t.n, t.d_psiinv + (size_t)prime_base * t.n, num, division,
                                    shifted(t.mods, prime_base, division), s);
    }
}

hipError_t fast_pointwise(const FastTables& t, u64* d_c, const u64* d_a, const u64* d_b, unsigned num, unsigned division,
                          hipStream_t s)
{
    unsigned gx = (t.n / 2 + 255) / 256;
    if (gx > 64) gx = 64;
    while ((unsigned long long)gx * num > 0x7fffffffull && gx > 1) gx /= 2;
    k_pointwise<<<gx * num, 256, 0, s>>>(d_c, d_a, d_b, t.n, division, reinterpret_cast<const PrimeDev*>(t.d_primes), gx);
    return hipGetLastError();
}

bool fast_polymul_epi_ok(const FastTables& t, unsigned num, unsigned division)
{
    static const bool off = std::getenv("MI355NTT_NO_FUSED_EPILOGUE") != nullptr;      // (A/B measurements)
    (void)division;
    return !(off || t.log_n != 15 || (t.hl & 15) == HL_LIT || num == 0);
}

// Head / tail cut of a fused product (tail_split_head) with the second operands of the tail: one per polynomial -> the same offset;
// shared by the batch -> the same `division` polynomials; shared per key group -> the group the tail lies in (one group: indexed
// from its start, no group arithmetic left), or, when the tail starts on a group boundary, the groups from there on.  0: no cut.
static unsigned polymul_split(const FastTables& t, unsigned num, unsigned division, bool shared_b, unsigned group, const u64* d_bhat,
                              const u64** b_tail, unsigned* group_tail)
{
    unsigned head = tail_split_head(t, num, division, true);
    *b_tail = d_bhat;
    *group_tail = group;
    if (!head) return 0;
    if (!shared_b) {
        *b_tail = d_bhat + (size_t)head * t.n;
    } else if (group) {
        const unsigned g0 = head / group, g1 = (num - 1) / group;
        if (g0 == g1) { *b_tail = d_bhat + (size_t)g0 * division * t.n; *group_tail = 0; }
        else if (head % group == 0) *b_tail = d_bhat + (size_t)g0 * division * t.n;
        else head = 0;                                       // (a tail across a group boundary that it does not start on: no cut)
    }
    return head;
}

hipError_t fast_polymul_batch_epi(const FastTables& t, int kind, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, hipStream_t s,
                                  bool shared_b, unsigned group, const u64* d_other, const void* d_consts)
{
    if (!fast_polymul_epi_ok(t, num, division)) return hipErrorNotSupported;
    if (shared_b && (division > kDivisionMask || group >= (kSharedB >> kSharedGroupShift) || (group && group % division))) return hipErrorInvalidValue;
    const u64* b_tail;
    unsigned group_tail;
    if (const unsigned head = polymul_split(t, num, division, shared_b, group, d_bhat, &b_tail, &group_tail)) {
        // the head on the persistent kernel (hipErrorNotSupported for a class that does not hold the epilogue: nothing has been launched
        // then and the caller runs the two steps), the tail on the small-batch kernels, which hold it for every class
        const hipError_t e = fast_polymul_batch_epi(t, kind, d_a, d_bhat, head, division, s, shared_b, group, d_other, d_consts);
        return e != hipSuccess ? e : fast_polymul_batch_epi(t, kind, d_a + (size_t)head * t.n, b_tail, num - head, division, s, shared_b, group_tail,
                                                            d_other + (size_t)head * t.n, d_consts);
    }
    if (shared_b) division |= kSharedB | (group << kSharedGroupShift);
    return fast_mul_epi_15(kind, t.hl, d_a, d_bhat, reinterpret_cast<const TwPair*>(t.d_fwd), reinterpret_cast<const TwPair*>(t.d_inv),
                           reinterpret_cast<const PrimeDev*>(t.d_primes), num, division, s, d_other, d_consts);
}

hipError_t fast_polymul_batch(const FastTables& t, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, hipStream_t s,
                              bool shared_b, unsigned group)
{
    const unsigned plain_division = division;
    if (shared_b && (division > kDivisionMask || group >= (kSharedB >> kSharedGroupShift) || (group && group % division))) return hipErrorInvalidValue;
    const u64* b_tail;
    unsigned group_tail;
    if (const unsigned head = polymul_split(t, num, division, shared_b, group, d_bhat, &b_tail, &group_tail)) {
        const hipError_t e = fast_polymul_batch(t, d_a, d_bhat, head, division, s, shared_b, group);
        return e != hipSuccess ? e : fast_polymul_batch(t, d_a + (size_t)head * t.n, b_tail, num - head, division, s, shared_b, group_tail);
    }
    if (shared_b) division |= kSharedB | (group << kSharedGroupShift);
    const TwPair* twf = reinterpret_cast<const TwPair*>(t.d_fwd);
    const TwPair* twi = reinterpret_cast<const TwPair*>(t.d_inv);
    const PrimeDev* pr = reinterpret_cast<const PrimeDev*>(t.d_primes);
    switch (t.log_n) {   // (n = 2^15 switches to a three-launch latency path for few polynomials by itself: launch_mul)
    case 11: return fast_mul_11(t.hl, d_a, d_bhat, twf, twi, pr, num, division, s);
    case 12: return fast_mul_12(t.hl, d_a, d_bhat, twf, twi, pr, num, division, s);
    case 13: return fast_mul_13(t.hl, d_a, d_bhat, twf, twi, pr, num, division, s);
    case 14: return fast_mul_14(t.hl, d_a, d_bhat, twf, twi, pr, num, division, s);
    case 15: return fast_mul_15(t.hl, d_a, d_bhat, twf, twi, pr, num, division, s);
    default: break;
    }
    if (shared_b) return hipErrorNotSupported;           // (sizes without a fused kernel: the callers compose the three steps)
    hipError_t e = fast_forward_batch(t, d_a, num, plain_division, 0, s);
    if (e != hipSuccess) return e;
    e = fast_pointwise(t, d_a, d_a, d_bhat, num, plain_division, s);
    if (e != hipSuccess) return e;
    return fast_inverse_batch(t, d_a, num, plain_division, 0, s);
}

}  // namespace mi355ntt
