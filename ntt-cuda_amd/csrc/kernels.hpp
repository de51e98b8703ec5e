// kernels.hpp -- host-callable launchers shared between the C ABI and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "hostparams.hpp"

namespace mi355ntt {

constexpr unsigned kMaxPrimes = 16;
// n = 2^15: tails of at most this many polynomials behind full rounds of the persistent grid run on the small-batch kernels
// (kernels_fast.hip, tail_split_head; measured, tools/probe/tail_split_ab.py: fused product 195 against 229 us at 512 + 32 polynomials,
// 207 / 229 at + 64, 221 / 234 at + 96, no gain from + 128 on; forward + inverse pairs 198 / 212, 214 / 223, 221 / 224)
constexpr unsigned kTailSplitMax = 96, kTailSplitMaxFused = 100;

// compute units of the CURRENT device (every launching entry point has switched to the context's device: device_scope.hpp);
// sizes the persistent grids.  Cached per host thread and device; 256 (MI355X) only if the query itself fails.
inline unsigned current_device_cus()
{
    static thread_local int cached_dev = -1;
    static thread_local unsigned cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != cached_dev) {
        int v = 0;
        cached_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (unsigned)v : 256u;
        cached_dev = dev;
    }
    return cached_cus;
}

// Per-launch copy of the moduli: replaces the reference's __constant__ q_cons / mu_cons / q_bit_cons
// (ntt_60bit.cuh:8-10).  Passed by value in the kernel argument segment (SGPR-resident).
struct ModSet {
    u64 q[kMaxPrimes];
    u64 mu[kMaxPrimes];
    unsigned k[kMaxPrimes];
};

// Checked raw calls (capi.cpp): the caller's table is compared with the cached context's on the device in front of every
// transform.  Guard words g = {current epoch, epoch of the last mismatch}: the throughput kernels launched with
// kGuardBit set in prime_base return at once when g[0] == g[1] (this call's table is not the cached one), the literal
// kernels launched with a guard pointer return at once when g[0] != g[1] -- exactly one of the two transforms the data.
constexpr unsigned kGuardBit = 0x80000000u;
// fused products: OR-ed into `division`, the second operand holds `division` polynomials shared by the whole batch
// (polynomial y multiplies with bhat[y % division]) -- the batched BFV drivers multiply every ciphertext with the same key
constexpr unsigned kSharedB = 0x80000000u;
// with kSharedB: bits 8..30 of the division word = polynomials per key group (0: one group); polynomial y then multiplies
// with bhat[(y / group) * division + y % division] -- the two components of a batch of ciphertexts in one launch
constexpr unsigned kSharedGroupShift = 8, kDivisionMask = 0xffu;
// k_inverse15 only: bit 30 of the division word = "this launch streams" -- the batch is several times the memory-side cache
// (kInvStreamLoadsMin polynomials of 256 KiB = 1 GiB), so the 16-byte row loads carry the non-temporal hint (kernels_fast_impl.cuh,
// MI355NTT_INV15_AUX_LD; measured there)
constexpr unsigned kStreamLoads = 0x40000000u, kInvStreamLoadsMin = 4096;
// k_polymul15, second operands that are NOT shared (with kSharedB bit 30 belongs to the group field): the same bit = "a and bhat together
// exceed the memory-side cache" -- more than kMulStreamLoadsAbove polynomials -- and the bhat loads carry the non-temporal hint
constexpr unsigned kMulStreamLoadsAbove = 512;

// ---- literal stage-per-launch kernels (kernels_compat.hip) ----
hipError_t compat_forward_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s, const unsigned* guard = nullptr);
hipError_t compat_inverse_batch(u64* d_a, unsigned n, const u64* d_tabs, unsigned num, unsigned division, const ModSet& m,
                                hipStream_t s, const unsigned* guard = nullptr);
// one stage of the above (stage `length` of the reference's loop) over the batch
hipError_t compat_ct_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s);
hipError_t compat_gs_stage(u64* d_a, unsigned n, const u64* d_tabs, unsigned length, unsigned num, unsigned division, const ModSet& m,
                           hipStream_t s);
hipError_t compat_pointwise(u64* d_c, const u64* d_a, const u64* d_b, unsigned n, unsigned num, unsigned division,
                            const ModSet& m, hipStream_t s, bool shared_b = false, unsigned group = 0);
hipError_t compat_pointwise_scalar(u64* d_a, u64 b, unsigned n, u64 q, u64 mu, unsigned k, hipStream_t s);
// the element-wise wrappers of poly_arithmetic.cuh:312-352, reference arithmetic word for word (kernels_compat.hip): d_a[i] <- op(d_a[i], d_b[i] or scalar)
enum { kEwAdd = 0, kEwAddInteger = 1, kEwSub = 2, kEwNegate = 3, kEwMulIntT = 4 };
hipError_t compat_elementwise(int op, u64* d_a, const u64* d_b, u64 scalar, u64 q_or_t, size_t count, hipStream_t s);
// a[y][i] = splitmix64(seed_base + y)_i mod q[y % division]: the synthetic inputs of SURVEY.md 4.2 / 8d, generated on the device
hipError_t compat_synth_splitmix(u64* d_a, unsigned n, unsigned num, unsigned division, const ModSet& m, u64 seed_base, hipStream_t s);
// *d_flag |= 1 when two sets of `count` reference-format tables differ in an entry the transforms read (index != 0)
hipError_t compat_tables_differ(const u64* d_x, const u64* d_y, unsigned n, unsigned count, unsigned* d_flag, hipStream_t s);
// the same as a stream-ordered check: guard[0] = epoch, and guard[1] = epoch when the tables differ
// (d_host_word: device address of a host-mapped word that is set to 1 when they differ, or null)
hipError_t compat_tables_check(const u64* d_x, const u64* d_y, unsigned n, unsigned count, unsigned* d_guard, unsigned epoch, hipStream_t s,
                               unsigned* d_host_word = nullptr);

// ---- throughput kernels (kernels_fast.hip) ----
// Device tables private to the fast path.  Built once per context from the reference-format tables.
struct FastTables {
    unsigned n = 0, log_n = 0, num_primes = 0;
    PrimeParams prime[kMaxPrimes];
    ModSet mods;
    // [P][n] pairs {w, floor(w*2^64/q)}: twiddle and its Shoup companion, interleaved so one 16-byte
    // load fetches both.  Same indexing as the reference tables (entry length+p for stage `length`).
    u64* d_fwd = nullptr;      // psi^bitrev(i)
    u64* d_inv = nullptr;      // psi^-bitrev(i)
    void* d_primes = nullptr;  // [P] PrimeDev records (ntt_core.cuh); the record BEFORE it holds the guard words {current epoch,
                               // epoch of the last table mismatch} of the checked raw calls (kGuardBit, capi.cpp)
    void* d_primes_alloc = nullptr;
    int hl = 6;                // bits 0-3: headroom class = min over primes of (64 - bit length), capped at 6; bit 4: all primes near 2^k;
                               // 0 (HL_LIT): the reference's literal arithmetic in the single-pass kernels
    const u64* d_psi = nullptr;     // reference-format tables owned by the context (fallback path)
    const u64* d_psiinv = nullptr;
};

// split_fwd (n = 2^16 contexts only, else null): per (virtual) prime the twiddle of the stage that couples the two halves
// literal: kernel class HL_LIT (hl = 0) -- the transforms of this context return the reference's own words (singleBarrett with one
// conditional subtraction carried through the stages, ntt_core.cuh), for contexts with a Barrett-inexact modulus
size_t fast_prime_record_bytes();     // sizeof(PrimeDev): d_primes_alloc holds num_primes + 1 of them, [0] = guard record
hipError_t fast_tables_create(FastTables* t, unsigned n, unsigned num_primes, const PrimeParams* prime, const u64* h_psi,
                              const u64* h_psiinv, const u64* d_psi, const u64* d_psiinv, const u64* split_fwd = nullptr,
                              const u64* split_inv = nullptr, bool literal = false);
// n = 2^16 forward as two launches over half-size transforms, the coupling stage fused into the first (1.5 instead of 2 passes
// over memory); _ok: the batch is large enough for the persistent kernels
bool fast_forward_split16_ok(const FastTables& t, unsigned num);
bool fast_inverse_split16_ok(const FastTables& t, unsigned num, bool product);   // (product: forward + inverse-with-factor launches)
hipError_t fast_forward_split16(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s);
// (same condition; d_bhat: null, or a factor in the NTT domain -- one polynomial per polynomial of d_a -- multiplied in on the way in)
hipError_t fast_inverse_split16(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s,
                                const u64* d_bhat = nullptr);
void fast_tables_destroy(FastTables* t);
// measurement helper: a stream-ordered clock probe (20 us) and the shader clock it measured (the read synchronises)
hipError_t fast_clock_probe(const FastTables& t, hipStream_t s);
hipError_t fast_probed_clock_mhz(const FastTables& t, double* mhz);
// test / measurement helper: `workgroups` whole CUs held for `microseconds` (k_occupy), stream-ordered, touches no memory
hipError_t fast_occupy(unsigned workgroups, unsigned microseconds, hipStream_t s);
// polynomial y of the batch uses prime (prime_base + y % division)
hipError_t fast_forward_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s);
hipError_t fast_inverse_batch(const FastTables& t, u64* d_a, unsigned num, unsigned division, unsigned prime_base, hipStream_t s);
hipError_t fast_pointwise(const FastTables& t, u64* d_c, const u64* d_a, const u64* d_b, unsigned num, unsigned division,
                          hipStream_t s);
hipError_t fast_polymul_batch(const FastTables& t, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, hipStream_t s,
                              bool shared_b = false, unsigned group = 0);
// The fused product of n = 2^15 with a per-prime element-wise epilogue in its store path (kernels_epi.cuh; the batched BFV drivers):
// kind 1: a <- (a bhat + other, `>`) k1 mod q (k2 = floor(k1 2^64 / q), the Shoup companion) for the primes whose record is `on` (decryption, bfv_decryption.cuh:98-122; the only kind
// built -- the encryption's `+ e` was measured slower fused than in k_encrypt_tail: profiles/r06_bfv_batch.txt).  other: a buffer of the shape of d_a; d_consts: `division` records
// {u64 k1, k2; unsigned on, pad} on the device.  _ok: ring degree and class take the fused form at all (else, and when fast_polymul_batch_epi
// returns hipErrorNotSupported -- a class whose persistent kernel does not hold the epilogue --, the callers run the product and the
// element-wise kernel one after the other).  A batch of k full rounds of the grid plus a short tail is cut as fast_polymul_batch cuts it.
bool fast_polymul_epi_ok(const FastTables& t, unsigned num, unsigned division);
hipError_t fast_polymul_batch_epi(const FastTables& t, int kind, u64* d_a, const u64* d_bhat, unsigned num, unsigned division, hipStream_t s,
                                  bool shared_b, unsigned group, const u64* d_other, const void* d_consts);

// ---- device-wide ownership of the pair flags (kernels_fast.hip): one pair kernel -- 60-bit k_forward15_pair or 30-bit k_ntt30x PAIR --
// in flight per device.  pair_acquire: the slot when stream s may launch a pair kernel now (not capturing, every CU available to it, no
// other stream's pair launch still in flight), else null: the caller runs its single-workgroup / stage-launch form.  Between
// pair_acquire and pair_release (which records the launch on s) the caller holds a mutex: launch and nothing else.
struct PairSlot;
constexpr unsigned kPairFlagWords = 2048;               // zeroed 32-bit words; flags[2 pair + role], grid <= 1024 workgroups
hipError_t pair_init_current_device();                  // allocates the current device's slot (idempotent)
// status (may be null): hipErrorLaunchFailure when an EARLIER pair launch on this device gave up on a partner (see the watchdog below) --
// the slot is then cleaned (flags zeroed in stream order) and null returned: the caller reports the error instead of launching
PairSlot* pair_acquire(hipStream_t s, hipError_t* status = nullptr);
void pair_release(PairSlot* slot, hipStream_t s);
unsigned* pair_flags(PairSlot* slot);
unsigned long long pair_fault_count(int device);        // launches that gave up on a partner on this device so far (never reset)
// Watchdog of the spinning workgroups, in ticks of the 100 MHz constant clock (s_memrealtime): a partner that has not become resident
// after this long means another tenant holds the CUs indefinitely.  The workgroup then GIVES UP instead of hanging (or trapping, which
// would take the process's device context with it): it sets the last word of the flag buffer -- a workgroup of this or of a later
// pair launch that finds it set while waiting for its own partner ends after 100 us instead of 30 s -- and the host-mapped error word; the library reports
// MI355NTT_EHIP (hipErrorLaunchFailure) from the next call that wants the pair slot and cleans it.  The data of the launch that
// gave up is invalid (as after any asynchronous device fault).  MI355NTT_PAIR_WATCHDOG_MS in the environment shortens it (tests).
constexpr unsigned long long kPairWatchdogTicks = 30ull * 100000000ull;      // 30 s
// tail of the flag buffer (grids use at most kPairFlagWords / 2 words): what a workgroup that gives up needs, written once by the host
constexpr unsigned kPairDeadWord = kPairFlagWords - 1;        // != 0: a workgroup of this or an earlier launch gave up
constexpr unsigned kPairErrPtrWord = kPairFlagWords - 4;      // u64: device pointer of the host-mapped error word
constexpr unsigned kPairTicksWord = kPairFlagWords - 6;       // u64: the watchdog in 100 MHz ticks
constexpr unsigned kPairLiveWords = kPairFlagWords - 8;       // words [0, kPairLiveWords) are the flags proper
unsigned long long pair_watchdog_ticks();
#ifdef __HIPCC__
// device side: the poll loop's slow path, on the SCALAR unit (s_load ... glc: coherent, no VGPR): formed as vector loads the three
// tail addresses are loop invariants, hoisted out of the polynomial loop into VGPR pairs that stay live through every round --
// measured: k_forward15_pair went from 112-120 VGPRs without scratch to 128 VGPRs with 24-52 bytes of it.
__device__ __forceinline__ unsigned pair_tail_u32(const unsigned* flags, unsigned word)
{
    unsigned v;
    asm volatile("s_load_dword %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(flags), "s"(word * 4u) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long pair_tail_u64(const unsigned* flags, unsigned word)
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(flags), "s"(word * 4u) : "memory");
    return v;
}
__device__ __forceinline__ bool pair_launch_is_dead(const unsigned* flags) { return pair_tail_u32(flags, kPairDeadWord) != 0; }
// called from the poll loop: ENDS THE WAVE (s_endpgm: noreturn, like the trap it replaces) when the launch is dead or the watchdog has
// expired
// (the scalar loads go to L2 and wait: microseconds under load.  A poll loop therefore calls this only once it has been waiting for
// kPairSlowPollTicks -- until then it costs what it cost with the trap, one s_memrealtime per iteration; polled every iteration the
// pair forward lost a third: 0.280 against 0.192 ms per 512 polynomials)
constexpr unsigned long long kPairSlowPollTicks = 10000ull;      // 100 us
__device__ __forceinline__ void pair_watchdog_check(unsigned* flags, unsigned long long t0)
{
    if (__builtin_amdgcn_s_memrealtime() - t0 <= kPairSlowPollTicks) return;
    if (pair_launch_is_dead(flags)) __builtin_amdgcn_endpgm();
    if (__builtin_amdgcn_s_memrealtime() - t0 <= pair_tail_u64(flags, kPairTicksWord)) return;
    unsigned* err = reinterpret_cast<unsigned*>(pair_tail_u64(flags, kPairErrPtrWord));
    __hip_atomic_store(flags + kPairDeadWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(err, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_endpgm();
}
#endif

// ---- the reference's 30-bit path (kernels_ntt30.hip): 32-bit words, single prime, `num` polynomials of n words ----
// ninv_native: m^-1 mod q (m = n, or n / 2 at n = 2^16) when the call may run the native kernels, 0 = literal kernels only
hipError_t ntt30_forward(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, unsigned ninv_native,
                         hipStream_t s);
hipError_t ntt30_inverse(unsigned* d_a, unsigned n, const unsigned* d_tab, unsigned num, unsigned q, unsigned mu, int bits, unsigned ninv_native,
                         hipStream_t s);
hipError_t ntt30_barrett(unsigned* d_a, const unsigned* d_b, size_t count, unsigned q, unsigned mu, int bits, hipStream_t s);

}  // namespace mi355ntt
