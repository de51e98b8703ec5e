// threads_test.cpp -- the header's thread-safety contract (include/mi355ntt.h: "no global mutable state ... an immutable context
// object, usable from any host thread") exercised from compiled C++: THREADS std::threads share ONE context (a stream each) and
// call the library at the same time.  Four sections, every result compared word for word with the CPU oracle's, computed up front
// (the oracle is the checker here, as everywhere under tests/):
//   1. n = 2^15, four 60-bit primes: the persistent single-pass kernels (192 polynomials per call) and the small-batch path (8)
//   2. the reference's own decryption_test.cu moduli, n = 4096: a class-0 context -- prime 1 is Barrett-inexact: its polynomials take the
//      reference's own butterflies, the others the lazy ones, in one launch (kernels_lit.cuh; until round 5 a gather buffer that the
//      streams handed over by an event)
//   3. n = 2^16, forward over 96 polynomials: the cooperating-workgroup launch whose flag buffer belongs to one stream at a time
//      (pair_acquire): the other threads' calls must fall back to the single-workgroup kernel, with the same words
//   4. the reference-signature raw API on 40 distinct tables: more than the 32 cached contexts, so entries are evicted and
//      re-derived while other threads use the cache (the mutex-guarded LRU of capi.cpp); every thread calls on a stream of its own and
//      destroys it when it is done, while the others still call with the same tables (the entries keep a guard record per stream)
// tests/test_threads.py builds and runs it.   usage: threads_test [threads = 8] [iterations = 4]
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mi355ntt.h"
#include "../../oracle/ntt_oracle.h"

typedef unsigned long long u64;

#define HIPCK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); fails++; return; } } while (0)
#define RCCK(x) do { int rc_ = (x); if (rc_) { printf("mi355ntt error %d (%s) line %d\n", rc_, mi355ntt_strerror(rc_), __LINE__); fails++; return; } } while (0)

static std::atomic<long> fails{0};
static unsigned THREADS = 8, ITERS = 4;

static u64 primitive_root_2n(u64 q, unsigned n)
{
    for (u64 c = 2;; c++) {
        const u64 x = mi355ntt_modpow(c, (q - 1) / (2ull * n), q);
        if (mi355ntt_modpow(x, n, q) == q - 1) return x;
    }
}

struct Tables {
    std::vector<u64> psi, psiinv;      // [P][n]
    std::vector<u64> q, mu;
    std::vector<unsigned> k;
};
static Tables make_tables(unsigned n, const std::vector<u64>& qs, const std::vector<u64>& psis)
{
    Tables t;
    const size_t P = qs.size();
    t.psi.resize(P * n); t.psiinv.resize(P * n); t.q = qs; t.mu.resize(P); t.k.resize(P);
    for (size_t i = 0; i < P; i++) {
        t.k[i] = orc_bit_length(qs[i]);
        t.mu[i] = orc_mu(qs[i], t.k[i]);
        orc_fill_table(psis[i], qs[i], t.psi.data() + i * n, n);
        orc_fill_table(orc_modinv(psis[i], qs[i]), qs[i], t.psiinv.data() + i * n, n);
    }
    return t;
}
static std::vector<u64> synth(unsigned n, unsigned num, const std::vector<u64>& qs, u64 seed)
{
    std::vector<u64> a((size_t)num * n);
    for (unsigned y = 0; y < num; y++) orc_splitmix_fill(a.data() + (size_t)y * n, n, seed + y, qs[y % qs.size()]);
    return a;
}
// polynomial y of the result is polynomial (y + shift) % num of src (shift a multiple of the prime count: same moduli)
static std::vector<u64> rotated(const std::vector<u64>& src, unsigned n, unsigned num, unsigned shift)
{
    std::vector<u64> r(src.size());
    for (unsigned y = 0; y < num; y++) memcpy(r.data() + (size_t)y * n, src.data() + (size_t)((y + shift) % num) * n, (size_t)n * 8);
    return r;
}
static long mismatches(const std::vector<u64>& a, const std::vector<u64>& b, size_t words)
{
    long m = 0;
    for (size_t i = 0; i < words; i++) m += a[i] != b[i];
    return m;
}

// ---- sections 1-3: a shared context, forward / inverse batches against the oracle -------------------------------------------------
static void context_section(const char* name, unsigned n, const std::vector<u64>& qs, const std::vector<u64>& psis, unsigned num, unsigned small,
                            int expect_routing)
{
    const unsigned P = (unsigned)qs.size();
    mi355ntt_ctx* ctx = nullptr;
    RCCK(mi355ntt_ctx_create(&ctx, n, P, qs.data(), psis.data(), 0));
    if (expect_routing >= 0 && mi355ntt_ctx_uses_literal_kernels(ctx) != expect_routing) {
        printf("%s: routing %d, expected %d\n", name, mi355ntt_ctx_uses_literal_kernels(ctx), expect_routing);
        fails++;
    }
    const Tables tb = make_tables(n, qs, psis);
    const std::vector<u64> base = synth(n, num, qs, 31337);
    std::vector<u64> fwd = base;
    orc_forward_batch(fwd.data(), n, tb.psi.data(), num, P, tb.q.data(), tb.mu.data(), tb.k.data(), 8);
    std::vector<u64> back = fwd;                       // (the inverse of the reference's forward words: for exact primes the input again)
    orc_inverse_batch(back.data(), n, tb.psiinv.data(), num, P, tb.q.data(), tb.mu.data(), tb.k.data(), 8);
    const long before = fails.load();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < THREADS; t++) {
        th.emplace_back([&, t]() {
            const unsigned shift = (t * P * 3) % num;
            const std::vector<u64> in = rotated(base, n, num, shift), want_f = rotated(fwd, n, num, shift), want_b = rotated(back, n, num, shift);
            const size_t words = (size_t)num * n, small_words = (size_t)small * n;
            std::vector<u64> got(words);
            u64 *d = nullptr, *d_small = nullptr;
            hipStream_t s;
            HIPCK(hipSetDevice(0));
            HIPCK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            HIPCK(hipMalloc(&d, words * 8));
            HIPCK(hipMalloc(&d_small, small_words * 8 + 8));
            for (unsigned it = 0; it < ITERS; it++) {
                HIPCK(hipMemcpyAsync(d, in.data(), words * 8, hipMemcpyHostToDevice, s));
                HIPCK(hipMemcpyAsync(d_small, in.data(), small_words * 8, hipMemcpyHostToDevice, s));
                RCCK(mi355ntt_forward_batch(ctx, d, num, P, s));
                RCCK(mi355ntt_forward_batch(ctx, d_small, small, P, s));            // (the small-batch path next to the large one)
                HIPCK(hipMemcpyAsync(got.data(), d, words * 8, hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                long m = mismatches(got, want_f, words);
                HIPCK(hipMemcpyAsync(got.data(), d_small, small_words * 8, hipMemcpyDeviceToHost, s));
                RCCK(mi355ntt_inverse_batch(ctx, d, num, P, s));
                RCCK(mi355ntt_inverse_batch(ctx, d_small, small, P, s));
                HIPCK(hipStreamSynchronize(s));
                m += mismatches(got, want_f, small_words);
                HIPCK(hipMemcpyAsync(got.data(), d, words * 8, hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                m += mismatches(got, want_b, words);
                HIPCK(hipMemcpyAsync(got.data(), d_small, small_words * 8, hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                m += mismatches(got, want_b, small_words);
                if (m) { printf("%s: thread %u iteration %u: %ld words differ from the oracle\n", name, t, it, m); fails += m; }
            }
            (void)hipFree(d); (void)hipFree(d_small); (void)hipStreamDestroy(s);
        });
    }
    for (auto& x : th) x.join();
    RCCK(mi355ntt_ctx_destroy(ctx));
    printf("%-58s %u threads x %u iterations: %s\n", name, THREADS, ITERS, fails.load() == before ? "ok" : "FAILED");
}

// ---- section 4: the raw API on more tables than the cache holds ---------------------------------------------------------------------
static void raw_section()
{
    const unsigned n = 2048, NT = 40;
    u64 q, psi, psiinv, ninv;
    unsigned bits;
    RCCK(mi355ntt_get_params(n, &q, &psi, &psiinv, &ninv, &bits));
    const u64 mu = mi355ntt_barrett_mu(q, bits);
    std::vector<u64*> d_tab(NT, nullptr);
    std::vector<std::vector<u64>> want(NT);
    std::vector<u64> in(n);
    orc_splitmix_fill(in.data(), n, 99, q);
    for (unsigned k = 0; k < NT; k++) {
        std::vector<u64> tab(n);
        orc_fill_table(mi355ntt_modpow(psi, 2 * k + 1, q), q, tab.data(), n);       // psi^(odd): another primitive 2n-th root, another table
        HIPCK(hipMalloc(&d_tab[k], (size_t)n * 8));
        HIPCK(hipMemcpy(d_tab[k], tab.data(), (size_t)n * 8, hipMemcpyHostToDevice));
        want[k] = in;
        orc_forward(want[k].data(), n, q, mu, bits, tab.data());
    }
    RCCK(mi355ntt_raw_cache_clear());
    const long before = fails.load();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < THREADS; t++) {
        th.emplace_back([&, t]() {
            u64* d = nullptr;
            hipStream_t s;
            std::vector<u64> got(n);
            HIPCK(hipSetDevice(0));
            HIPCK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            HIPCK(hipMalloc(&d, (size_t)n * 8));
            for (unsigned it = 0; it < ITERS * NT; it++) {
                const unsigned k = (t * 7 + it * 3) % NT;
                HIPCK(hipMemcpyAsync(d, in.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
                RCCK(mi355ntt_forward_raw(d, n, s, q, mu, (int)bits, d_tab[k]));
                HIPCK(hipMemcpyAsync(got.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                const long m = mismatches(got, want[k], n);
                if (m) { printf("raw: thread %u call %u table %u: %ld words differ from the oracle\n", t, it, k, m); fails += m; }
            }
            (void)hipFree(d); (void)hipStreamDestroy(s);
        });
    }
    for (auto& x : th) x.join();
    RCCK(mi355ntt_raw_cache_clear());
    for (auto p : d_tab) (void)hipFree(p);
    printf("%-58s %u threads x %u calls: %s\n", "raw API, 40 tables through a cache of 32", THREADS, ITERS * NT, fails.load() == before ? "ok" : "FAILED");
}

// a crash must not be silent: where it happened goes to stderr (the Python side prints it with the assertion)
static void on_fatal_signal(int sig)
{
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "threads_test: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char** argv)
{
    signal(SIGSEGV, on_fatal_signal);
    signal(SIGBUS, on_fatal_signal);
    signal(SIGABRT, on_fatal_signal);
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (argc > 1) THREADS = (unsigned)atoi(argv[1]);
    if (argc > 2) ITERS = (unsigned)atoi(argv[2]);
    const std::vector<u64> q60 = {1152921504606584833ULL, 1152921504598720513ULL, 1152921504597016577ULL, 1152921504595968001ULL};
    const std::vector<u64> psi60 = {4443670208963ULL, 100545759574150ULL, 31693996050849ULL, 88651361085495ULL};
    context_section("n = 32768, 4 x 60 bits: persistent + small-batch kernels", 32768, q60, psi60, 192, 8, 0);
    // decryption_test.cu:47-48 (primes 0 and 2 exact, prime 1 not): per-prime routing = 2; 900 polynomials = several polynomials of each kind per workgroup
    context_section("n = 4096, decryption_test.cu moduli: mixed context", 4096, {68719403009ULL, 68719230977ULL, 137438822401ULL},
                    {24250113ULL, 29008497ULL, 8625844ULL}, 900, 6, 2);
    const std::vector<u64> q16 = {q60[0], q60[1]};
    context_section("n = 65536, 2 x 60 bits: pair launches (one stream at a time)", 65536, q16,
                    {primitive_root_2n(q16[0], 65536), primitive_root_2n(q16[1], 65536)}, 96, 2, 0);
    raw_section();
    printf("errors = %ld\n", fails.load());
    return fails.load() ? 1 : 0;
}
