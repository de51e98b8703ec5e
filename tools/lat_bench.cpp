// lat_bench.cpp -- latencies of the small-batch paths timed from compiled code through the C ABI only (no Python in the
// launch path): HIP events around back-to-back calls, the way the reference's published figures were taken
// (Article.pdf p25 Table 6: one polynomial, n = 2^15: 39 us NTT / 23 us INTT on V100; p26 Table 7: BFV drivers).
//
//   hipcc -std=c++17 -O2 -x hip --offload-arch=gfx950 tools/lat_bench.cpp -x none -L ntt-cuda_amd -lmi355ntt \
//         -Wl,-rpath,$PWD/ntt-cuda_amd -o ntt-cuda_amd/build/lat_bench        (the package Makefile does this: `make lat_bench`)
//   ./lat_bench [calls=200] [rounds=7] [n=32768]      -> one JSON object on stdout (bench.py runs it and embeds the object)
//
// "stream" figures: `calls` back-to-back calls on one stream between two events (launches pipeline; per-call time is the
// steady-state cost of one call).  "graph" figures: the same calls captured once into a hipGraph (stream capture of the C ABI
// calls -- every entry point is capture-safe: no allocation, no synchronisation, no host read after context creation) and the
// graph launched between two events.  "sync" figures: one call + stream synchronise, wall clock (the latency a caller sees).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../include/mi355ntt.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define RC(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "mi355ntt error %d at line %d\n", r_, __LINE__); exit(3); } } while (0)

typedef unsigned long long u64;

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double vmin(const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); }

struct Timer {
    hipStream_t s;
    hipEvent_t e0, e1;
    int calls, rounds;
    // per-call microseconds: median and min over `rounds` samples of `calls` back-to-back calls
    void stream(const std::function<void()>& f, double& med, double& mn)
    {
        for (int i = 0; i < calls; i++) f();                      // warm (clocks, code objects)
        std::vector<double> t;
        for (int r = 0; r < rounds; r++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < calls; i++) f();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3 / calls);
        }
        med = median(t); mn = vmin(t);
    }
    // the same calls as one captured graph of `per_graph` calls, launched calls / per_graph times per sample
    void graph(const std::function<void()>& f, int per_graph, double& med, double& mn)
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < per_graph; i++) f();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const int launches = std::max(1, calls / per_graph);
        for (int i = 0; i < launches; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<double> t;
        for (int r = 0; r < rounds; r++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < launches; i++) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3 / (launches * per_graph));
        }
        med = median(t); mn = vmin(t);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // one call, then wait for it: wall-clock microseconds (median / min over calls)
    void sync(const std::function<void()>& f, double& med, double& mn)
    {
        std::vector<double> t;
        for (int i = 0; i < calls + 20; i++) {
            CK(hipStreamSynchronize(s));
            const auto a = std::chrono::steady_clock::now();
            f();
            CK(hipStreamSynchronize(s));
            const auto b = std::chrono::steady_clock::now();
            if (i >= 20) t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
        }
        med = median(t); mn = vmin(t);
    }
};

static u64 splitmix(u64& x) { u64 z = (x += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

int main(int argc, char** argv)
{
    Timer T;
    T.calls = argc > 1 ? atoi(argv[1]) : 200;
    T.rounds = argc > 2 ? atoi(argv[2]) : 7;
    const unsigned n = argc > 3 ? (unsigned)atoi(argv[3]) : 32768;      // ring degree (2048 .. 32768); the BFV part runs at any of them
    // BASELINE configs[1]: one 60-bit prime; configs[4]: 4 x 60-bit + the special prime
    const u64 q5[5] = {1152921504606584833ULL, 1152921504598720513ULL, 1152921504597016577ULL, 1152921504595968001ULL, 1152921504595640321ULL};
    u64 psi5[5] = {4443670208963ULL, 100545759574150ULL, 31693996050849ULL, 88651361085495ULL, 9679305630873ULL};
    if (n != 32768)             // a primitive 2n-th root for the smaller ring: psi_32768^(32768 / n)
        for (int i = 0; i < 5; i++) psi5[i] = mi355ntt_modpow(psi5[i], 32768 / n, q5[i]);
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&T.s));
    CK(hipEventCreate(&T.e0)); CK(hipEventCreate(&T.e1));

    mi355ntt_ctx* ctx = nullptr;
    RC(mi355ntt_ctx_create(&ctx, n, 1, q5, psi5, 0));
    std::vector<u64> h(n);
    u64 seed = 1;
    for (auto& v : h) v = splitmix(seed) % q5[0];
    u64 *d_a, *d_b;
    CK(hipMalloc(&d_a, n * 8)); CK(hipMalloc(&d_b, n * 8));
    CK(hipMemcpy(d_a, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h.data(), n * 8, hipMemcpyHostToDevice));

    auto fwd = [&] { RC(mi355ntt_forward(ctx, d_a, 0, T.s)); };
    auto inv = [&] { RC(mi355ntt_inverse(ctx, d_a, 0, T.s)); };
    auto pair = [&] { fwd(); inv(); };
    u64* d_p;                                                  // the fused product works on a buffer of its own
    CK(hipMalloc(&d_p, n * 8));
    CK(hipMemcpy(d_p, h.data(), n * 8, hipMemcpyHostToDevice));
    auto mul = [&] { RC(mi355ntt_polymul_batch(ctx, d_p, d_b, 1, 1, T.s)); };
    double m[12][2];
    T.stream(fwd, m[0][0], m[0][1]);
    T.stream(inv, m[1][0], m[1][1]);
    T.stream(pair, m[2][0], m[2][1]);
    T.stream(mul, m[3][0], m[3][1]);
    T.graph(fwd, 20, m[4][0], m[4][1]);
    T.graph(inv, 20, m[5][0], m[5][1]);
    T.graph(pair, 10, m[6][0], m[6][1]);
    T.sync(fwd, m[7][0], m[7][1]);
    T.sync(inv, m[8][0], m[8][1]);
    // d_a has seen the same number of forward and inverse transforms on every route -- stream calls, replayed graphs (the
    // captured C ABI calls: every launching entry point is capture-safe), call-and-wait -- so it must hold the input again
    std::vector<u64> back(n);
    CK(hipStreamSynchronize(T.s));
    CK(hipMemcpy(back.data(), d_a, n * 8, hipMemcpyDeviceToHost));
    const bool ok1 = back == h;

    // ---- BFV drivers after the samplers, one ciphertext (configs[4]) ----
    const unsigned R = 5;
    mi355ntt_bfv* bfv = nullptr;
    RC(mi355ntt_bfv_create(&bfv, n, R, q5, psi5, 1024, 2305843009213683713ULL, 0, 0));
    const size_t poly = (size_t)n, ct = 2 * R * poly;
    u64 *d_sk, *d_pk, *d_e, *d_c, *d_m, *d_c0;
    CK(hipMalloc(&d_sk, R * poly * 8)); CK(hipMalloc(&d_pk, ct * 8)); CK(hipMalloc(&d_e, ct * 8)); CK(hipMalloc(&d_c, ct * 8));
    CK(hipMalloc(&d_c0, ct * 8)); CK(hipMalloc(&d_m, poly * 8));
    {   // small-norm inputs in RNS form: ternary secret / u, small errors, a message below t; uniform second public-key half
        std::vector<u64> sk(R * poly), pk(ct), e(ct), u(ct), msg(poly);
        for (size_t i = 0; i < poly; i++) {
            const int s3 = (int)(splitmix(seed) % 3) - 1, u3 = (int)(splitmix(seed) % 3) - 1;
            const int e0 = (int)(splitmix(seed) % 7) - 3, e1 = (int)(splitmix(seed) % 7) - 3;
            msg[i] = i % 10;
            for (unsigned j = 0; j < R; j++) {
                sk[j * poly + i] = s3 < 0 ? q5[j] - 1 : (u64)s3;
                u[j * poly + i] = u[(R + j) * poly + i] = u3 < 0 ? q5[j] - 1 : (u64)u3;
                e[j * poly + i] = e0 < 0 ? q5[j] + e0 : (u64)e0;
                e[(R + j) * poly + i] = e1 < 0 ? q5[j] + e1 : (u64)e1;
                pk[(R + j) * poly + i] = splitmix(seed) % q5[j];
            }
        }
        CK(hipMemcpy(d_sk, sk.data(), sk.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pk, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_e, e.data(), e.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_c0, u.data(), u.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_m, msg.data(), msg.size() * 8, hipMemcpyHostToDevice));
    }
    RC(mi355ntt_bfv_keygen(bfv, d_sk, d_pk, d_e, T.s));
    CK(hipStreamSynchronize(T.s));
    u64 *d_sk2, *d_pk2;                       // scratch copies for the timed keygen calls (keygen transforms its inputs in place)
    CK(hipMalloc(&d_sk2, R * poly * 8)); CK(hipMalloc(&d_pk2, ct * 8));
    CK(hipMemcpy(d_sk2, d_sk, R * poly * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(d_pk2, d_pk, ct * 8, hipMemcpyDeviceToDevice));
    auto keygen = [&] { RC(mi355ntt_bfv_keygen(bfv, d_sk2, d_pk2, d_e, T.s)); };
    auto enc = [&] { RC(mi355ntt_bfv_encrypt(bfv, d_c, d_pk, d_e, d_m, T.s)); };
    auto dec = [&] { RC(mi355ntt_bfv_decrypt(bfv, d_c, d_sk, T.s)); };
    // correctness first: encrypt a fresh u, decrypt, read the plaintext
    CK(hipMemcpyAsync(d_c, d_c0, ct * 8, hipMemcpyDeviceToDevice, T.s));
    enc(); dec();
    CK(hipStreamSynchronize(T.s));
    std::vector<u64> pt(poly);
    CK(hipMemcpy(pt.data(), d_c + poly * (R - 2), poly * 8, hipMemcpyDeviceToHost));
    bool ok2 = true;
    for (size_t i = 0; i < poly; i++) ok2 = ok2 && pt[i] == i % 10;
    double b[9][2];
    T.stream(keygen, b[0][0], b[0][1]);
    T.stream(enc, b[1][0], b[1][1]);
    T.stream(dec, b[2][0], b[2][1]);
    T.graph(keygen, 8, b[3][0], b[3][1]);
    T.graph(enc, 8, b[4][0], b[4][1]);
    T.graph(dec, 8, b[5][0], b[5][1]);
    T.sync(keygen, b[6][0], b[6][1]);
    T.sync(enc, b[7][0], b[7][1]);
    T.sync(dec, b[8][0], b[8][1]);
    {   // a captured encrypt -> decrypt graph must decrypt to the message as the direct calls did
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(T.s, hipStreamCaptureModeRelaxed));
        CK(hipMemcpyAsync(d_c, d_c0, ct * 8, hipMemcpyDeviceToDevice, T.s));
        enc(); dec();
        CK(hipStreamEndCapture(T.s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipMemsetAsync(d_c, 0, ct * 8, T.s));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, T.s));
        CK(hipStreamSynchronize(T.s));
        CK(hipMemcpy(pt.data(), d_c + poly * (R - 2), poly * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < poly; i++) ok2 = ok2 && pt[i] == i % 10;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }

    printf("{\"how\": \"compiled C++ through the C ABI (tools/lat_bench.cpp), %d back-to-back calls per sample, median [min] of %d samples, microseconds per call\", "
           "\"n\": %u, \"round_trip_ok\": %s, \"bfv_round_trip_ok\": %s, "
           "\"batch1_stream_us\": {\"forward\": [%.2f, %.2f], \"inverse\": [%.2f, %.2f], \"pair\": [%.2f, %.2f], \"fused_polymul\": [%.2f, %.2f]}, "
           "\"batch1_graph_us\": {\"forward\": [%.2f, %.2f], \"inverse\": [%.2f, %.2f], \"pair\": [%.2f, %.2f]}, "
           "\"batch1_call_and_wait_us\": {\"forward\": [%.2f, %.2f], \"inverse\": [%.2f, %.2f]}, "
           "\"bfv_4plus1_primes_stream_us\": {\"keygen\": [%.2f, %.2f], \"encrypt\": [%.2f, %.2f], \"decrypt\": [%.2f, %.2f]}, "
           "\"bfv_4plus1_primes_graph_us\": {\"keygen\": [%.2f, %.2f], \"encrypt\": [%.2f, %.2f], \"decrypt\": [%.2f, %.2f]}, "
           "\"bfv_4plus1_primes_call_and_wait_us\": {\"keygen\": [%.2f, %.2f], \"encrypt\": [%.2f, %.2f], \"decrypt\": [%.2f, %.2f]}}\n",
           T.calls, T.rounds, n, ok1 ? "true" : "false", ok2 ? "true" : "false",
           m[0][0], m[0][1], m[1][0], m[1][1], m[2][0], m[2][1], m[3][0], m[3][1],
           m[4][0], m[4][1], m[5][0], m[5][1], m[6][0], m[6][1],
           m[7][0], m[7][1], m[8][0], m[8][1],
           b[0][0], b[0][1], b[1][0], b[1][1], b[2][0], b[2][1],
           b[3][0], b[3][1], b[4][0], b[4][1], b[5][0], b[5][1],
           b[6][0], b[6][1], b[7][0], b[7][1], b[8][0], b[8][1]);
    return (ok1 && ok2) ? 0 : 1;
}
