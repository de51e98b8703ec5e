// kbench.hip -- standalone timing of the single-pass kernels (n = 2^15) with optional ablations
// (-DMI355NTT_ABLATE_EXCHANGE / _TWIDDLE / _GLOBAL: results are wrong, timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntt-cuda_amd/csrc -I include [-D...] tools/kbench.hip \
//         ntt-cuda_amd/csrc/hostparams.cpp -o tools/kbench_X
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_fast_impl.cuh"

using namespace mi355ntt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int LOGN = 15;
    const unsigned n = 1u << LOGN;
    unsigned num = argc > 1 ? atoi(argv[1]) : 1024;
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    const u64 q = 1152921504606584833ULL, psi = 4443670208963ULL;
    PrimeParams pp;
    if (derive_prime(n, q, psi, &pp)) { printf("bad prime\n"); return 1; }
    std::vector<u64> tab(n);
    fill_table(psi, q, n, tab.data());
    std::vector<TwPair> tw(n);
    for (unsigned i = 0; i < n; i++) tw[i] = TwPair{tab[i], shoup(tab[i], q)};   // layout irrelevant for timing
    PrimeDev d{};
    d.q = q; d.nq = 0ULL - q;
    d.mu = pp.mu; d.k = pp.k; d.red_sh1 = pp.k - 17; d.red_sh2 = 16; d.red_c = (u32)((((u128)1) << (31 + pp.k)) / q);
    for (int j = 0; j < 32; j++) d.twn[j] = TwPair{tab[j] % q, shoup(tab[j] % q, q)};   // (timing only)
    d.delta = (u32)((1ull << pp.k) - q); d.near_sh = pp.k - 32; d.near_mask = (u32)((1ull << (pp.k - 32)) - 1);
    const int HLSEL = argc > 3 ? atoi(argv[3]) : 4;
    const int warm = argc > 4 ? atoi(argv[4]) : 3;   // untimed launches first (the chip needs tens of ms of load to settle its clocks)
    u64* a; TwPair* dtw; PrimeDev* dp;
    CK(hipMalloc(&a, (size_t)num * n * 8));
    CK(hipMalloc(&dtw, n * sizeof(TwPair)));
    CK(hipMalloc(&dp, 2 * sizeof(PrimeDev)));      // [0]: the guard / clock record in front of the array
    CK(hipMemset(dp, 0, sizeof(PrimeDev)));
    dp += 1;
    std::vector<u64> h((size_t)num * n);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % q; }
    CK(hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, tw.data(), n * sizeof(TwPair), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, &d, sizeof(d), hipMemcpyHostToDevice));
#ifdef MI355NTT_STAMPS
    unsigned long long* dstamp;
    CK(hipMalloc(&dstamp, (size_t)256 * 16 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dstamp, sizeof(dstamp)));
    unsigned long long* dwg;
    CK(hipMalloc(&dwg, (size_t)256 * 8 * 8));
    CK(hipMemset(dwg, 0, (size_t)256 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_buf), &dwg, sizeof(dwg)));
    unsigned long long* dlog;
    const size_t log_words = 2 + 2 * (size_t)kWgLogCap;
    CK(hipMalloc(&dlog, log_words * 8));
    CK(hipMemset(dlog, 0, log_words * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_log), &dlog, sizeof(dlog)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // which: 0 forward, 1 inverse, 2 (KB_PAIR=1) forward then inverse over the same buffer, as bench.py's step
    const int nwhich = getenv("KB_PAIR") ? 3 : 2;
    for (int which = 0; which < nwhich; which++) {
        for (int i = 0; i < warm; i++) {
            if (which != 1) (void)launch_fwd<LOGN>(HLSEL, a, dtw, dp, num, 1, 0, 0);
            if (which != 0) (void)launch_inv<LOGN>(HLSEL, a, dtw, dp, num, 1, 0, 0);
        }
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
#ifdef MI355NTT_STAMPS
        CK(hipMemset(dlog, 0, 16));
#endif
        // KB_B2B = L > 1: each sample times L back-to-back launches (the queue stays full, as in bench.py), else one
        // isolated launch per sample (which includes ~10 us of host launch latency after the start event)
        const int b2b = getenv("KB_B2B") ? atoi(getenv("KB_B2B")) : 1;
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < b2b; l++) {
                if (which != 1) (void)launch_fwd<LOGN>(HLSEL, a, dtw, dp, num, 1, 0, 0);
                if (which != 0) (void)launch_inv<LOGN>(HLSEL, a, dtw, dp, num, 1, 0, 0);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / b2b);
        }
        std::sort(ts.begin(), ts.end());
#ifdef MI355NTT_STAMPS
        {
            // forward15: per-wave sums over all iterations of the time between marks:
            // slot 0: (store+issue of previous iteration ->) loop top, 1: R1 incl. load wait, 2: wait at sync, 3: exchange,
            // 4: R2, 5: T5->0 + R3 + canon, 6: row store + issue next loads
            unsigned nb = num < 256 ? num : 256;
            std::vector<unsigned long long> st((size_t)nb * 16 * 8);
            CK(hipMemcpy(st.data(), dstamp, st.size() * 8, hipMemcpyDeviceToHost));
#if MI355NTT_STAMPS == 2
            const char* nmf[] = {"(loop top)", "wait for the loaded polynomial", "R1", "sync + exchange 10->5", "R2 + T5->0 + R3 + canon", "row store (both halves)", "issue next loads", "-"};
#else
            const char* nmf[] = {"(loop top)", "R1 incl. load wait", "wait at sync", "exchange 10->5", "R2", "T5->0 + R3 + canon", "row store + issue next loads", "-"};
#endif
            // inverse15: 0 loop top, 1 R1' (bit 0 round), 2 T0->5 + R2', 3 wait at sync, 4 exchange 5->10, 5 R3' + canon, 6 store + next row loads (waited for)
            const char* nmi[] = {"(loop top)", "R1'", "T0->5 + R2'", "wait at sync", "exchange 5->10", "R3' + canon", "store + next row loads (incl. wait)", "-"};
            const char** nm = which == 0 ? nmf : nmi;
            double iters = (double)num / nb, tot = 0;
            for (int ph = 0; ph < 7; ph++) {
                double acc = 0, mn = 1e18, mx = 0;
                for (size_t w = 0; w < (size_t)nb * 16; w++) { double d = (double)st[w * 8 + ph] / iters; acc += d; mn = d < mn ? d : mn; mx = d > mx ? d : mx; }
                printf("    %-30s %9.0f cycles/poly  [wave min %8.0f .. max %8.0f]\n", nm[ph], acc / (nb * 16.0), mn, mx);
                tot += acc / (nb * 16.0);
            }
            printf("    total                          %9.0f cycles per polynomial per wave\n", tot);
            { double c = 0; for (size_t w = 0; w < (size_t)nb * 16; w++) c += (double)st[w * 8 + 7] / 16777216.0; printf("    in-kernel clock (memtime/memrealtime x 100 MHz): %.3f GHz\n", c / (nb * 16.0) * 0.1); }
            printf(which == 0 ? "    WG0 per wave (cycles/poly): R1+wait | sync | xchg | R2 | R3 | store\n"
                              : "    WG0 per wave (cycles/poly): R1' | T+R2' | sync | xchg | R3' | store+load\n");
            for (int w = 0; w < 16; w++) {
                printf("     w%02d", w);
                for (int ph = 1; ph < 7; ph++) printf(" %8.0f", (double)st[w * 8 + ph] / iters);
                printf("\n");
            }
            // workgroup timeline of the LAST launch (s_memrealtime, 10 ns ticks), relative to the earliest entry
            std::vector<unsigned long long> wg((size_t)nb * 8);
            CK(hipMemcpy(wg.data(), dwg, wg.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (unsigned b = 0; b < nb; b++) t0 = std::min(t0, wg[b * 8]);
            auto stat = [&](const char* name, auto f) {
                double mn = 1e18, mx = -1e18, sum = 0; unsigned cnt = 0;
                for (unsigned b = 0; b < nb; b++) { double v = f(b); if (v < -1e17) continue; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; cnt++; }
                printf("    wg %-34s mean %8.2f us  min %8.2f  max %8.2f  (%u wgs)\n", name, sum / cnt * 0.01, mn * 0.01, mx * 0.01, cnt);
            };
            stat("entry (after first entry)", [&](unsigned b) { return (double)(wg[b * 8] - t0); });
            stat("stagger sleep", [&](unsigned b) { return (double)(wg[b * 8 + 1] - wg[b * 8]); });
            int nit = (int)std::min<double>(5, iters);
            for (int i = 0; i < nit; i++) {
                char nmb[64]; snprintf(nmb, sizeof nmb, "iteration %d duration", i);
                stat(nmb, [&](unsigned b) { return (double)(wg[b * 8 + 2 + i] - (i ? wg[b * 8 + 1 + i] : wg[b * 8 + 1])); });
            }
            stat("exit (after first entry)", [&](unsigned b) { return (double)(wg[b * 8 + 7] - t0); });
            stat("exit - last iteration end", [&](unsigned b) { return (double)(wg[b * 8 + 7] - wg[b * 8 + 1 + nit]); });
            if (getenv("KB_WGDUMP")) {      // per workgroup: XCD (blockIdx % 8), stagger phase, iteration durations, exit time (us)
                for (int x = 0; x < 8; x++) {
                    double sum[6] = {0, 0, 0, 0, 0, 0}; unsigned cnt2 = 0;
                    for (unsigned b = x; b < nb; b += 8) {
                        for (int i = 0; i < nit; i++) sum[i] += (double)(wg[b * 8 + 2 + i] - (i ? wg[b * 8 + 1 + i] : wg[b * 8 + 1])) * 0.01;
                        sum[5] += (double)(wg[b * 8 + 7] - t0) * 0.01; cnt2++;
                    }
                    printf("    xcd %d: mean iteration us", x);
                    for (int i = 0; i < nit; i++) printf(" %6.2f", sum[i] / cnt2);
                    printf("   mean exit %7.2f\n", sum[5] / cnt2);
                }
                for (int ph = 0; ph < 8; ph++) {
                    double sum[6] = {0, 0, 0, 0, 0, 0}, mx = 0; unsigned cnt2 = 0;
                    for (unsigned b = 0; b < nb; b++) {
                        if (((b >> 3) & 7u) != (unsigned)ph) continue;
                        for (int i = 0; i < nit; i++) sum[i] += (double)(wg[b * 8 + 2 + i] - (i ? wg[b * 8 + 1 + i] : wg[b * 8 + 1])) * 0.01;
                        const double ex = (double)(wg[b * 8 + 7] - t0) * 0.01; sum[5] += ex; mx = std::max(mx, ex); cnt2++;
                    }
                    printf("    phase %d: mean iteration us", ph);
                    for (int i = 0; i < nit; i++) printf(" %6.2f", sum[i] / cnt2);
                    printf("   mean exit %7.2f  max exit %7.2f\n", sum[5] / cnt2, mx);
                }
            }
            // launch log of the timed back-to-back launches: per launch first entry / last entry / first exit / last exit,
            // and the gap from the last exit of the previous launch to the first entry of this one
            std::vector<unsigned long long> lg(log_words);
            CK(hipMemcpy(lg.data(), dlog, log_words * 8, hipMemcpyDeviceToHost));
            const size_t ne = std::min<size_t>(lg[0], kWgLogCap), nx = std::min<size_t>(lg[1], kWgLogCap);
            std::vector<unsigned long long> en(lg.begin() + 2, lg.begin() + 2 + ne), ex(lg.begin() + 2 + kWgLogCap, lg.begin() + 2 + kWgLogCap + nx);
            std::sort(en.begin(), en.end()); std::sort(ex.begin(), ex.end());
            const size_t per = nb * (which == 2 ? 1 : 1), nl = std::min(ne, nx) / per;
            double gap = 0, span = 0, ramp = 0, tail = 0; unsigned cnt = 0;
            for (size_t l = 1; l < nl; l++) {
                if (b2b > 1 && l % b2b == 0) continue;               // (first launch of a sample follows an event wait, not a launch)
                gap += (double)en[l * per] - (double)ex[l * per - 1];
                span += (double)ex[(l + 1) * per - 1] - (double)en[l * per];
                ramp += (double)en[(l + 1) * per - 1] - (double)en[l * per];
                tail += (double)ex[(l + 1) * per - 1] - (double)ex[l * per];
                cnt++;
            }
            if (cnt) printf("    launch log (%u back-to-back launches): gap last exit -> next first entry %.2f us, first entry -> last exit %.2f us, "
                            "entries spread over %.2f us, exits over %.2f us\n", cnt, gap / cnt * 0.01, span / cnt * 0.01, ramp / cnt * 0.01, tail / cnt * 0.01);
        }
#endif
        if (which == 2) {
            printf("pair     num=%u  median %.4f ms  min %.4f ms  => %.3f M fwd+inv pairs/s\n", num, ts[ts.size() / 2], ts[0], num / (ts[ts.size() / 2] * 1e-3) / 1e6);
            continue;
        }
        printf("%s  num=%u  median %.4f ms  min %.4f ms  => %.3f M transforms/s  (%.1f%% of 15.26M)\n", which ? "inverse" : "forward", num,
               ts[ts.size() / 2], ts[0], num / (ts[ts.size() / 2] * 1e-3) / 1e6, num / (ts[ts.size() / 2] * 1e-3) / 15.26e6 * 100);
    }
    // KB_STREAMS = k > 1: the batch split into k sub-batches, each transformed forward then inverse on a stream of its own, the k
    // streams fed round-robin (the launch ramp of one sub-batch overlaps the tail of another); host clock over `reps` steps
    if (getenv("KB_STREAMS") && atoi(getenv("KB_STREAMS")) > 1) {
        const int k = atoi(getenv("KB_STREAMS"));
        std::vector<hipStream_t> st(k);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const unsigned per = num / k;
        auto step = [&]() {
            for (int s = 0; s < k; s++) {
                (void)launch_fwd<LOGN>(HLSEL, a + (size_t)s * per * n, dtw, dp, per, 1, 0, st[s]);
                (void)launch_inv<LOGN>(HLSEL, a + (size_t)s * per * n, dtw, dp, per, 1, 0, st[s]);
            }
        };
        for (int rnd = 0; rnd < 3; rnd++) {
            for (int i = 0; i < warm; i++) step();
            CK(hipDeviceSynchronize());
            for (int i = 0; i < 30; i++) step();          // (flows into the timed steps: no clock ramp)
            // timed region between two events on stream 0 that every stream joins (wall time of the whole region, chip hot)
            hipEvent_t j0[16], j1[16], es, ee;
            CK(hipEventCreate(&es)); CK(hipEventCreate(&ee));
            for (int s = 1; s < k; s++) { CK(hipEventCreateWithFlags(&j0[s], hipEventDisableTiming)); CK(hipEventRecord(j0[s], st[s])); CK(hipStreamWaitEvent(st[0], j0[s], 0)); }
            CK(hipEventRecord(es, st[0]));
            for (int s = 1; s < k; s++) CK(hipStreamWaitEvent(st[s], es, 0));
            for (int i = 0; i < reps; i++) step();
            for (int s = 1; s < k; s++) { CK(hipEventCreateWithFlags(&j1[s], hipEventDisableTiming)); CK(hipEventRecord(j1[s], st[s])); CK(hipStreamWaitEvent(st[0], j1[s], 0)); }
            CK(hipEventRecord(ee, st[0]));
            CK(hipDeviceSynchronize());
            float worst = 0;
            CK(hipEventElapsedTime(&worst, es, ee));
            printf("streams=%d x %u polynomials: %.4f ms per step of %u pairs => %.3f M fwd+inv pairs/s\n", k, per, worst / reps, per * k, per * k / (worst / reps * 1e-3) / 1e6);
        }
    }
    return 0;
}
