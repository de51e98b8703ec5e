// power_model.hip -- package power, shader clock, core voltage and the firmware's throttle residencies as a function of VALU
// utilisation and HBM traffic (round 5, VERDICT r04 item 2: a model of the power cap that the numbers can refute).
//
// One persistent kernel, 256 workgroups x 1024 threads (one per CU, 4 waves per SIMD, as the n = 2^15 kernels):
//   waves 0..11 (three per SIMD)  the lazy Shoup butterfly stream of ntt_core.cuh (mad chain + fold, as ct_round), duty-cycled by the
//                                 shader clock: active in `busy` of every 16 slices of 8192 cycles, chip-wide in step
//   waves 12..15 (one per SIMD)   stream 16-byte loads and stores over a region of a 4 GiB buffer (past the 256 MB memory-side
//                                 cache), 1 read : 1 write as a transform, throttled by s_sleep `msleep` (0xffff: no memory work)
// Every wave runs until a common deadline (s_memrealtime) and reports what it did.  The host runs windows back to back for
// `seconds` and reads the SMU's gpu_metrics table (rocm_smi: rsmi_dev_gpu_metrics_info_get) before, during and after:
// socket power, gfx clock per XCD, gfx voltage, hotspot / HBM temperature, energy accumulator, and the accumulated
// throttler residencies (PPT = package power tracking, socket thermal, HBM thermal, VR thermal, PROCHOT) -- PVIOL % and
// TVIOL % over the run are (residency B - residency A) * 100 / (accumulation counter B - A), as rocm_smi.h documents.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ntt-cuda_amd/csrc -I include tools/power_model.hip -lrocm_smi64 -o tools/power_model
//   ./tools/power_model [seconds per point = 2.5] [mode: grid | valu | mem | idle]
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ntt_core.cuh"

using namespace mi355ntt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Out { unsigned long long work, t_first, t_last, clk; };

constexpr int CH = 16;

__global__ void __launch_bounds__(1024) k_power(Out* out, unsigned long long window_ticks, u64 q, u64 wseed, unsigned busy, unsigned vsleep,
                                                uint4* mem, unsigned long long region_vec, unsigned msleep)
{
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_end = t0 + window_ticks;
    unsigned long long n = 0, tl = t0;
    if (wave < 12) {
        u64 x[CH], y[CH], w[4], wp[4];
        const u64 nq = 0 - q, cq = 4 * q;
#pragma unroll
        for (int u = 0; u < CH; u++) {
            x[u] = (threadIdx.x * 1315423911ULL + u * 7919ULL) % q;
            y[u] = (x[u] * 2654435761ULL + wseed) % q;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            w[u] = (wseed * (u + 3) + threadIdx.x) % q;
            wp[u] = w[u] * 31 + 7;
        }
        // duty cycle by the shader clock: every wave of the chip works while ((s_memtime >> 13) & 15) < busy (slices of 8192
        // cycles, 16 per period) and sleeps otherwise -- all waves of a SIMD pause together, so the VALU utilisation follows the
        // duty (with three waves per SIMD a wave sleeping on its own would leave the SIMD saturated); vsleep is unused
        (void)vsleep;
        for (;;) {
            if (((__builtin_amdgcn_s_memtime() >> 13) & 15u) < busy) {
#pragma unroll 1
                for (int rep = 0; rep < 2; rep++) {
#pragma unroll
                    for (int u = 0; u < CH; u++) {
                        const u64 U = x[u];
                        u64 D = (U << 1) + cq;
                        asm("" : "+v"(D));
                        const u64 A = mul_shoup4m_acc<false>(y[u], w[u & 3], wp[u & 3], nq, U);
                        x[u] = A;
                        y[u] = D - A;
                        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                n += 2 * CH;
            } else {
                __builtin_amdgcn_s_sleep(8);
            }
            tl = __builtin_amdgcn_s_memrealtime();
            if (tl >= t_end) break;
        }
        u64 s = 0;
#pragma unroll
        for (int u = 0; u < CH; u++) s ^= x[u] ^ y[u];
        if (s == 0x1234567) out[0].work = s;
    } else if (msleep != 0xffffu) {
        __builtin_amdgcn_s_setprio(3);          // (the memory waves issue little: never starved by the butterfly waves)
        // this workgroup's region: region_vec 16-byte vectors; the four memory waves interleave 8 KiB blocks (8 x 1 KiB per wave-step)
        uint4* base = mem + (unsigned long long)blockIdx.x * region_vec;
        const unsigned long long half = region_vec / 2;           // first half is read, second half written
        unsigned long long pos = (unsigned long long)(wave - 12) * 1024;
        for (;;) {
            uint4 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { const v4u32 t_ = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(base + pos + k * 64 + lane)); v[k] = uint4{t_.x, t_.y, t_.z, t_.w}; }
#pragma unroll
            for (int k = 0; k < 16; k++) { v[k].x ^= (unsigned)n; base[half + pos + k * 64 + lane] = v[k]; }
            n += 16 * 1024 * 2;                                    // bytes moved by this wave (read + written)
            pos += 4 * 1024;
            if (pos + 1024 > half) pos = (unsigned long long)(wave - 12) * 1024;
            for (unsigned s = 0; s < msleep; s++) __builtin_amdgcn_s_sleep(4);
            tl = __builtin_amdgcn_s_memrealtime();
            if (tl >= t_end) break;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[1 + blockIdx.x * 16 + wave] = Out{n, t0, tl, c1 - c0};
}

static uint32_t g_rsmi = 0;
struct Sample { double w, mhz, mv, hot, hbm; };
static bool metrics(rsmi_gpu_metrics_t* m) { return rsmi_dev_gpu_metrics_info_get(g_rsmi, m) == RSMI_STATUS_SUCCESS; }
static Sample sample_of(const rsmi_gpu_metrics_t& m)
{
    double clk = 0; int nc = 0;
    for (int i = 0; i < RSMI_MAX_NUM_GFX_CLKS; i++) if (m.current_gfxclks[i] != 0xffff && m.current_gfxclks[i] != 0) { clk += m.current_gfxclks[i]; nc++; }
    double hbm = 0; int nh = 0;
    for (int i = 0; i < RSMI_NUM_HBM_INSTANCES; i++) if (m.temperature_hbm[i] != 0xffff && m.temperature_hbm[i] != 0) { hbm = hbm > m.temperature_hbm[i] ? hbm : m.temperature_hbm[i]; nh++; }
    return Sample{(double)(m.current_socket_power != 0xffff ? m.current_socket_power : m.average_socket_power), nc ? clk / nc : (double)m.current_gfxclk,
                  (double)(m.voltage_gfx != 0xffff ? m.voltage_gfx : 0), (double)(m.temperature_hotspot != 0xffff ? m.temperature_hotspot : 0), hbm};
}

struct Point { double util, tbs, w, mhz, mv, hot, hbm, pviol, tviol, hviol, vrviol, proch, ghz_kernel, bfly_rate, joules_per_s; };

static Point run_point(Out* d_out, uint4* d_mem, unsigned long long region_vec, unsigned busy, unsigned vsleep, unsigned msleep, double seconds, double full_rate_per_ghz)
{
    const int grid = 256;
    const unsigned long long window_us = 20000;
    rsmi_gpu_metrics_t a{}, b{};
    std::atomic<bool> stop{false};
    std::vector<Sample> samples;
    // 0.4 s of the same load first (clocks and the SMU's averages settle), then the measured run
    auto launch = [&]() { hipLaunchKernelGGL(k_power, dim3(grid), dim3(1024), 0, 0, d_out, window_us * 100ull, 1152921504606584833ULL, 4443670208963ULL, busy, vsleep, d_mem, region_vec, msleep); };
    for (int i = 0; i < 20; i++) launch();
    CK(hipDeviceSynchronize());
    metrics(&a);
    std::thread smp([&]() {
        while (!stop.load()) {
            rsmi_gpu_metrics_t m{};
            if (metrics(&m)) samples.push_back(sample_of(m));
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    });
    const int nwin = (int)(seconds * 1e6 / window_us);
    double bfly = 0, bytes = 0, clk = 0, span = 0;
    std::vector<Out> h(1 + grid * 16);
    for (int i = 0; i < nwin; i++) launch();
    CK(hipDeviceSynchronize());
    metrics(&b);
    stop.store(true);
    smp.join();
    CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Out), hipMemcpyDeviceToHost));      // (the last window stands for the run)
    unsigned long long tmin = ~0ull, tmax = 0; int nw = 0;
    for (int g = 0; g < grid; g++) for (int w = 0; w < 16; w++) {
        const Out& o = h[1 + g * 16 + w];
        if (o.t_last <= o.t_first) continue;
        tmin = tmin < o.t_first ? tmin : o.t_first; tmax = tmax > o.t_last ? tmax : o.t_last;
        if (w < 12) { bfly += (double)o.work * 64.0; clk += (double)o.clk / (double)(o.t_last - o.t_first); nw++; }
        else bytes += (double)o.work;
    }
    span = (double)(tmax - tmin) * 1e-8;
    Point p{};
    p.ghz_kernel = nw ? clk / nw * 0.1 : 0;
    p.bfly_rate = bfly / span;
    p.util = full_rate_per_ghz > 0 && p.ghz_kernel > 0 ? p.bfly_rate / (full_rate_per_ghz * p.ghz_kernel) : 0;
    p.tbs = bytes / span * 1e-12;
    for (const Sample& s : samples) { p.w += s.w; p.mhz += s.mhz; p.mv += s.mv; p.hot = p.hot > s.hot ? p.hot : s.hot; p.hbm = p.hbm > s.hbm ? p.hbm : s.hbm; }
    if (!samples.empty()) { p.w /= samples.size(); p.mhz /= samples.size(); p.mv /= samples.size(); }
    const double acc = (double)(b.accumulation_counter - a.accumulation_counter);
    if (acc > 0) {
        p.pviol = (double)(b.ppt_residency_acc - a.ppt_residency_acc) * 100.0 / acc;
        p.tviol = (double)(b.socket_thm_residency_acc - a.socket_thm_residency_acc) * 100.0 / acc;
        p.hviol = (double)(b.hbm_thm_residency_acc - a.hbm_thm_residency_acc) * 100.0 / acc;
        p.vrviol = (double)(b.vr_thm_residency_acc - a.vr_thm_residency_acc) * 100.0 / acc;
        p.proch = (double)(b.prochot_residency_acc - a.prochot_residency_acc) * 100.0 / acc;
    }
    const double dt = (double)(b.firmware_timestamp - a.firmware_timestamp) * 1e-8;
    if (dt > 0) p.joules_per_s = (double)(b.energy_accumulator - a.energy_accumulator) * 15.259e-6 * 10.0 / dt;   // (x 10: the header's unit reads a tenth of the sampled power on this part)
    return p;
}

static void print_point(const char* tag, unsigned busy, unsigned vsleep, unsigned msleep, const Point& p)
{
    printf("%-8s busy %3u vsleep %3u msleep %5u | VALU util %5.3f  %6.3f TB/s | %6.1f W (energy counter %6.1f W)  gfxclk %6.1f MHz (in-kernel %5.3f GHz)  vddgfx %5.0f mV  "
           "hotspot %3.0f C  hbm %3.0f C | PVIOL %5.1f %%  TVIOL %5.1f %%  HBM-thm %4.1f %%  VR-thm %4.1f %%  PROCHOT %4.1f %%\n",
           tag, busy, vsleep, msleep, p.util, p.tbs, p.w, p.joules_per_s, p.mhz, p.ghz_kernel, p.mv, p.hot, p.hbm, p.pviol, p.tviol, p.hviol, p.vrviol, p.proch);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    const char* mode = argc > 2 ? argv[2] : "grid";
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("rsmi_init failed\n"); return 1; }
    uint32_t ndev = 0;
    rsmi_num_monitor_devices(&ndev);
    // the rocm_smi index of HIP device 0 (by PCI bus id)
    char bus[64] = {0};
    CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
    unsigned dom = 0, bs = 0, dv = 0, fn = 0;
    sscanf(bus, "%x:%x:%x.%x", &dom, &bs, &dv, &fn);
    for (uint32_t i = 0; i < ndev; i++) {
        uint64_t bdf = 0;
        if (rsmi_dev_pci_id_get(i, &bdf) == RSMI_STATUS_SUCCESS && ((bdf >> 8) & 0xff) == bs && ((bdf >> 32) & 0xffffffff) == dom) g_rsmi = i;
    }
    uint64_t cap = 0, capmin = 0, capmax = 0;
    rsmi_dev_power_cap_get(g_rsmi, 0, &cap);
    rsmi_dev_power_cap_range_get(g_rsmi, 0, &capmax, &capmin);
    rsmi_gpu_metrics_t m0{};
    const bool ok = metrics(&m0);
    printf("# rocm_smi devices %u, HIP device 0 = %s = rsmi %u; power cap %.0f W (range %.0f .. %.0f); gpu_metrics %s v%u.%u\n", ndev, bus, g_rsmi, cap * 1e-6,
           capmin * 1e-6, capmax * 1e-6, ok ? "ok" : "UNAVAILABLE", m0.common_header.format_revision, m0.common_header.content_revision);
    if (ok) { Sample s = sample_of(m0); printf("# idle: %.0f W  gfxclk %.0f MHz  vddgfx %.0f mV  hotspot %.0f C  throttle_status 0x%x indep 0x%llx\n", s.w, s.mhz, s.mv, s.hot, m0.throttle_status, (unsigned long long)m0.indep_throttle_status); }

    Out* d_out; uint4* d_mem;
    const unsigned long long total_bytes = 4ull << 30, region_vec = total_bytes / 16 / 256;
    CK(hipMalloc(&d_out, (1 + 256 * 16) * sizeof(Out)));
    CK(hipMalloc(&d_mem, total_bytes));
    CK(hipMemset(d_mem, 1, total_bytes));

    // calibration: butterflies per second per GHz with every issue slot used and no memory work (short, the clock is read in-kernel)
    Point cal = run_point(d_out, d_mem, region_vec, 16, 0, 0xffffu, 0.6, 0);
    const double full_rate_per_ghz = cal.bfly_rate / cal.ghz_kernel;
    printf("# calibration (12 waves per CU of butterflies, no sleep, no memory): %.3e butterflies/s at %.3f GHz -> %.3e per GHz = %.2f cycles per wave-butterfly and SIMD\n",
           cal.bfly_rate, cal.ghz_kernel, full_rate_per_ghz, 1e9 / (full_rate_per_ghz / 64.0 / 1024.0));
    if (!strcmp(mode, "idle")) {
        std::this_thread::sleep_for(std::chrono::seconds(2));
        rsmi_gpu_metrics_t a{}, b{};
        metrics(&a); std::this_thread::sleep_for(std::chrono::seconds(2)); metrics(&b);
        printf("idle 2 s: energy acc %.1f W\n", (double)(b.energy_accumulator - a.energy_accumulator) * 15.259e-6 / ((double)(b.firmware_timestamp - a.firmware_timestamp) * 1e-8));
        return 0;
    }
    // busy = active slices of 16 (VALU duty); msleep = pause of the memory waves per 32 KiB moved, in units of 256 cycles
    const unsigned du[] = {16, 14, 13, 12, 11, 10, 8};
    const unsigned ms[] = {0xffffu, 160, 80, 45, 20, 0};
    if (!strcmp(mode, "valu")) {
        for (unsigned v : du) print_point("valu", v, 0, 0xffffu, run_point(d_out, d_mem, region_vec, v, 0, 0xffffu, seconds, full_rate_per_ghz));
    } else if (!strcmp(mode, "mem")) {
        for (unsigned mm : ms) print_point("mem", 0, 0, mm, run_point(d_out, d_mem, region_vec, 0, 0, mm, seconds, full_rate_per_ghz));
    } else {
        for (unsigned mm : ms)
            for (unsigned v : du) print_point("grid", v, 0, mm, run_point(d_out, d_mem, region_vec, v, 0, mm, seconds, full_rate_per_ghz));
        for (unsigned mm : ms) print_point("memonly", 0, 0, mm, run_point(d_out, d_mem, region_vec, 0, 0, mm, seconds, full_rate_per_ghz));
    }
    rsmi_shut_down();
    return 0;
}
