// smi_watch.cpp -- prints the SMU's gpu_metrics of one GPU as JSON lines (rocm_smi): socket power, gfx clock per XCD, temperatures,
// the energy accumulator and the accumulated throttler residencies.  PVIOL % (package power tracking) between two samples A, B is
// (ppt_acc B - ppt_acc A) * 100 / (acc_counter B - acc_counter A), likewise TVIOL % from thm_acc (rocm_smi.h, rsmi_gpu_metrics_t).
//   g++ -O2 -std=c++17 tools/smi_watch.cpp -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/smi_watch
//   tools/smi_watch [interval_ms = 200] [count = 0: until killed] [pci bus id of the device, e.g. 0000:0a:00.0; default: rsmi device 0]
// Measurement helper (bench.py's extras.power_sustained, tools/exp_r05*.sh); not part of the library.
#include <rocm_smi/rocm_smi.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

int main(int argc, char** argv)
{
    const int interval_ms = argc > 1 ? atoi(argv[1]) : 200;
    const long count = argc > 2 ? atol(argv[2]) : 0;
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("{\"error\": \"rsmi_init failed\"}\n"); return 1; }
    uint32_t ndev = 0, dev = 0;
    rsmi_num_monitor_devices(&ndev);
    if (argc > 3) {
        unsigned dom = 0, bs = 0, dv = 0, fn = 0;
        sscanf(argv[3], "%x:%x:%x.%x", &dom, &bs, &dv, &fn);
        for (uint32_t i = 0; i < ndev; i++) {
            uint64_t bdf = 0;
            if (rsmi_dev_pci_id_get(i, &bdf) == RSMI_STATUS_SUCCESS && ((bdf >> 8) & 0xff) == bs && ((bdf >> 32) & 0xffffffff) == dom) dev = i;
        }
    }
    uint64_t cap = 0;
    rsmi_dev_power_cap_get(dev, 0, &cap);
    for (long i = 0; count == 0 || i < count; i++) {
        rsmi_gpu_metrics_t m;
        memset(&m, 0, sizeof m);
        if (rsmi_dev_gpu_metrics_info_get(dev, &m) != RSMI_STATUS_SUCCESS) { printf("{\"error\": \"gpu_metrics unavailable\"}\n"); fflush(stdout); return 2; }
        double clk = 0; int nc = 0;
        for (int k = 0; k < RSMI_MAX_NUM_GFX_CLKS; k++) if (m.current_gfxclks[k] != 0xffff && m.current_gfxclks[k] != 0) { clk += m.current_gfxclks[k]; nc++; }
        const auto now = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        printf("{\"t_us\": %lld, \"power_w\": %u, \"power_cap_w\": %.0f, \"gfxclk_mhz\": %.1f, \"hotspot_c\": %u, \"mem_c\": %u, \"energy_acc\": %llu, "
               "\"fw_ts\": %llu, \"acc_counter\": %llu, \"ppt_acc\": %llu, \"thm_acc\": %llu, \"hbm_thm_acc\": %llu, \"vr_thm_acc\": %llu, \"prochot_acc\": %llu, "
               "\"gfx_activity\": %u, \"umc_activity\": %u}\n",
               (long long)now, (unsigned)(m.current_socket_power != 0xffff ? m.current_socket_power : m.average_socket_power), cap * 1e-6, nc ? clk / nc : (double)m.current_gfxclk,
               (unsigned)m.temperature_hotspot, (unsigned)m.temperature_mem, (unsigned long long)m.energy_accumulator, (unsigned long long)m.firmware_timestamp,
               (unsigned long long)m.accumulation_counter, (unsigned long long)m.ppt_residency_acc, (unsigned long long)m.socket_thm_residency_acc,
               (unsigned long long)m.hbm_thm_residency_acc, (unsigned long long)m.vr_thm_residency_acc, (unsigned long long)m.prochot_residency_acc,
               (unsigned)m.average_gfx_activity, (unsigned)m.average_umc_activity);
        fflush(stdout);
        if (count == 0 || i + 1 < count) std::this_thread::sleep_for(std::chrono::milliseconds(interval_ms));
    }
    rsmi_shut_down();
    return 0;
}
