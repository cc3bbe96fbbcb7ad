// How does the dispatcher spread a launch's workgroups over XCDs, shader engines and CUs?  (Does every shader engine of an XCD
// have the same number of active CUs, and does an engine with fewer CUs get fewer workgroups?  DESIGN.md 4.11: a CU-masked
// queue runs at the pace of the engine left with the fewest CUs.)  Workgroups of 256 threads with 72 KiB of LDS (two per CU,
// like the conv kernel) spin for a fixed number of cycles and record where and when they ran.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void work(unsigned long long* out, int spin)
{
    extern __shared__ float lds[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long w0 = wall_clock64();
    const long long t0 = clock64();
    lds[threadIdx.x] = (float)t0;
    while (clock64() - t0 < spin) {}
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x] = xcc & 0xf;
        out[4 * blockIdx.x + 1] = hw;
        out[4 * blockIdx.x + 2] = w0;
        out[4 * blockIdx.x + 3] = w1 + (lds[0] == 1.5f);
    }
}
static void run(const char* name, hipStream_t st, int nwg, unsigned long long* d)
{
    std::vector<unsigned long long> h((size_t)nwg * 4);
    hipLaunchKernelGGL(work, dim3(nwg), dim3(256), 72 * 1024, st, d, 400000);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, (size_t)nwg * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nwg; i++) { tmin = std::min(tmin, h[4 * i + 2]); tmax = std::max(tmax, h[4 * i + 3]); }
    // per (xcc, se): CUs seen, workgroups run, last end
    struct Acc { std::set<unsigned> cus; int wgs = 0; unsigned long long end = 0; };
    std::map<unsigned, Acc> per;       // key = xcc * 8 + se
    std::map<unsigned, int> per_cu;    // key = xcc << 16 | se << 8 | sh << 4 | cu
    for (int i = 0; i < nwg; i++) {
        const unsigned xcc = (unsigned)h[4 * i], hw = (unsigned)h[4 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        Acc& a = per[xcc * 8 + se];
        a.cus.insert(sh << 4 | cu);
        a.wgs++;
        a.end = std::max(a.end, h[4 * i + 3]);
        per_cu[xcc << 16 | se << 8 | sh << 4 | cu]++;
    }
    printf("%s: %d workgroups, %zu CUs, wall %.1f us (100 MHz clock)\n", name, nwg, per_cu.size(), (tmax - tmin) / 100.0);
    for (unsigned xcc = 0; xcc < 8; xcc++) {
        printf("  xcc%u:", xcc);
        int cus = 0, wgs = 0;
        for (unsigned se = 0; se < 8; se++) {
            auto it = per.find(xcc * 8 + se);
            if (it == per.end()) continue;
            printf("  se%u %zu CUs %d wgs (%.2f/CU) end %.0f us |", se, it->second.cus.size(), it->second.wgs, it->second.wgs / (double)it->second.cus.size(),
                   (it->second.end - tmin) / 100.0);
            cus += (int)it->second.cus.size();
            wgs += it->second.wgs;
        }
        printf("  total %d CUs %d wgs\n", cus, wgs);
    }
    int mn = 1 << 30, mx = 0;
    for (auto& kv : per_cu) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
    printf("  workgroups per CU: min %d max %d\n", mn, mx);
}
int main()
{
    unsigned long long* d;
    hipMalloc(&d, 16384 * 32);
    hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    run("warm-up", 0, 2048, d);
    run("2048 workgroups (4.0 per slot)", 0, 2048, d);
    run("8192 workgroups", 0, 8192, d);
    for (int k = 1; k <= 8; k += (k == 1 ? 2 : k == 3 ? 1 : k == 4 ? 1 : 3)) {   // k = 1, 3, 4, 5, 8 partition CUs per XCD masked OFF
        uint32_t mask[8];
        for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
        for (int b = 0; b < 8 * k; b++) mask[b / 32] &= ~(1u << (b % 32));
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("mask k=%d: create failed\n", k); continue; }
        char name[64];
        snprintf(name, sizeof name, "8192 workgroups, first %d mask bits off (k = %d)", 8 * k, k);
        run(name, st, 8192, d);
        // (the stream is left alone: hipStreamDestroy of a CU-masked stream can hang the runtime on ROCm 7.2)
    }
    return 0;
}
