// Which CUs does a CU-masked stream use?  (hipExtStreamCreateWithCUMask bit order vs XCD / CU ids on gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <map>
#include <vector>
__global__ void where(unsigned* out, int spin)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
int main()
{
    const int nwg = 4096;
    unsigned* d; hipMalloc(&d, nwg * 8);
    std::vector<unsigned> h(nwg * 2);
    for (int test = 0; test < 5; test++) {
        uint32_t mask[8] = {0};
        const char* name = "";
        if (test == 0) { name = "all"; for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu; }
        if (test == 1) { name = "bits 0..15"; mask[0] = 0xffffu; }
        if (test == 2) { name = "bits 0,8,16,..,120"; for (int i = 0; i < 128; i += 8) mask[i / 32] |= 1u << (i % 32); }
        if (test == 3) { name = "bits 0..1 of every 32"; for (int i = 0; i < 8; i++) mask[i] = 0x3u; }
        if (test == 4) { name = "all but bits 0..15"; for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu; mask[0] = 0xffff0000u; }
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
        if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); continue; }
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, st, d, 20000);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per;
        for (int i = 0; i < nwg; i++) per[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xff | ((h[2 * i + 1] >> 13) & 0x7) << 8);   // cu_id bits 8..11, sh 12, se 13..15
        int tot = 0;
        printf("%s:", name);
        for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  -> %d CUs\n", tot);
        hipStreamDestroy(st);
    }
    return 0;
}
