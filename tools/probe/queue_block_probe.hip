// Does a long kernel on one CU-masked stream keep OTHER streams' kernels from starting?  (round 5: forward lanes stood still for the length of
// a beam search in some runs of tools/policy_probe.py -- see DESIGN_LOG.md.)  N masked streams are created one after the other (as
// pipe_reads.hip does: two complement-mask lanes, then the partition stream, then spares); a kernel that spins `ms` milliseconds on ONE workgroup
// is launched on stream i, and right behind it a trivial kernel on every other stream j; printed: how long after its launch each trivial kernel
// finished.  A row with ~ms entries = stream i blocks stream j (they share a hardware pipe / queue slot); ~0 = independent.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/qprobe tools/probe/queue_block_probe.hip ; run: /tmp/qprobe [n_streams=8] [ms=20] [follow=0|1]
//   follow=1: a dependent trivial kernel is queued BEHIND the spinner on stream i as well (a packet waiting on the running kernel)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin(long long cycles, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
__global__ void tiny(int* out) { out[threadIdx.x] = 1; }
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const double ms = argc > 2 ? atof(argv[2]) : 20.0;
    const int follow = argc > 3 ? atoi(argv[3]) : 0;
    int* d;
    hipMalloc(&d, 4096);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; i++) {
        uint32_t mask[8];
        const bool part = (i == 2);                       // stream 2 plays the decode partition (first 4 CUs of every XCD), the others the complement
        for (int w = 0; w < 8; w++) mask[w] = 0;
        for (int b = 0; b < 256; b++)
            if ((b < 32) == part) mask[b / 32] |= 1u << (b % 32);
        if (hipExtStreamCreateWithCUMask(&st[i], 8, mask) != hipSuccess) { printf("stream %d: create failed\n", i); return 1; }
    }
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    const long long cycles = (long long)(ms * rate);
    std::vector<hipEvent_t> ev(n);
    for (int j = 0; j < n; j++) hipEventCreate(&ev[j]);
    for (int j = 0; j < n; j++) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[j], d + 64 * j); }
    hipDeviceSynchronize();
    printf("spinner %.0f ms on stream i (rows), trivial kernel on stream j (columns): ms until the trivial kernel finished; follow=%d\n      ", ms, follow);
    for (int j = 0; j < n; j++) printf("   j=%d ", j);
    printf("\n");
    for (int i = 0; i < n; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], cycles, (int*)nullptr);
        if (follow) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], d + 2048);
        for (int j = 0; j < n; j++)
            if (j != i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[j], d + 64 * j); hipEventRecord(ev[j], st[j]); }
        std::vector<double> done(n, -1.0);
        int left = n - 1;
        while (left > 0) {
            for (int j = 0; j < n; j++)
                if (j != i && done[j] < 0 && hipEventQuery(ev[j]) == hipSuccess) {
                    done[j] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    left--;
                }
        }
        hipDeviceSynchronize();
        printf("i=%d : ", i);
        for (int j = 0; j < n; j++) j == i ? printf("    -  ") : printf(" %6.2f", done[j]);
        printf("\n");
    }
    return 0;
}
