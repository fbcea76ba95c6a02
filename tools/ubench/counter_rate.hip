// What does s_memtime (__builtin_readcyclecounter) count on gfx950?  Compare it with s_memrealtime (100 MHz) and
// with hipEvent wall time over a long busy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long *out, int iters, float seed)
{
    float a = seed, b = seed * 0.5f;
    const long long c0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = fmaf(a, b, 1e-9f);
    }
    const long long c1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = w1 - w0; out[3] = (long long)a; }
}
int main()
{
    long long *d, h[4];
    (void)hipMalloc(&d, 32);
    int rate = 0;
    (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0;
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate attr %d kHz, clock rate attr %d kHz\n", rate, clk);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, d, 200000, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("kernel %.3f ms: s_memtime delta %lld (%.1f MHz if it spans the kernel), s_memrealtime delta %lld (%.1f MHz), wall_clock64 delta %lld; memtime/realtime = %.3f; dependent v_fma_f32 = %.2f memtime ticks each\n",
               ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), h[2], (double)h[0] / h[1], (double)h[0] / (200000.0 * 64));
    }
    return 0;
}
