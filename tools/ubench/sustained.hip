// sustained.hip -- do the streaming rates of load_patterns / store_patterns hold when the kernels run back to back for seconds?
// (bench.py's kernels run in a sustained loop; the pattern microbenchmarks time 5 isolated launches)
// mode 0: read [N][2304] bf16 (16 rows x 64 B per instruction)   mode 1: write [N][2304] fp32 (8 rows x 128 B per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int SYNC, int COLROT>
__global__ __launch_bounds__(512) void rdk(const char *x, unsigned *sink, long long n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (long long grp = (long long)blockIdx.x * 8 + wave; grp * 32 < n; grp += (long long)gridDim.x * 8) {
        const char *base = x + grp * 32 * 4608;
        u32x4 v[8];
        // COLROT: every wave starts its sweep at a different column (rotated by its group index), so that the waves of the chip
        // are not all at the same column at the same time
        const int rot = COLROT ? (int)((grp * 13) % 72) : 0;
        auto addr = [&](int i) { return base + (size_t)(16 * (i & 1) + (lane & 15)) * 4608 + (((i >> 1) + rot) % 72) * 64 + 16 * (lane >> 4); };
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4 *>(addr(j));
        for (int i = 8; i < 152; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc ^= v[j]; v[j] = *reinterpret_cast<const u32x4 *>(addr(i + j < 144 ? i + j : 143)); }
            if (SYNC) __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = acc[0];
}
// the project kernel's load code transplanted: asm loads (SGPR base + per-lane 32-bit offset + immediate), groups of 8 into a
// 3-deep register ring, group s+2 issued after group s landed, one asm statement holding the wait and naming the registers
template <int LDS_BYTES>
__global__ __launch_bounds__(512) void rdasm(const char *x, unsigned *sink, long long n)
{
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, nn = lane & 15;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 xr[3][8];
    const long long ntiles = n / 256, mytiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long S = mytiles * 18;
    long long iti = 0; int icc = 0;
    auto issue = [&](auto slot_c, u32x4 (&xr)[3][8]) {
        constexpr int SLOT = decltype(slot_c)::value;
        const long long tile = blockIdx.x + iti * gridDim.x;
        const int cc = icc;
        if (!(iti == mytiles - 1 && icc == 17)) { if (++icc == 18) { icc = 0; ++iti; } }
        const char *xb = x + (size_t)tile * 256 * 4608 + (size_t)cc * 256;
        unsigned voff[2];
        for (int T = 0; T < 2; ++T) voff[T] = (unsigned)((wave * 32 + 16 * T + nn) * 4608 + g * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int T = 0; T < 2; ++T)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xr[SLOT][k * 2 + T]) : "v"(voff[T]), "s"(xb), "i"(k * 64) : "memory");
    };
    auto landed = [&](auto slot_c, u32x4 (&xr)[3][8]) {
        constexpr int SLOT = decltype(slot_c)::value;
        u32x4 *r = &xr[SLOT][0];
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
    };
    issue(std::integral_constant<int, 0>{}, xr);
    issue(std::integral_constant<int, 1>{}, xr);
    for (long long s0 = 0; s0 < S; s0 += 3) {
        landed(std::integral_constant<int, 0>{}, xr);
        for (int j = 0; j < 8; ++j) acc ^= xr[0][j];
        issue(std::integral_constant<int, 2>{}, xr);
        landed(std::integral_constant<int, 1>{}, xr);
        for (int j = 0; j < 8; ++j) acc ^= xr[1][j];
        issue(std::integral_constant<int, 0>{}, xr);
        landed(std::integral_constant<int, 2>{}, xr);
        for (int j = 0; j < 8; ++j) acc ^= xr[2][j];
        issue(std::integral_constant<int, 1>{}, xr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LDS_BYTES && acc[0] == 0x9999u) smem[lane] = 1;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = acc[0];
}
__global__ __launch_bounds__(512) void wrk(float *out, long long n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (long long grp = (long long)blockIdx.x * 8 + wave; grp * 32 < n; grp += (long long)gridDim.x * 8)
        for (int c0 = 0; c0 < 2304; c0 += 32)
            for (int h = 0; h < 4; ++h) *reinterpret_cast<f32x4 *>(out + (grp * 32 + 8 * h + (lane & 7)) * 2304 + c0 + 4 * (lane >> 3)) = v;
}
int main(int argc, char **argv)
{
    const long long n = 262144;
    const int reps = argc > 1 ? atoi(argv[1]) : 3000;
    char *x; float *out; unsigned *sink;
    hipMalloc(&x, n * 4608); hipMalloc(&out, n * 9216); hipMalloc(&sink, 64);
    hipMemset(x, 1, n * 4608);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rdasm<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 108 * 1024);
    for (int mode = 0; mode < 8; ++mode) {
        for (int phase = 0; phase < 2; ++phase) {
            const int r = phase == 0 ? 5 : reps;
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < r; ++i) {
                if (mode == 0 || mode == 2) hipLaunchKernelGGL((rdk<0, 0>), dim3(256), dim3(512), 0, 0, x, sink, n);
                if (mode == 3) hipLaunchKernelGGL((rdk<1, 0>), dim3(256), dim3(512), 0, 0, x, sink, n);
                if (mode == 4) hipLaunchKernelGGL((rdk<1, 1>), dim3(256), dim3(512), 0, 0, x, sink, n);
                if (mode == 5) hipLaunchKernelGGL((rdk<0, 1>), dim3(256), dim3(512), 0, 0, x, sink, n);
                if (mode == 6) hipLaunchKernelGGL((rdasm<0>), dim3(256), dim3(512), 0, 0, x, sink, n);
                if (mode == 7) hipLaunchKernelGGL((rdasm<1>), dim3(256), dim3(512), 108 * 1024, 0, x, sink, n);
                if (mode == 1 || mode == 2) hipLaunchKernelGGL(wrk, dim3(256), dim3(512), 0, 0, out, n);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (mode == 1 ? 9216.0 : mode == 2 ? 13824.0 : 4608.0) * n;
            const char *nm[] = {"read 1.2 GB", "write 2.4 GB", "read then write", "read, barrier per 8 loads", "read, barrier + rotated columns", "read, rotated columns", "read, project kernel's asm schedule", "read, asm schedule + 108 KiB LDS"};
            printf("%s, %5d launches back to back: %.3f ms each, %.2f TB/s\n", nm[mode], r, ms / r, bytes / (ms / r) / 1e9);
        }
    }
    return 0;
}
