// dma_rate.hip -- how fast does one CU take LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave instruction)?
// Every workgroup (8 waves, one per CU) streams pieces into a per-wave LDS ring, G pieces per group, two groups in flight,
// waiting with a counted vmcnt for the older group (the fused kernel's phase A / phase C pattern):
//   src 0: a 64 KiB table every workgroup shares (L2-resident, the weight images)
//   src 1: private rows from a 1.2 GB buffer (HBM, the x stream), 8 rows x 128 bytes per piece
//   src 2: both, alternating
// form 0: the kernels' dma_piece (M0 saved and restored around every piece, s_nop 4)
// form 1: M0 written once per group, pieces addressed by the instruction offset
// build: hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate ; run: ./dma_rate [n_workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ const void *uniform_ptr(const void *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned long long u = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
    return reinterpret_cast<const void *>(u);
}
__device__ __forceinline__ void dma_piece(const void *g_uniform, unsigned lds_addr, unsigned lane_byte_off)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(g_uniform), "s"(lds_addr) : "memory");
}

template <int SRC, int FORM, int G>
__global__ __launch_bounds__(512) void k(const char *table, const char *big, unsigned *sink, int groups, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem + wave * (2 * G * 1024);
    const unsigned lane16 = lane * 16;
    const unsigned rowoff = (unsigned)((lane >> 3) * 4608 + (lane & 7) * 16);      // 8 rows x 128 B of a [N][4608 B] matrix
    const char *mine = big + ((size_t)blockIdx.x * 8 + wave) * (size_t)groups * G * 8 * 4608 / 36;   // private region
    const long long t0 = __builtin_amdgcn_s_memtime();
    auto issue = [&](int grp, int half) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const bool l2 = SRC == 0 || (SRC == 2 && (i & 1));
            const long long pi = (long long)grp * G + i;
            const void *src = l2 ? (const void *)(table + ((pi * 8 + wave) & 63) * 1024)
                                 : (const void *)(mine + (pi / 36) * (8 * 4608) + (pi % 36) * 128);
            const unsigned off = l2 ? lane16 : rowoff;
            if (FORM == 0) dma_piece(uniform_ptr(src), lds0 + (half * G + i) * 1024, off);
            else {
                unsigned keep;
                const void *u = uniform_ptr(src);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(u), "s"(lds0 + (half * G + i) * 1024) : "memory");
            }
        }
    };
    issue(0, 0);
    issue(1, 1);
    u32x4 acc = {0, 0, 0, 0};
    for (int grp = 0; grp < groups; grp += 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        acc ^= *reinterpret_cast<const u32x4 *>(smem + wave * (2 * G * 1024) + lane * 16);
        issue(grp + 2, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        acc ^= *reinterpret_cast<const u32x4 *>(smem + wave * (2 * G * 1024) + G * 1024 + lane * 16);
        issue(grp + 3, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if ((acc[0] ^ acc[1]) == 0x12345678u) sink[0] = acc[0];
}

int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int groups = 288;                     // per wave
    char *table, *big; unsigned *sink; long long *cyc;
    const size_t bigbytes = (size_t)nwg * 8 * groups * 8 * 8 * 4608 / 36 + (64 << 20);
    hipMalloc(&table, 64 * 1024); hipMalloc(&big, bigbytes); hipMalloc(&sink, 64); hipMalloc(&cyc, nwg * 8 * 8);
    hipMemset(table, 1, 64 * 1024); hipMemset(big, 2, bigbytes);
    long long *h = (long long *)malloc(nwg * 8 * 8);
    auto run = [&](auto kern, const char *name, int G) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * G * 1024);
        float best = 1e9f;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 8 * 2 * G * 1024, 0, table, big, sink, groups, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(h, cyc, nwg * 8 * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < nwg * 8; ++i) m += h[i]; m /= nwg * 8;
        const double pieces = (double)(groups + 2) * G;
        printf("%-34s G=%d: %.3f ms, %.0f cycles per piece per wave, %.1f B/cycle per CU, %.1f GB/s per CU\n", name, G, best, m / pieces,
               8 * 1024.0 / (m / pieces), 8.0 * pieces * 1024 / best / 1e6);
    };
    run(k<0, 0, 4>, "L2 table, M0 save/restore", 4);
    run(k<0, 1, 4>, "L2 table, s_nop 0", 4);
    run(k<1, 0, 4>, "HBM rows, M0 save/restore", 4);
    run(k<2, 0, 4>, "HBM + L2 alternating", 4);
    run(k<0, 0, 8>, "L2 table, M0 save/restore", 8);
    run(k<1, 0, 8>, "HBM rows, M0 save/restore", 8);
    run(k<2, 0, 8>, "HBM + L2 alternating", 8);
    return 0;
}
