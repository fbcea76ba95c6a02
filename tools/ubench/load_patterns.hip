// load_patterns.hip -- how fast can 256 CUs read a [N][2304] bf16 matrix once, by load shape?
// Each wave owns 32 token rows (as the project kernel does) and sweeps them left to right, DEPTH instructions in flight:
//   mode 0: per instruction 16 rows x 64 B, lane (g, n) = row n, bytes 16 g  (the MFMA B-operand layout read directly)
//   mode 1: per instruction  8 rows x 128 B to registers
//   mode 2: per instruction  2 rows x 512 B to registers
//   mode 3: mode 1 by LDS-DMA (global_load_lds_dwordx4), pieces land in a per-wave LDS ring
//   mode 4: mode 2 by LDS-DMA
//   mode 5: mode 0 addresses in the project kernel's schedule (groups of DEPTH loads, two groups in flight, wait for a whole group)
// build: hipcc -O3 --offload-arch=gfx950 load_patterns.hip -o load_patterns ; run: ./load_patterns [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int D = 2304;            // bf16 elements per row = 4608 B
constexpr int ROWB = D * 2;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const char *x, unsigned *sink, long long n_tokens)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    u32x4 acc = {0, 0, 0, 0};
    char *ring = smem + wave * DEPTH * 1024;
    for (long long grp = (long long)blockIdx.x * (blockDim.x >> 6) + wave; grp * 32 < n_tokens; grp += nwaves) {
        const char *base = x + grp * 32 * ROWB;
        // a group = 32 rows x 4608 B = 144 KiB = 144 instructions of 1 KiB
        constexpr int NI = 144;
        auto addr = [&](int i) -> const char * {
            if (MODE == 0 || MODE == 5) { const int h = i & 1, c = i >> 1; return base + (size_t)(16 * h + (lane & 15)) * ROWB + c * 64 + 16 * (lane >> 4); }
            if (MODE == 1 || MODE == 3) { const int h = i & 3, c = i >> 2; return base + (size_t)(8 * h + (lane >> 3)) * ROWB + c * 128 + 16 * (lane & 7); }
            { const int h = i & 15, c = i >> 4; return base + (size_t)(2 * h + (lane >> 5)) * ROWB + c * 512 + 16 * (lane & 31); }
        };
        if (MODE == 5) {
            // the project kernel's schedule: groups of DEPTH loads (mode 0 addresses), two groups in flight, wait for ALL of
            // the older group, then issue the next one
            u32x4 a[DEPTH], b[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) a[j] = *reinterpret_cast<const u32x4 *>(addr(j));
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) b[j] = *reinterpret_cast<const u32x4 *>(addr(DEPTH + j));
            for (int i = 2 * DEPTH; i < NI + 2 * DEPTH; i += 2 * DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) acc ^= a[j];
                asm volatile("" : "+v"(acc)::"memory");
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) a[j] = *reinterpret_cast<const u32x4 *>(addr(i + j < NI ? i + j : NI - 1));
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) acc ^= b[j];
                asm volatile("" : "+v"(acc)::"memory");
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) b[j] = *reinterpret_cast<const u32x4 *>(addr(i + DEPTH + j < NI ? i + DEPTH + j : NI - 1));
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc ^= a[j] ^ b[j];
        } else if (MODE <= 2) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) v[j] = *reinterpret_cast<const u32x4 *>(addr(j));
            for (int i = DEPTH; i < NI + DEPTH; i += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    acc ^= v[j];
                    const int ii = i + j < NI ? i + j : NI - 1;
                    v[j] = *reinterpret_cast<const u32x4 *>(addr(ii));
                }
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc ^= v[j];
        } else {
            for (int i = 0; i < NI; i += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)addr(i + j < NI ? i + j : NI - 1),
                                                     (__attribute__((address_space(3))) void *)(ring + j * 1024), 16, 0, 0);
                // half the ring is read back while the other half is in flight is the real kernel's job; here: drain
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc ^= *reinterpret_cast<const u32x4 *>(ring + lane * 16);
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = acc[0];
}

template <int MODE, int DEPTH>
void run(const char *x, unsigned *sink, long long n, hipEvent_t e0, hipEvent_t e1)
{
    for (int wpb : {4, 8}) {
        for (int bpc : {1, 2, 4}) {
            const int grid = 256 * bpc;
            const size_t lds = MODE >= 3 ? (size_t)wpb * DEPTH * 1024 : 0;
            if (lds * bpc > 160 * 1024) continue;
            hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(grid), dim3(wpb * 64), lds, 0, x, sink, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mode %d depth %2d waves/WG %d WG/CU %d (%2d waves/CU, %3d KiB in flight/CU): %.3f ms  %.2f TB/s\n", MODE, DEPTH, wpb, bpc,
                   wpb * bpc, wpb * bpc * DEPTH, best, (double)n * ROWB / best / 1e9);
        }
    }
}

__global__ void fill_random(unsigned *p, size_t nwords)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)i * 2654435761u + 12345u;
        v ^= v << 13; v ^= v >> 17; v ^= v << 5;
        p[i] = v;
    }
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 262144;
    const int random_data = argc > 2 ? atoi(argv[2]) : 0;      // 0: constant bytes, 1: pseudo-random words
    char *x; unsigned *sink;
    hipMalloc(&x, (size_t)n * ROWB);
    hipMalloc(&sink, 64);
    if (random_data) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(x), (size_t)n * ROWB / 4);
    else hipMemset(x, 1, (size_t)n * ROWB);
    printf("data: %s\n", random_data ? "pseudo-random" : "constant");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    run<0, 8>(x, sink, n, e0, e1);
    run<5, 8>(x, sink, n, e0, e1);
    run<5, 4>(x, sink, n, e0, e1);
    run<5, 16>(x, sink, n, e0, e1);
    return 0;
}
