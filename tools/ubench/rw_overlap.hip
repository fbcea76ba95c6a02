// rw_overlap.hip -- does one CU's read stream share a limit with its write stream?
// The project kernel streams x at ~24 GB/s per CU and the reconstruct kernel stores at ~22 GB/s per CU; with every CU capped
// (outstanding misses x latency), chip bandwidth needs every CU streaming all the time.  If a CU's loads and stores are
// limited separately, a kernel that does both at once moves more bytes per CU than either stream alone.
// Runs on NCU workgroups (one per CU, far below what HBM sustains) so that the per-CU limits are what is measured:
//   mode 0: 8 waves read         mode 1: 8 waves write        mode 2: every wave alternates 8 loads / 8 stores
//   mode 3: waves 0-3 read, waves 4-7 write                   mode 4: 16 waves, 8 read + 8 write
//   mode 5: 8 waves read HBM + 8 waves read an L2-resident table (the weight stream)
// build: hipcc -O3 --offload-arch=gfx950 rw_overlap.hip -o rw_overlap ; run: ./rw_overlap [n_workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr size_t CHUNK = 1 << 20;       // bytes one wave walks per pass (1 KiB per instruction)

__device__ __forceinline__ void rd(const char *src, size_t bytes, int lane, u32x4 &acc)
{
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4 *>(src + (size_t)j * 1024 + lane * 16);
    for (size_t o = 8192; o < bytes + 8192; o += 8192) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc ^= v[j];
            const size_t a = o + (size_t)j * 1024 < bytes ? o + (size_t)j * 1024 : bytes - 1024;
            v[j] = *reinterpret_cast<const u32x4 *>(src + a + lane * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
}
__device__ __forceinline__ void wr(char *dst, size_t bytes, int lane, u32x4 val)
{
    for (size_t o = 0; o < bytes; o += 1024) *reinterpret_cast<u32x4 *>(dst + o + lane * 16) = val;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(const char *x, char *out, const char *table, unsigned *sink, int passes)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u32x4 acc = {0, 0, 0, (unsigned)lane};
    for (int p = 0; p < passes; ++p) {
        const size_t slot = ((size_t)p * gridDim.x + blockIdx.x) * nw + wave;
        const char *src = x + slot * CHUNK;
        char *dst = out + slot * CHUNK;
        if (MODE == 0) rd(src, CHUNK, lane, acc);
        else if (MODE == 1) wr(dst, CHUNK, lane, acc);
        else if (MODE == 2) {
            for (size_t o = 0; o < CHUNK; o += 8192) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4 *>(src + o + (size_t)j * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4 *>(dst + o + (size_t)j * 1024 + lane * 16) = v[j] ^ acc;
            }
        } else if (MODE == 3 || MODE == 4) {
            if (wave < nw / 2) rd(src, CHUNK, lane, acc);
            else wr(dst, CHUNK, lane, acc);
        } else {
            if (wave < nw / 2) rd(src, CHUNK, lane, acc);
            else rd(table + (size_t)((p * 7 + wave) & 1) * CHUNK, CHUNK, lane, acc);     // 2 MiB table: stays in L2
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2]) == 0x12345678u) sink[0] = acc[0];
}

int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 64;
    const int passes = 16;
    const size_t bytes = (size_t)passes * nwg * 16 * CHUNK;
    char *x, *out, *table; unsigned *sink;
    hipMalloc(&x, bytes); hipMalloc(&out, bytes); hipMalloc(&table, 2 * CHUNK); hipMalloc(&sink, 64);
    hipMemset(x, 1, bytes); hipMemset(out, 0, bytes); hipMemset(table, 2, 2 * CHUNK);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"8 waves read", "8 waves write", "8 waves load+store", "4 read + 4 write", "8 read + 8 write", "8 read HBM + 8 read L2"};
    for (int mode = 0; mode < 6; ++mode) {
        const int nwaves = mode >= 4 ? 16 : 8;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(nwg), dim3(nwaves * 64), 0, 0, x, out, table, sink, passes); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double per = (double)passes * CHUNK;     // bytes per wave
        double rdb = 0, wrb = 0, l2b = 0;
        if (mode == 0) rdb = 8 * per; else if (mode == 1) wrb = 8 * per; else if (mode == 2) { rdb = 8 * per; wrb = 8 * per; }
        else if (mode == 3) { rdb = 4 * per; wrb = 4 * per; } else if (mode == 4) { rdb = 8 * per; wrb = 8 * per; } else { rdb = 8 * per; l2b = 8 * per; }
        printf("%-26s %d WGs: %.3f ms   per CU: read %.1f GB/s  write %.1f GB/s  L2-hit read %.1f GB/s\n", names[mode], nwg, best,
               rdb / best / 1e6, wrb / best / 1e6, l2b / best / 1e6);
    }
    return 0;
}
