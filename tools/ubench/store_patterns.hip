// store_patterns.hip -- how fast can 256 CUs write a [N][2304] fp32 matrix, by store shape?
// Each wave owns 32 token rows (as the reconstruct kernel does) and writes them with dwordx4 stores:
//   mode 0: per instruction 16 rows x 64 B   (raw MFMA 16x16 D layout), full rows swept left to right
//   mode 1: per instruction  8 rows x 128 B
//   mode 2: per instruction  4 rows x 256 B
//   mode 3: per instruction  1 row  x 1 KiB
//   mode 4: mode 1, but d cut into 8 slices of 1152 B owned by different workgroups (weights-stationary layout)
//   mode 5: mode 1 with nontemporal stores      mode 6: mode 4 with nontemporal stores
// build: hipcc -O3 --offload-arch=gfx950 store_patterns.hip -o store_patterns ; run: ./store_patterns [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int D = 2304;

#define ST(p, v) do { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); else *reinterpret_cast<f32x4 *>(p) = v; } while (0)
template <int MODE_, bool NT = (MODE_ == 5 || MODE_ == 6), int MODE = (MODE_ == 5 ? 1 : MODE_ == 6 ? 4 : MODE_)>
__global__ __launch_bounds__(512) void k(float *out, long long n_tokens, int ngroups_per_wave_stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * 8;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    if (MODE == 4) {
        const int b = blockIdx.x, xcd = b & 7, w = b >> 3, slice = w & 7, lq = w >> 3, LQ = gridDim.x / 64;
        for (long long pi = (long long)lq * 8 + wave;; pi += (long long)LQ * 8) {
            const long long pair = xcd + 8 * pi;
            if (pair * 32 >= n_tokens) break;
            for (int i = 0; i < 9; ++i)       // 9 pairs of 16-column tiles = 288 columns
                for (int h = 0; h < 4; ++h) { // 32 tokens = 4 x 8
                    const long long tok = pair * 32 + 8 * h + (lane & 7);
                    const int col = slice * 288 + i * 32 + 4 * (lane >> 3);
                    ST(out + tok * D + col, v);
                }
        }
        return;
    }
    for (long long grp = (long long)blockIdx.x * 8 + wave; grp * 32 < n_tokens; grp += nwaves) {
        const long long t0 = grp * 32;
        if (MODE == 0) {
            for (int c0 = 0; c0 < D; c0 += 16)
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<f32x4 *>(out + (t0 + 16 * h + (lane & 15)) * D + c0 + 4 * (lane >> 4)) = v;
        } else if (MODE == 1) {
            for (int c0 = 0; c0 < D; c0 += 32)
                for (int h = 0; h < 4; ++h)
                    ST(out + (t0 + 8 * h + (lane & 7)) * D + c0 + 4 * (lane >> 3), v);
        } else if (MODE == 2) {
            for (int c0 = 0; c0 < D; c0 += 64)
                for (int h = 0; h < 8; ++h)
                    *reinterpret_cast<f32x4 *>(out + (t0 + 4 * h + (lane & 3)) * D + c0 + 4 * (lane >> 2)) = v;
        } else if (MODE == 7) {       // per instruction 2 rows x 512 B
            for (int c0 = 0; c0 < D; c0 += 128)
                for (int h = 0; h < 16; ++h)
                    if (c0 + 4 * (lane & 31) < D) *reinterpret_cast<f32x4 *>(out + (t0 + 2 * h + (lane >> 5)) * D + c0 + 4 * (lane & 31)) = v;
        } else {
            for (int r = 0; r < 32; ++r)
                for (int c0 = 0; c0 < D; c0 += 256)
                    *reinterpret_cast<f32x4 *>(out + (t0 + r) * D + c0 + 4 * lane) = v;
        }
    }
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 262144;
    float *out;
    hipMalloc(&out, (size_t)n * D * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 8; ++mode) {
        for (int grid : {32, 64, 128, 256}) {
            if ((mode == 4 || mode == 6) && grid < 64) continue;   // the slice layout needs 8 XCDs x 8 slices
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(512), 0, 0, out, n, 0); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mode %d grid %4d: %.3f ms  %.2f TB/s  %.1f GB/s per workgroup\n", mode, grid, best, (double)n * D * 4 / best / 1e9, (double)n * D * 4 / best / 1e6 / grid); fflush(stdout);
        }
    }
    return 0;
}
