// Issue-rate probe for gfx950: cycles per wave-instruction of the candidates for "fp16 value -> fp32 accumulate" (the similarity
// search's inner operation) and of the fp32 matrix instructions beside packed fp32 FMA (the exact arm's sweeps).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mix_rate.hip -o mix_rate && ./mix_rate
// Every mode issues 16 independent instructions per loop trip; the table is cycles per instruction and SIMD at 1, 2 and 4 waves
// per SIMD (256 / 512 / 1024 threads on one CU... launched on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(1024) probe(long long *out, float *sink, int iters, float seed)
{
    float a[16];
    f2 p2[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p2[i] = (f2){seed + i, seed - i}; }
    unsigned h = __builtin_bit_cast(unsigned, seed) | 0x3c003c00u;
    f2 x2 = {seed, seed * 0.5f};
    float x = seed, w = seed * 0.25f;
    f4 c4[4];
    f16v c16[2];
    for (int i = 0; i < 4; ++i) c4[i] = (f4){seed, seed, seed, seed};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = seed;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define R16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
        if (MODE == 0) {
#define S(i) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h));
            R16(S)
#undef S
        } else if (MODE == 1) {
#define S(i) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h));
            R16(S)
#undef S
        } else if (MODE == 2) {
#define S(i) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(a[i]) : "v"(h));
            R16(S)
#undef S
        } else if (MODE == 3) {
#define S(i) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h));
            R16(S)
#undef S
        } else if (MODE == 4) {
#define S(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            R16(S)
#undef S
        } else if (MODE == 5) {
#define S(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p2[i]) : "v"(x2));
            R16(S)
#undef S
        } else if (MODE == 6) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(w));
            R16(S)
#undef S
        } else if (MODE == 7) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2[i]) : "v"(x2), "v"(x2));
            R16(S)
#undef S
        } else if (MODE == 8) {   // 16x16x4 f32: 1024 FMA per instruction, 4 independent accumulators
#define S(i) c4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w, c4[i & 3], 0, 0, 0);
            R16(S)
#undef S
        } else if (MODE == 9) {   // 32x32x2 f32: 2048 FMA per instruction, 2 independent accumulators
#define S(i) c16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w, c16[i & 1], 0, 0, 0);
            R16(S)
#undef S
        } else if (MODE == 10) {  // v_dot2_f32_f16: two fp16 products into an fp32 accumulator
#define S(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(h), "v"(h));
            R16(S)
#undef S
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + p2[i].x + p2[i].y;
    for (int i = 0; i < 4; ++i) s += c4[i].x;
    for (int i = 0; i < 2; ++i) s += c16[i][0];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
double run(int threads, int iters)
{
    long long *out;
    float *sink;
    const int nb = 256;
    (void)hipMalloc(&out, nb * 16 * sizeof(long long));
    (void)hipMalloc(&sink, 4);
    hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(threads), 0, 0, out, sink, 100, 1.5f);
    hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(threads), 0, 0, out, sink, iters, 1.5f);
    (void)hipDeviceSynchronize();
    long long h[nb * 16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    const int nw = threads / 64;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < nw; ++w) sum += (double)h[b * 16 + w];
    (void)hipFree(out);
    (void)hipFree(sink);
    const double per_wave = sum / (nb * nw);                 // s_memtime-like counter: 100 MHz on this part -> scale below
    return per_wave / ((double)iters * 16) * (nw / 4.0 > 1 ? 1.0 : 1.0);
}

int main()
{
    const char *names[] = {"v_fma_mix_f32 (lo half)", "v_fma_mix_f32 (hi half)", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa WORD_1", "v_add_f32",
                           "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32", "v_dot2_f32_f16"};
    printf("counter ticks per wave-instruction (per wave; divide the SIMD's cycles among its waves), 1 / 2 / 4 waves per SIMD\n");
    printf("(the counter is __builtin_readcyclecounter: s_memtime; ratios between rows are what matters)\n");
#define ROW(M) printf("%-28s %8.3f %8.3f %8.3f\n", names[M], run<M>(256, 4000), run<M>(512, 4000), run<M>(1024, 4000));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10)
    return 0;
}
