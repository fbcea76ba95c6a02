// VALU issue-rate probe for gfx950: cycles per instruction for pk_fma variants, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ void __launch_bounds__(1024) probe(long long *out, float *sink, int iters, float seed)
{
    __shared__ f4 lds[1024];
    lds[threadIdx.x] = (f4){seed, seed, seed, seed};
    __syncthreads();
    f2 a[16];
    for (int i = 0; i < 16; ++i) a[i] = (f2){seed + i, seed - i};
    f2 bb[16], cc[16];
    for (int i = 0; i < 16; ++i) { bb[i] = (f2){seed * i, seed + 2 * i}; cc[i] = (f2){seed - 3 * i, seed * 0.5f * i}; }
    f2 x = {seed, seed * 0.5f};
    f2 w = {seed * 0.25f, seed * 0.125f};
    f2 sx = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed))),
             __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed * 3.f)))};
    float s1 = seed;
    typedef float f16v __attribute__((ext_vector_type(16)));
    f4 c4[8];
    for (int i = 0; i < 8; ++i) c4[i] = (f4){seed, seed, seed, seed};
    f16v c16[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = seed;
    f4 q = lds[threadIdx.x];
    const f4 *lp = &lds[threadIdx.x & 63];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 16 independent pk_fma, VGPR operands
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 1) {   // op_sel broadcast of w.lo
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 2) {   // SGPR pair operand + broadcast (lin_out form)
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(sx), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 3) {   // plain v_fma_f32
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(x.x), "v"(w.x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 4) {   // pk_add
#define S(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 5) {   // 16 pk_fma + 1 ds_read_b128 (pass-like ratio ~ 13:1)
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
            asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const f4 *)lp));
            S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            w.x += q.x * 0.f;
#undef S
        } else if (MODE == 6) {   // dependent chain of 4 (y chain in lin_out: 4 chains of 4)
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 7) {   // fully dependent chain
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 8) {   // MFMA 4x4x1 x16 independent
            f4 c = q;
#define S(i) c = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.x, c, 0, 0, 0);
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
            q = c;
        } else if (MODE == 10) {  // 8 independent MFMA 4x4x1 chains
#define S(i) c4[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.x, c4[i & 7], 0, 0, 0);
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 11) {  // 16 pk_fma interleaved with 8 MFMA 4x4x1 (all independent)
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(w));
#define M(i) c4[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.x, c4[i & 7], 0, 0, 0);
            S(0) S(1) M(0) S(2) S(3) M(1) S(4) S(5) M(2) S(6) S(7) M(3) S(8) S(9) M(4) S(10) S(11) M(5) S(12) S(13) M(6) S(14) S(15) M(7)
#undef S
#undef M
        } else if (MODE == 12) {  // 16 pk_fma interleaved with 16 MFMA 4x4x1
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(w));
#define M(i) c4[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.x, c4[i & 7], 0, 0, 0);
            S(0) M(0) S(1) M(1) S(2) M(2) S(3) M(3) S(4) M(4) S(5) M(5) S(6) M(6) S(7) M(7) S(8) M(8) S(9) M(9) S(10) M(10) S(11) M(11) S(12) M(12) S(13) M(13) S(14) M(14) S(15) M(15)
#undef S
#undef M
        } else if (MODE == 13) {  // 16 independent-ish MFMA 16x16x1 (4 chains)
#define M(i) c16[i & 3] = __builtin_amdgcn_mfma_f32_16x16x1f32(x.x, w.x, c16[i & 3], 0, 0, 0);
            M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#undef M
        } else if (MODE == 14) {  // 16 pk_fma + 4 MFMA 16x16x1
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(w));
#define M(i) c16[i & 3] = __builtin_amdgcn_mfma_f32_16x16x1f32(x.x, w.x, c16[i & 3], 0, 0, 0);
            S(0) S(1) S(2) S(3) M(0) S(4) S(5) S(6) S(7) M(1) S(8) S(9) S(10) S(11) M(2) S(12) S(13) S(14) S(15) M(3)
#undef S
#undef M
        } else if (MODE == 15) {  // 32 plain v_fma_f32 (same FMA count as 16 pk)
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %3, %1, %2, %3" : "+v"(a[i].x), "+v"(a[i].y) : "v"(x.x), "v"(w.x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 16) {  // v_fma_f64, 16 independent (a[i] reinterpreted as a double)
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 17) {  // v_cvt_f32_f64
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i].x) : "v"(x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 18) {  // v_cvt_f16_f32
#define S(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a[i].x) : "v"(x.x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 19) {  // v_fma_f64 with 8 chains of 2
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i & 7]) : "v"(x), "v"(w));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 20) {  // v_fma_f64, every operand a different register pair
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb[i]), "v"(cc[i]));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 21) {  // v_pk_fma_f32, every operand a different register pair
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb[i]), "v"(cc[i]));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 22) {  // v_pk_fma_f32 acc and weights distinct, x shared + broadcast (the pass's form)
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(bb[i]), "v"(x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 23) {  // v_fma_f64: accumulators distinct, one operand shared (the table kernel's form)
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(cc[i]));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 24) {  // v_fma_f32, every operand a different register
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(bb[i].x), "v"(cc[i].y));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 9) {   // v_mov_b32 (cheap VALU)
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(x.x));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = q.x + s1;
    for (int i = 0; i < 8; ++i) acc += c4[i].x + c4[i].w;
    for (int i = 0; i < 4; ++i) acc += c16[i][0] + c16[i][15];
    for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y + bb[i].x + cc[i].y;
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, long long *d_out, float *d_sink)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 4; ++wps) {   // waves per SIMD: block = 4*wps waves on one CU
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256 * wps), 0, 0, d_out, d_sink, iters, 1.0f);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int i = 0; i < 4 * wps; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-44s waves/SIMD=%d  cycles/instr/wave=%6.2f  SIMD cycles per instr=%5.2f\n", name, wps,
               (double)mx / (iters * 16.0), (double)mx / (iters * 16.0 * wps));
    }
}

int main()
{
    long long *d_out; float *d_sink;
    hipMalloc(&d_out, 16 * 16 * sizeof(long long));
    hipMalloc(&d_sink, 64);
    run<0>("pk_fma vgpr x16 independent", d_out, d_sink);
    run<1>("pk_fma op_sel bcast", d_out, d_sink);
    run<2>("pk_fma sgpr-pair + bcast", d_out, d_sink);
    run<3>("v_fma_f32", d_out, d_sink);
    run<4>("v_pk_add_f32", d_out, d_sink);
    run<5>("16 pk_fma + ds_read_b128 + wait", d_out, d_sink);
    run<6>("pk_fma 4 chains of 4 dependent", d_out, d_sink);
    run<7>("pk_fma fully dependent", d_out, d_sink);
    run<8>("mfma 4x4x1 dependent chain", d_out, d_sink);
    run<9>("v_mov_b32", d_out, d_sink);
    run<10>("mfma 4x4x1 x16, 8 independent chains", d_out, d_sink);
    run<11>("16 pk_fma + 8 mfma 4x4x1 (per 16 pk)", d_out, d_sink);
    run<12>("16 pk_fma + 16 mfma 4x4x1 (per 16 pk)", d_out, d_sink);
    run<13>("mfma 16x16x1 x16, 4 chains", d_out, d_sink);
    run<14>("16 pk_fma + 4 mfma 16x16x1 (per 16 pk)", d_out, d_sink);
    run<15>("32 v_fma_f32 (per 16 pairs)", d_out, d_sink);
    run<16>("v_fma_f64 x16 independent", d_out, d_sink);
    run<19>("v_fma_f64 8 chains of 2", d_out, d_sink);
    run<20>("v_fma_f64 all operands distinct", d_out, d_sink);
    run<23>("v_fma_f64 acc + one operand distinct", d_out, d_sink);
    run<21>("v_pk_fma_f32 all operands distinct", d_out, d_sink);
    run<22>("v_pk_fma_f32 acc + w distinct, x bcast", d_out, d_sink);
    run<24>("v_fma_f32 all operands distinct", d_out, d_sink);
    run<17>("v_cvt_f32_f64", d_out, d_sink);
    run<18>("v_cvt_f16_f32", d_out, d_sink);
    return 0;
}
