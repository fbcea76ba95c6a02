// rqae_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the RQAE hot path.
//
// What runs here (reference file:line relative to /root/reference):
//   rqae_hip_encode         RQAE.forward level loop            rqae/model.py:199-230
//   rqae_hip_quantize       RQAE.quantize at temperature 0     rqae/model.py:187-197
//   rqae_hip_decode[_values] RQAE.decode                       rqae/model.py:232-252
//   rqae_hip_gemma_norm / _denorm_patch   the steps either side of forward in RQAE.hook   rqae/model.py:276-289
//   rqae_hip_intensity[_prepare]          RQAEFeature.intensity                             rqae/feature.py:102-129
//
// Arithmetic contract (restated on the CPU by
// oracle/rqae_oracle.c, which these kernels match bit for bit):
//   * everything is fp32, one rounding per written operation; compiled with
//     -ffp-contract=off, fused multiply-adds appear only as explicit fmaf()
//   * lin_in  : d is zero-padded to dp = 256*P and cut into 4 slices of 64*P;
//               element elem_of(P, s, i, lane) belongs to (slice s, lane, step i); each
//               (slice, lane) runs a fmaf chain over i from +0; the 64 lanes are
//               summed by the pairing lane^32, lane^16, lane^8, 7-mirror, lane^1,
//               lane^2; slices as ((s0+s1)+(s2+s3)); bias last
//   * ||q||   : ATen's 2-norm order (8-wide fma chunks, one 4-chunk of rounded
//               squares, fma tail), IEEE sqrt; q/||q|| IEEE division
//   * cos-sim : k-ordered fmaf chain from +0; argmax = first maximum
//   * STE     : q + (cv - q);  lin_out: k-ordered fmaf chain from +0, then + bias
//
// Work decomposition of the encode kernels: one workgroup owns TG tokens for all
// levels.  Wave (slice, half) keeps the residual and the running reconstruction of
// its TPW tokens x 64*P elements in VGPRs for the whole level loop (the N x d
// residual never goes back to HBM); weights stream from L2 once per level per
// workgroup (LDS-DMA into LDS in rqae_encode_lds_kernel, the main path); the only
// exchange per level is 4c partial sums in and c floats out per token through LDS.
// Kernels, in file order: pack; rqae_encode_kernel (generic shapes); rqae_encode_lds_kernel
// (main); rqae_quantize_kernel; rqae_decode_kernel / rqae_decode_lds_kernel; the hook
// kernels; intensity.  DESIGN.md section 5 describes each with its roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/rqae_hip.h"

namespace {

constexpr int NSLICE = 4;
constexpr int MAGIC = 0x52514145;  // "RQAE"
constexpr int MAX_D = 8192;
constexpr int MAX_C = 16;
constexpr int MAX_MASK_LEVELS = 2048;
constexpr size_t LDS_CODEBOOK_BUDGET = 96 * 1024;

}  // namespace
// Two translation units are built from this file (__graft_entry__.py): the main path's kernels -- the LDS
// encode kernel for d = 2304, packed-VALU arm -- with the ILP-first machine scheduler
// (-mllvm -amdgpu-sched-strategy=max-ilp, -DRQAE_TU_MAIN: +2-4 % on them, but it slows the small shapes and makes
// the generic kernels spill), everything else with the default one.  They share the error text and one launcher.
#ifdef RQAE_TU_MAIN
extern __attribute__((visibility("hidden"))) thread_local char rqae_g_err[512];
#else
__attribute__((visibility("hidden"))) thread_local char rqae_g_err[512] = "";
#endif
extern "C" __attribute__((visibility("hidden"))) int rqae_main_tu_launch(int c, const void *enc_params, void *stream);
namespace {
#define g_err rqae_g_err

int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// ---------------------------------------------------------------- packed block
struct Header {
    int magic, d, c, K, L, dp, shared_cb, pad;
};

// quads (float4) per lane per slice of a pass record: see make_layout
constexpr int quads_of(int floats) { return (floats + 3) / 4; }
constexpr int rec_out_quads(int P, int C) { return quads_of(P * (C + 1)); }   // {w_out[0..C-1], b_out} per element
constexpr int rec_in_quads(int P, int C) { return quads_of(P * C); }          // {w_in[0..C-1]} per element
constexpr int rec_quads(int P, int C) { return rec_out_quads(P, C) + rec_in_quads(P, C); }

// floats of `rows` codebook rows in the LDS kernel's layout (stage_rows below)
constexpr int cb_lds_floats(int C, int rows) { return rows * (C <= 4 ? 4 : 4 + (C - 4)); }

struct Layout {
    size_t w_in, w_out, b_out, b_in, cb, rec_p, cb_img, total;
    int cb_img_pieces;  // 1-KiB pieces of the level-0 codebook in LDS layout
    int dp, nq;  // nq: float4s per lane per slice in a pass record
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

Layout make_layout(int d, int c, int K, int L)
{
    Layout lo;
    lo.dp = ((d + 255) / 256) * 256;
    size_t off = 256;
    lo.w_in = off;  off += align256(sizeof(float) * (size_t)L * c * lo.dp);
    lo.w_out = off; off += align256(sizeof(float) * (size_t)L * c * lo.dp);
    lo.b_out = off; off += align256(sizeof(float) * (size_t)L * lo.dp);
    lo.b_in = off;  off += align256(sizeof(float) * (size_t)L * c);
    lo.cb = off;    off += align256(sizeof(float) * (size_t)L * K * c);
    // Pass records for the LDS-staged kernels.  Pass pl (0..L) applies level pl-1 and projects onto
    // level pl; its weights are one contiguous block, cut into 1-KiB pieces that LDS-DMA copies
    // verbatim:  rec_p[pl][slice][quad][lane][4].  A lane's floats are two packed streams over its
    // P element steps (el = elem_of(P, slice, i, lane)):
    //   quads [0, QOUT):    n = i*(c+1) + k   { w_out[pl-1][el][0..c-1], b_out[pl-1][el] }
    //   quads [QOUT, NQ):   n = i*c + j       { w_in[pl][j][el] }
    lo.nq = rec_quads(lo.dp / 256, c);
    lo.rec_p = off; off += align256(sizeof(float) * (size_t)(L + 1) * NSLICE * lo.nq * 256);
    // Level-0 codebook exactly as the LDS kernel keeps it ([K][4] head columns, then [K][c-4]), padded to whole
    // 1-KiB pieces, so a workgroup stages it by LDS-DMA without touching a register.
    lo.cb_img_pieces = (cb_lds_floats(c, K) + 255) / 256;
    lo.cb_img = off; off += align256(sizeof(float) * (size_t)lo.cb_img_pieces * 256);
    lo.total = off;
    return lo;
}

// Canonical lane<->element map (identical to elem_of in oracle/rqae_oracle.c):
// slice s owns [s*64P, (s+1)*64P); a lane's P element steps are laid out in memory in runs of 4, then
// 2, then 1 consecutive elements, so tiles load and store with dwordx4 / dwordx2.
__host__ __device__ __forceinline__ int elem_of(int P, int s, int i, int lane)
{
    const int base = s * 64 * P;
    const int q4 = (P / 4) * 4;
    if (i < q4) return base + 256 * (i / 4) + 4 * lane + (i % 4);
    const int rem = P - q4, j = i - q4;
    if (rem >= 2 && j < 2) return base + 64 * q4 + 2 * lane + j;
    return base + 64 * q4 + (rem >= 2 ? 128 : 0) + lane;
}

bool shape_ok(int d, int c, int K, int L)
{
    return d >= 1 && d <= MAX_D && c >= 1 && c <= MAX_C && K >= 1 && L >= 1;
}

// Fill the pass records (see make_layout).  One thread per float.
__global__ void pack_records_kernel(Header h, int nq, int qout, float *rec_p, const float *w_in, const float *w_out,
                                    const float *b_out)
{
    const int P = h.dp / 256;
    const size_t total = (size_t)(h.L + 1) * NSLICE * nq * 256;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += stride) {
        const int e4 = (int)(u & 3), lane = (int)((u >> 2) & 63);
        size_t rest = u >> 8;                      // (pl*4 + slice)*nq + quad
        const int quad = (int)(rest % nq); rest /= nq;
        const int slice = (int)(rest & 3);
        const size_t pl = rest >> 2;
        float val = 0.0f;
        if (quad < qout) {                         // lin_out stream of level pl-1
            const int n = quad * 4 + e4, i = n / (h.c + 1), k = n % (h.c + 1);
            if (i < P && pl >= 1) {
                const int el = elem_of(P, slice, i, lane);
                if (el < h.d) val = k < h.c ? w_out[((pl - 1) * h.d + el) * h.c + k] : b_out[(pl - 1) * h.d + el];
            }
        } else {                                   // lin_in stream of level pl
            const int n = (quad - qout) * 4 + e4, i = n / h.c, j = n % h.c;
            if (i < P && pl < (size_t)h.L) {
                const int el = elem_of(P, slice, i, lane);
                if (el < h.d) val = w_in[(pl * h.c + j) * h.d + el];
            }
        }
        rec_p[u] = val;
    }
}

// Level-0 codebook in the LDS layout of stage_rows (see make_layout).  One thread per float of the image.
__global__ void pack_cb_image_kernel(int c, int K, int n_img, float *img, const float *codebook)
{
    const int tail = c > 4 ? c - 4 : 0;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_img; u += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (u < K * 4) {
            const int r = u >> 2, k = u & 3;
            if (k < c) v = codebook[(size_t)r * c + k];
        } else if (u < K * 4 + K * tail) {
            const int r = (u - K * 4) / tail, k = 4 + (u - K * 4) % tail;
            v = codebook[(size_t)r * c + k];
        }
        img[u] = v;
    }
}

__global__ void pack_kernel(Header *hdr, Header h, float *w_in_p, float *w_out_p, float *b_out_p,
                            const float *w_in, const float *w_out, const float *b_out)
{
    const size_t n_w = (size_t)h.L * h.c * h.dp;
    const size_t n_b = (size_t)h.L * h.dp;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_w; i += stride) {
        const int e = (int)(i % h.dp);
        const size_t lj = i / h.dp;  // l*c + j
        const int j = (int)(lj % h.c);
        const size_t l = lj / h.c;
        w_in_p[i] = e < h.d ? w_in[lj * h.d + e] : 0.0f;
        w_out_p[i] = e < h.d ? w_out[(l * h.d + e) * h.c + j] : 0.0f;  // transposed: [l][k][e]
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_b; i += stride) {
        const int e = (int)(i % h.dp);
        const size_t l = i / h.dp;
        b_out_p[i] = e < h.d ? b_out[l * h.d + e] : 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = h;
}

// ------------------------------------------------------------ cross-lane moves
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// DPP controls: row_ror:8 (lane^8 inside a row of 16), row_half_mirror (i <-> 7-i inside 8),
// quad_perm [1,0,3,2] (lane^1), quad_perm [2,3,0,1] (lane^2)
constexpr int DPP_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;

// v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b.  (The builtin's second result is
// miscompiled by this toolchain, hence asm; s_nop 1 = the VALU-write -> permlane-read hazard.)
__device__ __forceinline__ void swap32(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// v_permlane16_swap: odd rows (of 16) of a <-> even rows of b
__device__ __forceinline__ void swap16(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// One halving step of the reduce-scatter: two registers of per-lane partials in, one out.
// Lanes whose STAGE bit is 0 keep value a (own + partner), the others keep value b.
template <int OP>
__device__ __forceinline__ float rs_op(float a, float b)
{
    if constexpr (OP == 0) return a + b;
    else return fmaxf(a, b);
}

template <int STAGE, int OP = 0>
__device__ __forceinline__ float rs_pair(float a, float b, int lane)
{
    if constexpr (STAGE == 0) {
        swap32(a, b);
        return rs_op<OP>(a, b);
    } else if constexpr (STAGE == 1) {
        swap16(a, b);
        return rs_op<OP>(a, b);
    } else {
        constexpr int bit = STAGE == 2 ? 8 : STAGE == 3 ? 4 : STAGE == 4 ? 1 : 2;
        constexpr int ctrl = STAGE == 2 ? DPP_ROR8 : STAGE == 3 ? DPP_HALF_MIRROR
                                                  : STAGE == 4 ? DPP_XOR1 : DPP_XOR2;
        const bool lo = (lane & bit) == 0;
        const float keep = lo ? a : b;
        const float give = lo ? b : a;
        return rs_op<OP>(keep, dpp_mov<ctrl>(give));
    }
}

template <int STAGE, int OP = 0>
__device__ __forceinline__ float rs_single(float a)
{
    if constexpr (STAGE == 0) {
        float b = a;
        swap32(a, b);
        return rs_op<OP>(a, b);
    } else if constexpr (STAGE == 1) {
        float b = a;
        swap16(a, b);
        return rs_op<OP>(a, b);
    } else {
        constexpr int ctrl = STAGE == 2 ? DPP_ROR8 : STAGE == 3 ? DPP_HALF_MIRROR
                                                  : STAGE == 4 ? DPP_XOR1 : DPP_XOR2;
        return rs_op<OP>(a, dpp_mov<ctrl>(a));
    }
}

// N per-lane partial sums -> ceil(N/64) registers; register f, lane -> the complete
// 64-lane sum of value 64*f + rs_value_of_lane(lane) (lanes whose index is >= N hold copies).
template <int N, int STAGE, int NF, int OP = 0>
struct ReduceScatter {
    static __device__ __forceinline__ void run(const float (&in)[N], float (&fin)[NF], int lane)
    {
        if constexpr (STAGE == 6) {
            static_assert(N == NF, "six halvings leave ceil(N0/64) registers");
#pragma unroll
            for (int f = 0; f < NF; ++f) fin[f] = in[f];
        } else {
            constexpr int M = (N + 1) / 2;
            float out[M];
#pragma unroll
            for (int m = 0; m < N / 2; ++m) out[m] = rs_pair<STAGE, OP>(in[2 * m], in[2 * m + 1], lane);
            if constexpr (N & 1) out[M - 1] = rs_single<STAGE, OP>(in[N - 1]);
            ReduceScatter<M, STAGE + 1, NF, OP>::run(out, fin, lane);
        }
    }
};

__device__ __forceinline__ int rs_value_of_lane(int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) |
           (((lane >> 2) & 1) << 3) | ((lane & 1) << 4) | (((lane >> 1) & 1) << 5);
}

// ------------------------------------------------------------ per-token scalar math
// ||q||_2 in ATen's order (see file header); cp = live length (== C unless GEN).
template <int C>
__device__ __forceinline__ float aten_norm2(const float (&q)[C], int cp)
{
    const int n8 = (cp / 8) * 8;
    float acc = 0.0f;
    if (n8) {
        float buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int d0 = 0; d0 + 8 <= C; d0 += 8)
            if (d0 + 8 <= n8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) buf[j] = fmaf(q[d0 + j], q[d0 + j], buf[j]);
            }
        acc = buf[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc = acc + buf[j];
    }
    int tail = n8;
#pragma unroll
    for (int d0 = 0; d0 + 4 <= C; d0 += 8)
        if (d0 == n8 && cp - d0 >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sq = q[d0 + j] * q[d0 + j];
                acc = acc + sq;
            }
            tail = d0 + 4;
        }
#pragma unroll
    for (int e = 0; e < C; ++e)
        if (e >= tail && e < cp) acc = fmaf(q[e], q[e], acc);
    return __builtin_sqrtf(acc);
}

// Running (best, index) over the codebook rows staged in LDS; lane owns rows lane, lane+64, ...
// in ascending order and updates on strictly-greater only => first maximum inside the lane.
template <int C, int TQ>
__device__ __forceinline__ void scan_rows(const float (&xn)[TQ][C], const float *cb_lds, int rows,
                                          int row_base, int cp, int lane, float (&best)[TQ],
                                          int (&bidx)[TQ])
{
    for (int r0 = 0; r0 < rows; r0 += 64) {
        const int row = r0 + lane;
        if (row < rows) {
            float cbv[C];
#pragma unroll
            for (int k = 0; k < C; ++k) cbv[k] = k < cp ? cb_lds[row * cp + k] : 0.0f;
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < C; ++k)
                    if (k < cp) s = fmaf(xn[tq][k], cbv[k], s);
                if (s > best[tq]) {
                    best[tq] = s;
                    bidx[tq] = row_base + row;
                }
            }
        }
    }
}

// 64-lane argmax with lowest-index tie rule (rqae/model.py:182: torch.argmax).
__device__ __forceinline__ void wave_argmax(float &best, int &bidx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
}

__device__ __forceinline__ float load_x(const void *x, int x_dtype, size_t off)
{
    if (x_dtype == RQAE_X_F32) return reinterpret_cast<const float *>(x)[off];
    if (x_dtype == RQAE_X_BF16) {
        const uint32_t u = reinterpret_cast<const uint16_t *>(x)[off];
        return __builtin_bit_cast(float, u << 16);
    }
    return (float)reinterpret_cast<const _Float16 *>(x)[off];
}

// All threads of the workgroup copy n floats global -> LDS.
__device__ __forceinline__ void stage_floats(float *dst, const float *src, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------ encode kernel
struct EncParams {
    const float *w_in, *w_out, *b_out, *b_in, *cb, *rec_p;
    const Header *hdr;
    const void *x;
    float *q_out;
    long long *idx;
    long long n_tokens;
    int d, dp, c, K, levels, x_dtype;
    int kch;  // codebook rows staged in LDS per chunk (multiple of 64, or >= K)
    int aligned16;  // x and q_out 16-byte aligned: full tiles may use vector loads/stores
    int ablate;  // RQAE_HIP_ABLATE bits: timing experiments only, results are wrong when != 0
    int tl_lds_floats;  // RQAE_TIMELINE builds: float offset of the stamp buffer in LDS
    const float *cb_img;  // level-0 codebook in LDS layout (make_layout), cb_img_pieces KiB
    int cb_img_pieces;
    const float *norm_w;  // NORM kernels: Gemma RMSNorm weight [d], eps, and where 1/rms per token goes
    float *inv_rms;
    float norm_eps;
};

#ifndef RQAE_TU_MAIN   // (the main path's translation unit carries only rqae_encode_lds_kernel, see the end of this section)
// P  : elements per lane per slice (dp = 256*P)      C : codebook_dim
// TPW: tokens held by one wave                        NH: token halves (waves = 4*NH)
// GEN: P and C are upper bounds, the live sizes come from EncParams (slow generic path)
template <int P, int C, int TPW, int NH, bool RECON, bool GEN>
__global__ void __launch_bounds__(256 * NH, NH)
rqae_encode_kernel(const EncParams p)
{
    constexpr int NW = NSLICE * NH;
    constexpr int TG = TPW * NH;
    constexpr int TQ = (TG + NW - 1) / NW;
    constexpr int NV = TPW * C;
    constexpr int NF = (NV + 63) / 64;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *part = smem;                       // [NH][NSLICE][NV]
    float *cvs = part + NH * NSLICE * NV;     // [TG][C]
    float *cbl = cvs + ((TG * C + 3) & ~3);   // [kch][cp]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = wave & 3;
    const int half = wave >> 2;
    const int cp = GEN ? p.c : C;
    const int Pp = GEN ? p.dp / 256 : P;
    const int dp = GEN ? p.dp : 256 * P;
    auto el = [&](int i) { return elem_of(Pp, slice, i, lane); };   // this lane's element at step i
    const long long tok0 = (long long)blockIdx.x * TG;  // first token of the workgroup
    const long long tokw = tok0 + half * TPW;           // first token of this wave

    const bool shared_cb = p.hdr->shared_cb != 0;
    const int nchunk = (p.K + p.kch - 1) / p.kch;
    const bool stage_once = shared_cb && nchunk == 1;

    float r[P][TPW];
    float out[RECON ? P : 1][RECON ? TPW : 1];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int e = el(i);
            const long long tok = tokw + t;
            const bool ok = (!GEN || i < Pp) && e < p.d && tok < p.n_tokens;
            r[i][t] = ok ? load_x(p.x, p.x_dtype, (size_t)tok * p.d + e) : 0.0f;
            if constexpr (RECON) out[i][t] = 0.0f;
        }

    for (int l = 0; l < p.levels; ++l) {
        // ---- lin_in partials: acc[t*C + j] = sum_i r[i][t] * W_in[l][j][e_i]   (model.py:211)
        float acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0f;
        {
            const float *wl = p.w_in + (size_t)l * cp * dp;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (GEN && i >= Pp) continue;
                float w[C];
#pragma unroll
                for (int j = 0; j < C; ++j) w[j] = (!GEN || j < cp) ? wl[(size_t)j * dp + el(i)] : 0.0f;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (!GEN || j < cp) acc[t * C + j] = fmaf(r[i][t], w[j], acc[t * C + j]);
            }
        }
        // ---- 64-lane tree per value, results scattered over lanes, then to LDS
        {
            float fin[NF];
            ReduceScatter<NV, 0, NF>::run(acc, fin, lane);
            const int v0 = rs_value_of_lane(lane);
            float *dst = part + (half * NSLICE + slice) * NV;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (64 * f + v0 < NV) dst[64 * f + v0] = fin[f];
        }
        if (l == 0 ? true : !stage_once) {
            const int rows = p.kch < p.K ? p.kch : p.K;
            stage_floats(cbl, p.cb + (size_t)(shared_cb ? 0 : l) * p.K * cp, rows * cp);
        }
        __syncthreads();

        // ---- per token: q, ||q||, cosine argmax, STE value            (model.py:187-193, 218-220)
        {
            float q[TQ][C], xn[TQ][C], best[TQ];
            int bidx[TQ];
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const int tl = wave * TQ + tq;  // token inside the workgroup
                const int th = tl / TPW, tt = tl % TPW;
                const float *ps = part + (size_t)th * NSLICE * NV + tt * C;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if ((GEN && j >= cp) || tl >= TG) { q[tq][j] = 0.0f; continue; }
                    const float s01 = ps[0 * NV + j] + ps[1 * NV + j];
                    const float s23 = ps[2 * NV + j] + ps[3 * NV + j];
                    q[tq][j] = (s01 + s23) + p.b_in[l * cp + j];
                }
                const float nrm = aten_norm2<C>(q[tq], cp);
#pragma unroll
                for (int j = 0; j < C; ++j) xn[tq][j] = q[tq][j] / nrm;
                best[tq] = -INFINITY;
                bidx[tq] = 0;
            }
            for (int ch = 0; ch < nchunk; ++ch) {
                const int row_base = ch * p.kch;
                const int rows = (p.K - row_base) < p.kch ? (p.K - row_base) : p.kch;
                if (ch > 0) {
                    __syncthreads();
                    stage_floats(cbl, p.cb + ((size_t)(shared_cb ? 0 : l) * p.K + row_base) * cp, rows * cp);
                    __syncthreads();
                }
                if (wave * TQ < TG) scan_rows<C, TQ>(xn, cbl, rows, row_base, cp, lane, best, bidx);
            }
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const int tl = wave * TQ + tq;
                if (tl >= TG) continue;
                wave_argmax(best[tq], bidx[tq]);
                const float *row = p.cb + ((size_t)(shared_cb ? 0 : l) * p.K + bidx[tq]) * cp;
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (!GEN || j < cp) {
                            const float diff = row[j] - q[tq][j];
                            cvs[tl * C + j] = q[tq][j] + diff;
                        }
                    const long long tok = tok0 + tl;
                    if (tok < p.n_tokens) p.idx[tok * p.levels + l] = bidx[tq];
                }
            }
        }
        __syncthreads();

        // ---- lin_out, residual and reconstruction update               (model.py:221-224)
        {
            float cv[TPW][C];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int k = 0; k < C; ++k) cv[t][k] = cvs[(half * TPW + t) * C + k];
            const float *wl = p.w_out + (size_t)l * cp * dp;
            const float *bl = p.b_out + (size_t)l * dp;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (GEN && i >= Pp) continue;
                float w[C];
#pragma unroll
                for (int k = 0; k < C; ++k) w[k] = (!GEN || k < cp) ? wl[(size_t)k * dp + el(i)] : 0.0f;
                const float b = bl[el(i)];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    float y = 0.0f;
#pragma unroll
                    for (int k = 0; k < C; ++k)
                        if (!GEN || k < cp) y = fmaf(cv[t][k], w[k], y);
                    y = y + b;
                    r[i][t] = r[i][t] - y;
                    if constexpr (RECON) out[i][t] = out[i][t] + y;  // level 0: 0 + y (model.py:201)
                }
            }
        }
        // the next level's writes to `part` happen after this level's reads (barrier above);
        // its reads of `cvs` happen after the barrier that follows the next writes.
    }

    if constexpr (RECON) {
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int e = el(i);
                const long long tok = tokw + t;
                if ((!GEN || i < Pp) && e < p.d && tok < p.n_tokens) p.q_out[(size_t)tok * p.d + e] = out[i][t];
            }
    }
}
#endif  // !RQAE_TU_MAIN



// ---- token-tile load/store for the fused kernel: lane e-offsets 64*i are immediates, the
// dtype switch and the bounds predicates are hoisted out of the element loops
template <typename XT>
__device__ __forceinline__ float widen(XT v);
template <>
__device__ __forceinline__ float widen<float>(float v) { return v; }
template <>
__device__ __forceinline__ float widen<uint16_t>(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
template <>
__device__ __forceinline__ float widen<_Float16>(_Float16 v) { return (float)v; }

template <typename XT, int N>
struct VecOf { typedef XT type __attribute__((ext_vector_type(N))); };

// One token row segment -> registers.  FULL: the lane's runs of 4 / 2 / 1 consecutive elements are
// fetched with one load each (elem_of); otherwise element by element with bounds predicates.
template <int P, int TPW, typename XT, bool FULL>
__device__ __forceinline__ void load_tile(float (&dst)[P][TPW], const XT *x, long long tokw, long long n_tokens,
                                          int d, int slice, int lane)
{
    constexpr int Q4 = (P / 4) * 4, REM = P - Q4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const long long tok = tokw + t;
        const XT *xr = x + (size_t)(FULL ? tok : (tok < n_tokens ? tok : 0)) * d;
        if constexpr (FULL) {
#pragma unroll
            for (int q = 0; q < Q4; q += 4) {
                const auto v = *reinterpret_cast<const typename VecOf<XT, 4>::type *>(xr + elem_of(P, slice, q, lane));
                dst[q][t] = widen<XT>(v.x); dst[q + 1][t] = widen<XT>(v.y);
                dst[q + 2][t] = widen<XT>(v.z); dst[q + 3][t] = widen<XT>(v.w);
            }
            if constexpr (REM >= 2) {
                const auto v = *reinterpret_cast<const typename VecOf<XT, 2>::type *>(xr + elem_of(P, slice, Q4, lane));
                dst[Q4][t] = widen<XT>(v.x); dst[Q4 + 1][t] = widen<XT>(v.y);
            }
            if constexpr (REM & 1) dst[P - 1][t] = widen<XT>(xr[elem_of(P, slice, P - 1, lane)]);
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int e = elem_of(P, slice, i, lane);
                dst[i][t] = (tok < n_tokens && e < d) ? widen<XT>(xr[e]) : 0.0f;
            }
        }
    }
}

// Store the run of a lane's consecutive elements that ends at element step i (if one ends there):
// v(k) yields the value of step k.  row = start of the token's output row.
template <int P, typename F>
__device__ __forceinline__ void store_run_ending_at(float *row, int slice, int lane, int i, F v)
{
    constexpr int Q4 = (P / 4) * 4, REM = P - Q4;
    if (i < Q4) {
        if ((i & 3) == 3) {
            typedef float f4s __attribute__((ext_vector_type(4)));
            *reinterpret_cast<f4s *>(row + elem_of(P, slice, i - 3, lane)) = (f4s){v(i - 3), v(i - 2), v(i - 1), v(i)};
        }
    } else if (REM >= 2 && i - Q4 < 2) {
        if (i - Q4 == 1) {
            typedef float f2s __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f2s *>(row + elem_of(P, slice, Q4, lane)) = (f2s){v(i - 1), v(i)};
        }
    } else {
        row[elem_of(P, slice, i, lane)] = v(i);
    }
}



// =============================================================== LDS-staged encode kernel
// Same arithmetic as rqae_encode_kernel, organised for throughput (exact shapes, d = 256*P):
//   * 8 waves = 4 slices x 2 token halves, 8 tokens per wave, 16 tokens per workgroup; the
//     residual and the reconstruction of a wave (2 x 72 VGPRs at P = 9) stay in the 256
//     VALU-addressable registers for all levels (AGPRs are not VALU operands on gfx950)
//   * lin_out(l), r -= y, out += y and lin_in(l+1) are ONE pass over those registers, in packed
//     fp32 math over token pairs (v_pk_fma_f32 / v_pk_add_f32)
//   * a pass's weights (one record, make_layout) are copied L2 -> LDS by LDS-DMA while the
//     quantise phase of the previous level runs, and read back with ds_read_b128; both token
//     halves share them, so each weight byte crosses the CU's L1/TA once per level
//   * the quantise phase is lane-parallel: lane (token, j) owns q[token][j] (one IEEE divide per
//     lane), unit vectors reach every lane as scalars (v_readlane), codebook rows come from LDS
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 splat2(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
__device__ __forceinline__ float lane_bcast_first(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Codebook rows in LDS for the LDS kernel: columns 0..3 of every row as one float4 (conflict-free
// ds_read_b128, 16-byte lane stride), the remaining C-4 columns in a second array behind them.
template <int C>
__device__ __forceinline__ void stage_rows(float *dst, const float *src, int rows)
{
    constexpr int CB = C > 4 ? C - 4 : 1;
    constexpr int BATCH = 8;   // independent loads in flight per thread: one L2 latency per batch, not per float
    float *tailp = dst + (size_t)rows * 4;
    const int n = rows * C;
    for (int u0 = threadIdx.x; u0 < n; u0 += BATCH * blockDim.x) {
        float v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int u = u0 + b * blockDim.x;
            v[b] = u < n ? src[u] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int u = u0 + b * blockDim.x;
            if (u < n) {
                const int r = u / C, k = u - r * C;
                if (k < 4) dst[r * 4 + k] = v[b];
                else tailp[r * CB + (k - 4)] = v[b];
            }
        }
    }
    if constexpr (C < 4) {   // zero the unused columns (they are read as part of the float4)
        for (int u = threadIdx.x; u < rows * (4 - C); u += blockDim.x) {
            const int r = u / (4 - C), k = C + u % (4 - C);
            dst[r * 4 + k] = 0.0f;
        }
    }
}
template <int C>
__device__ __forceinline__ void load_row(float (&cbv)[C], const float *cbl_base, int row, int rows_staged)
{
    const float4 h = *reinterpret_cast<const float4 *>(cbl_base + row * 4);
    const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int k = 0; k < C && k < 4; ++k) cbv[k] = hv[k];
    if constexpr (C > 4) {
        const float *tailp = cbl_base + (size_t)rows_staged * 4 + row * (C - 4);
#pragma unroll
        for (int k = 4; k < C; ++k) cbv[k] = tailp[k - 4];
    }
}
template <int C>
__device__ __forceinline__ float row_elem(const float *cbl_base, int row, int k, int rows_staged)
{
    if (C <= 4 || k < 4) return cbl_base[row * 4 + k];
    return cbl_base[(size_t)rows_staged * 4 + row * (C - 4) + (k - 4)];
}

// NQ consecutive 1-KiB pieces by LDS-DMA with scalar addressing: global source = uniform base +
// lane*16 + immediate, LDS destination = M0 + the SAME immediate + lane*16 (the instruction offset
// applies to both sides: probed on hardware), so consecutive pieces need only a new immediate.  Written in
// asm on purpose: after the builtin form hipcc puts s_waitcnt vmcnt(0) in front of the next ds_read,
// which serialises the transfer with the quantise phase it is meant to hide behind (measured: the
// builtin cost 0.4 ms per launch whether or not the quantise phase ran).  M0 is saved/restored; the
// reader side waits with dma_wait_all() before the barrier that publishes the data.
template <int NQ>
__device__ __forceinline__ void dma_pieces(const float *g_uniform_in, unsigned lds_byte_addr_in, unsigned lane_byte_off)
{
    static_assert(NQ * 1024 <= 4096, "immediate offset range of global_load_lds");
    // "s" operands must be provably wave-uniform for hipcc: say so explicitly
    const unsigned long long ga = reinterpret_cast<unsigned long long>(g_uniform_in);
    const unsigned long long gu = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga) |
                                  ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) << 32);
    const float *g_uniform = reinterpret_cast<const float *>(gu);
    const unsigned lds_byte_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte_addr_in);
    unsigned keep;
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_byte_off), "s"(g_uniform), "s"(lds_byte_addr), "i"(qd * 1024)
                     : "memory");
    }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const float *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p;
}

// MFMA: lin_out and the cosine contraction run on the matrix pipe as v_mfma_f32_4x4x1_16b_f32 with
// the A operand broadcast from block 0 (cbsz=4): D[lane][i] = A[lane i] * B[lane] + C[lane][i], i.e.
// lane = element (or codebook row), the four accumulator registers = four tokens.  K = 1 per
// instruction, so a chain of c instructions is the k-ordered fmaf chain of the contract, bit for bit
// (checked on hardware by the parity tests), with no padding of c.
// Timeline probe (build with -DRQAE_TIMELINE, scratch/timeline.sh; never part of the shipped library):
// every wave of 4 sample workgroups stamps s_memtime at its phase boundaries.
#ifdef RQAE_TIMELINE
__device__ long long g_timeline[4][8][64];
#define TL_STAMP()                                                        \
    do {                                                                  \
        if (tl_on) {                                                      \
            const long long tl_t = __builtin_readcyclecounter();          \
            if (lane == 0 && tl_n < 64) tl_buf[wave * 64 + tl_n] = tl_t;  \
        }                                                                 \
        ++tl_n;                                                           \
    } while (0)
#else
#define TL_STAMP() do { } while (0)
#endif

// Lane-private weight streams of a pass record: the lane's floats [i*W, (i+1)*W) for element step i,
// stored as quads q at base[q * 64] (f4 units).  All indices are compile-time after unrolling.
constexpr int last_quad_of_step(int i, int W) { return ((i + 1) * W - 1) / 4; }
__device__ __forceinline__ float f4_get(const f4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
// acc += a * broadcast(component c of quad) as ONE v_pk_fma_f32 whose op_sel picks the component out of the
// quad's aligned register pair.  Written in asm because with four users per weight hipcc materialises the
// splat (two v_mov per weight, +40 VGPRs of pressure and spills) instead of folding it into op_sel.
// c must be a constant after unrolling.
__device__ __forceinline__ void pk_fma_bcast(f2 &acc, f2 a, const f4 &quad, int c)
{
    const f2 pair = c < 2 ? __builtin_shufflevector(quad, quad, 0, 1) : __builtin_shufflevector(quad, quad, 2, 3);
    if ((c & 1) == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(pair));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(pair));
}
// The same with a wave-uniform multiplicand in an SGPR pair (lin_out: y += cv * w), as `asm volatile`: the
// order in which these are written is the order in which they issue.  lin_out's four token pairs are written
// k-outermost, so that consecutive instructions are independent -- left to itself hipcc runs the four
// eight-deep dependent chains one after the other (one issue per ~10 cycles per wave instead of ~6).
// first: the chain starts from +0 (y = cv * w + 0).
__device__ __forceinline__ void pk_fma_bcast_s(f2 &y, f2 cv_uniform, const f4 &quad, int c, bool first)
{
    const f2 pair = c < 2 ? __builtin_shufflevector(quad, quad, 0, 1) : __builtin_shufflevector(quad, quad, 2, 3);
    if (first) {
        if ((c & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(y) : "s"(cv_uniform), "v"(pair));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(y) : "s"(cv_uniform), "v"(pair));
    } else {
        if ((c & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(y) : "s"(cv_uniform), "v"(pair));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(y) : "s"(cv_uniform), "v"(pair));
    }
}
// y += broadcast(component c of quad)
__device__ __forceinline__ void pk_add_bcast(f2 &y, const f4 &quad, int c)
{
    const f2 pair = c < 2 ? __builtin_shufflevector(quad, quad, 0, 1) : __builtin_shufflevector(quad, quad, 2, 3);
    if ((c & 1) == 0) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(y) : "v"(pair));
    else asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(y) : "v"(pair));
}

// NORM: x is the hooked LLM layer's output; the Gemma RMSNorm of RQAE.hook (rqae/model.py:279, rqae/llm.py:65-66)
// is applied while the tile is in registers (rqae_hip_encode_gemma), normed never exists in HBM.
template <int P, int C, bool RECON, bool MFMA, bool NORM = false>
__global__ void __launch_bounds__(512, 2)
rqae_encode_lds_kernel(const EncParams p)
{
    constexpr int TPW = 8, NW = 8, TG = 16, TQ = 2, JW = 8, NG = 2, NPAIR = 4;
    static_assert(C <= 8, "quantise phase uses 8 lanes per token");
    constexpr int QOUT = rec_out_quads(P, C), QIN = rec_in_quads(P, C), NQ = QOUT + QIN;
    constexpr int NPIECE = NSLICE * NQ;         // 1-KiB pieces in one pass record
    constexpr int WFLOATS = NPIECE * 256;
    constexpr int dp = 256 * P;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wts = smem;                       // [slice][quad][lane][4]
    float *part = wts + WFLOATS;             // [TG][C][NSLICE]: the 4 slice partials of one (token, j) are contiguous
    float *cvs = part + NSLICE * TG * C;     // [C][TG]  (token pairs contiguous)
    float *cbl = cvs + ((TG * C + 3) & ~3);  // [kch][C]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = wave & 3;
    const int half = wave >> 2;
#ifdef RQAE_TIMELINE
    const bool tl_on = (blockIdx.x % 4096) == 2053 && blockIdx.x / 4096 < 4;
    int tl_n = 0;
    long long *tl_buf = reinterpret_cast<long long *>(smem + p.tl_lds_floats);
    if (tl_on) tl_buf[wave * 64 + lane] = 0;
    TL_STAMP();
#endif
    const long long tok0 = (long long)blockIdx.x * TG;
    const long long tokw = tok0 + half * TPW;

    const bool shared_cb = p.hdr->shared_cb != 0;
    const int nchunk = (p.K + p.kch - 1) / p.kch;
    const bool stage_once = shared_cb && nchunk == 1;

    // quantise-phase role of this lane: lanes 0..7 -> token 2*wave, lanes 8..15 -> token 2*wave+1
    const int tq = lane / JW, jj = lane % JW;
    const int tl_q = wave * TQ + (tq < TQ ? tq : 0);
    const bool act = tq < TQ && jj < C;

    const unsigned wts_addr = lds_addr_of(wts);
    // The record of pass pl goes to `wts` in 1-KiB pieces; every wave owns a contiguous run of them and issues
    // it a few pieces at a time from inside its quantise phase (dma_trip, once per 256 codebook rows): a burst
    // of ~13 global_load_lds per wave blocks on the TA's address queue for ~1 800 cycles per level (measured,
    // profiles/r01_timeline_*).  Four consecutive pieces share one M0 set-up (immediate offsets).
    int dma_next = 0, dma_end = 0;
    const float *dma_src = p.rec_p;
    auto dma_begin = [&](int pl) {
        constexpr int PER = (NPIECE + NW - 1) / NW;
        dma_src = p.rec_p + (size_t)pl * WFLOATS;
        dma_next = wave * PER;
        dma_end = dma_next + PER < NPIECE ? dma_next + PER : NPIECE;
    };
    auto dma_trip = [&]() __attribute__((always_inline)) {
        if (dma_end - dma_next >= 4) {
            dma_pieces<4>(dma_src + dma_next * 256, wts_addr + dma_next * 1024, lane * 16);
            dma_next += 4;
        } else if (dma_next < dma_end) {
            dma_pieces<1>(dma_src + dma_next * 256, wts_addr + dma_next * 1024, lane * 16);
            dma_next += 1;
        }
    };
    auto dma_flush = [&]() { while (dma_next < dma_end) dma_trip(); };

    {   // record 0: only its lin_in part is read
        for (int s = 0; s < NSLICE; ++s)
            for (int q = QOUT + wave; q < NQ; q += NW)
                dma_pieces<1>(p.rec_p + (s * NQ + q) * 256, wts_addr + (s * NQ + q) * 1024, lane * 16);
        if (nchunk == 1) {   // level-0 codebook: its LDS image by DMA -- no registers, no wait in front of the x loads
            const unsigned cbl_addr = lds_addr_of(cbl);
            for (int pc = wave; pc < p.cb_img_pieces; pc += NW)
                dma_pieces<1>(p.cb_img + pc * 256, cbl_addr + pc * 1024, lane * 16);
        } else {
            stage_rows<C>(cbl, p.cb, p.kch);
        }
    }

    f2 r2[P][NPAIR];
    f2 out2[RECON ? P : 1][RECON ? NPAIR : 1];
    const bool full_tile = (tokw + TPW <= p.n_tokens) && (p.d == dp) && p.aligned16;  // wave-uniform: vector loads/stores
    {
        float xt[P][TPW];
        if (full_tile) {
            if (p.x_dtype == RQAE_X_BF16) load_tile<P, TPW, uint16_t, true>(xt, (const uint16_t *)p.x, tokw, p.n_tokens, p.d, slice, lane);
            else if (p.x_dtype == RQAE_X_F32) load_tile<P, TPW, float, true>(xt, (const float *)p.x, tokw, p.n_tokens, p.d, slice, lane);
            else load_tile<P, TPW, _Float16, true>(xt, (const _Float16 *)p.x, tokw, p.n_tokens, p.d, slice, lane);
        } else {
            if (p.x_dtype == RQAE_X_BF16) load_tile<P, TPW, uint16_t, false>(xt, (const uint16_t *)p.x, tokw, p.n_tokens, p.d, slice, lane);
            else if (p.x_dtype == RQAE_X_F32) load_tile<P, TPW, float, false>(xt, (const float *)p.x, tokw, p.n_tokens, p.d, slice, lane);
            else load_tile<P, TPW, _Float16, false>(xt, (const _Float16 *)p.x, tokw, p.n_tokens, p.d, slice, lane);
        }
        if constexpr (NORM) {
            // normed = (x * rsqrt(mean(x^2) + eps)) * (1 + w); the row sum of squares in the canonical order of
            // section 3 (this wave's quarter: chain over the element steps, 64-lane tree; slices through LDS)
            float ssv[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                float a = 0.0f;
#pragma unroll
                for (int i = 0; i < P; ++i) a = fmaf(xt[i][t], xt[i][t], a);
                ssv[t] = a;
            }
            float fin[1];
            ReduceScatter<TPW, 0, 1>::run(ssv, fin, lane);     // lane holds token (bit5 + 2 bit4 + 4 bit3)
            if ((lane & 7) == 0) {
                const int t = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);
                part[(half * TPW + t) * NSLICE + slice] = fin[0];
            }
            float w1[P][1];                                     // 1 + norm weight of this lane's elements
            if (full_tile) load_tile<P, 1, float, true>(w1, p.norm_w, 0, 1, p.d, slice, lane);
            else load_tile<P, 1, float, false>(w1, p.norm_w, 0, 1, p.d, slice, lane);
            __syncthreads();
            float rs_mine = 0.0f;
            if (lane < TPW) {
                const f4 ps = *reinterpret_cast<const f4 *>(part + (half * TPW + lane) * NSLICE);
                const float total = (ps.x + ps.y) + (ps.z + ps.w);
                const float ms = total / (float)p.d;
                rs_mine = 1.0f / __builtin_sqrtf(ms + p.norm_eps);
                if (slice == 0 && tokw + lane < p.n_tokens) p.inv_rms[tokw + lane] = rs_mine;
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float rs = lane_bcast(rs_mine, t);
#pragma unroll
                for (int i = 0; i < P; ++i) xt[i][t] = (xt[i][t] * rs) * (1.0f + w1[i][0]);
            }
        }
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) r2[i][pr] = (f2){xt[i][2 * pr], xt[i][2 * pr + 1]};
    }
#ifdef RQAE_TIMELINE
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");   // (also waits for record 0: stamp 1 = x + record landed)
#endif
    TL_STAMP();

    const f4 *wl = reinterpret_cast<const f4 *>(wts) + (size_t)slice * NQ * 64 + lane;

    // ---- sweep 1 of a pass: apply a level.  y = lin_out(cv); r -= y; out += y, for the wave's 8 tokens per
    // element step, so every weight quad is read from LDS once per wave (read per token group, LDS bandwidth
    // bounds the pass).  FIRST: the reconstruction starts as 0 + y (rqae/model.py:201,224).  LAST: the
    // reconstruction is final -- store it from inside the sweep.
    auto pass_out = [&](auto first_t, auto last_t) {
        constexpr bool FIRST = decltype(first_t)::value, LAST = decltype(last_t)::value;
        constexpr int W = C + 1;
        f2 cv2[NG][2][C];     // wave-uniform straight-through values as SGPR pairs, not 40 VGPRs
        float a_cv[NG][C];    // MFMA A operands: lane i (0..3) holds the value of token 4g+i
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if constexpr (MFMA) {
#pragma unroll
                for (int k = 0; k < C; ++k) a_cv[g][k] = cvs[k * TG + half * TPW + 4 * g + (lane & 3)];
            } else {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const f2 v = *reinterpret_cast<const f2 *>(cvs + k * TG + half * TPW + 4 * g + 2 * pr);
                        cv2[g][pr][k] = (f2){lane_bcast_first(v.x), lane_bcast_first(v.y)};
                    }
            }
        }
        if constexpr (!MFMA) asm volatile("s_nop 3" ::: "memory");   // v_readfirstlane -> SGPR read by the asm below
        f4 wq[QOUT];          // SSA values: each quad lives from its read (one step ahead) to its last use
#pragma unroll
        for (int q = 0; q < QOUT; ++q)
            if (q <= last_quad_of_step(0, W)) wq[q] = wl[q * 64];
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int q = 0; q < QOUT; ++q)
                if (i + 1 < P && q > last_quad_of_step(i, W) && q <= last_quad_of_step(i + 1, W)) wq[q] = wl[q * 64];
            f2 yy[NG][2];
            if constexpr (MFMA) {
                float w[W];
#pragma unroll
                for (int k = 0; k < W; ++k) w[k] = f4_get(wq[(i * W + k) / 4], (i * W + k) % 4);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    f4 y4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < C; ++k) y4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cv[g][k], w[k], y4, 4, 0, 0);
                    y4 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, w[C], y4, 4, 0, 0);   // + bias, one rounding
                    yy[g][0] = (f2){y4.x, y4.y};
                    yy[g][1] = (f2){y4.z, y4.w};
                }
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr)
                            pk_fma_bcast_s(yy[g][pr], cv2[g][pr][k], wq[(i * W + k) / 4], (i * W + k) % 4, k == 0);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) pk_add_bcast(yy[g][pr], wq[(i * W + C) / 4], (i * W + C) % 4);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f2 &rr = r2[i][2 * g + pr];
                    const f2 y = yy[g][pr];
                    rr = rr - y;
                    if constexpr (RECON) {
                        if constexpr (FIRST) out2[i][2 * g + pr] = splat2(0.0f) + y;
                        else if constexpr (MFMA)   // y is an MFMA result: stay behind `rr - y`, whose hazard nops hipcc places
                            asm("v_pk_add_f32 %0, %0, %2" : "+v"(out2[i][2 * g + pr]), "+v"(rr) : "v"(y));
                        else  // one opaque instruction: written as C++ the allocator needs ~60 more VGPRs and spills
                            asm("v_pk_add_f32 %0, %0, %1" : "+v"(out2[i][2 * g + pr]) : "v"(y));
                        if constexpr (LAST) {
                            if (full_tile) {
                                const int t0 = 4 * g + 2 * pr;
                                float *o0 = p.q_out + (size_t)(tokw + t0) * p.d;
                                store_run_ending_at<P>(o0, slice, lane, i, [&](int k) { return out2[k][2 * g + pr].x; });
                                store_run_ending_at<P>(o0 + p.d, slice, lane, i, [&](int k) { return out2[k][2 * g + pr].y; });
                            }
                        }
                    }
                }
            }
        }
    };

    // ---- sweep 2 of a pass: project the residual onto the next level (lin_in partials over this wave's
    // quarter of d -> 64-lane tree -> LDS), again with one LDS read of each weight quad for all 8 tokens.
    auto pass_in = [&]() {
        constexpr int W = C;
        const f4 *wi = wl + QOUT * 64;
        f2 acc2[NG][2][C];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int j = 0; j < C; ++j) acc2[g][pr][j] = splat2(0.0f);
        f4 wq[QIN];
#pragma unroll
        for (int q = 0; q < QIN; ++q)
            if (q <= last_quad_of_step(0, W)) wq[q] = wi[q * 64];
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int q = 0; q < QIN; ++q)
                if (i + 1 < P && q > last_quad_of_step(i, W) && q <= last_quad_of_step(i + 1, W)) wq[q] = wi[q * 64];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        pk_fma_bcast(acc2[g][pr][j], r2[i][2 * g + pr], wq[(i * W + j) / 4], (i * W + j) % 4);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float vals[4 * C];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    vals[(2 * pr) * C + j] = acc2[g][pr][j].x;
                    vals[(2 * pr + 1) * C + j] = acc2[g][pr][j].y;
                }
            float fin[1];
            ReduceScatter<4 * C, 0, 1>::run(vals, fin, lane);
            const int v0 = rs_value_of_lane(lane);
            if (v0 < 4 * C) part[((half * TPW + 4 * g) * C + v0) * NSLICE + slice] = fin[0];
        }
    };
    using T = std::true_type;
    using F = std::false_type;

    // The quantise phase of level l (after the barrier that publishes `part`).
    auto quantise = [&](int l, float bin_own) {
        float q_own = 0.0f;
        if (act) {
            const f4 ps = *reinterpret_cast<const f4 *>(part + (tl_q * C + jj) * NSLICE);
            const float s01 = ps.x + ps.y;
            const float s23 = ps.z + ps.w;
            q_own = (s01 + s23) + bin_own;
        }
        float qk[C];
#pragma unroll
        for (int k = 0; k < C; ++k) qk[k] = __shfl(q_own, (lane / JW) * JW + k, 64);
        const float nrm = aten_norm2<C>(qk, C);
        const float xn_own = q_own / nrm;
        float best[TQ] = {-INFINITY, -INFINITY};
        int bidx[TQ] = {0, 0};
        f2 xs2[C];   // VALU path: wave-uniform unit vectors of the wave's two tokens
        float xa[C]; // MFMA path: A operands, lane 0 / lane 1 = token 0 / token 1
        if constexpr (MFMA) {
            const float xz = act ? xn_own : 0.0f;
#pragma unroll
            for (int k = 0; k < C; ++k) xa[k] = __shfl(xz, (lane & 3) * JW + k, 64);
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) xs2[k] = (f2){lane_bcast(xn_own, k), lane_bcast(xn_own, JW + k)};
        }
        TL_STAMP();   // unit vectors ready
        for (int ch = 0; ch < nchunk; ++ch) {
            const int row_base = ch * p.kch;
            const int rows = (p.K - row_base) < p.kch ? (p.K - row_base) : p.kch;
            if (ch > 0) {
                __syncthreads();
                stage_rows<C>(cbl, p.cb + ((size_t)(shared_cb ? 0 : l) * p.K + row_base) * C, rows);
                __syncthreads();
            }
            auto scan_block = [&](int r0, bool in) __attribute__((always_inline)) {
                const int row = r0 + lane;
                float cbv[C];
                load_row<C>(cbv, cbl, in ? row : 0, rows);
                float s0, s1;
                if constexpr (MFMA) {
                    f4 s4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < C; ++k) s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[k], cbv[k], s4, 4, 0, 0);
                    s0 = s4.x; s1 = s4.y;
                } else {
                    f2 s = splat2(0.0f);
#pragma unroll
                    for (int k = 0; k < C; ++k) s = fma2(xs2[k], splat2(cbv[k]), s);
                    s0 = s.x; s1 = s.y;
                }
                if (in && s0 > best[0]) { best[0] = s0; bidx[0] = row_base + row; }
                if (in && s1 > best[1]) { best[1] = s1; bidx[1] = row_base + row; }
            };
            // four 64-row blocks per trip, written out (hipcc will not unroll this loop on request)
            const int full4 = rows & ~255, full = rows & ~63;
            for (int r0 = 0; r0 < full4; r0 += 256) {
                dma_trip();   // a few pieces of the next record per 256 codebook rows
                if constexpr (MFMA) {
                    scan_block(r0, true);
                    scan_block(r0 + 64, true);
                    scan_block(r0 + 128, true);
                    scan_block(r0 + 192, true);
                } else {
                    // Written as three phases with scheduling fences: left to itself hipcc (at the register limit)
                    // runs the four blocks one after the other, each waiting out its own LDS read and its own
                    // five-deep fma chain -- ~210 cycles per block, the largest single item of the quantise phase.
                    float cb4[4][C];
#pragma unroll
                    for (int b = 0; b < 4; ++b) load_row<C>(cb4[b], cbl, r0 + 64 * b + lane, rows);
                    __builtin_amdgcn_sched_barrier(0);
                    f2 s4[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) s4[b] = splat2(0.0f);
#pragma unroll
                    for (int k = 0; k < C; ++k)
#pragma unroll
                        for (int b = 0; b < 4; ++b) s4[b] = fma2(xs2[k], splat2(cb4[b][k]), s4[b]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int row = row_base + r0 + 64 * b + lane;
                        if (s4[b].x > best[0]) { best[0] = s4[b].x; bidx[0] = row; }
                        if (s4[b].y > best[1]) { best[1] = s4[b].y; bidx[1] = row; }
                    }
                }
            }
            for (int r0 = full4; r0 < full; r0 += 64) scan_block(r0, true);
            if (full < rows) scan_block(full, full + lane < rows);
        }
        dma_flush();   // small codebooks: fewer scan blocks than pieces
        TL_STAMP();   // scan done
        // wave-wide first maximum per token: max by reduce-scatter, then the lowest row among its holders
        float mx[1];
        ReduceScatter<TQ, 0, 1, 1>::run(best, mx, lane);
        int win[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const float m = lane_bcast(mx[0], t << 5);  // lanes with bit5 == t hold token t's maximum
            const unsigned long long tie = __ballot(best[t] == m);
            if (__popcll(tie) != 1) {  // duplicate codebook rows (or no ordered maximum: NaN): lowest index
                int cand = best[t] == m ? bidx[t] : 0x7fffffff;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const int o = __shfl_xor(cand, off, 64);
                    cand = o < cand ? o : cand;
                }
                win[t] = __builtin_amdgcn_readfirstlane(cand);
                if (tie == 0ull) win[t] = 0;       // every similarity is NaN: argmax returns 0
            } else {
                win[t] = __builtin_amdgcn_readlane(bidx[t], __builtin_ctzll(tie));
            }
        }
        TL_STAMP();   // wave-wide winners known
        if (act) {
            const int mine = tq == 1 ? win[1] : win[0];
            // two separate loads: a pointer select here becomes one flat_load, whose wait is vmcnt(0) -- i.e.
            // for the whole record in flight behind this phase
            // (the LDS read in asm: hipcc otherwise still folds both sides into one flat_load)
            float rowv;
            if (nchunk == 1) {
                const int rs = p.kch < p.K ? p.kch : p.K;
                const float *src = (C <= 4 || jj < 4) ? cbl + mine * 4 + jj : cbl + (size_t)rs * 4 + mine * (C - 4) + (jj - 4);
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rowv) : "v"(lds_addr_of(src)) : "memory");
            } else {
                rowv = p.cb[((size_t)(shared_cb ? 0 : l) * p.K + mine) * C + jj];
            }
            const float diff = rowv - q_own;                  // straight-through value (model.py:218-220)
            cvs[jj * TG + tl_q] = q_own + diff;
            const long long tok = tok0 + tl_q;
            if (jj == 0 && tok < p.n_tokens) p.idx[tok * p.levels + l] = mine;
        }
    };

    // lin_in bias of level l for this lane's (token, j).  It is loaded before the pass and pinned
    // (land_bias) before the LDS-DMA is issued: a vmcnt wait on it placed later would also wait for
    // the whole DMA and serialise it with the quantise phase.
    auto bias_of = [&](int l) { return act ? p.b_in[l * C + jj] : 0.0f; };
    auto land_bias = [&](float &b) { asm volatile("" : "+v"(b)); };
    auto slot_end = [&]() { dma_wait_all(); __syncthreads(); };
    auto restage = [&](int l) {   // per-level codebooks: level l's rows replace level l-1's
        if (!stage_once) {
            const int rows = p.kch < p.K ? p.kch : p.K;
            stage_rows<C>(cbl, p.cb + (size_t)(shared_cb ? 0 : l) * p.K * C, rows);
        }
    };
    // one level slot: [barrier: partials published, wts free] quantise(l) with record l+1 streaming in
    auto slot = [&](int l, float bin_own) __attribute__((always_inline)) {
        land_bias(bin_own);
        TL_STAMP();
        __syncthreads();
        TL_STAMP();
        if (!(p.ablate & 1)) dma_begin(l + 1);
        if (!(p.ablate & 2)) quantise(l, bin_own);
        else dma_flush();
        TL_STAMP();
        slot_end();                                       // cvs published, record l+1 landed
        TL_STAMP();
    };

    // ---- level 0 projection
    float bin_own = bias_of(0);
    slot_end();                                           // record 0 + codebook landed
    TL_STAMP();
    pass_in();
    slot(0, bin_own);

    if (p.levels > 1) {                                   // level 0 applied, level 1 projected
        bin_own = bias_of(1);
        pass_out(T{}, F{});
        pass_in();
        restage(1);
        slot(1, bin_own);
    }
    for (int l = 1; l + 1 < p.levels; ++l) {
        bin_own = bias_of(l + 1);
        if (!(p.ablate & 4)) {                            // apply level l, project onto level l+1
            pass_out(F{}, F{});
            pass_in();
        }
        restage(l + 1);
        slot(l + 1, bin_own);
    }
    if (p.levels > 1) pass_out(F{}, T{});                 // apply the last level
    else pass_out(T{}, T{});
    TL_STAMP();
#ifdef RQAE_TIMELINE
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");
    TL_STAMP();
    if (tl_on) g_timeline[blockIdx.x / 4096][wave][lane] = tl_buf[wave * 64 + lane];
#endif

    if constexpr (RECON) {
        if (!full_tile) {   // ragged tile: predicated stores after the loop (full tiles stored inside the last pass)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const long long tok = tokw + t;
                float *op = p.q_out + (size_t)(tok < p.n_tokens ? tok : 0) * p.d;
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const float v = (t & 1) ? out2[i][t >> 1].y : out2[i][t >> 1].x;
                    const int e = elem_of(P, slice, i, lane);
                    if (tok < p.n_tokens && e < p.d) op[e] = v;
                }
            }
        }
    }
}

template <int P, int C>
int launch_encode_lds(const EncParams &p, hipStream_t stream, bool use_mfma)
{
    constexpr int TG = 16;
    EncParams q = p;
    const size_t fixed = sizeof(float) * ((size_t)NSLICE * rec_quads(P, C) * 256 + NSLICE * TG * C + ((TG * C + 3) & ~3));
    const size_t cb_budget = 160 * 1024 - fixed - 1024;
    int kch = (int)(cb_budget / (sizeof(float) * cb_lds_floats(C, 1)));
    kch = (kch / 64) * 64;
    if (kch < 64) return fail("encode: no LDS left for the codebook");
    if (kch >= p.K) kch = ((p.K + 63) / 64) * 64;
    q.kch = kch;
    {
        const char *ab = getenv("RQAE_HIP_ABLATE");
        q.ablate = ab ? atoi(ab) : 0;
    }
    const int rows = kch < p.K ? kch : p.K;
#ifdef RQAE_TIMELINE
    const size_t cbf_tl = kch >= p.K ? (size_t)p.cb_img_pieces * 256 : (size_t)cb_lds_floats(C, rows);
    q.tl_lds_floats = (int)((fixed / sizeof(float) + cbf_tl + 3) & ~(size_t)3);
    const size_t lds = sizeof(float) * (size_t)q.tl_lds_floats + 8 * 64 * sizeof(long long);
#else
    const size_t cbf = kch >= p.K ? (size_t)p.cb_img_pieces * 256 : (size_t)cb_lds_floats(C, rows);   // whole pieces when DMA-staged
    const size_t lds = fixed + sizeof(float) * cbf;
#endif
    const long long nblk = (p.n_tokens + TG - 1) / TG;
    if (nblk > 0x7fffffffLL) return fail("encode: too many tokens for one launch");
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, stream, q);
        HIP_TRY(hipGetLastError());
        return 0;
    };
#if !defined(RQAE_TU_MAIN) && !defined(RQAE_SINGLE_TU)
    if constexpr (P == 9) {
        if (!use_mfma) return rqae_main_tu_launch(C, &p, stream);   // compiled in the other translation unit
    }
#endif
    if (p.norm_w) {
        if (!p.q_out) return fail("encode_gemma: the reconstruction output is required");
        if (use_mfma) return fail("encode_gemma: no MFMA arm");
#if defined(RQAE_TU_MAIN) || defined(RQAE_SINGLE_TU)
        return go(rqae_encode_lds_kernel<P, C, true, false, true>);
#else
        if constexpr (P != 9) return go(rqae_encode_lds_kernel<P, C, true, false, true>);
#endif
    }
#ifndef RQAE_TU_MAIN
    if (use_mfma) {
        if (p.q_out) return go(rqae_encode_lds_kernel<P, C, true, true>);
        return go(rqae_encode_lds_kernel<P, C, false, true>);
    }
#endif
#if defined(RQAE_TU_MAIN) || defined(RQAE_SINGLE_TU)
    if (p.q_out) return go(rqae_encode_lds_kernel<P, C, true, false>);
    return go(rqae_encode_lds_kernel<P, C, false, false>);
#else
    if constexpr (P != 9) {
        if (p.q_out) return go(rqae_encode_lds_kernel<P, C, true, false>);
        return go(rqae_encode_lds_kernel<P, C, false, false>);
    }
    return fail("encode: unreachable");
#endif
}

#ifdef RQAE_TU_MAIN
}  // namespace
extern "C" __attribute__((visibility("hidden"))) int rqae_main_tu_launch(int c, const void *enc_params, void *stream)
{
    const EncParams &p = *static_cast<const EncParams *>(enc_params);
    if (c == 5) return launch_encode_lds<9, 5>(p, static_cast<hipStream_t>(stream), false);
    if (c == 4) return launch_encode_lds<9, 4>(p, static_cast<hipStream_t>(stream), false);
    return fail("encode: no main-path kernel for c=%d", c);
}
namespace {
#endif

#ifndef RQAE_TU_MAIN
template <int P, int C, int TPW, int NH, bool GEN>
int launch_encode(const EncParams &p, hipStream_t stream)
{
    constexpr int TG = TPW * NH;
    constexpr int NV = TPW * C;
    EncParams q = p;
    const size_t fixed = sizeof(float) * (NH * NSLICE * NV + ((TG * C + 3) & ~3));
    int kch = (int)((LDS_CODEBOOK_BUDGET) / (sizeof(float) * p.c));
    kch = (kch / 64) * 64;
    if (kch >= p.K) kch = ((p.K + 63) / 64) * 64;
    q.kch = kch;
    const int rows = kch < p.K ? kch : p.K;
    const size_t lds = fixed + sizeof(float) * (size_t)rows * p.c;
    const long long nblk = (p.n_tokens + TG - 1) / TG;
    if (nblk > 0x7fffffffLL) return fail("encode: too many tokens for one launch");
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256 * NH), lds, stream, q);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (p.q_out) return go(rqae_encode_kernel<P, C, TPW, NH, true, GEN>);
    return go(rqae_encode_kernel<P, C, TPW, NH, false, GEN>);
}

enum Variant { V_AUTO = 0, V_A = 1, V_B = 2, V_GEN = 3, V_M = 4 };

Variant env_variant()
{
    const char *s = getenv("RQAE_HIP_VARIANT");
    if (!s || !*s) return V_AUTO;
    if (!strcmp(s, "A")) return V_A;
    if (!strcmp(s, "B")) return V_B;
    if (!strcmp(s, "GEN")) return V_GEN;
    if (!strcmp(s, "M")) return V_M;
    return V_AUTO;
}

// Which instantiation serves (P, c).  Returns a name; launches when p != nullptr.
const char *dispatch_encode(int P, int c, const EncParams *p, hipStream_t stream, int *rc)
{
    const Variant v = env_variant();
#ifdef RQAE_EXPERIMENT_ONLY_95   // compile-time experiments: one instantiation, fast rebuilds
    {
        if (p) *rc = launch_encode_lds<9, 5>(*p, stream, v == V_M);
        (void)v; (void)P; (void)c;
        return "experiment";
    }
#else
#define RUN(name, ...)                                  \
    do {                                                \
        if (p) *rc = launch_encode<__VA_ARGS__>(*p, stream); \
        return name;                                    \
    } while (0)
#define RUNF(name, ...)                                       \
    do {                                                      \
        if (p) *rc = launch_encode_lds<__VA_ARGS__>(*p, stream, v == V_M); \
        return name;                                          \
    } while (0)
    // M = the arm that puts lin_out and the cosine contraction on the matrix pipe (v_mfma_f32_4x4x1_16b).
    // Bit-identical output; measured 10-17 % slower than the packed-VALU arm on MI355X.
    if (v == V_AUTO || v == V_M) {
        const bool m = v == V_M;
        if (P == 9 && c == 5) RUNF(m ? "encode_lds<P=9,C=5> MFMA 4x4x1 arm" : "encode_lds<P=9,C=5> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 9, 5);
        if (P == 9 && c == 4) RUNF(m ? "encode_lds<P=9,C=4> MFMA 4x4x1 arm" : "encode_lds<P=9,C=4> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 9, 4);
        if (P == 1 && c == 4) RUNF(m ? "encode_lds<P=1,C=4> MFMA 4x4x1 arm" : "encode_lds<P=1,C=4> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 1, 4);
    }
#undef RUNF
    if (v != V_GEN) {
        if (P == 9 && c == 5) {
            if (v == V_B) RUN("encode<P=9,C=5,TPW=8,NH=2> 8 waves, 16 tokens/WG", 9, 5, 8, 2, false);
            RUN("encode<P=9,C=5,TPW=16,NH=1> 4 waves, 16 tokens/WG", 9, 5, 16, 1, false);
        }
        if (P == 9 && c == 4) RUN("encode<P=9,C=4,TPW=16,NH=1> 4 waves, 16 tokens/WG", 9, 4, 16, 1, false);
        if (P == 1 && c == 4) RUN("encode<P=1,C=4,TPW=16,NH=1> 4 waves, 16 tokens/WG", 1, 4, 16, 1, false);
        if (P == 14 && c == 13) RUN("encode<P=14,C=13,TPW=8,NH=1> 4 waves, 8 tokens/WG", 14, 13, 8, 1, false);
    }
    if (P <= 8) RUN("encode<generic P<=8,C<=16,TPW=2>", 8, 16, 2, 1, true);
    RUN("encode<generic P<=32,C<=16,TPW=2>", 32, 16, 2, 1, true);
#undef RUN
#endif
}

// ---------------------------------------------------------------- quantize kernel
struct QuantParams {
    const float *p, *cb;
    float *cv;
    long long *idx;
    long long n;
    int c, K, kch, iters;
};

// One workgroup = 4 waves; a wave quantises TQ tokens at a time.  When the codebook fits in LDS in one chunk it is
// staged once per workgroup and the waves walk `iters` token groups against it (staging 20 KB per 16 tokens was
// 25x the algorithmic bytes); a chunked codebook keeps one group per workgroup (the running maxima of a group
// live in registers across the chunks).
template <int C>
__global__ void __launch_bounds__(256)
rqae_quantize_kernel(const QuantParams p)
{
    constexpr int TQ = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = p.c;
    const int nchunk = (p.K + p.kch - 1) / p.kch;
    if (nchunk == 1) {
        stage_floats(smem, p.cb, p.K * cp);
        __syncthreads();
    }
    for (int it = 0; it < p.iters; ++it) {
        const long long tokw = (((long long)blockIdx.x * p.iters + it) * 4 + wave) * TQ;
        if (nchunk == 1 && tokw >= p.n) break;                // wave-uniform; (a chunked codebook has barriers below: every wave stays)
        float q[TQ][C], xn[TQ][C], best[TQ];
        int bidx[TQ];
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
            const long long tok = tokw + tq;
#pragma unroll
            for (int j = 0; j < C; ++j) q[tq][j] = (j < cp && tok < p.n) ? p.p[tok * cp + j] : 0.0f;
            const float nrm = aten_norm2<C>(q[tq], cp);
#pragma unroll
            for (int j = 0; j < C; ++j) xn[tq][j] = q[tq][j] / nrm;
            best[tq] = -INFINITY;
            bidx[tq] = 0;
        }
        if (nchunk == 1) {
            scan_rows<C, TQ>(xn, smem, p.K, 0, cp, lane, best, bidx);
        } else {
            for (int ch = 0; ch < nchunk; ++ch) {             // iters == 1 here: every wave of the block takes part
                const int row_base = ch * p.kch;
                const int rows = (p.K - row_base) < p.kch ? (p.K - row_base) : p.kch;
                if (ch > 0) __syncthreads();
                stage_floats(smem, p.cb + (size_t)row_base * cp, rows * cp);
                __syncthreads();
                scan_rows<C, TQ>(xn, smem, rows, row_base, cp, lane, best, bidx);
            }
        }
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
            wave_argmax(best[tq], bidx[tq]);
            const long long tok = tokw + tq;
            if (tok < p.n) {
                if (lane == 0) p.idx[tok] = bidx[tq];
                if (lane < cp) p.cv[tok * cp + lane] = p.cb[(size_t)bidx[tq] * cp + lane];
            }
        }
    }
}

// ------------------------------------------------------------------ decode kernel
struct DecParams {
    const float *w_out, *b_out, *cb0;
    const void *src;  // indices [n][nq] or cv [n][nq][c]
    float *out;
    long long n_tokens;
    int d, dp, c, K, nlev, nq, idx_dtype;
    const float *rec_p;   // pass records (rqae_decode_lds_kernel reads the lin_out quads of record l+1)
    int aligned16;        // out 16-byte aligned: full tiles use vector stores
    uint32_t mask[MAX_MASK_LEVELS / 32];
};

__device__ __forceinline__ long long load_idx(const void *src, int dt, size_t off)
{
    if (dt == RQAE_IDX_I64) return reinterpret_cast<const long long *>(src)[off];
    if (dt == RQAE_IDX_I32) return reinterpret_cast<const int *>(src)[off];
    return reinterpret_cast<const uint16_t *>(src)[off];
}

// One workgroup = 4 waves (one per slice) x TPW tokens; the running sum stays in VGPRs,
// lin_out weights stream once per workgroup.  LCH levels of codebook values are staged in LDS.
template <int P, int C, int TPW, bool FROM_IDX, bool GEN>
__global__ void __launch_bounds__(256)
rqae_decode_kernel(const DecParams p)
{
    constexpr int LCH = 16;
    __shared__ float cvl[TPW * LCH * C];
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = GEN ? p.c : C;
    const int Pp = GEN ? p.dp / 256 : P;
    const int dp = GEN ? p.dp : 256 * P;
    auto el = [&](int i) { return elem_of(Pp, slice, i, lane); };
    const long long tok0 = (long long)blockIdx.x * TPW;

    float acc[P][TPW];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[i][t] = 0.0f;
    bool first = true;

    for (int l0 = 0; l0 < p.nlev; l0 += LCH) {
        const int lch = (p.nlev - l0) < LCH ? (p.nlev - l0) : LCH;
        __syncthreads();
        for (int u = threadIdx.x; u < TPW * lch; u += blockDim.x) {
            const int t = u / lch, ll = u % lch, l = l0 + ll;
            const long long tok = tok0 + t;
            const bool on = (p.mask[l >> 5] >> (l & 31)) & 1u;
            if (!on || tok >= p.n_tokens) continue;
            const float *row;
            if constexpr (FROM_IDX) {
                const long long ix = load_idx(p.src, p.idx_dtype, (size_t)tok * p.nq + l);
                row = p.cb0 + (size_t)ix * cp;  // level-0 codebook for every level (model.py:234)
            } else {
                row = reinterpret_cast<const float *>(p.src) + ((size_t)tok * p.nq + l) * cp;
            }
            for (int k = 0; k < cp; ++k) cvl[(t * LCH + ll) * C + k] = row[k];
        }
        __syncthreads();
        for (int ll = 0; ll < lch; ++ll) {
            const int l = l0 + ll;
            if (!((p.mask[l >> 5] >> (l & 31)) & 1u)) continue;
            const float *wl = p.w_out + (size_t)l * cp * dp;
            const float *bl = p.b_out + (size_t)l * dp;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (GEN && i >= Pp) continue;
                float w[C];
#pragma unroll
                for (int k = 0; k < C; ++k) w[k] = (!GEN || k < cp) ? wl[(size_t)k * dp + el(i)] : 0.0f;
                const float b = bl[el(i)];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    float y = 0.0f;
#pragma unroll
                    for (int k = 0; k < C; ++k)
                        if (!GEN || k < cp) y = fmaf(cvl[(t * LCH + ll) * C + k], w[k], y);
                    y = y + b;
                    acc[i][t] = first ? y : acc[i][t] + y;  // first level assigns (model.py:244-247)
                }
            }
            first = false;
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int e = el(i);
            const long long tok = tok0 + t;
            if ((!GEN || i < Pp) && e < p.d && tok < p.n_tokens) p.out[(size_t)tok * p.d + e] = acc[i][t];
        }
}

template <int P, int C, int TPW, bool GEN>
int launch_decode(const DecParams &p, bool from_idx, hipStream_t stream)
{
    const long long nblk = (p.n_tokens + TPW - 1) / TPW;
    if (nblk > 0x7fffffffLL) return fail("decode: too many tokens for one launch");
    if (from_idx)
        hipLaunchKernelGGL((rqae_decode_kernel<P, C, TPW, true, GEN>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((rqae_decode_kernel<P, C, TPW, false, GEN>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}


// =============================================================== LDS-staged decode kernel
// Decode has no dependency between levels and no cross-lane reduction: out = sum_l (lin_out_l(cv_l)),
// ascending, first selected level assigns (rqae/model.py:236-248).  Same geometry as the encode
// kernel (8 waves = 4 slices x 2 token halves, 8 tokens per wave, packed fp32 over token pairs); the
// lin_out weights of a level are the first two quads of every element step of pass record l+1
// (w_out[0..c-1], b_out), copied L2 -> LDS by LDS-DMA into one of TWO buffers, so level l+1 loads while
// level l computes.  Codebook values of the tile (16 tokens x 32 levels x c) are gathered into LDS.
// NBUF = 2: the next level's weights load into a second LDS buffer behind the current level (one workgroup per CU).
// NBUF = 1: one buffer, 66 KB of LDS and <= 128 VGPRs, so TWO workgroups share a CU and cover each other's load and
//           barrier waits (the kernel is VALU-bound once those are hidden).
template <int P, int C, bool FROM_IDX, int NBUF>
__global__ void __launch_bounds__(512, NBUF == 1 ? 4 : 2)
rqae_decode_lds_kernel(const DecParams p)
{
    constexpr int TPW = 8, NW = 8, TG = 16, NPAIR = 4, NG = 2, LCH = 32;
    constexpr int NQ = rec_quads(P, C);          // quads per lane per slice in a pass record
    constexpr int QD = rec_out_quads(P, C);      // its leading quads: the packed {w_out[0..C-1], b_out} stream
    constexpr int W = C + 1;
    constexpr int BUF = NSLICE * QD * 256;       // floats per weight buffer
    constexpr int dp = 256 * P;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf = smem;                          // [NBUF][slice][QD][lane][4]
    float *cvl = wbuf + NBUF * BUF;                 // [LCH][C][TG]  (token pairs contiguous)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = wave & 3, half = wave >> 2;
    const long long tok0 = (long long)blockIdx.x * TG;
    const long long tokw = tok0 + half * TPW;
    const bool full_tile = (tokw + TPW <= p.n_tokens) && (p.d == dp) && p.aligned16;
    const unsigned wbuf_addr = lds_addr_of(wbuf);

    auto level_on = [&](int l) { return ((p.mask[l >> 5] >> (l & 31)) & 1u) != 0; };
    auto next_on = [&](int l) { while (l < p.nlev && !level_on(l)) ++l; return l; };   // uniform scalar loop
    // lin_out quads of level l (record l+1) -> buffer b; the 8 waves share the (slice, i) chunks
    auto issue_dma = [&](int l, int b) {
        const float *rec = p.rec_p + (size_t)(l + 1) * NSLICE * NQ * 256;
        for (int pc = wave; pc < NSLICE * QD; pc += NW) {
            const int s = pc / QD, q = pc - s * QD;
            dma_pieces<1>(rec + (size_t)(s * NQ + q) * 256, wbuf_addr + (b * BUF + pc * 256) * 4, lane * 16);
        }
    };

    f2 acc[P][NPAIR];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) acc[i][pr] = splat2(0.0f);
    bool first = true;

    int l = next_on(0), buf = 0;
    if (l < p.nlev) issue_dma(l, 0);
    int staged_from = -1;                         // first level of the codebook-value chunk in LDS
    bool first_level = true;
    while (l < p.nlev) {
        const int ln = next_on(l + 1);
        if constexpr (NBUF == 1) {
            if (!first_level) {
                __syncthreads();                              // every wave is done with the previous level's weights
                issue_dma(l, 0);
            }
            first_level = false;
        }
        if (l >= staged_from + LCH || staged_from < 0) {
            // gather this tile's codebook values for levels [l, l+LCH)
            if (staged_from >= 0) __syncthreads();            // previous chunk no longer read
            staged_from = l;
            for (int u = threadIdx.x; u < TG * LCH; u += blockDim.x) {
                const int t = u % TG, ll = u / TG, lv = l + ll;
                const long long tok = tok0 + t;
                if (lv >= p.nlev || !level_on(lv) || tok >= p.n_tokens) continue;
                const float *row;
                if constexpr (FROM_IDX) {
                    const long long ix = load_idx(p.src, p.idx_dtype, (size_t)tok * p.nq + lv);
                    row = p.cb0 + (size_t)ix * C;             // level-0 codebook for every level (model.py:234)
                } else {
                    row = reinterpret_cast<const float *>(p.src) + ((size_t)tok * p.nq + lv) * C;
                }
#pragma unroll
                for (int k = 0; k < C; ++k) cvl[(ll * C + k) * TG + t] = row[k];
            }
        }
        dma_wait_all();
        __syncthreads();                                      // weights of level l + codebook values visible
        // next selected level loads behind this one: one piece per element step (a burst of all pieces waits
        // on the TA's address queue, see the encode kernel)
        int dpc = (NBUF == 2 && ln < p.nlev) ? wave : NSLICE * QD;
        const float *drec = p.rec_p + (size_t)(ln + 1) * NSLICE * NQ * 256;
        auto dma_step = [&]() __attribute__((always_inline)) {
            if (dpc < NSLICE * QD) {
                const int s = dpc / QD, q = dpc - s * QD;
                dma_pieces<1>(drec + (size_t)(s * NQ + q) * 256, wbuf_addr + ((buf ^ 1) * BUF + dpc * 256) * 4, lane * 16);
                dpc += NW;
            }
        };

        const f4 *wl = reinterpret_cast<const f4 *>(wbuf + buf * BUF) + (size_t)slice * QD * 64 + lane;
        const float *cvp = cvl + (l - staged_from) * C * TG + half * TPW;
        // element step outermost: every weight quad is read from LDS once for the wave's 8 tokens
        f2 cv2[NG][2][C];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const f2 v = *reinterpret_cast<const f2 *>(cvp + k * TG + 4 * g + 2 * pr);
                    cv2[g][pr][k] = (f2){lane_bcast_first(v.x), lane_bcast_first(v.y)};
                }
        asm volatile("s_nop 3" ::: "memory");   // v_readfirstlane -> SGPR read by the asm below
        f4 wq[QD];
#pragma unroll
        for (int q = 0; q < QD; ++q)
            if (q <= last_quad_of_step(0, W)) wq[q] = wl[q * 64];
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int q = 0; q < QD; ++q)
                if (i + 1 < P && q > last_quad_of_step(i, W) && q <= last_quad_of_step(i + 1, W)) wq[q] = wl[q * 64];
            dma_step();
            f2 yy[NG][2];   // the four chains interleaved, as in the encode kernel's sweep 1
#pragma unroll
            for (int k = 0; k < C; ++k)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
                        pk_fma_bcast_s(yy[g][pr], cv2[g][pr][k], wq[(i * W + k) / 4], (i * W + k) % 4, k == 0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    pk_add_bcast(yy[g][pr], wq[(i * W + C) / 4], (i * W + C) % 4);
                    if (first) acc[i][2 * g + pr] = yy[g][pr];     // first selected level assigns (model.py:244-247)
                    else acc[i][2 * g + pr] = acc[i][2 * g + pr] + yy[g][pr];
                }
        }
        while (dpc < NSLICE * QD) dma_step();                 // more pieces per wave than element steps
        first = false;
        l = ln;
        if constexpr (NBUF == 2) buf ^= 1;
    }

#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const long long tok = tokw + t;
        float *op = p.out + (size_t)(tok < p.n_tokens ? tok : 0) * p.d;
        auto val = [&](int k) { return (t & 1) ? acc[k][t >> 1].y : acc[k][t >> 1].x; };
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if (full_tile) store_run_ending_at<P>(op, slice, lane, i, val);
            else {
                const int e = elem_of(P, slice, i, lane);
                if (tok < p.n_tokens && e < p.d) op[e] = val(i);
            }
        }
    }
}

template <int P, int C>
int launch_decode_lds(const DecParams &p, bool from_idx, hipStream_t stream)
{
    constexpr int QD = rec_out_quads(P, C);
    const char *nb = getenv("RQAE_HIP_DECODE_NBUF");     // 2 = the double-buffered form (measured slower: 1.42 vs 1.16 ms)
    const int nbuf = nb && atoi(nb) == 2 ? 2 : 1;
    const size_t lds = sizeof(float) * ((size_t)nbuf * NSLICE * QD * 256 + 32 * C * 16);
    const long long nblk = (p.n_tokens + 15) / 16;
    if (nblk > 0x7fffffffLL) return fail("decode: too many tokens for one launch");
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, stream, p);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (nbuf == 2) {
        if (from_idx) return go(rqae_decode_lds_kernel<P, C, true, 2>);
        return go(rqae_decode_lds_kernel<P, C, false, 2>);
    }
    if (from_idx) return go(rqae_decode_lds_kernel<P, C, true, 1>);
    return go(rqae_decode_lds_kernel<P, C, false, 1>);
}

int dispatch_decode(int P, int c, const DecParams &p, bool from_idx, hipStream_t stream)
{
#ifdef RQAE_EXPERIMENT_ONLY_95
    return fail("experiment build");
#else
    if (env_variant() == V_AUTO) {
        if (P == 9 && c == 5) return launch_decode_lds<9, 5>(p, from_idx, stream);
        if (P == 9 && c == 4) return launch_decode_lds<9, 4>(p, from_idx, stream);
        if (P == 1 && c == 4) return launch_decode_lds<1, 4>(p, from_idx, stream);
    }
    if (env_variant() != V_GEN) {
        if (P == 9 && c == 5) return launch_decode<9, 5, 16, false>(p, from_idx, stream);
        if (P == 9 && c == 4) return launch_decode<9, 4, 16, false>(p, from_idx, stream);
        if (P == 1 && c == 4) return launch_decode<1, 4, 16, false>(p, from_idx, stream);
        if (P == 14 && c == 13) return launch_decode<14, 13, 4, false>(p, from_idx, stream);
    }
    if (P <= 8) return launch_decode<8, 16, 4, true>(p, from_idx, stream);
    return launch_decode<32, 16, 4, true>(p, from_idx, stream);
#endif
}


// =============================================================== hook pre/post
// The step either side of forward() in RQAE.hook (rqae/model.py:276-289) for a Gemma-2 LLM
// (rqae/llm.py:65-73):  normed = (hs * rsqrt(mean(hs^2) + eps)) * (1 + w)   and afterwards
// new = (q / (1 + w)) / rsqrt(mean(hs^2) + eps), new[:, 0] = hs[:, 0], written back in the LLM's dtype.
// One wave per token.  Order of the mean: lane owns elements 4*(lane + 64*j) + {0..3}, j ascending,
// fma chain of squares from +0 per lane, lanes summed by the tree of section 3, divided by d;
// rs = 1 / sqrt(ms + eps) with IEEE sqrt and divide (oracle/oracle.py:gemma_norm_canonical restates it).
// Sum of the four slice totals of every lane's value (in: per-lane partial of slice s in ss[s]) in the order of
// section 3: the 64-lane tree per slice, then (s0+s1)+(s2+s3).  Every lane returns the row total.
__device__ __forceinline__ float canonical_row_total(const float (&ss)[NSLICE], int lane)
{
    float fin[1];
    ReduceScatter<NSLICE, 0, 1>::run(ss, fin, lane);      // lane holds the tree sum of slice (bit5 + 2*bit4)
    const float s0 = lane_bcast(fin[0], 0), s1 = lane_bcast(fin[0], 32);
    const float s2 = lane_bcast(fin[0], 16), s3 = lane_bcast(fin[0], 48);
    return (s0 + s1) + (s2 + s3);
}

// One wave per token, any d (runtime P = ceil(d / 256)), scalar loads.
template <typename HT>
__global__ void __launch_bounds__(256)
gemma_norm_kernel(const HT *hs, const float *w, float eps, long long n_tokens, int d, float *normed, float *rs_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const int P = (d + 255) / 256;
    const HT *row = hs + (size_t)tok * d;
    float ss[NSLICE];
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        float a = 0.0f;
        for (int i = 0; i < P; ++i) {
            const int e = elem_of(P, s, i, lane);
            const float x = e < d ? widen<HT>(row[e]) : 0.0f;
            a = fmaf(x, x, a);
        }
        ss[s] = a;
    }
    const float total = canonical_row_total(ss, lane);
    const float ms = total / (float)d;
    const float rs = 1.0f / __builtin_sqrtf(ms + eps);
    if (lane == 0) rs_out[tok] = rs;
    float *orow = normed + (size_t)tok * d;
    for (int e = lane; e < d; e += 64) {
        const float t = widen<HT>(row[e]) * rs;
        orow[e] = t * (1.0f + w[e]);
    }
}

// Vector form for d == 256 * P with 16-byte aligned rows: hs is read ONCE with the tile loads of the encode
// kernel (runs of 4 / 2 / 1 consecutive elements per lane), kept in registers across the reduction, and normed
// is written with the matching vector stores.  Same arithmetic and order as gemma_norm_kernel.
template <typename HT, int P>
__global__ void __launch_bounds__(256)
gemma_norm_vec_kernel(const HT *hs, const float *w, float eps, long long n_tokens, int d, float *normed, float *rs_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    float x[NSLICE][P][1];
    float ss[NSLICE];
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        load_tile<P, 1, HT, true>(x[s], hs, tok, n_tokens, d, s, lane);
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < P; ++i) a = fmaf(x[s][i][0], x[s][i][0], a);
        ss[s] = a;
    }
    const float total = canonical_row_total(ss, lane);
    const float ms = total / (float)d;
    const float rs = 1.0f / __builtin_sqrtf(ms + eps);
    if (lane == 0) rs_out[tok] = rs;
    float *orow = normed + (size_t)tok * d;
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        float w1[P][1];
        load_tile<P, 1, float, true>(w1, w, 0, 1, d, s, lane);
#pragma unroll
        for (int i = 0; i < P; ++i)
            store_run_ending_at<P>(orow, s, lane, i, [&](int k) { return (x[s][k][0] * rs) * (1.0f + w1[k][0]); });
    }
}

template <typename HT>
__device__ __forceinline__ HT narrow(float v);
template <>
__device__ __forceinline__ _Float16 narrow<_Float16>(float v) { return (_Float16)v; }
template <>
__device__ __forceinline__ float narrow<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint16_t narrow<uint16_t>(float v)   // round-to-nearest-even bf16 (as torch .to(bfloat16))
{
    return __builtin_bit_cast(uint16_t, (__bf16)v);
}

template <typename HT>
__global__ void __launch_bounds__(256)
gemma_denorm_patch_kernel(const float *q, const float *w, const float *rs, long long n_tokens, int d, int seq_len,
                          int skip_bos, HT *hs_inout, float *new_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const bool bos = skip_bos && (tok % seq_len) == 0;
    const float r = rs[tok];
    const float *qrow = q + (size_t)tok * d;
    HT *hrow = hs_inout + (size_t)tok * d;
    float *nrow = new_out ? new_out + (size_t)tok * d : nullptr;
    for (int e = lane; e < d; e += 64) {
        float v;
        if (bos) v = widen<HT>(hrow[e]);                       // q_out[:, 0] = hs[:, 0]  (model.py:285-286)
        else {
            const float t = qrow[e] / (1.0f + w[e]);           // rqae/llm.py:69
            v = t / r;                                          // rqae/llm.py:70-72
        }
        if (nrow) nrow[e] = v;
        if (!bos) hrow[e] = narrow<HT>(v);                      // output[0].data.copy_(q_out)  (model.py:289)
    }
}

// Vector form of the patch kernel (d % 4 == 0, 16-byte aligned rows): float4 in, 4 elements of the LLM's
// dtype out per lane and step.  Same arithmetic as gemma_denorm_patch_kernel.
template <typename HT>
__global__ void __launch_bounds__(256)
gemma_denorm_patch_vec_kernel(const float *q, const float *w, const float *rs, long long n_tokens, int d, int seq_len,
                              int skip_bos, HT *hs_inout, float *new_out)
{
    typedef typename VecOf<HT, 4>::type H4;
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const bool bos = skip_bos && (tok % seq_len) == 0;      // wave-uniform
    const float r = rs[tok];
    const float *qrow = q + (size_t)tok * d;
    HT *hrow = hs_inout + (size_t)tok * d;
    float *nrow = new_out ? new_out + (size_t)tok * d : nullptr;
    if (bos) {                                               // q_out[:, 0] = hs[:, 0]  (model.py:285-286): hs stays
        if (nrow)
            for (int e0 = 4 * lane; e0 < d; e0 += 256) {
                const H4 h = *reinterpret_cast<const H4 *>(hrow + e0);
                *reinterpret_cast<f4 *>(nrow + e0) = (f4){widen<HT>(h.x), widen<HT>(h.y), widen<HT>(h.z), widen<HT>(h.w)};
            }
        return;
    }
    for (int e0 = 4 * lane; e0 < d; e0 += 256) {
        const f4 qv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(qrow + e0));
        const f4 wv = *reinterpret_cast<const f4 *>(w + e0);
        f4 v;
        v.x = (qv.x / (1.0f + wv.x)) / r; v.y = (qv.y / (1.0f + wv.y)) / r;   // rqae/llm.py:69-72
        v.z = (qv.z / (1.0f + wv.z)) / r; v.w = (qv.w / (1.0f + wv.w)) / r;
        if (nrow) *reinterpret_cast<f4 *>(nrow + e0) = v;
        H4 o;
        o.x = narrow<HT>(v.x); o.y = narrow<HT>(v.y); o.z = narrow<HT>(v.z); o.w = narrow<HT>(v.w);
        *reinterpret_cast<H4 *>(hrow + e0) = o;             // output[0].data.copy_(q_out)  (model.py:289)
    }
}

// =============================================================== feature intensity
// RQAEFeature.intensity (rqae/feature.py:102-129): for every token n and every requested level l
//     sims[n][m] = codebook_sims[center[m], code[n][m]] * w[m]            (fp16 product)
//     out[n][l]  = fp16( fp16(sum_{m<=l} sims[n][m]) / wcum[l] )           (fp32 running sum, sequential)
// with torch-CPU's fp16 semantics written out (oracle/oracle.py:intensity restates them; bit-equal to the
// reference's output).  One lane = one token, levels strictly in order (the sum is a sequential fp32 chain, so
// there is no parallel scan).  HBM-bound on the code matrix: a wave reads its 64 tokens x TL levels tile with
// fully coalesced 16-byte loads (one tile ahead, in registers), transposes it through LDS and walks its own
// row; the per-level table rows (codebook_sims[center[m], :], K fp16 each) sit in LDS a chunk of levels at a time.
struct IntensityParams {
    const _Float16 *table;    // [M][K]  = codebook_sims[center[m], :]
    const _Float16 *w, *wcum; // [M]
    const void *idx;
    const int *layers;        // [n_layers] device
    _Float16 *out;            // [n_tokens][n_layers]
    long long n_tokens;
    int nq, M, K, LC, n_layers, vec_ok;
};

__global__ void intensity_prepare_kernel(const _Float16 *sims, const int *center, const _Float16 *w, int M, int K,
                                         _Float16 *table, _Float16 *wcum)
{
    const size_t n = (size_t)M * K;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(u / K), k = (int)(u % K);
        table[u] = sims[(size_t)center[m] * K + k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // layer_weights[:M].cumsum(): fp32 accumulator, fp16 prefixes
        float acc = 0.0f;
        for (int m = 0; m < M; ++m) {
            acc += (float)w[m];
            wcum[m] = (_Float16)acc;
        }
    }
}

template <typename IT>
__global__ void __launch_bounds__(256)
intensity_kernel(const IntensityParams p)
{
    constexpr int TL = 16;                          // levels per index tile
    constexpr int EPC = 16 / sizeof(IT);            // index elements per 16-byte chunk
    constexpr int CPR = TL / EPC;                   // chunks per tile row
    constexpr int RS = CPR + 1;                     // row stride in chunks (odd: conflict-free row reads)
    typedef IT ITV __attribute__((ext_vector_type(EPC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tab_bytes = ((p.LC * p.K * 2 + 15) & ~15);
    _Float16 *tab = reinterpret_cast<_Float16 *>(smem_raw);                       // [LC][K]
    float *wf = reinterpret_cast<float *>(smem_raw + tab_bytes);                   // [32] w, [32] wcum
    unsigned *omask = reinterpret_cast<unsigned *>(wf + 64);                       // output levels of the tile
    ITV *tiles = reinterpret_cast<ITV *>(smem_raw + tab_bytes + 64 * 4 + 16);      // [4 waves][64 rows][RS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tokw = (long long)blockIdx.x * 256 + wave * 64;
    const long long tok = tokw + lane;
    ITV *tile = tiles + (size_t)wave * 64 * RS;
    const IT *idx = reinterpret_cast<const IT *>(p.idx);

    // this lane's share of the wave's tile: chunk c of row r for the CPR load steps
    auto load_tile = [&](int m0, ITV (&regs)[CPR]) {
#pragma unroll
        for (int j = 0; j < CPR; ++j) {
            const int u = j * 64 + lane, r = u / CPR, c = u % CPR;
            long long t = tokw + r;
            if (t >= p.n_tokens) t = p.n_tokens - 1;
            const IT *src = idx + (size_t)t * p.nq + m0 + c * EPC;
            if (p.vec_ok && m0 + TL <= p.nq) {
                regs[j] = *reinterpret_cast<const ITV *>(src);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) regs[j][e] = (m0 + c * EPC + e < p.nq) ? src[e] : (IT)0;
            }
        }
    };
    // table chunk [c0, c0+LC) -> registers (whole 16-byte pieces; at most 12 per thread for a 48-KiB chunk)
    constexpr int TPT = 12;
    auto load_table = [&](int c0, f4 (&tregs)[TPT]) {
        const int nlev = (p.M - c0) < p.LC ? (p.M - c0) : p.LC;
        const int n16 = (nlev * p.K * 2 + 15) / 16;
        const f4 *src = reinterpret_cast<const f4 *>(p.table + (size_t)c0 * p.K);
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int u = j * 256 + threadIdx.x;
            if (u < n16) tregs[j] = src[u];
        }
    };
    ITV regs[CPR];
    f4 tregs[TPT];
    load_tile(0, regs);
    load_table(0, tregs);
    float acc = 0.0f;

    for (int c0 = 0; c0 < p.M; c0 += p.LC) {
        const int m0 = c0 & ~(TL - 1);
        const bool new_tile = c0 == m0;
        __syncthreads();                                  // previous table chunk (and tile) no longer read
        if (new_tile) {
#pragma unroll
            for (int j = 0; j < CPR; ++j) {
                const int u = j * 64 + lane, r = u / CPR, c = u % CPR;
                tile[r * RS + c] = regs[j];
            }
            if (threadIdx.x < 64) {                       // weights of the tile's levels; which of them are outputs
                const int m = m0 + threadIdx.x;
                bool hit = false;
                if (threadIdx.x < TL) {
                    wf[threadIdx.x] = m < p.M ? (float)p.w[m] : 0.0f;
                    wf[32 + threadIdx.x] = m < p.M ? (float)p.wcum[m] : 1.0f;
                    for (int j = 0; j < p.n_layers; ++j) hit |= p.layers[j] == m;
                }
                const unsigned long long b = __ballot(hit);
                if (threadIdx.x == 0) *omask = (unsigned)b;
            }
        }
        {
            const int nlev = (p.M - c0) < p.LC ? (p.M - c0) : p.LC;
            const int n16 = (nlev * p.K * 2 + 15) / 16;
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const int u = j * 256 + threadIdx.x;
                if (u < n16) reinterpret_cast<f4 *>(tab)[u] = tregs[j];
            }
        }
        __syncthreads();
        if (c0 + p.LC < p.M) {                            // next chunk (and tile): in flight behind this one
            load_table(c0 + p.LC, tregs);
            if (((c0 + p.LC) & (TL - 1)) == 0) load_tile(c0 + p.LC, regs);
        }
        const unsigned mask = *omask;
        int hi = c0 + p.LC;
        if (hi > p.M) hi = p.M;
        auto emit = [&](int m, int rel) __attribute__((always_inline)) {   // output level: wave-uniform branch
            const _Float16 pref = (_Float16)acc;                              // cumsum's prefixes are fp16
            const _Float16 o = (_Float16)((float)pref / wf[32 + rel]);
            if (tok < p.n_tokens)
                for (int j = 0; j < p.n_layers; ++j)
                    if (p.layers[j] == m) p.out[(size_t)tok * p.n_layers + j] = o;
        };
        int m = c0;
        // whole 16-byte groups of codes: one row read, EPC gathers in flight together, then the sequential sum
        for (; m + EPC <= hi && ((m - m0) % EPC) == 0; m += EPC) {
            const int rel = m - m0;
            const ITV codes = tile[lane * RS + rel / EPC];
            float sv[EPC], wv[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) sv[e] = (float)tab[(m + e - c0) * p.K + (int)codes[e]];
#pragma unroll
            for (int e = 0; e < EPC; ++e) wv[e] = wf[rel + e];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const _Float16 prod = (_Float16)(sv[e] * wv[e]);             // sims *= layer_weights
                acc += (float)prod;                                            // cumsum: fp32 accumulator
                if ((mask >> (rel + e)) & 1u) emit(m + e, rel + e);
            }
        }
        for (; m < hi; ++m) {
            const int rel = m - m0;
            const IT code = reinterpret_cast<const IT *>(tile + lane * RS)[rel];
            const float sv = (float)tab[(m - c0) * p.K + (int)code];
            const _Float16 prod = (_Float16)(sv * wf[rel]);
            acc += (float)prod;
            if ((mask >> rel) & 1u) emit(m, rel);
        }
    }
}

struct PackedView {
    const Header *hdr;
    const float *w_in, *w_out, *b_out, *b_in, *cb, *rec_p, *cb_img;
    int dp, cb_img_pieces;
};

PackedView view_of(const void *packed, int d, int c, int K, int L)
{
    const Layout lo = make_layout(d, c, K, L);
    const char *b = static_cast<const char *>(packed);
    PackedView v;
    v.hdr = reinterpret_cast<const Header *>(b);
    v.w_in = reinterpret_cast<const float *>(b + lo.w_in);
    v.w_out = reinterpret_cast<const float *>(b + lo.w_out);
    v.b_out = reinterpret_cast<const float *>(b + lo.b_out);
    v.b_in = reinterpret_cast<const float *>(b + lo.b_in);
    v.cb = reinterpret_cast<const float *>(b + lo.cb);
    v.cb_img = reinterpret_cast<const float *>(b + lo.cb_img);
    v.cb_img_pieces = lo.cb_img_pieces;
    v.rec_p = reinterpret_cast<const float *>(b + lo.rec_p);
    v.dp = lo.dp;
    return v;
}

int decode_common(const void *packed, int d, int c, int K, int L, const void *src, bool from_idx,
                  int idx_dtype, int64_t n_tokens, int nq, const uint8_t *level_mask, float *out,
                  void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("decode: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !src || !out) return fail("decode: null pointer");
    if (n_tokens < 0 || nq < 1) return fail("decode: bad sizes n_tokens=%lld nq=%d", (long long)n_tokens, nq);
    if (L > MAX_MASK_LEVELS) return fail("decode: L=%d exceeds %d", L, MAX_MASK_LEVELS);
    if (from_idx && (idx_dtype < RQAE_IDX_I64 || idx_dtype > RQAE_IDX_U16)) return fail("decode: bad idx_dtype %d", idx_dtype);
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    DecParams p;
    memset(&p, 0, sizeof(p));
    int used = 0, top = 0;
    for (int l = 0; l < L; ++l) {
        const bool on = level_mask ? level_mask[l] != 0 : l < nq;
        if (!on) continue;
        if (l >= nq) return fail("decode: level %d selected but indices hold only %d levels", l, nq);
        p.mask[l >> 5] |= 1u << (l & 31);
        ++used;
        top = l + 1;
    }
    if (!used) return fail("decode: no level selected");
    p.w_out = v.w_out; p.b_out = v.b_out; p.cb0 = v.cb; p.rec_p = v.rec_p;
    p.src = src; p.out = out; p.n_tokens = n_tokens;
    p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.nlev = top; p.nq = nq; p.idx_dtype = idx_dtype;
    p.aligned16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    return dispatch_decode(v.dp / 256, c, p, from_idx, static_cast<hipStream_t>(stream));
}

}  // namespace

// =================================================================== C ABI
#ifdef RQAE_TIMELINE
extern "C" int rqae_hip_debug_timeline(long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), sizeof(long long) * 4 * 8 * 64);
}
#endif

extern "C" {

int rqae_hip_abi_version(void) { return RQAE_HIP_ABI_VERSION; }

const char *rqae_hip_last_error(void) { return g_err; }

size_t rqae_hip_packed_bytes(int d, int c, int K, int L)
{
    if (!shape_ok(d, c, K, L)) {
        fail("unsupported shape d=%d c=%d K=%d L=%d (need 1<=d<=%d, 1<=c<=%d)", d, c, K, L, MAX_D, MAX_C);
        return 0;
    }
    return make_layout(d, c, K, L).total;
}

int rqae_hip_pack(void *packed, int d, int c, int K, int L, const float *w_in, const float *b_in,
                  const float *codebook, const float *w_out, const float *b_out, int codebook_shared,
                  void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("pack: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !w_in || !b_in || !codebook || !w_out || !b_out) return fail("pack: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Layout lo = make_layout(d, c, K, L);
    char *b = static_cast<char *>(packed);
    Header h = {MAGIC, d, c, K, L, lo.dp, codebook_shared ? 1 : 0, 0};
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<Header *>(b), h,
                       reinterpret_cast<float *>(b + lo.w_in), reinterpret_cast<float *>(b + lo.w_out),
                       reinterpret_cast<float *>(b + lo.b_out), w_in, w_out, b_out);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pack_records_kernel, dim3(2048), dim3(256), 0, s, h, lo.nq, rec_out_quads(lo.dp / 256, c),
                       reinterpret_cast<float *>(b + lo.rec_p), w_in, w_out, b_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b + lo.b_in, b_in, sizeof(float) * (size_t)L * c, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + lo.cb, codebook, sizeof(float) * (size_t)L * K * c, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(pack_cb_image_kernel, dim3(256), dim3(256), 0, s, c, K, lo.cb_img_pieces * 256,
                       reinterpret_cast<float *>(b + lo.cb_img), codebook);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_encode(const void *packed, int d, int c, int K, int L, const void *x, int x_dtype,
                    int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("encode: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !x || !indices) return fail("encode: null pointer");
    if (levels < 1 || levels > L) return fail("encode: levels=%d outside 1..%d", levels, L);
    if (x_dtype < RQAE_X_F32 || x_dtype > RQAE_X_F16) return fail("encode: bad x_dtype %d", x_dtype);
    if (n_tokens < 0) return fail("encode: negative n_tokens");
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    EncParams p;
    memset(&p, 0, sizeof(p));
    p.w_in = v.w_in; p.w_out = v.w_out; p.b_out = v.b_out; p.b_in = v.b_in; p.cb = v.cb; p.hdr = v.hdr;
    p.rec_p = v.rec_p; p.cb_img = v.cb_img; p.cb_img_pieces = v.cb_img_pieces;
    p.x = x; p.q_out = q_out; p.idx = reinterpret_cast<long long *>(indices);
    p.n_tokens = n_tokens; p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.levels = levels; p.x_dtype = x_dtype;
    p.aligned16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q_out)) & 15) == 0;
    int rc = 0;
    dispatch_encode(v.dp / 256, c, &p, static_cast<hipStream_t>(stream), &rc);
    return rc;
}

// shapes whose encode kernel can apply the Gemma norm on load (the LDS kernel's instantiations)
static bool gemma_fusable(int d, int c)
{
    const int P = (d + 255) / 256;
    return env_variant() == V_AUTO && d % 256 == 0 && ((P == 9 && (c == 5 || c == 4)) || (P == 1 && c == 4));
}

int rqae_hip_encode_gemma_supported(int d, int c, int K, int L)
{
    return shape_ok(d, c, K, L) && gemma_fusable(d, c) ? 1 : 0;
}

int rqae_hip_encode_gemma(const void *packed, int d, int c, int K, int L, const void *hs, int hs_dtype,
                          const float *norm_weight, float eps, int64_t n_tokens, int levels, float *q_out,
                          int64_t *indices, float *inv_rms, void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("encode_gemma: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!gemma_fusable(d, c)) return fail("encode_gemma: no fused kernel for d=%d c=%d (see rqae_hip_encode_gemma_supported)", d, c);
    if (!packed || !hs || !indices || !norm_weight || !inv_rms || !q_out) return fail("encode_gemma: null pointer");
    if (levels < 1 || levels > L) return fail("encode_gemma: levels=%d outside 1..%d", levels, L);
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("encode_gemma: bad hs_dtype %d", hs_dtype);
    if (n_tokens < 0) return fail("encode_gemma: negative n_tokens");
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    EncParams p;
    memset(&p, 0, sizeof(p));
    p.w_in = v.w_in; p.w_out = v.w_out; p.b_out = v.b_out; p.b_in = v.b_in; p.cb = v.cb; p.hdr = v.hdr;
    p.rec_p = v.rec_p; p.cb_img = v.cb_img; p.cb_img_pieces = v.cb_img_pieces;
    p.x = hs; p.q_out = q_out; p.idx = reinterpret_cast<long long *>(indices);
    p.n_tokens = n_tokens; p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.levels = levels; p.x_dtype = hs_dtype;
    p.norm_w = norm_weight; p.inv_rms = inv_rms; p.norm_eps = eps;
    p.aligned16 = ((reinterpret_cast<uintptr_t>(hs) | reinterpret_cast<uintptr_t>(q_out) |
                    reinterpret_cast<uintptr_t>(norm_weight)) & 15) == 0;
    int rc = 0;
    dispatch_encode(v.dp / 256, c, &p, static_cast<hipStream_t>(stream), &rc);
    return rc;
}

int rqae_hip_kernel_info(int d, int c, int K, int L, char *buf, size_t buflen)
{
    if (!shape_ok(d, c, K, L)) return fail("kernel_info: unsupported shape");
    if (!buf || !buflen) return fail("kernel_info: null buffer");
    int rc = 0;
    const char *name = dispatch_encode(((d + 255) / 256), c, nullptr, nullptr, &rc);
    snprintf(buf, buflen, "%s", name);
    return 0;
}

int rqae_hip_quantize(const float *pin, int64_t n, int c, int K, const float *codebook_l, float *cv,
                      int64_t *idx, void *stream)
{
    if (c < 1 || c > MAX_C || K < 1) return fail("quantize: unsupported c=%d K=%d", c, K);
    if (!pin || !codebook_l || !cv || !idx) return fail("quantize: null pointer");
    if (n < 0) return fail("quantize: negative n");
    if (n == 0) return 0;
    QuantParams p;
    p.p = pin; p.cb = codebook_l; p.cv = cv; p.idx = reinterpret_cast<long long *>(idx);
    p.n = n; p.c = c; p.K = K;
    int kch = (int)(LDS_CODEBOOK_BUDGET / (sizeof(float) * c));
    kch = (kch / 64) * 64;
    if (kch >= K) kch = ((K + 63) / 64) * 64;
    p.kch = kch;
    const int rows = kch < K ? kch : K;
    const size_t lds = sizeof(float) * (size_t)rows * c;
    // token groups per workgroup: enough workgroups to fill the chip several times over, at most 16 groups each
    int iters = 1;
    if (kch >= K) {
        const long long groups = (n + 15) / 16;
        while (iters < 16 && groups / (iters * 2) >= 256 * 8) iters *= 2;
    }
    p.iters = iters;
    const long long nblk = (n + 16LL * iters - 1) / (16LL * iters);
    if (nblk > 0x7fffffffLL) return fail("quantize: too many rows for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (c <= 4) return go(rqae_quantize_kernel<4>);
    if (c <= 8) return go(rqae_quantize_kernel<8>);
    return go(rqae_quantize_kernel<16>);
}

int rqae_hip_decode(const void *packed, int d, int c, int K, int L, const void *indices, int idx_dtype,
                    int64_t n_tokens, int idx_levels, const uint8_t *level_mask, float *out, void *stream)
{
    return decode_common(packed, d, c, K, L, indices, true, idx_dtype, n_tokens, idx_levels, level_mask, out, stream);
}

int rqae_hip_decode_values(const void *packed, int d, int c, int K, int L, const float *cv, int64_t n_tokens,
                           int nq, const uint8_t *level_mask, float *out, void *stream)
{
    return decode_common(packed, d, c, K, L, cv, false, 0, n_tokens, nq, level_mask, out, stream);
}

int rqae_hip_gemma_norm(const void *hs, int hs_dtype, const float *norm_weight, float eps, int64_t n_tokens, int d,
                        float *normed, float *inv_rms, void *stream)
{
    if (!hs || !norm_weight || !normed || !inv_rms) return fail("gemma_norm: null pointer");
    if (n_tokens < 0 || d < 1) return fail("gemma_norm: bad sizes");
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("gemma_norm: bad hs_dtype %d", hs_dtype);
    if (n_tokens == 0) return 0;
    const long long nblk = (n_tokens + 3) / 4;
    if (nblk > 0x7fffffffLL) return fail("gemma_norm: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t hsz = hs_dtype == RQAE_X_F32 ? 4 : 2;
    const int P = d / 256;
    const bool vec = d % 256 == 0 && (P == 1 || P == 4 || P == 9 || P == 14) &&
                     (reinterpret_cast<uintptr_t>(hs) % (4 * hsz)) == 0 && (reinterpret_cast<uintptr_t>(normed) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(norm_weight) & 15) == 0;
    if (vec) {
#define NORM_VEC(HT, PP) hipLaunchKernelGGL((gemma_norm_vec_kernel<HT, PP>), dim3((unsigned)nblk), dim3(256), 0, s, \
                                            (const HT *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms)
#define NORM_VEC_D(HT) do { if (P == 1) NORM_VEC(HT, 1); else if (P == 4) NORM_VEC(HT, 4); else if (P == 9) NORM_VEC(HT, 9); else NORM_VEC(HT, 14); } while (0)
        if (hs_dtype == RQAE_X_F16) NORM_VEC_D(_Float16);
        else if (hs_dtype == RQAE_X_BF16) NORM_VEC_D(uint16_t);
        else NORM_VEC_D(float);
#undef NORM_VEC_D
#undef NORM_VEC
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (hs_dtype == RQAE_X_F16)
        hipLaunchKernelGGL(gemma_norm_kernel<_Float16>, dim3((unsigned)nblk), dim3(256), 0, s, (const _Float16 *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    else if (hs_dtype == RQAE_X_BF16)
        hipLaunchKernelGGL(gemma_norm_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), 0, s, (const uint16_t *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    else
        hipLaunchKernelGGL(gemma_norm_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, (const float *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_gemma_denorm_patch(const float *q_out, const float *norm_weight, const float *inv_rms, int64_t n_tokens,
                                int d, int seq_len, int skip_bos, void *hs_inout, int hs_dtype, float *new_out,
                                void *stream)
{
    if (!q_out || !norm_weight || !inv_rms || !hs_inout) return fail("gemma_denorm_patch: null pointer");
    if (n_tokens < 0 || d < 1 || seq_len < 1) return fail("gemma_denorm_patch: bad sizes");
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("gemma_denorm_patch: bad hs_dtype %d", hs_dtype);
    if (n_tokens == 0) return 0;
    const long long nblk = (n_tokens + 3) / 4;
    if (nblk > 0x7fffffffLL) return fail("gemma_denorm_patch: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t hsz = hs_dtype == RQAE_X_F32 ? 4 : 2;
    const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(hs_inout) % (4 * hsz)) == 0 &&
                     (reinterpret_cast<uintptr_t>(q_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(norm_weight) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(new_out) & 15) == 0;
    if (vec) {
#define PATCH_VEC(HT) hipLaunchKernelGGL(gemma_denorm_patch_vec_kernel<HT>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, \
                                         norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (HT *)hs_inout, new_out)
        if (hs_dtype == RQAE_X_F16) PATCH_VEC(_Float16);
        else if (hs_dtype == RQAE_X_BF16) PATCH_VEC(uint16_t);
        else PATCH_VEC(float);
#undef PATCH_VEC
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (hs_dtype == RQAE_X_F16)
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<_Float16>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (_Float16 *)hs_inout, new_out);
    else if (hs_dtype == RQAE_X_BF16)
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (uint16_t *)hs_inout, new_out);
    else
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (float *)hs_inout, new_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_intensity_prepare(const void *codebook_sims, const int *center, const void *layer_weights, int M, int K,
                               void *table, void *wcum, void *stream)
{
    if (!codebook_sims || !center || !layer_weights || !table || !wcum) return fail("intensity_prepare: null pointer");
    if (M < 1 || K < 1) return fail("intensity_prepare: bad sizes M=%d K=%d", M, K);
    hipLaunchKernelGGL(intensity_prepare_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const _Float16 *)codebook_sims, center, (const _Float16 *)layer_weights, M, K,
                       (_Float16 *)table, (_Float16 *)wcum);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_intensity(const void *table, const void *layer_weights, const void *wcum, int M, int K,
                       const void *indices, int idx_dtype, int64_t n_tokens, int nq, const int *layers_dev,
                       int n_layers, void *out, void *stream)
{
    if (!table || !layer_weights || !wcum || !indices || !layers_dev || !out) return fail("intensity: null pointer");
    if (M < 1 || K < 1 || nq < M || n_layers < 1) return fail("intensity: bad sizes M=%d K=%d nq=%d n_layers=%d", M, K, nq, n_layers);
    if (idx_dtype < RQAE_IDX_I64 || idx_dtype > RQAE_IDX_U16) return fail("intensity: bad idx_dtype %d", idx_dtype);
    if (n_tokens < 0) return fail("intensity: negative n_tokens");
    if (n_tokens == 0) return 0;
    IntensityParams p;
    p.table = (const _Float16 *)table; p.w = (const _Float16 *)layer_weights; p.wcum = (const _Float16 *)wcum;
    p.idx = indices; p.layers = layers_dev; p.out = (_Float16 *)out;
    p.n_tokens = n_tokens; p.nq = nq; p.M = M; p.K = K; p.n_layers = n_layers;
    const size_t es = idx_dtype == RQAE_IDX_I64 ? 8 : idx_dtype == RQAE_IDX_I32 ? 4 : 2;
    const int TL = 16;
    int LC = TL;
    while (LC > 1 && (size_t)LC * K * 2 > 48 * 1024) LC >>= 1;
    if ((size_t)LC * K * 2 > 48 * 1024) return fail("intensity: K=%d too large for the LDS table", K);
    if (((size_t)LC * K * 2) % 16 != 0) return fail("intensity: K=%d needs LC*K*2 %% 16 == 0 (LC=%d)", K, LC);
    p.LC = LC;
    p.vec_ok = (reinterpret_cast<uintptr_t>(indices) & 15) == 0 && ((size_t)nq * es) % 16 == 0 &&
               (reinterpret_cast<uintptr_t>(table) & 15) == 0;
    if ((reinterpret_cast<uintptr_t>(table) & 15) != 0) return fail("intensity: table must be 16-byte aligned");
    const int RS = (int)(TL / (16 / es)) + 1;
    const size_t lds = (((size_t)LC * K * 2 + 15) & ~(size_t)15) + 64 * 4 + 16 + (size_t)4 * 64 * RS * 16;
    const long long nblk = (n_tokens + 255) / 256;
    if (nblk > 0x7fffffffLL) return fail("intensity: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (idx_dtype == RQAE_IDX_I64) return go(intensity_kernel<long long>);
    if (idx_dtype == RQAE_IDX_I32) return go(intensity_kernel<int>);
    return go(intensity_kernel<uint16_t>);
}

}  // extern "C"
#else   // RQAE_TU_MAIN
}  // namespace
#endif  // !RQAE_TU_MAIN
