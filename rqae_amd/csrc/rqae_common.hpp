// rqae_common.hpp -- shared by the translation units of librqae_hip.so: constants, the error text, the packed
// parameter block's layout, the canonical lane<->element map, cross-lane moves, per-token scalar math, tile
// loads/stores, LDS-DMA and packed-fp32 helpers.  Everything is internal to a translation unit (anonymous
// namespace) except the error text and the main-path launcher declared below.
#pragma once
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
// librqae_hip.so is two translation units (__graft_entry__.py): encode_main.hip -- the LDS encode kernel for
// d = 2304, packed-VALU arm -- built with the ILP-first machine scheduler, and rqae_hip.hip -- everything else --
// with the default one.  They share the error text and one launcher.
#ifdef RQAE_TU_MAIN
extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];   // __thread, not thread_local: no TLS init wrapper (a hidden weak _ZTH call crashes)
#else
__attribute__((visibility("hidden"))) __thread char rqae_g_err[512] = "";
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

__attribute__((unused)) Layout make_layout(int d, int c, int K, int L)
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

__attribute__((unused)) bool shape_ok(int d, int c, int K, int L)
{
    return d >= 1 && d <= MAX_D && c >= 1 && c <= MAX_C && K >= 1 && L >= 1;
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
// Timeline probe (build with -DRQAE_TIMELINE, tools/timeline.py; never part of the shipped library):
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

}  // namespace

