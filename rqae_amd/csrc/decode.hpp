// decode.hpp -- rqae_quantize_kernel (RQAE.quantize) and the decode kernels (RQAE.decode, rqae/model.py:232-252) with their launchers; included by rqae_hip.hip.
#pragma once
#include "rqae_common.hpp"

namespace {

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
        float best[TQ];
        int bidx[TQ];
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) { best[tq] = -INFINITY; bidx[tq] = 0; }
        // c <= 8: lane (token, j) = (lane / 8, lane % 8) divides its own component once (the encode kernel's quantise
        // phase), the unit vectors reach every lane as SGPR pairs and the scan is packed over the two token pairs.
        // c > 8: every lane normalises all four tokens itself and scans with plain fmas.
        float xn[C > 8 ? TQ : 1][C];
        f2 xs2[C > 8 ? 1 : TQ / 2][C];
        if constexpr (C <= 8) {
            const int tq_l = lane >> 3, jj = lane & 7;
            const long long tok_own = tokw + (tq_l < TQ ? tq_l : 0);
            const bool mine = tq_l < TQ && tok_own < p.n;
            float qk[C];
#pragma unroll
            for (int k = 0; k < C; ++k) qk[k] = (k < cp && mine) ? p.p[tok_own * cp + k] : 0.0f;
            const float nrm = aten_norm2<C>(qk, cp);
            const float q_own = (jj < cp && mine) ? p.p[tok_own * cp + jj] : 0.0f;
            const float xn_own = q_own / nrm;
#pragma unroll
            for (int pr = 0; pr < TQ / 2; ++pr)
#pragma unroll
                for (int k = 0; k < C; ++k)   // components beyond c are exact zeros: their fmas add +0
                    xs2[pr][k] = k < cp ? (f2){lane_bcast(xn_own, 16 * pr + k), lane_bcast(xn_own, 16 * pr + 8 + k)} : splat2(0.0f);
        } else {
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const long long tok = tokw + tq;
                float q[C];
#pragma unroll
                for (int j = 0; j < C; ++j) q[j] = (j < cp && tok < p.n) ? p.p[tok * cp + j] : 0.0f;
                const float nrm = aten_norm2<C>(q, cp);
#pragma unroll
                for (int j = 0; j < C; ++j) xn[tq][j] = q[j] / nrm;
            }
        }
        auto scan = [&](int rows, int row_base) {
            if constexpr (C <= 8) {
                for (int r0 = 0; r0 < rows; r0 += 64) {
                    const int row = r0 + lane;
                    const bool in = row < rows;
                    float cbv[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const float v = smem[(in ? row : 0) * cp + (k < cp ? k : cp - 1)];
                        cbv[k] = k < cp ? v : 0.0f;
                    }
#pragma unroll
                    for (int pr = 0; pr < TQ / 2; ++pr) {
                        f2 sv = splat2(0.0f);
#pragma unroll
                        for (int k = 0; k < C; ++k) sv = fma2(xs2[pr][k], splat2(cbv[k]), sv);
                        const bool u0 = in && sv.x > best[2 * pr], u1 = in && sv.y > best[2 * pr + 1];   // selects, no branches
                        best[2 * pr] = u0 ? sv.x : best[2 * pr];
                        bidx[2 * pr] = u0 ? row_base + row : bidx[2 * pr];
                        best[2 * pr + 1] = u1 ? sv.y : best[2 * pr + 1];
                        bidx[2 * pr + 1] = u1 ? row_base + row : bidx[2 * pr + 1];
                    }
                }
            } else {
                scan_rows<C, TQ>(xn, smem, rows, row_base, cp, lane, best, bidx);
            }
        };
        if (nchunk == 1) {
            scan(p.K, 0);
        } else {
            for (int ch = 0; ch < nchunk; ++ch) {             // iters == 1 here: every wave of the block takes part
                const int row_base = ch * p.kch;
                const int rows = (p.K - row_base) < p.kch ? (p.K - row_base) : p.kch;
                if (ch > 0) __syncthreads();
                stage_floats(smem, p.cb + (size_t)row_base * cp, rows * cp);
                __syncthreads();
                scan(rows, row_base);
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
    int *status;          // include/rqae_hip.h "Status word" (a code outside -K..K-1), or NULL
    uint32_t mask[MAX_MASK_LEVELS / 32];
};

__device__ __forceinline__ long long load_idx(const void *src, int dt, size_t off)
{
    if (dt == RQAE_IDX_I64) return reinterpret_cast<const long long *>(src)[off];
    if (dt == RQAE_IDX_I32) return reinterpret_cast<const int *>(src)[off];
    return reinterpret_cast<const uint16_t *>(src)[off];
}

// torch indexing as `self.codebook[0][indices]` applies it (rqae/model.py:234): -K <= code < K, a negative code counts from the
// last row; anything else raises IndexError there -- here it reads row 0 and flags the call (never an out-of-bounds gather)
__device__ __forceinline__ long long checked_code(long long ix, int K, int *status)
{
    ix = ix < 0 ? ix + K : ix;
    if ((unsigned long long)ix >= (unsigned long long)K) {
        if (status) atomicOr(status, RQAE_STATUS_CODE_RANGE);
        ix = 0;
    }
    return ix;
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
                const long long ix = checked_code(load_idx(p.src, p.idx_dtype, (size_t)tok * p.nq + l), p.K, p.status);
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
__global__ void __launch_bounds__(512, 4)
rqae_decode_lds_kernel(const DecParams p)
{
    constexpr int TPW = 8, NW = 8, TG = 64, NPAIR = 4, NG = 2, LCH = 8;
    constexpr int NQ = rec_quads(P, C);          // quads per lane per slice in a pass record
    constexpr int QD = rec_out_quads(P, C);      // its leading quads: the packed {w_out[0..C-1], b_out} stream
    constexpr int W = C + 1;
    constexpr int BUF = QD * 256;                // floats per weight buffer: one slice of one level
    constexpr int dp = 256 * P;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf = smem;                          // [NBUF][QD][lane][4]
    float *cvl = wbuf + NBUF * BUF;                 // [LCH][C][TG]  (token pairs contiguous)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // decode is elementwise in (token, element): a workgroup owns ONE slice (a quarter of d) of 64 tokens, its
    // 8 waves 8 tokens each, so a level's weights cross L2 -> LDS once per 64 tokens and fit in 14 KB
    const int slice = blockIdx.x & 3;
    const long long tok0 = (long long)(blockIdx.x >> 2) * TG;
    const long long tokw = tok0 + wave * TPW;
    const bool full_tile = (tokw + TPW <= p.n_tokens) && (p.d == dp) && p.aligned16;
    const unsigned wbuf_addr = lds_addr_of(wbuf);

    auto level_on = [&](int l) { return ((p.mask[l >> 5] >> (l & 31)) & 1u) != 0; };
    auto next_on = [&](int l) { while (l < p.nlev && !level_on(l)) ++l; return l; };   // uniform scalar loop
    // lin_out quads of level l (record l+1), this slice -> buffer b; the 8 waves share the pieces
    auto issue_dma = [&](int l, int b) {
        const float *rec = p.rec_p + ((size_t)(l + 1) * NSLICE + slice) * NQ * 256;
        for (int pc = wave; pc < QD; pc += NW)
            dma_pieces<1>(rec + (size_t)pc * 256, wbuf_addr + (b * BUF + pc * 256) * 4, lane * 16);
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
                    const long long ix = checked_code(load_idx(p.src, p.idx_dtype, (size_t)tok * p.nq + lv), p.K, p.status);
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
        int dpc = (NBUF == 2 && ln < p.nlev) ? wave : QD;
        const float *drec = p.rec_p + ((size_t)(ln + 1) * NSLICE + slice) * NQ * 256;
        auto dma_step = [&]() __attribute__((always_inline)) {
            if (dpc < QD) {
                dma_pieces<1>(drec + (size_t)dpc * 256, wbuf_addr + ((buf ^ 1) * BUF + dpc * 256) * 4, lane * 16);
                dpc += NW;
            }
        };

        const f4 *wl = reinterpret_cast<const f4 *>(wbuf + buf * BUF) + lane;
        const float *cvp = cvl + (l - staged_from) * C * TG + wave * TPW;
        // element step outermost: every weight quad is read from LDS once for the wave's 8 tokens
        f2 cv2[NG][2][C];
        const float cv_lane = lane < C * TPW ? cvp[(lane / TPW) * TG + (lane % TPW)] : 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int k = 0; k < C; ++k)   // lane k*8 + t holds the value of (k, token t): one LDS read per level
                    cv2[g][pr][k] = (f2){lane_bcast(cv_lane, k * TPW + 4 * g + 2 * pr), lane_bcast(cv_lane, k * TPW + 4 * g + 2 * pr + 1)};
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
        while (dpc < QD) dma_step();                          // more pieces per wave than element steps
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
    const char *nb = getenv("RQAE_HIP_DECODE_NBUF");     // 1 = single weight buffer (a barrier + exposed load per level)
    const int nbuf = nb && atoi(nb) == 1 ? 1 : 2;
    const size_t lds = sizeof(float) * ((size_t)nbuf * QD * 256 + 8 * C * 64);
    const long long nblk = NSLICE * ((p.n_tokens + 63) / 64);
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
        if (P == 14 && c == 13) return launch_decode<14, 13, 2, false>(p, from_idx, stream);   // (4 tokens per wave parks values in AGPRs)
    }
    if (P <= 8) return launch_decode<8, 16, 4, true>(p, from_idx, stream);
    return launch_decode<32, 16, 4, true>(p, from_idx, stream);
#endif
}

}  // namespace
