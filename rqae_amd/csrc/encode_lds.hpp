// encode_lds.hpp -- the main encode kernel (rqae_encode_lds_kernel) and its launcher.  Included by both
// translation units: encode_main.hip instantiates it for d = 2304 (packed-VALU arm) under the ILP-first machine
// scheduler, rqae_hip.hip for the other shapes and for the MFMA arm under the default one.
#pragma once
#include "rqae_common.hpp"

namespace {

// NORM: x is the hooked LLM layer's output; the Gemma RMSNorm of RQAE.hook (rqae/model.py:279, rqae/llm.py:65-66)
// is applied while the tile is in registers (rqae_hip_encode_gemma), normed never exists in HBM.
// NH: token halves of 8 tokens (4 waves each).  2 with the reconstruction (its registers allow 2 waves per SIMD);
// the codes-only kernel of d = 2304 runs 3 (12 waves, 24 tokens per workgroup, 168 VGPRs): the latency-bound
// quantise phase then serves 24 tokens per level instead of 16 and the sweeps issue at the 3-wave rate.
// TPWT: tokens per wave, 8 or 4 (4 only for the smallest batches: one token group per wave, 4 tokens per workgroup).
template <int P, int C, bool RECON, bool MFMA, bool NORM = false, int NH = 2, int TPWT = 8>
__global__ void __launch_bounds__(256 * NH, NH)
rqae_encode_lds_kernel(const EncParams p)
{
    constexpr int TPW = TPWT, NW = 4 * NH, TG = TPW * NH, TQ = 2, JW = 8, NG = TPW / 4, NPAIR = TPW / 2;
    static_assert(TPW == 8 || (TPW == 4 && !NORM && !MFMA), "token groups of four; the norm prologue is written for 8");
#ifdef RQAE_ABLATE_BUILD   // timing experiments only (tools/timeline.py): RQAE_HIP_ABLATE bits skip parts of the kernel
    const int ABLATE = p.ablate;
#else
    // d = 2304: compiled out.  The other shapes keep the branches on a value that is always 0 at run time
    // (launch_encode_lds): with them folded away hipcc schedules the d = 256 kernel 14 % slower (0.42 -> 0.48 ms
    // per 262 144 tokens, measured either side of the change).
    const int ABLATE = P == 9 ? 0 : p.ablate;
#endif
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
    // Codes-only launches are persistent: one workgroup per CU (the LDS footprint allows no more) walks the
    // 16-token tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the level-0 codebook image stays in LDS and the
    // workgroup launch and its first instruction fetches are paid once per CU instead of once per tile (2.68 ->
    // 2.60 ms).  With the reconstruction the loop keeps the kernel arguments live across its back-edge and hipcc
    // then spills SGPRs inside the sweeps (720 more v_readlane, 2.98 -> 3.40 ms; the same with the tile index in
    // a VGPR), so that instantiation keeps one tile per workgroup.
    constexpr bool PERSIST = !RECON;
    const int ntiles = (int)((p.n_tokens + TG - 1) / TG);      // (the launcher bounds it)

    const bool shared_cb = p.hdr->shared_cb != 0;
    const int nchunk = (p.K + p.kch - 1) / p.kch;
    const bool stage_once = shared_cb && nchunk == 1;

    // quantise-phase role of this lane: lanes 0..7 -> token 2*wave, lanes 8..15 -> token 2*wave+1
    const int tq = lane / JW, jj = lane % JW;
    const int tl_q = wave * TQ + (tq < TQ ? tq : 0);
    const bool act = tq < TQ && jj < C && tl_q < TG;   // (with 4 tokens per workgroup only waves 0 and 1 quantise)

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

    int tile = blockIdx.x;
    do {
    const long long tok0 = (long long)tile * TG;
    const long long tokw = tok0 + half * TPW;
    const bool first_tile = tile == (int)blockIdx.x;
    {   // record 0: only its lin_in part is read (that part of `wts` was last read before the previous tile's last
        // quantise phase, i.e. two barriers ago)
        for (int s = 0; s < NSLICE; ++s)
            for (int q = QOUT + wave; q < NQ; q += NW)
                dma_pieces<1>(p.rec_p + (s * NQ + q) * 256, wts_addr + (s * NQ + q) * 1024, lane * 16);
        if (nchunk == 1) {   // level-0 codebook: its LDS image by DMA -- no registers, no wait in front of the x loads
            if (first_tile || !stage_once) {   // (a shared codebook that fits stays for every tile)
                const unsigned cbl_addr = lds_addr_of(cbl);
                for (int pc = wave; pc < p.cb_img_pieces; pc += NW)
                    dma_pieces<1>(p.cb_img + pc * 256, cbl_addr + pc * 1024, lane * 16);
            }
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
        if (!(ABLATE & 1) && (RECON || l + 1 < p.levels)) dma_begin(l + 1);   // (codes only: the last record is never applied)
        if (!(ABLATE & 2)) quantise(l, bin_own);
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
        if (!(ABLATE & 4)) {                              // apply level l, project onto level l+1
            pass_out(F{}, F{});
            pass_in();
        }
        restage(l + 1);
        slot(l + 1, bin_own);
    }
    if constexpr (RECON) {                                // apply the last level (codes only: nothing left to do)
        if (p.levels > 1) pass_out(F{}, T{});
        else pass_out(T{}, T{});
    }
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
    } while (PERSIST && (tile += gridDim.x) < ntiles);
}

template <int P, int C>
int launch_encode_lds(const EncParams &p, hipStream_t stream, bool use_mfma)
{
#if defined(RQAE_TU_MAIN) && !defined(RQAE_TIMELINE)
    // codes only: 12 waves x 8 tokens.  Small batches with the reconstruction (the hook is called with 4 x 128
    // tokens, scripts/1_create_activations.py:143-186): 4 waves x 8 tokens, so that twice as many CUs take part
    // and a level's sweeps run one wave per SIMD.
    static int n_cu_small = 0;
    if (n_cu_small == 0) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&n_cu_small, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const bool plain = !use_mfma && !p.norm_w;
    const int NHsel = (plain && !p.q_out) ? 3 : (!use_mfma && p.q_out && p.n_tokens <= 8LL * n_cu_small) ? 1 : 2;
    const int TPWsel = (NHsel == 1 && plain && p.n_tokens <= 4LL * n_cu_small) ? 4 : 8;
#else
    const int NHsel = 2, TPWsel = 8;
#endif
    const int TG = TPWsel * NHsel;
    EncParams q = p;
    const size_t fixed = sizeof(float) * ((size_t)NSLICE * rec_quads(P, C) * 256 + NSLICE * TG * C + ((TG * C + 3) & ~3));
    const size_t cb_budget = 160 * 1024 - fixed - 1024;
    int kch = (int)(cb_budget / (sizeof(float) * cb_lds_floats(C, 1)));
    kch = (kch / 64) * 64;
    if (kch < 64) return fail("encode: no LDS left for the codebook");
    if (kch >= p.K) kch = ((p.K + 63) / 64) * 64;
    q.kch = kch;
#ifdef RQAE_ABLATE_BUILD
    {
        const char *ab = getenv("RQAE_HIP_ABLATE");
        q.ablate = ab ? atoi(ab) : 0;
    }
#else
    q.ablate = 0;
#endif
    const int rows = kch < p.K ? kch : p.K;
#ifdef RQAE_TIMELINE
    const size_t cbf_tl = kch >= p.K ? (size_t)p.cb_img_pieces * 256 : (size_t)cb_lds_floats(C, rows);
    q.tl_lds_floats = (int)((fixed / sizeof(float) + cbf_tl + 3) & ~(size_t)3);
    const size_t lds = sizeof(float) * (size_t)q.tl_lds_floats + 8 * 64 * sizeof(long long);
#else
    const size_t cbf = kch >= p.K ? (size_t)p.cb_img_pieces * 256 : (size_t)cb_lds_floats(C, rows);   // whole pieces when DMA-staged
    const size_t lds = fixed + sizeof(float) * cbf;
#endif
    long long nblk = (p.n_tokens + TG - 1) / TG;
    if (nblk > 0x7fffffffLL) return fail("encode: too many tokens for one launch");
#ifndef RQAE_TIMELINE   // (the timeline build stamps one tile per workgroup)
    if (!p.q_out) {   // persistent workgroups: as many as can be resident (LDS allows floor(160 KB / lds) per CU), each walks tiles
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        long long per_cu = lds > 0 ? (160 * 1024) / (long long)lds : 1;   // LDS; at most 4 by wave slots (8 waves each)
        if (per_cu > 4) per_cu = 4;
        if (NHsel == 3 && per_cu > 2) per_cu = 2;
        const long long resident = (long long)n_cu * per_cu;
        if (resident >= 1 && nblk > resident) nblk = resident;
    }
#endif
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256 * NHsel), lds, stream, q);
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
#if defined(RQAE_TU_MAIN) && !defined(RQAE_TIMELINE)
        if (NHsel == 1) return go(rqae_encode_lds_kernel<P, C, true, false, true, 1>);
#endif
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
#if defined(RQAE_TU_MAIN) && !defined(RQAE_TIMELINE)
    if (p.q_out && NHsel == 1 && TPWsel == 4) return go(rqae_encode_lds_kernel<P, C, true, false, false, 1, 4>);
    if (p.q_out && NHsel == 1) return go(rqae_encode_lds_kernel<P, C, true, false, false, 1>);
#endif
    if (p.q_out) return go(rqae_encode_lds_kernel<P, C, true, false>);
#if defined(RQAE_TU_MAIN) && !defined(RQAE_TIMELINE)
    return go(rqae_encode_lds_kernel<P, C, false, false, false, 3>);
#else
    return go(rqae_encode_lds_kernel<P, C, false, false>);
#endif
#else
    if constexpr (P != 9) {
        if (p.q_out) return go(rqae_encode_lds_kernel<P, C, true, false>);
        return go(rqae_encode_lds_kernel<P, C, false, false>);
    }
    return fail("encode: unreachable");
#endif
}

}  // namespace

