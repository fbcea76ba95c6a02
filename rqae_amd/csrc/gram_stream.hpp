// gram_stream.hpp -- level chain of the matrix-core arm for codebooks too large to stay in LDS (included by gram.hip).
//
// gram_levels_kernel keeps the codebook's bf16-piece image (K x 64 bytes) and its fp32 rows resident in LDS: K <= ~1100 rows.
// BASELINE.json configs[4] (C5) has K = 4096, depth 32: the image is 256 KB.  Here the image streams through a 3-stage ring in
// LDS by LDS-DMA, once per level, shared by the workgroup's 8 waves (512 tokens): 256 KB of L2 -> LDS traffic per 2048
// matrix instructions per wave, so the refill is 8 bytes per clock and CU; the fp32 rows for the exact rescoring (16 rows per
// token and level) and the Gram blocks come straight from L2.  Arithmetic, tie rules and outputs are gram_levels_kernel's
// (same codes); what changes is the schedule: the 8 waves walk the stages together (one barrier per stage of 32 row tiles =
// 128 matrix instructions per wave), so every wave of a workgroup runs every level, with or without tokens.
constexpr int SL_NW = 8;            // waves per workgroup (4 when the batch would otherwise leave CUs idle)
constexpr int SL_STT = 32;          // row tiles (1 KiB each) per stage
constexpr int SL_NST = 3;           // ring slots

__host__ __device__ inline size_t stream_lds_bytes(int nw) { return (size_t)SL_NST * SL_STT * 1024 + (size_t)nw * 4096; }

template <int CP, bool MULTI, int NW>
__global__ __launch_bounds__(NW * 64) void gram_levels_stream_kernel(LevParams p)
{
    if (p.lat_flag && *reinterpret_cast<const volatile int *>(p.lat_flag) >= 2) return;      // the lattice form ran (gram_wide.hpp)
    constexpr int GL = 8;
    constexpr int GP = ((GL * CP + 15) / 16) * 16;
    constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;
    constexpr int GB = (CP * CP + 3) & ~3;
    static_assert(6 * CP <= 32, "one k-step per cosine contraction");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int c = p.c;
    char *xch = smem + SL_NST * SL_STT * 1024 + wave * 4096;
    const float *cbf = p.cb_f32;                       // global (L2): K x c fp32, + 64 zero rows
    const int nstage = (p.KTP + SL_STT - 1) / SL_STT;
    constexpr int PW = SL_STT / NW;                    // pieces per wave and stage
    // stage s = row tiles [32 s, 32 s + 32) (tiles past KTP repeat the last one: never read)
    auto issue = [&](int s) {
        const int slot = s % SL_NST;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int tl = s * SL_STT + wave * PW + i;
            tl = tl < p.KTP ? tl : p.KTP - 1;
            dma_piece(uniform_ptr(p.cb_img + (size_t)tl * 512), lds0 + (slot * SL_STT + wave * PW + i) * 1024, (unsigned)lane * 16u);
        }
    };
    const int nlg = (p.levels + GL - 1) / GL;
    const int wq = (lane >> 2) & 3;
    const long long tiles_per_round = (long long)gridDim.x * NW;
    issue(0);
    if (nstage > 1) issue(1);
    for (long long tbase = (long long)blockIdx.x * NW; tbase < p.ntiles; tbase += tiles_per_round) {
        const bool live = tbase + wave < p.ntiles;                        // a wave without tokens keeps the workgroup's schedule
        const long long tile = live ? tbase + wave : p.ntiles - 1;
        const long long t = tile * 64 + lane;                             // < tok_pad
        for (int lg = 0; lg < nlg; ++lg) {
            float Q[GL][CP];
            {
                float buf[GP];
#pragma unroll
                for (int u = 0; u < GP / 16; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(p.P + (((size_t)(lg * (GP / 16) + u) * p.tok_pad + t) * 16 + 4 * v));
                        const f32x4 kc = *reinterpret_cast<const f32x4 *>(p.kconst + (lg * GP + 16 * u + 4 * v));
                        buf[16 * u + 4 * v + 0] = w[0] + kc[0];
                        buf[16 * u + 4 * v + 1] = w[1] + kc[1];
                        buf[16 * u + 4 * v + 2] = w[2] + kc[2];
                        buf[16 * u + 4 * v + 3] = w[3] + kc[3];
                    }
#pragma unroll
                for (int li = 0; li < GL; ++li)
#pragma unroll
                    for (int j = 0; j < CP; ++j) Q[li][j] = buf[li * CP + j];
            }
            float cvh[GL][CP];
            int codes[GL];
            unsigned img[GS / 2];
#pragma unroll
            for (int i = 0; i < GS / 2; ++i) img[i] = 0u;

            static_for<0, GL>([&](auto li_c) {
                constexpr int li = decltype(li_c)::value;
                const int l = lg * GL + li;
                if (l < p.levels) {
                    float q[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) q[j] = Q[li][j];
                    float n2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < CP; ++j) n2 = __builtin_fmaf(q[j], q[j], n2);
                    const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
                    float xn[CP];
                    unit_dir<CP>(q, c, xn);      // the reference's q / ||q|| (rqae/model.py:188): what the exact rescoring multiplies the rows by
                    {
                        unsigned dw[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) dw[i] = 0u;
#pragma unroll
                        for (int j = 0; j < CP; ++j) {
                            unsigned h, m, lo;
                            split3(q[j], h, m, lo);
                            dw[3 * j] = pack_hi16(h, m);
                            dw[3 * j + 1] = pack_hi16(h, lo);
                            dw[3 * j + 2] = dw[3 * j];
                        }
#pragma unroll
                        for (int pc = 0; pc < 4; ++pc) {
                            const int pos = (pc + 2 * wq) & 3;
                            *reinterpret_cast<u32x4 *>(xch + lane * 64 + pos * 16) = u32x4{dw[4 * pc], dw[4 * pc + 1], dw[4 * pc + 2], dw[4 * pc + 3]};
                        }
                    }
                    wave_sync_lds();
                    u32x4 B[4];
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        const int pos = (g + 2 * ((n >> 2) & 3)) & 3;
                        B[T] = *reinterpret_cast<const u32x4 *>(xch + (16 * T + n) * 64 + pos * 16);
                    }
                    // ---- scan: the image passes through the ring stage by stage; running maximum per (lane, token tile) over groups
                    // of 4 row tiles, as in gram_levels_kernel
                    float best[4];
                    int bgrp[4];
#pragma unroll
                    for (int T = 0; T < 4; ++T) { best[T] = -INFINITY; bgrp[T] = 0; }
                    for (int s = 0; s < nstage; ++s) {
                        // this wave's pieces of stage s have landed; younger: the pieces of stage s + 1 when it was issued (anything the
                        // chain itself loaded or stored since is older or only over-waits)
                        if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();      // everyone's pieces of stage s landed; everyone has left stage s - 1
                        if (s + 2 < nstage) issue(s + 2);
                        const char *sb = smem + (s % SL_NST) * (SL_STT * 1024) + lane * 16;
                        int gcount = (p.KTP - s * SL_STT) / 4;
                        gcount = gcount < SL_STT / 4 ? gcount : SL_STT / 4;
                        u32x4 an[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(sb + i * 1024);
                        for (int gi = 0; gi < gcount; ++gi) {
                            float mx[4];
                            u32x4 a[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) a[i] = an[i];
                            const int gn = gi + 1 < gcount ? gi + 1 : gi;          // next group's fragments while this one multiplies
#pragma unroll
                            for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(sb + (gn * 4 + i) * 1024);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f32x4 sc[4];
#pragma unroll
                                for (int T = 0; T < 4; ++T) sc[T] = mfma16(a[i], B[T], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                                for (int T = 0; T < 4; ++T) {
                                    const float m012 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(sc[T][0], sc[T][1]), sc[T][2]);
                                    if (i == 0) mx[T] = __builtin_elementwise_maximum(m012, sc[T][3]);
                                    else mx[T] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m012, sc[T][3]), mx[T]);
                                }
                            }
                            const int grp = s * (SL_STT / 4) + gi;
#pragma unroll
                            for (int T = 0; T < 4; ++T) {
                                const bool up = mx[T] > best[T];
                                best[T] = up ? mx[T] : best[T];
                                bgrp[T] = up ? grp : bgrp[T];
                            }
                        }
                    }
                    __builtin_amdgcn_s_barrier();          // every wave has left the last stages: the ring is free
                    issue(0);                              // the next level's (or tile's) first stages land behind the serial part
                    if (nstage > 1) issue(1);
                    // ---- the four lanes (g) of a token hand their (maximum, group) to the token's own lane
                    wave_sync_lds();
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        *reinterpret_cast<u32x2 *>(xch + ((16 * T + n) * 4 + g) * 8) = u32x2{u32_of(best[T]), (unsigned)bgrp[T]};
                    }
                    wave_sync_lds();
                    const u32x4 h0 = *reinterpret_cast<const u32x4 *>(xch + lane * 32);
                    const u32x4 h1 = *reinterpret_cast<const u32x4 *>(xch + lane * 32 + 16);
                    wave_sync_lds();
                    float bv = f32_of(h0[0]);
                    int bgr = (int)h0[1], bg = 0;
                    bool amb = false;
                    {
                        const float vv[3] = {f32_of(h0[2]), f32_of(h1[0]), f32_of(h1[2])};
                        const int gg[3] = {(int)h0[3], (int)h1[1], (int)h1[3]};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const bool gt = vv[k] > bv, eq = vv[k] == bv;
                            const bool take = gt || (eq && gg[k] < bgr);
                            amb = take ? false : (amb || (eq && gg[k] == bgr));
                            bv = take ? vv[k] : bv;
                            bg = take ? k + 1 : bg;
                            bgr = take ? gg[k] : bgr;
                        }
                    }
                    if (!okq) { bgr = 0; bg = 0; amb = false; }
                    // ---- the 16 rows behind that maximum, rescored in fp32 (k-ordered fma chain): first maximum wins
                    int brow = 64 * bgr + 4 * bg;
                    float bsc = -INFINITY;
                    auto rescore4 = [&](int r0) {
                        float f[4 * CP];
                        if (c == CP) {
                            const f32x4 *qp = reinterpret_cast<const f32x4 *>(cbf + (size_t)r0 * CP);      // 4 rows = CP aligned float4s
#pragma unroll
                            for (int w = 0; w < CP; ++w) {
                                const f32x4 v = qp[w];
                                f[4 * w] = v[0]; f[4 * w + 1] = v[1]; f[4 * w + 2] = v[2]; f[4 * w + 3] = v[3];
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int j = 0; j < CP; ++j) f[r * CP + j] = j < c ? cbf[(size_t)(r0 + r) * c + j] : 0.0f;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sc = 0.0f;
#pragma unroll
                            for (int j = 0; j < CP; ++j) sc = __builtin_fmaf(xn[j], f[r * CP + j], sc);
                            const bool up = (sc > bsc) && (r0 + r < p.K);
                            bsc = up ? sc : bsc;
                            brow = up ? r0 + r : brow;
                        }
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i) rescore4(16 * (4 * bgr + i) + 4 * bg);
                    if (__builtin_amdgcn_ballot_w64(amb) != 0) {
                        if (amb) {
                            brow = 64 * bgr;
                            bsc = -INFINITY;
#pragma unroll 1
                            for (int r0 = 64 * bgr; r0 < 64 * bgr + 64; r0 += 4) rescore4(r0);
                        }
                    }
                    if (!okq) brow = 0;
                    codes[li] = brow;
                    // ---- straight-through value (rqae/model.py:218-220) and its bf16 pieces for the reconstruction
                    float cvp[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) {
                        const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                        const float df = cv - q[j];
                        cvp[j] = q[j] + df;
                        cvh[li][j] = cvp[j];
                        unsigned h, m, lo;
                        split3(cvp[j], h, m, lo);
                        const int s0 = (li * CP + j) * 3;
                        img[s0] = pack_hi16(h, m);
                        img[s0 + 1] = pack_hi16(h, lo);
                        img[s0 + 2] = img[s0];
                    }
                    // ---- later levels of this group see the level just applied
#pragma unroll
                    for (int li2 = li + 1; li2 < GL; ++li2) {
                        const int l2 = lg * GL + li2;
                        if (l2 < p.levels) {
                            const float *gb = p.gneg + ((size_t)l * p.L + l2) * GB;
#pragma unroll
                            for (int j = 0; j < CP; ++j)
#pragma unroll
                                for (int i = 0; i < CP; ++i) Q[li2][j] = __builtin_fmaf(gb[j * CP + i], cvp[i], Q[li2][j]);
                        }
                    }
                    // ---- reconstruction operand rows: every k-step this level completed (all of them after the last level)
                    if (live) {
                        const int ks_lo = (3 * CP * li) / 16;
                        const int ks_hi = (l == p.levels - 1 || li == GL - 1) ? (3 * CP * (li + 1) + 15) / 16 : (3 * CP * (li + 1)) / 16;
#pragma unroll
                        for (int ks = ks_lo; ks < GS / 32; ++ks)
                            if (ks < ks_hi) {
                                unsigned *dst = p.cvimg + (((size_t)(lg * (GS / 32) + ks) * p.tok_pad + t) * 16);
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    *reinterpret_cast<u32x4 *>(dst + 4 * v) = u32x4{img[16 * ks + 4 * v], img[16 * ks + 4 * v + 1], img[16 * ks + 4 * v + 2], img[16 * ks + 4 * v + 3]};
                            }
                    }
                } else {
                    codes[li] = 0;
#pragma unroll
                    for (int j = 0; j < CP; ++j) cvh[li][j] = 0.0f;
                }
            });
            if (live && t < p.n_tokens) {
#pragma unroll
                for (int li = 0; li < GL; ++li)
                    if (lg * GL + li < p.levels) p.idx[t * p.levels + lg * GL + li] = (long long)codes[li];
            }
            // ---- projections of the later level groups see this group's levels
            if (MULTI && live)
                for (int l2 = (lg + 1) * GL; l2 < p.levels; ++l2) {
                    const int o0 = (l2 / GL) * GP + (l2 % GL) * CP;
                    float v[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) v[j] = p.P[((size_t)((o0 + j) >> 4) * p.tok_pad + t) * 16 + ((o0 + j) & 15)];
#pragma unroll
                    for (int li = 0; li < GL; ++li) {
                        const float *gb = p.gneg + ((size_t)(lg * GL + li) * p.L + l2) * GB;
#pragma unroll
                        for (int j = 0; j < CP; ++j)
#pragma unroll
                            for (int i = 0; i < CP; ++i) v[j] = __builtin_fmaf(gb[j * CP + i], cvh[li][i], v[j]);
                    }
#pragma unroll
                    for (int j = 0; j < CP; ++j) p.P[((size_t)((o0 + j) >> 4) * p.tok_pad + t) * 16 + ((o0 + j) & 15)] = v[j];
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the pieces issued after the last scan: nothing in flight at exit
}
