// gram_wide.hpp -- the matrix-core arm for wide codebook vectors: 9 <= c <= 13 (included by gram.hip inside its namespace).
//
// BASELINE.json configs[3] (C4: d 3584, c 13, K 8192, depth 16; rqae/model.py:187-193,208-224 with the reference's shape-agnostic
// loop).  The six bf16 x bf16 products per component that reproduce an fp32 x fp32 product fill 6 c = 78 k-slots: THREE k-steps
// of v_mfma_f32_16x16x32_bf16 per cosine contraction (NKC = 3), a 1.5 MB codebook image (K x 192 bytes) that streams through an
// LDS ring once per level, and reconstruction rows of 40 k-steps (4 level groups x 10).  Same arithmetic and tie rules as the
// other level chains (MFMA scores pick a block of 16 rows per token, the reference's fp32 fma chain on q / ||q|| rescans it, first
// maximum wins); what differs is the geometry:
//   gram_levels_wide_kernel   4 levels per level group (52 projection registers per token), one wave = 64 tokens, the image in
//                             stages of 8 row tiles x 3 k-steps (24 KiB) through a 3-slot ring shared by the workgroup's waves;
//                             straight-through values leave as fp32 planes [level][token][16] (the reconstruction splits them)
//   gram_recon_wide_kernel    token-stationary: a wave keeps the bf16 pieces of its 16 tokens for all 40 k-steps in registers (B
//                             operands), the W_out image streams through an LDS ring one element tile (40 KiB) per stage
// The projection is gram_project2_kernel (any number of output groups).  Tier-2 parity like the rest of the arm.

constexpr int W_STT = 8;        // row tiles per ring stage
constexpr int W_NST = 3;        // ring slots

template <int CP> struct WideGeom {
    static constexpr int GL = 4;                                   // levels per level group (gl_of)
    static constexpr int GP = ((GL * CP + 15) / 16) * 16;          // projection outputs per group, whole 16-wide tiles
    static constexpr int NKC = (6 * CP + 31) / 32;                 // k-steps of the cosine contraction
    static constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;      // reconstruction slots per group
    static constexpr int NKG = GS / 32;                            // reconstruction k-steps per group
    static constexpr int GB = (CP * CP + 3) & ~3;
    static constexpr int STAGE = W_STT * NKC * 1024;               // bytes per ring stage
    static constexpr int CVS = 16;                                 // floats per token and level in the value planes (c <= 13, padded)
};

__host__ __device__ inline size_t wide_levels_lds_bytes(int nkc, int nw) { return (size_t)W_NST * W_STT * nkc * 1024 + (size_t)nw * 4096; }

// p.cvimg is used as float planes [levels][tok_pad][16]
template <int CP, int NW>
__global__ __launch_bounds__(NW * 64) void gram_levels_wide_kernel(LevParams p)
{
    using G = WideGeom<CP>;
    constexpr int GL = G::GL, GP = G::GP, NKC = G::NKC, GB = G::GB, STAGE = G::STAGE, CVS = G::CVS;
    static_assert(NKC == 3, "three k-steps per cosine contraction");
    static_assert((W_STT * NKC) % NW == 0, "every wave issues the same number of pieces per stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int c = p.c;
    char *xch = smem + W_NST * STAGE + wave * 4096;
    const float *cbf = p.cb_f32;                       // global (L2): K x c fp32, + 64 zero rows
    float *cvf = reinterpret_cast<float *>(p.cvimg);
    const int nstage = (p.KTP + W_STT - 1) / W_STT;    // KTP is a multiple of 4; W_STT of 4 too: stages hold whole groups of 4 row tiles
    constexpr int PW = (W_STT * NKC) / NW;             // pieces per wave and stage
    const char *img = reinterpret_cast<const char *>(p.cb_img);
    const size_t img_bytes = (size_t)p.KTP * NKC * 1024;
    auto issue = [&](int s) {
        const int slot = s % W_NST;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave * PW + i;
            size_t off = ((size_t)s * W_STT * NKC + pc) * 1024;
            off = off < img_bytes ? off : img_bytes - 1024;      // pieces past the image repeat its last one: never read
            dma_piece(uniform_ptr(img + off), lds0 + slot * STAGE + pc * 1024, (unsigned)lane * 16u);
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
                    // ---- B operand rows of the wave's 64 tokens: 96 slots = 3 k-steps x 64 bytes per token, one k-step per trip
                    // through the exchange buffer (component j fills dwords 3j .. 3j+2: hi|mid, hi|lo, hi|mid)
                    u32x4 B[4][NKC];
                    {
                        unsigned dw[16 * NKC];
#pragma unroll
                        for (int i = 0; i < 16 * NKC; ++i) dw[i] = 0u;
#pragma unroll
                        for (int j = 0; j < CP; ++j) {
                            unsigned h, m, lo;
                            split3(q[j], h, m, lo);
                            dw[3 * j] = pack_hi16(h, m);
                            dw[3 * j + 1] = pack_hi16(h, lo);
                            dw[3 * j + 2] = dw[3 * j];
                        }
#pragma unroll
                        for (int ks = 0; ks < NKC; ++ks) {
#pragma unroll
                            for (int pc = 0; pc < 4; ++pc) {
                                const int pos = (pc + 2 * wq) & 3;   // bank-conflict-free placement of the four 16-byte pieces
                                *reinterpret_cast<u32x4 *>(xch + lane * 64 + pos * 16) =
                                    u32x4{dw[16 * ks + 4 * pc], dw[16 * ks + 4 * pc + 1], dw[16 * ks + 4 * pc + 2], dw[16 * ks + 4 * pc + 3]};
                            }
                            wave_sync_lds();
#pragma unroll
                            for (int T = 0; T < 4; ++T) {
                                const int pos = (g + 2 * ((n >> 2) & 3)) & 3;
                                B[T][ks] = *reinterpret_cast<const u32x4 *>(xch + (16 * T + n) * 64 + pos * 16);
                            }
                            wave_sync_lds();
                        }
                    }
                    // ---- scan: the image passes through the ring stage by stage; running maximum per (lane, token tile) over groups
                    // of 4 row tiles (16 rows per lane)
                    float best[4];
                    int bgrp[4];
#pragma unroll
                    for (int T = 0; T < 4; ++T) { best[T] = -INFINITY; bgrp[T] = 0; }
                    for (int s = 0; s < nstage; ++s) {
                        // this wave's pieces of stage s have landed; younger: the pieces of stage s + 1 when it was issued
                        if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();      // everyone's pieces of stage s landed; everyone has left stage s - 1
                        if (s + 2 < nstage) issue(s + 2);
                        const char *sb = smem + (s % W_NST) * STAGE + lane * 16;
                        int gcount = (p.KTP - s * W_STT) / 4;
                        gcount = gcount < W_STT / 4 ? gcount : W_STT / 4;
                        for (int gi = 0; gi < gcount; ++gi) {
                            float mx[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                u32x4 a[NKC];
#pragma unroll
                                for (int ks = 0; ks < NKC; ++ks) a[ks] = *reinterpret_cast<const u32x4 *>(sb + ((gi * 4 + i) * NKC + ks) * 1024);
                                // k-step outermost: four independent accumulators (one per token tile) between dependent instructions
                                f32x4 sc[4];
#pragma unroll
                                for (int T = 0; T < 4; ++T) sc[T] = mfma16(a[0], B[T][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                                for (int ks = 1; ks < NKC; ++ks)
#pragma unroll
                                    for (int T = 0; T < 4; ++T) sc[T] = mfma16(a[ks], B[T][ks], sc[T]);
#pragma unroll
                                for (int T = 0; T < 4; ++T) {
                                    const float m012 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(sc[T][0], sc[T][1]), sc[T][2]);
                                    if (i == 0) mx[T] = __builtin_elementwise_maximum(m012, sc[T][3]);
                                    else mx[T] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m012, sc[T][3]), mx[T]);
                                }
                            }
                            const int grp = s * (W_STT / 4) + gi;
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
                    // ---- the 16 rows behind that maximum, rescored in fp32 on the reference's q / ||q|| (k-ordered fma chain): first
                    // maximum wins
                    float xn[CP];
                    unit_dir<CP>(q, c, xn);
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
#pragma unroll 1
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
                    // ---- straight-through value (rqae/model.py:218-220), as fp32 for the reconstruction
                    float cvp[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) {
                        const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                        const float df = cv - q[j];
                        cvp[j] = q[j] + df;
                        cvh[li][j] = cvp[j];
                    }
                    if (live) {
                        float *dst = cvf + ((size_t)l * p.tok_pad + t) * CVS;
#pragma unroll
                        for (int v = 0; v < CVS / 4; ++v)
                            *reinterpret_cast<f32x4 *>(dst + 4 * v) = f32x4{4 * v < CP ? cvp[4 * v < CP ? 4 * v : 0] : 0.0f,
                                                                             4 * v + 1 < CP ? cvp[4 * v + 1 < CP ? 4 * v + 1 : 0] : 0.0f,
                                                                             4 * v + 2 < CP ? cvp[4 * v + 2 < CP ? 4 * v + 2 : 0] : 0.0f,
                                                                             4 * v + 3 < CP ? cvp[4 * v + 3 < CP ? 4 * v + 3 : 0] : 0.0f};
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
            if (live)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stages issued for a level that does not come
}

// ------------------------------------------------------------------------------------------------ reconstruction, wide
// out[tok][e] = bias[e] + sum over (level, component, term) slots of W_out piece x cv piece (rqae/model.py:221,224 summed as the
// decode sum).  One wave = 16 tokens (one MFMA column tile); the pieces of its tokens for ALL k-steps stay in registers as B
// operands (NLG x NKG k-steps x 4 registers: 160 at C4), built group by group from the fp32 value planes through the wave's LDS
// buffer; the W_out image streams through a 3-slot LDS ring, one element tile (all k-steps: 40 KiB at C4) per stage, shared by the
// workgroup's 8 waves.  D layout: lane (g, n) holds elements 4g..4g+3 of token n -- one 16-byte store per lane, 64 contiguous
// bytes per token row and element tile (two consecutive stages complete a 128-byte line).
struct RecWideParams {
    const uint16_t *wout_img;    // [NET][NKSC][512]
    const float *bias;           // bias_prefix[levels-1][d]
    const float *cvf;            // [levels][tok_pad][16]
    float *out;                  // [n_tokens][d]
    long long n_tokens, tok_pad;
    int d, NKSC, NET, levels;
};

template <int CP, int NLG>
__host__ __device__ constexpr size_t wide_recon_lds_bytes() { return (size_t)W_NST * NLG * WideGeom<CP>::NKG * 1024 + 8 * 16 * 208; }

template <int CP, int NLG>
__global__ __launch_bounds__(512) void gram_recon_wide_kernel(RecWideParams p)
{
    using G = WideGeom<CP>;
    constexpr int GL = G::GL, NKG = G::NKG, NKS = NLG * NKG, CVS = G::CVS, NW = 8;
    constexpr int STAGE = NKS * 1024;
    constexpr int XROW = 208;                         // bytes per token in the exchange buffer: 3 k-steps (192) + 16: conflict-free rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    char *xb = smem + W_NST * STAGE + wave * (16 * XROW);
    constexpr int PW = (NKS + NW - 1) / NW;           // pieces per wave and stage (waves past the last piece repeat it)
    auto issue = [&](int et_in, int slot) {
        const int et = et_in < p.NET ? et_in : p.NET - 1;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int q = wave + i * NW;
            q = q < NKS ? q : NKS - 1;
            dma_piece(uniform_ptr(p.wout_img + ((size_t)et * p.NKSC + q) * 512), lds0 + slot * STAGE + q * 1024, (unsigned)lane * 16u);
        }
    };
    const long long ntile = (p.n_tokens + NW * 16 - 1) / (NW * 16);
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long tw0 = tile * (NW * 16) + wave * 16;
        long long tok = tw0 + n;                                  // lanes 16g + n, g = 0..3, all work on token n of the wave
        const bool tok_ok = tok < p.n_tokens;
        tok = tok_ok ? tok : p.n_tokens - 1;
        issue(0, 0);
        issue(1, 1);
        // ---- B operands: slot of (level li of the group, component j, term) = (li CP + j) 6 + term, dword (li CP + j) 3 + {0: hi|mid,
        // 1: hi|lo, 2: hi|mid}; lane group g builds the dwords of level li = g of the group (GL = 4 levels: one per lane group)
        u32x4 cvB[NKS];
        static_assert(GL == 4, "one level of a group per 16-lane group");
#pragma unroll
        for (int lg = 0; lg < NLG; ++lg) {
            const int l = lg * GL + g;
            float cv[CP];
            {
                const float *src = p.cvf + ((size_t)(l < p.levels ? l : 0) * p.tok_pad + tok) * CVS;
#pragma unroll
                for (int v = 0; v < (CP + 3) / 4; ++v) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(src + 4 * v);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * v + e < CP) cv[4 * v + e] = l < p.levels ? w[e] : 0.0f;
                }
            }
            // level g's 3 CP dwords are dwords [3 CP g, 3 CP (g + 1)) of the token's group row (10 k-steps = 160 dwords at CP 13); the
            // row passes through LDS three k-steps (48 dwords) at a time.  The pieces are cut again in every trip (a few VALU) rather
            // than kept: next to the 160 registers of cvB there is no room for them.
            constexpr int NTRIP = (NKG + 2) / 3;
            const int dbase = 3 * CP * g;                     // this lane's first dword in the group row
#pragma unroll
            for (int trip = 0; trip < NTRIP; ++trip) {
                unsigned *rowp = reinterpret_cast<unsigned *>(xb + n * XROW);
                const int w0 = dbase - 48 * trip;             // window position of this lane's dword 0
#pragma unroll
                for (int j = 0; j < CP; ++j) {
                    unsigned h, m, lo;
                    split3(cv[j], h, m, lo);
                    const unsigned d0 = pack_hi16(h, m), d1 = pack_hi16(h, lo);
                    const int q0 = w0 + 3 * j;
                    if (q0 >= 0 && q0 < 48) rowp[q0] = d0;
                    if (q0 + 1 >= 0 && q0 + 1 < 48) rowp[q0 + 1] = d1;
                    if (q0 + 2 >= 0 && q0 + 2 < 48) rowp[q0 + 2] = d0;
                }
                if (trip == NTRIP - 1) {
                    // the pad dwords past the last level's (GS - 6 CP GL slots) are zeros: stale bytes never reach an MFMA
                    for (int dpos = 3 * CP * GL - 48 * trip + g; dpos < 48; dpos += 4)
                        if (dpos >= 0) rowp[dpos] = 0u;
                }
                wave_sync_lds();
#pragma unroll
                for (int kl = 0; kl < 3; ++kl)
                    if (3 * trip + kl < NKG) cvB[lg * NKG + 3 * trip + kl] = *reinterpret_cast<const u32x4 *>(xb + n * XROW + kl * 64 + 16 * g);
                wave_sync_lds();
            }
        }
        // ---- element tiles: stage e = all k-steps of element tile e
        int st = 0;
        for (int e = 0; e < p.NET; ++e) {
            // this wave's pieces of stage e have landed; younger: the pieces of stage e + 1 (stores never wait here: they are older)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(e + 2, st == 0 ? 2 : st - 1);        // slot (e + 2) % 3 was read last in stage e - 1
            const char *wb = smem + st * STAGE + lane * 16;
            f32x4 acc = *reinterpret_cast<const f32x4 *>(p.bias + 16 * e + 4 * g);
            f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            // four fragments at a time: left alone hipcc hoists all NKS fragment reads above the first MFMA (160 more registers)
#pragma unroll
            for (int k0 = 0; k0 < NKS; k0 += 4) {
                u32x4 fr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k0 + i < NKS) fr[i] = *reinterpret_cast<const u32x4 *>(wb + (k0 + i) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k0 + i < NKS) {
                        if (i & 1) acc2 = mfma16(fr[i], cvB[k0 + i], acc2);
                        else acc = mfma16(fr[i], cvB[k0 + i], acc);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = f32x4{acc[0] + acc2[0], acc[1] + acc2[1], acc[2] + acc2[2], acc[3] + acc2[3]};
            if (tok_ok) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(p.out + (size_t)tok * p.d + 16 * e + 4 * g));
            st = st == 2 ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated stages past the last element tile
        __builtin_amdgcn_s_barrier();                       // the ring is free for the next tile
    }
}
