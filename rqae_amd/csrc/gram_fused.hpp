// gram_fused.hpp -- the matrix-core arm as ONE persistent kernel (included by gram.hip inside its anonymous namespace).
//
// The three-kernel form (project, levels, reconstruct) moves 0.37 GB of intermediates through HBM per 262144 tokens and
// runs a read-bound, a compute-bound and a write-bound phase one after the other on the whole chip; HBM is what limits
// the two memory phases (profiles/r02_rw_overlap.txt: one CU alone reads 58 GB/s and writes 128 GB/s, 256 CUs together
// 24 GB/s each), so a compute phase that runs while OTHER workgroups stream costs nothing.  Here a workgroup takes a tile
// of 256 tokens through all three phases -- projections and straight-through values never leave the CU -- and the
// workgroups of the chip drift apart, so reads, matrix-core work and writes overlap across CUs.
//
//   phase A  P = x W_in'   : x streamed HBM -> registers (asm loads, 3-deep ring), W_in image L2 -> LDS ring by LDS-DMA,
//                            counted vmcnt + one raw barrier per chunk (as gram_project2_kernel)
//   phase B  level chain   : one token per lane in c-space, cosine contraction on MFMA against the LDS-resident codebook
//                            image, exact fp32 rescoring of the 16 best-candidate rows (as gram_levels_kernel)
//   phase C  out = W_out cv: cv pieces in registers (B operand), W_out image L2 -> LDS ring by LDS-DMA, 128-byte store rows
//
// One workgroup = 8 waves x 32 tokens.  LDS: codebook image + fp32 codebook + bias sums stay resident; one region is the
// W_in ring in phase A, the per-wave exchange buffers in phase B and the W_out ring in phase C.
// Serves one level group (levels <= 8) and a shared codebook; other shapes take the three-kernel form.

struct FusedParams {
    const void *x;
    const uint16_t *win_img;   // [KSA][NOG][3][512]
    const float *kconst;       // [NOG * 16]
    const float *gneg;         // [L][L][GB]
    const uint16_t *cb_img;    // [KTP][512]
    const float *cb_f32;       // [K + 64][c]
    const uint16_t *wout_img;  // [NET][NKSC][512]
    const float *bias;         // [d]: bias_prefix[levels - 1]
    float *out;                // [n_tokens][d], or NULL: codes only
    long long *idx;            // [n_tokens][levels]
    float *dump;               // 32 KiB: stores of rows / elements that do not exist
    unsigned long long *stamps; // tools build (-DRQAE_FUSED_TIMELINE) only: s_memtime at the phase boundaries, [wg < 8][wave][tile < 4][8]
    long long n_tokens;
    int d, K, KTP, c, levels, L, KSA, NOG, NET, NKSC;
    unsigned off_cbf, off_bias, off_union;   // LDS byte offsets (the codebook image is at 0)
};

constexpr int F_NW = 8;                 // waves per workgroup
constexpr int F_TILE = F_NW * 32;       // tokens per tile
constexpr int F_XCHW = 5120;            // exchange bytes per wave
constexpr int F_ETS = 4;                // element tiles per W_out stage

template <int CP> struct FusedGeom {
    static constexpr int GL = 8;
    static constexpr int GP = ((GL * CP + 15) / 16) * 16;
    static constexpr int NG = GP / 16;                       // projection groups
    static constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;
    static constexpr int NKS = GS / 32;                      // k-steps of the reconstruction
    static constexpr int GB = (CP * CP + 3) & ~3;
    static constexpr int KC = 2;                             // k-steps per phase-A chunk
    static constexpr int STAGE_A = KC * NG * 3 * 1024;
    static constexpr int STAGE_C = F_ETS * NKS * 1024;
    static constexpr int XCH_ALL = F_NW * F_XCHW + GL * GL * GB * 4;     // exchange buffers + Gram blocks (phase B)
    static constexpr int RINGS = 3 * STAGE_A > 2 * STAGE_C ? 3 * STAGE_A : 2 * STAGE_C;
    static constexpr int UNION = RINGS > XCH_ALL ? RINGS : XCH_ALL;
};

__host__ __device__ inline size_t fused_lds_bytes(int d, int c, int K)
{
    const int CP = cp_of(c);
    const int KTP = (((K + 15) / 16 + 3) / 4) * 4;
    const size_t img = (size_t)KTP * 1024, cbf = (((size_t)K + 64) * c * 4 + 15) & ~(size_t)15, bias = ((size_t)d * 4 + 15) & ~(size_t)15;
    const size_t un = CP == 4 ? FusedGeom<4>::UNION : FusedGeom<5>::UNION;
    return img + cbf + bias + un;
}

// Tile schedule.  Workgroup b of G takes tiles b, b + G, b + 2G, ...  The first tile of every odd workgroup is a HALF tile:
// all workgroups start together, and with equal tiles they would stay in step -- every CU reading, then every CU on the
// matrix cores, then every CU writing, HBM idle a third of the time.  A half tile puts the odd workgroups half a period
// ahead for the rest of the launch, so the phases of neighbouring CUs overlap.  ftile_off(j) = first token of tile j.
__host__ __device__ inline long long ftile_off(long long j, long long G)
{
    return j < G ? 192 * j + ((j & 1) ? 64 : 0) : 192 * G + 256 * (j - G);
}
__host__ inline long long ftile_count(long long n_tokens, long long G)
{
    if (n_tokens <= 192 * G) {
        long long j = n_tokens / 192;                       // off(j) <= 192 j + 64
        while (j > 0 && ftile_off(j, G) >= n_tokens) --j;
        while (ftile_off(j, G) < n_tokens) ++j;
        return j;
    }
    return G + (n_tokens - 192 * G + 255) / 256;
}

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef RQAE_FUSED_TIMELINE
#define F_STAMP(i)                                                                                                         \
    do {                                                                                                                   \
        if (blockIdx.x < 8 && tord < 4 && lane == 0)                                                                       \
            p.stamps[(((size_t)blockIdx.x * F_NW + wave) * 4 + tord) * 8 + (i)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
// per-chunk stamps of phase A (tile ordinal 1, workgroup 0): [wave][chunk < 36][4] after the 2048 phase stamps
#define F_STAMPA(s_, i)                                                                                                   \
    do {                                                                                                                   \
        if (blockIdx.x == 0 && tord == 1 && lane == 0 && (s_) < 36)                                                        \
            p.stamps[2048 + ((size_t)wave * 36 + (s_)) * 4 + (i)] = __builtin_amdgcn_s_memtime();                          \
    } while (0)
#else
#define F_STAMP(i) do { } while (0)
#define F_STAMPA(s_, i) do { } while (0)
#endif

// FULL: levels == 8 (no per-level depth checks); CEQ: c == CP (codebook rows are CP floats apart: aligned float4 row quads)
template <int XT, int CP, bool FULL, bool CEQ>
__global__ __launch_bounds__(F_NW * 64) void gram_fused_kernel(FusedParams p)
{
    using G = FusedGeom<CP>;
    constexpr int GL = G::GL, NG = G::NG, NKS = G::NKS, GB = G::GB, KC = G::KC, STAGE_A = G::STAGE_A, STAGE_C = G::STAGE_C;
    constexpr int NLD = XRaw<XT>::NLD, ESZ = XRaw<XT>::ESZ;
    constexpr int FPK = NG * 3;                    // W_in fragments per k-step
    constexpr int XLOADS = KC * A_TT * NLD;        // x loads per chunk and wave: 4 (2-byte x) or 8 (fp32 x)
    static_assert(A_TT == 2, "32 tokens per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int tl = lane & 31;                      // c-space: lane l works on token (l & 31) of the wave (upper half mirrors)
    const int c = CEQ ? CP : p.c;
    const int levels = FULL ? GL : p.levels;
    char *cbl = smem;
    const float *cbf = reinterpret_cast<const float *>(smem + p.off_cbf);
    const float *sbias = reinterpret_cast<const float *>(smem + p.off_bias);
    char *un = smem + p.off_union;
    char *xch = un + wave * F_XCHW;
    float *sgram = reinterpret_cast<float *>(un + F_NW * F_XCHW);     // phase B only: the level group's Gram blocks [li][li2][GB]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    const size_t rowb = (size_t)p.d * ESZ;

    // ---- resident tables: codebook image (LDS-DMA), fp32 codebook rows, bias sums
    for (int q = wave; q < p.KTP; q += F_NW) dma_piece(uniform_ptr(p.cb_img + (size_t)q * 512), lds0 + q * 1024, lane16);
    for (int i = tid; i < (p.K + 64) * c; i += F_NW * 64) const_cast<float *>(cbf)[i] = p.cb_f32[i];
    for (int i = tid; i < p.d; i += F_NW * 64) const_cast<float *>(sbias)[i] = p.bias[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nchunks = p.KSA / KC;
    const int wq = (tl >> 2) & 3;

    int tord = 0;
    for (long long tile = blockIdx.x; ftile_off(tile, gridDim.x) < p.n_tokens; tile += gridDim.x, ++tord) {
        const long long tile0 = ftile_off(tile, gridDim.x);
        long long tile_end = ftile_off(tile + 1, gridDim.x);          // rows past it belong to another tile: never stored
        tile_end = tile_end < p.n_tokens ? tile_end : p.n_tokens;
        F_STAMP(0);
        const long long tw0 = tile0 + (long long)wave * 32;           // first token of this wave
        // =========================================================================================== phase A
        f32x4 acc[NG][A_TT];
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int T = 0; T < A_TT; ++T) acc[u][T] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // per iteration s the wave issues G(s) = { x fragments of chunk s+3 (registers), its W_in pieces of chunk s+2 (LDS-DMA) }
            // in that order, then multiplies chunk s.  Chunk s needs x(s) from G(s-3) and W(s) from G(s-2): everything older
            // than G(s-1) must have landed, G(s-1) itself (XLOADS loads + this wave's pieces) may stay in flight.
            constexpr int NPB = (KC * FPK) / F_NW, NPX = (KC * FPK) % F_NW;    // pieces per wave: NPB, one more on waves < NPX
            u32x4 xr[4][KC][A_TT][NLD];
            unsigned voff[A_TT];
#pragma unroll
            for (int T = 0; T < A_TT; ++T) {
                long long r = tw0 + 16 * T + n;
                r = r < tile_end ? r : tile_end - 1;          // rows past the tile re-read its last row
                voff[T] = (unsigned)((r - tile0) * (long long)rowb) + (unsigned)(g * 8 * ESZ);
            }
            const char *xtile = static_cast<const char *>(p.x) + (size_t)tile0 * rowb;
            // chunks past the end repeat the last one so that the operation counts stay static
            auto issue_x = [&](int cc_in, auto slot_c, u32x4 (&xr)[4][KC][A_TT][NLD], const unsigned (&voff)[A_TT]) {
                constexpr int SLOT = decltype(slot_c)::value;
#ifdef FQ_NOX   /* tools build: every x load re-reads chunk 0 of the tile (same operation count, cache-hot lines) */
                const int cc = 0 * cc_in;
#else
                const int cc = cc_in < nchunks ? cc_in : nchunks - 1;
#endif
                const void *xb = uniform_ptr(xtile + (size_t)cc * KC * 32 * ESZ);
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int T = 0; T < A_TT; ++T)
#pragma unroll
                        for (int h = 0; h < NLD; ++h)
                            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3"
                                         : "=v"(xr[SLOT][k][T][h])
                                         : "v"(voff[T]), "s"(xb), "i"(k * 32 * ESZ + 16 * h)
                                         : "memory");
            };
            auto issue_w = [&](int cc_in, int stage) {
                const int cc = cc_in < nchunks ? cc_in : nchunks - 1;
#ifdef FQ_NOW   /* tools build: every W piece re-reads piece 0 (same operation count, L2-hot line) */
                for (int q = wave; q < KC * FPK; q += F_NW) dma_piece(uniform_ptr(p.win_img), lds0 + p.off_union + stage * STAGE_A + q * 1024, lane16);
                return;
#endif
                for (int q = wave; q < KC * FPK; q += F_NW) {
                    const int k = q / FPK, f = q - k * FPK;
                    const int ks = cc * KC + k;          // < KSA: the host sends only d % (32 KC) == 0 here
                    dma_piece(uniform_ptr(p.win_img + (((size_t)ks * p.NOG) * 3 + f) * 512), lds0 + p.off_union + stage * STAGE_A + (k * FPK + f) * 1024, lane16);
                }
            };
            issue_x(0, std::integral_constant<int, 0>{}, xr, voff);                       // G(-3)
            issue_x(1, std::integral_constant<int, 1>{}, xr, voff); issue_w(0, 0);        // G(-2)
            issue_x(2, std::integral_constant<int, 2>{}, xr, voff); issue_w(1, 1);        // G(-1)
            int wst = 0;                                                                  // W stage of chunk s = s % 3
            for (int s0 = 0; s0 < nchunks; s0 += 4) {
                static_for<0, 4>([&](auto ph_c) {
                    constexpr int PH = decltype(ph_c)::value;
                    const int s = s0 + PH;
                    // operand-free, per-wave count first; then ONE statement that names the registers of slot PH, so that no use
                    // or copy of them can be placed above the waits
                    F_STAMPA(s, 0);
                    if (wave < NPX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XLOADS + NPB + 1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XLOADS + NPB) : "memory");
                    {
                        u32x4 *r = &xr[PH][0][0][0];
                        if constexpr (XLOADS == 4)
                            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(XLOADS + NPB + 1) : "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(%8)"
                                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                                         : "n"(XLOADS + NPB + 1)
                                         : "memory");
                    }
                    __builtin_amdgcn_s_barrier();        // every wave's pieces of W(s) landed; every wave has left chunk s-1
                    F_STAMPA(s, 2);
                    // x(s+3) -> slot (s+3) % 4 (read last in chunk s-1 by this wave); W(s+2) -> stage (s+2) % 3 (read last in chunk s-1)
                    issue_x(s + 3, std::integral_constant<int, (PH + 3) % 4>{}, xr, voff);
                    F_STAMPA(s, 1);      // (re-used slot: after the x loads)
                    issue_w(s + 2, wst == 0 ? 2 : wst - 1);
                    F_STAMPA(s, 3);
                    if (s < nchunks) {
                        const char *wb = un + wst * STAGE_A + lane * 16;
                        constexpr int NF = KC * FPK, MF = A_TT * XFrag<XT>::NP, PF = 6;
                        u32x4 xp[KC][A_TT][3];
#pragma unroll
                        for (int k = 0; k < KC; ++k)
#pragma unroll
                            for (int T = 0; T < A_TT; ++T) xparts<XT>(xr[PH][k][T], xp[k][T]);
                        u32x4 fr[NF];
#pragma unroll
                        for (int i = 0; i < NF; ++i) fr[i] = *reinterpret_cast<const u32x4 *>(wb + i * 1024);
#pragma unroll
                        for (int i = 0; i < NF; ++i) {
                            const int k = i / FPK, u = (i % FPK) / 3;
#pragma unroll
                            for (int T = 0; T < A_TT; ++T)
#pragma unroll
                                for (int xq = 0; xq < XFrag<XT>::NP; ++xq) acc[u][T] = mfma16(fr[i], xp[k][T][xq], acc[u][T]);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                        for (int i = 0; i < NF - PF; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, PF * MF, 0);
                    }
                    wst = wst == 2 ? 0 : wst + 1;
                });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated groups past the last chunk
        }
        F_STAMP(1);
        __builtin_amdgcn_s_barrier();    // the union region turns from the W_in ring into the exchange buffers
        F_STAMP(2);
        // the Gram blocks of the level group (rewritten every tile: the rings of phases A and C run over them)
        for (int i = tid; i < GL * GL * GB; i += F_NW * 64) {
            const int blk = i / GB, li = blk / GL, li2 = blk - li * GL;
            sgram[i] = (li < p.L && li2 < p.L) ? p.gneg[((size_t)li * p.L + li2) * GB + (i - blk * GB)] : 0.0f;
        }

        // ---- D layout (lane (g, n): outputs 4g..4g+3 of token 16T+n) -> one token per lane, through LDS
        float Q[GL][CP];
        {
            float buf[NG * 16];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
#pragma unroll
                for (int T = 0; T < A_TT; ++T) *reinterpret_cast<f32x4 *>(xch + (16 * T + n) * 80 + 16 * g) = acc[u][T];
                wave_sync_lds();
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(xch + tl * 80 + 16 * v);
                    const f32x4 kc = *reinterpret_cast<const f32x4 *>(p.kconst + 16 * u + 4 * v);
                    buf[16 * u + 4 * v + 0] = w[0] + kc[0];
                    buf[16 * u + 4 * v + 1] = w[1] + kc[1];
                    buf[16 * u + 4 * v + 2] = w[2] + kc[2];
                    buf[16 * u + 4 * v + 3] = w[3] + kc[3];
                }
                wave_sync_lds();
            }
#pragma unroll
            for (int li = 0; li < GL; ++li)
#pragma unroll
                for (int j = 0; j < CP; ++j) Q[li][j] = buf[li * CP + j];
        }

        __syncthreads();      // Gram blocks staged
        F_STAMP(3);
        // =========================================================================================== phase B
        float cvh[GL][CP];
        int codes[GL];
        static_for<0, GL>([&](auto li_c) {
            constexpr int li = decltype(li_c)::value;
            if (FULL || li < levels) {
                float q[CP];
#pragma unroll
                for (int j = 0; j < CP; ++j) q[j] = Q[li][j];
                float n2 = 0.0f;
#pragma unroll
                for (int j = 0; j < CP; ++j) n2 = __builtin_fmaf(q[j], q[j], n2);
                const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
                // ---- B operand rows of the wave's 32 tokens: 32 slots = 64 bytes per token
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
                        const int pos = (pc + 2 * wq) & 3;   // bank-conflict-free placement of the four 16-byte pieces
                        *reinterpret_cast<u32x4 *>(xch + tl * 64 + pos * 16) = u32x4{dw[4 * pc], dw[4 * pc + 1], dw[4 * pc + 2], dw[4 * pc + 3]};
                    }
                }
                wave_sync_lds();
                u32x4 B[A_TT];
#pragma unroll
                for (int T = 0; T < A_TT; ++T) {
                    const int pos = (g + 2 * ((n >> 2) & 3)) & 3;
                    B[T] = *reinterpret_cast<const u32x4 *>(xch + (16 * T + n) * 64 + pos * 16);
                }
                // ---- scan: running maximum per (lane, token tile) over groups of 4 row tiles.  Rows past K repeat row 0 in
                // the image (gram_pack_cb_kernel), so no masking: a repeat never beats the row it repeats (lowest group wins ties)
                float best[A_TT];
                int bgrp[A_TT];
#pragma unroll
                for (int T = 0; T < A_TT; ++T) { best[T] = -INFINITY; bgrp[T] = 0; }
                const int ngrp = p.KTP / 4;
                u32x4 an[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(cbl + i * 1024 + lane * 16);
                for (int grp = 0; grp < ngrp; ++grp) {
                    u32x4 a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = an[i];
                    const int gn = grp + 1 < ngrp ? grp + 1 : grp;
#pragma unroll
                    for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(cbl + (gn * 4 + i) * 1024 + lane * 16);
                    f32x4 s[4][A_TT];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int T = 0; T < A_TT; ++T) s[i][T] = mfma16(a[i], B[T], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int T = 0; T < A_TT; ++T) {
                        // IEEE-754-2019 maximum (v_maximum3_f32): a NaN score poisons the maximum, which then never beats `best`;
                        // the level takes index 0 through okq
                        float mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s[0][T][0], s[0][T][1]), s[0][T][2]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[0][T][3]), s[1][T][0]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[1][T][1]), s[1][T][2]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[1][T][3]), s[2][T][0]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[2][T][1]), s[2][T][2]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[2][T][3]), s[3][T][0]);
                        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, s[3][T][1]), s[3][T][2]);
                        mx = __builtin_elementwise_maximum(mx, s[3][T][3]);
                        const bool up = mx > best[T];
                        best[T] = up ? mx : best[T];
                        bgrp[T] = up ? grp : bgrp[T];
                    }
                }
                // ---- the four lanes (g) of a token hand their (maximum, group) to the token's own lane
                wave_sync_lds();
#pragma unroll
                for (int T = 0; T < A_TT; ++T) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    *reinterpret_cast<u32x2 *>(xch + ((16 * T + n) * 4 + g) * 8) = u32x2{u32_of(best[T]), (unsigned)bgrp[T]};
                }
                wave_sync_lds();
                const u32x4 h0 = *reinterpret_cast<const u32x4 *>(xch + tl * 32);
                const u32x4 h1 = *reinterpret_cast<const u32x4 *>(xch + tl * 32 + 16);
                wave_sync_lds();      // the next level (or the hand-over to phase C) rewrites the exchange buffer
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
                    if (CEQ) {
                        const f32x4 *qp = reinterpret_cast<const f32x4 *>(cbf + (size_t)r0 * CP);   // a row quad is CP aligned float4s
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
                        for (int j = 0; j < CP; ++j) sc = __builtin_fmaf(f[r * CP + j], q[j], sc);
                        const bool up = (sc > bsc) && (r0 + r < p.K);
                        bsc = up ? sc : bsc;
                        brow = up ? r0 + r : brow;
                    }
                };
#pragma unroll
                for (int i = 0; i < 4; ++i) rescore4(16 * (4 * bgr + i) + 4 * bg);
                if (amb) {   // equal maxima in one 64-row block on several lanes (duplicate rows): rescan the block in row order
                    brow = 64 * bgr;
                    bsc = -INFINITY;
                    for (int r0 = 64 * bgr; r0 < 64 * bgr + 64; r0 += 4) rescore4(r0);
                }
                if (!okq) brow = 0;
                codes[li] = brow;
                // ---- straight-through value (rqae/model.py:218-220)
#pragma unroll
                for (int j = 0; j < CP; ++j) {
                    const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                    const float df = cv - q[j];
                    cvh[li][j] = q[j] + df;
                }
                // ---- later levels see the level just applied
#pragma unroll
                for (int li2 = li + 1; li2 < GL; ++li2) {
                    if (FULL || li2 < levels) {
                        const float *gb = sgram + (li * GL + li2) * GB;        // broadcast reads
                        float gbr[GB];
#pragma unroll
                        for (int w = 0; w < GB / 4; ++w) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(gb + 4 * w);
                            gbr[4 * w] = v[0]; gbr[4 * w + 1] = v[1]; gbr[4 * w + 2] = v[2]; gbr[4 * w + 3] = v[3];
                        }
#pragma unroll
                        for (int j = 0; j < CP; ++j)
#pragma unroll
                            for (int i = 0; i < CP; ++i) Q[li2][j] = __builtin_fmaf(gbr[j * CP + i], cvh[li][i], Q[li2][j]);
                    }
                }
            } else {
                codes[li] = 0;
#pragma unroll
                for (int j = 0; j < CP; ++j) cvh[li][j] = 0.0f;
            }
        });
        {
            const long long t = tw0 + tl;
            if (lane < 32 && t < tile_end) {
#pragma unroll
                for (int li = 0; li < GL; ++li)
                    if (FULL || li < levels) p.idx[t * levels + li] = (long long)codes[li];
            }
        }
        F_STAMP(4);
        if (p.out == nullptr) {
            __builtin_amdgcn_s_barrier();     // the union region turns back into the W_in ring
            continue;
        }

        // =========================================================================================== phase C
        // ---- cv pieces of the wave's tokens as B operands: slot of (level li, component j, term) = (li CP + j) 6 + term,
        // dword (li CP + j) 3 + {0: hi|mid, 1: hi|lo, 2: hi|mid}; two k-steps (32 dwords per token) per trip through LDS
        u32x4 cvB[NKS][A_TT];
        static_for<0, (NKS + 1) / 2>([&](auto r_c) {
            constexpr int R = decltype(r_c)::value;
            unsigned dwr[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int di = 32 * R + i, ci = di / 3, part = di % 3, li = ci / CP, j = ci % CP;
                unsigned v = 0u;
                if (ci < GL * CP) {
                    unsigned h, m, lo;
                    split3(cvh[li][j], h, m, lo);
                    v = part == 1 ? pack_hi16(h, lo) : pack_hi16(h, m);
                }
                dwr[i] = v;
            }
#pragma unroll
            for (int v = 0; v < 8; ++v)
                *reinterpret_cast<u32x4 *>(xch + tl * 144 + 16 * v) = u32x4{dwr[4 * v], dwr[4 * v + 1], dwr[4 * v + 2], dwr[4 * v + 3]};
            wave_sync_lds();
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
                if (2 * R + kl < NKS) {
#pragma unroll
                    for (int T = 0; T < A_TT; ++T) cvB[2 * R + kl][T] = *reinterpret_cast<const u32x4 *>(xch + (16 * T + n) * 144 + kl * 64 + 16 * g);
                }
            wave_sync_lds();
        });
        F_STAMP(5);
        __builtin_amdgcn_s_barrier();     // the union region turns from the exchange buffers into the W_out ring
        F_STAMP(6);

        {
            const int nstg = (p.NET + F_ETS - 1) / F_ETS;
            const bool lo8 = n < 8;
            // stage e = the W_out fragments of element tiles 4e .. 4e+3, all NKS k-steps: contiguous in the image (NKSC == NKS)
            auto issueC = [&](int e, int slot) {
                for (int q = wave; q < F_ETS * NKS; q += F_NW) {
                    int et = e * F_ETS + q / NKS;
                    et = et < p.NET ? et : p.NET - 1;
                    dma_piece(uniform_ptr(p.wout_img + ((size_t)et * p.NKSC + (q % NKS)) * 512), lds0 + p.off_union + slot * STAGE_C + q * 1024, lane16);
                }
            };
            // row bases of the 128-byte store rows: lanes n < 8 and n >= 8 of a 16-lane row trade one accumulator (DPP row_ror:8),
            // so one store instruction writes 128 contiguous bytes (a whole line) for each of 8 tokens
            float *ob[A_TT][2];
            bool live[A_TT][2];
#pragma unroll
            for (int T = 0; T < A_TT; ++T)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const long long tok = tw0 + 16 * T + 8 * h + (n & 7);
                    live[T][h] = tok < tile_end;
                    ob[T][h] = p.out + (size_t)(live[T][h] ? tok : 0) * p.d + 4 * g + (lo8 ? 0 : 16);
                }
            issueC(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int e0 = 0; e0 < nstg; e0 += 2) {
                static_for<0, 2>([&](auto sl_c) {
                    constexpr int SL = decltype(sl_c)::value;
                    const int e = e0 + SL;
                    if (e < nstg) {
                        // this wave's pieces of stage e landed: the 2 F_ETS stores of the previous stage are the only younger operations
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * F_ETS) : "memory");
                        __builtin_amdgcn_s_barrier();     // every wave's pieces landed; every wave has left stage e-1
                        if (e + 1 < nstg) issueC(e + 1, SL ^ 1);
                        const char *wb = un + SL * STAGE_C + lane * 16;
#pragma unroll
                        for (int tp = 0; tp < F_ETS / 2; ++tp) {
                            const int et = e * F_ETS + 2 * tp;                    // element tiles et, et + 1
                            const bool et_ok = (lo8 ? et : et + 1) < p.NET;
                            const int et1 = et + 1 < p.NET ? et + 1 : p.NET - 1, et0c = et < p.NET ? et : p.NET - 1;
                            const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(sbias + et0c * 16 + 4 * g);
                            const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(sbias + et1 * 16 + 4 * g);
                            f32x4 acc0[A_TT], acc1[A_TT];
#pragma unroll
                            for (int T = 0; T < A_TT; ++T) { acc0[T] = bias0; acc1[T] = bias1; }
#pragma unroll
                            for (int ks = 0; ks < NKS; ++ks) {
                                const u32x4 a0 = *reinterpret_cast<const u32x4 *>(wb + ((2 * tp) * NKS + ks) * 1024);
                                const u32x4 a1 = *reinterpret_cast<const u32x4 *>(wb + ((2 * tp + 1) * NKS + ks) * 1024);
#pragma unroll
                                for (int T = 0; T < A_TT; ++T) {
                                    acc0[T] = mfma16(a0, cvB[ks][T], acc0[T]);
                                    acc1[T] = mfma16(a1, cvB[ks][T], acc1[T]);
                                }
                                // keep hipcc from hoisting all 2 NKS fragment reads of the tile pair (64 VGPRs at NKS = 8)
                                if (ks % 2 == 1) asm volatile("" ::: "memory");
                            }
#pragma unroll
                            for (int T = 0; T < A_TT; ++T) {
                                f32x4 s0, s1;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v0 = acc0[T][r], v1 = acc1[T][r];
                                    const float snd = lo8 ? v1 : v0;
                                    const float rcv = f32_of((unsigned)__builtin_amdgcn_update_dpp(0, (int)u32_of(snd), 0x128, 0xf, 0xf, false));
                                    s0[r] = lo8 ? v0 : rcv;    // n < 8: (token n, tile et)      n >= 8: (token n - 8, tile et + 1)
                                    s1[r] = lo8 ? rcv : v1;    // n < 8: (token n + 8, tile et)  n >= 8: (token n, tile et + 1)
                                }
                                // exactly two store instructions per token tile (rows / elements that do not exist go to the dump
                                // row), so that the wait above can count them
                                float *dmp = p.dump + ((wave * 4 + 2 * T) & 31) * 256 + lane * 4;     // 32 KiB of dump rows
                                float *d0 = (live[T][0] && et_ok) ? ob[T][0] + et * 16 : dmp;
                                float *d1 = (live[T][1] && et_ok) ? ob[T][1] + et * 16 : dmp + 256;
                                *reinterpret_cast<f32x4 *>(d0) = s0;
                                *reinterpret_cast<f32x4 *>(d1) = s1;
                            }
                        }
                    }
                });
            }
        }
        F_STAMP(7);
        __builtin_amdgcn_s_barrier();     // the union region turns back into the W_in ring
    }
}
