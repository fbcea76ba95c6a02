// gram_fused.hpp -- the matrix-core arm as ONE persistent kernel (included by gram.hip inside its anonymous namespace).
//
// The three-kernel form (project, levels, reconstruct) moves 0.37 GB of intermediates through HBM per 262144 tokens and
// runs a read-bound, a compute-bound and a write-bound phase one after the other on the whole chip.  Here a workgroup takes
// a tile of 256 tokens through all three phases; projections and straight-through values never leave the CU.
//
//   phase A  P = x W_in'   : x AND the W_in image arrive by LDS-DMA (x in pieces of 8 rows x 128 bytes, XOR-swizzled through
//                            the per-lane SOURCE address so that the MFMA operand reads are conflict-free; reading x in the
//                            MFMA operand shape, 16 rows x 64 bytes per instruction, costs the CU's address path twice as
//                            much -- profiles/r02_fused_timeline_regloads.txt), counted vmcnt + one raw barrier per chunk
//   phase B  level chain   : one token per lane in c-space, cosine contraction on MFMA against the codebook image, exact fp32
//                            rescoring of the 16 best-candidate rows (as gram_levels_kernel)
//   phase C  out = W_out cv: cv pieces in registers (B operand), W_out image by LDS-DMA through a 3-stage ring, accumulator
//                            tiles transposed through LDS so that every store instruction writes 2 rows x 512 contiguous
//                            bytes (a CU stores 132 GB/s in that shape and 34 GB/s in rows of 128 or 256 bytes:
//                            profiles/r02_store_patterns_per_cu.txt)
//
// One workgroup = 8 waves x 32 tokens, one workgroup per CU.  The three phases need different tables, so LDS is re-mapped
// at every phase boundary (one barrier each) and the codebook image is staged again for every tile (86 KB from L2 per 256
// tokens -- 2 % of the tile's traffic):
//   A: [W_in ring 3 x 18 KB][x rings 8 waves x 3 x 4 KB]                          = 150 KB
//   B: [codebook image 64 KB][fp32 codebook 22 KB][exchange 8 x 5 KB][Gram 7 KB]  = 133 KB
//   C: [bias sums 9 KB][W_out ring 3 x 32 KB][transposition 8 x 4.1 KB]           = 138 KB      (sizes at C2)
// Serves 2-byte x (bf16 / fp16), one level group (levels <= 8) and a shared codebook; other shapes take the three kernels.

struct FusedParams {
    const void *x;
    const uint16_t *win_img;   // [KSA][NOG][3][512]
    const float *kconst;       // [NOG * 16]
    const float *gneg;         // [L][L][GB]
    const uint16_t *cb_img;    // [KTP][512]
    const float *cb_f32;       // [K + 64][c], zero padded to whole KiB
    const uint16_t *wout_img;  // [NET][NKSC][512]
    const float *bias;         // [d] (+ slack to a whole KiB): bias_prefix[levels - 1]
    float *out;                // [n_tokens][d], or NULL: codes only
    long long *idx;            // [n_tokens][levels]
    float *dump;               // 32 KiB: stores of rows / elements that do not exist
    unsigned long long *stamps; // tools build (-DRQAE_FUSED_TIMELINE) only: s_memtime at the phase boundaries
    long long n_tokens;
    int d, K, KTP, c, levels, L, KSA, NOG, NET, NKSC;
    int stagger;               // 1: the first tile of every odd workgroup is a half tile (ftile_off)
    // the single-phase forms (MODE 1 / 2 below) exchange the intermediates of the three-kernel path with gram_levels_kernel:
    float *P;                  // MODE 1 writes the projections [NOG][tok_pad][16]
    const unsigned *cvimg;     // MODE 2 reads the reconstruction operand rows [k-step][tok_pad][16 dwords]
    long long tok_pad;
    int ug0;                   // MODE 1: first projection group of this launch (NG groups per launch)
    int ny;                    // MODE 1: > 1 = one launch covers ny blocks of NG groups; the ny workgroups of a tile share an XCD
    int nks_live;              // MODE 2: k-steps the levels in use fill (the rest are zero)
    const float *cvf;          // MODE 3: fp32 straight-through values, level planes [8][tok_pad][CP]
    const void *codes;         // MODE 4: code indices [n_tokens][levels] (int64 / int32 / uint16 by the XT parameter: RQAE_IDX_*)
    int *status;               // MODE 4: include/rqae_hip.h "Status word" (a code outside -K..K-1), or NULL
    float *inv_rms;            // MODE 5: 1 / sqrt(mean(x^2) + eps) per token, written for the way back; MODE 6 reads it
    float eps;                 // MODE 5
    // MODE 6 (MODE 3 with Gemma's denorm, BOS skip and the write-back into the layer output in the store path):
    void *hs_out;              // [n_tokens][d] fp16 / bf16 (hs_dtype: RQAE_X_F16 / RQAE_X_BF16), the hooked layer's output, patched in place
    const float *inv1pw;       // [d] (+ slack to a whole KiB): 1 / (1 + norm.weight)
    long long tok0;            // global index of this launch's first token (BOS rows: (tok0 + t) % S == 0)
    int S, skip_bos, hs_dtype;
};

constexpr int F_NW = 8;                 // waves per workgroup
constexpr int F_TILE = F_NW * 32;       // tokens per tile
constexpr int F_XCHW = 5120;            // phase B: exchange bytes per wave
constexpr int F_ETS = 4;                // phase C: element tiles per W_out stage
constexpr int F_XSLOT = 4096;           // phase A: one chunk of a wave's x (32 rows x 128 bytes)
constexpr int F_TRS = 528;              // phase C: row stride of the transposition buffer (512 + 16: conflict-free writes)
constexpr int F_TRW = 4608;             // phase C: transposition bytes per wave (8 token rows of F_TRS; MODE 3 also passes 32 x 144 bytes through it)



template <int CP> struct FusedGeom {
    static constexpr int GL = 8;
    static constexpr int GP = ((GL * CP + 15) / 16) * 16;
    static constexpr int NG = GP / 16;                       // projection groups
    static constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;
    static constexpr int NKS = GS / 32;                      // k-steps of the reconstruction
    static constexpr int GB = (CP * CP + 3) & ~3;
    static constexpr int KC = 2;                             // k-steps per phase-A chunk: 64 elements = 128 bytes of a 2-byte row
    static constexpr int STAGE_A = KC * NG * 3 * 1024;
    static constexpr int STAGE_C = F_ETS * NKS * 1024;
    static constexpr size_t LDS_A = 3 * (size_t)STAGE_A + F_NW * 3 * (size_t)F_XSLOT;
};

__host__ __device__ inline size_t fused_lds_bytes(int d, int c, int K)
{
    const int CP = cp_of(c);
    const size_t KTP = (((size_t)(K + 15) / 16 + 3) / 4) * 4;
    const size_t gb = (CP * CP + 3) & ~3, nks = CP == 4 ? FusedGeom<4>::NKS : FusedGeom<5>::NKS;
    const size_t a = CP == 4 ? FusedGeom<4>::LDS_A : FusedGeom<5>::LDS_A;
    const size_t b = KTP * 1024 + pad1k(((size_t)K + 64) * c * 4) + F_NW * F_XCHW + 64 * gb * 4;
    const size_t cc = pad1k((size_t)d * 4) + 3 * F_ETS * nks * 1024 + F_NW * F_TRW;
    return a > b ? (a > cc ? a : cc) : (b > cc ? b : cc);
}

// Tile schedule.  Workgroup b of G takes tiles b, b + G, b + 2G, ...  With stagger the first tile of every odd workgroup is a
// HALF tile: all workgroups start together and would otherwise stay in step (every CU reading, then every CU on the matrix
// cores, then every CU writing); a half tile puts the odd workgroups half a period ahead.  ftile_off(j) = first token of tile j.
__host__ __device__ inline long long ftile_off(long long j, long long G, int stagger)
{
    if (!stagger) return 256 * j;
    return j < G ? 192 * j + ((j & 1) ? 64 : 0) : 192 * G + 256 * (j - G);
}
__host__ inline long long ftile_count(long long n_tokens, long long G, int stagger)
{
    if (!stagger) return (n_tokens + 255) / 256;
    if (n_tokens <= 192 * G) {
        long long j = n_tokens / 192;
        while (j > 0 && ftile_off(j, G, 1) >= n_tokens) --j;
        while (ftile_off(j, G, 1) < n_tokens) ++j;
        return j;
    }
    return G + (n_tokens - 192 * G + 255) / 256;
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

// CEQ: c == CP (codebook rows are CP floats apart: aligned float4 row quads)
// MODE 0: all three phases.  MODE 1: phase A alone (projections to global memory: the first kernel of the three-kernel
// path).  MODE 4: phase C alone from code indices (decode; XT then names the index type).  MODE 2 / 3: phase C alone (its third kernel; operand rows as bf16 pieces from gram_levels_kernel / as fp32 values from
// gram_levels2_kernel).  Same code, same bits.
// MODE 5: MODE 1 with Gemma's RMSNorm folded in (rqae/llm.py:65-66 in front of forward, rqae/model.py:279-281): the norm is
// linear in front of lin_in -- normed = (x rs) (1 + w) with rs = 1 / sqrt(mean(x^2) + eps) per token -- so (1 + w) is folded into
// the W_in image at pack time (rqae_hip_gram_pack's in_scale), the kernel sums x^2 while the tile's fragments pass through its
// registers (v_dot2c_f32_bf16 / _f16, four per fragment) and scales the projections by rs before they leave; `normed` never exists.
template <int XT, int CP, bool CEQ, int MODE = 0>
__global__ __launch_bounds__(F_NW * 64) void gram_fused_kernel(FusedParams p)
{
    using G = FusedGeom<CP>;
    constexpr int GL = G::GL, NG = G::NG, NKS = G::NKS, GB = G::GB, KC = G::KC, STAGE_A = G::STAGE_A, STAGE_C = G::STAGE_C;
    constexpr bool PROJ = MODE == 1 || MODE == 5, NORM = MODE == 5;
    static_assert((MODE >= 2 && !PROJ) || XRaw<XT>::ESZ == 2, "x rows are staged as 128-byte pieces of 64 two-byte elements");
    constexpr int FPK = NG * 3;                    // W_in fragments per k-step
    constexpr int XP = 4;                          // x pieces per chunk and wave (32 rows x 128 bytes)
    static_assert(A_TT == 2, "32 tokens per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int tl = lane & 31;                      // c-space: lane l works on token (l & 31) of the wave (upper half mirrors)
    const int c = CEQ ? CP : p.c;
    const int levels = p.levels;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    const size_t rowb = (size_t)p.d * 2;
    // ---- LDS maps
    // A
    char *wring = smem;
    const unsigned off_xw = 3u * STAGE_A + (unsigned)wave * 3u * F_XSLOT;
    char *xw = smem + off_xw;
    // B
    char *cbl = smem;
    const unsigned off_cbf = (unsigned)p.KTP * 1024u;
    const unsigned cbf_bytes = (unsigned)pad1k(((size_t)p.K + 64) * c * 4);
    const float *cbf = reinterpret_cast<const float *>(smem + off_cbf);
    char *xch = smem + off_cbf + cbf_bytes + wave * F_XCHW;
    float *sgram = reinterpret_cast<float *>(smem + off_cbf + cbf_bytes + F_NW * F_XCHW);
    // C
    const float *sbias = reinterpret_cast<const float *>(smem);
    const unsigned bias_bytes = (unsigned)pad1k((size_t)p.d * 4);
    const unsigned off_wo = MODE == 6 ? 2 * bias_bytes : bias_bytes;      // MODE 6 keeps 1 / (1 + w) behind the bias sums
    char *trb = smem + off_wo + 3 * STAGE_C + wave * F_TRW;

    const int nchunks = p.KSA / KC;
    const int wq = (tl >> 2) & 3;
    long long Gn = gridDim.x, tile_first = blockIdx.x;
    int ug0 = p.ug0;
    bool x_shared = false;
    if constexpr (PROJ) {
        if (p.ny > 1) {
            // XCD-aware mapping: consecutive workgroup ids go round the 8 XCDs, so ids that are equal mod 8 share an L2.  The ny
            // workgroups that take the same token tile through different output groups get such ids and run at the same time:
            // the tile's x comes from HBM once and from that XCD's L2 ny - 1 times (no non-temporal hint on it then).
            const unsigned bid = blockIdx.x, xcd = bid & 7u, within = bid >> 3;
            const unsigned ysel = within % (unsigned)p.ny;
            Gn = gridDim.x / (unsigned)p.ny;
            tile_first = (within / (unsigned)p.ny) * 8u + xcd;
            ug0 = p.ug0 + (int)ysel * NG;
            x_shared = true;
        }
    }

    int tord = 0;
    for (long long tile = tile_first; ftile_off(tile, Gn, p.stagger) < p.n_tokens; tile += Gn, ++tord) {
        const long long tile0 = ftile_off(tile, Gn, p.stagger);
        long long tile_end = ftile_off(tile + 1, Gn, p.stagger);      // rows past it belong to another tile: never stored
        tile_end = tile_end < p.n_tokens ? tile_end : p.n_tokens;
        F_STAMP(0);
        const long long tw0 = tile0 + (long long)wave * 32;           // first token of this wave
        // a wave without tokens (half tiles, the ragged last tile) keeps issuing its LDS-DMA pieces and meeting the barriers,
        // and skips the arithmetic
        const bool active = tw0 < tile_end;
        // =========================================================================================== phase A
        f32x4 acc[NG][A_TT];
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int T = 0; T < A_TT; ++T) acc[u][T] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ssq[A_TT] = {0.0f, 0.0f};          // NORM: this lane's share of sum(x^2) of its token in each token tile
        if constexpr (MODE < 2 || PROJ) {
            // Piece pc of a chunk = rows 8 pc .. 8 pc + 7 of the wave x 128 bytes; lane i = (row r = i >> 3, slot jj = i & 7)
            // lands at piece + 16 i and fetches the row's 16-byte chunk jj ^ r: LDS row r holds chunk j at slot j ^ r, and the
            // operand reads below (lane (g, n): row n, chunk 4k + g) touch 16 different bank groups per 16-lane read group.
            unsigned voff[XP];
#pragma unroll
            for (int pc = 0; pc < XP; ++pc) {
                long long r = tw0 + 8 * pc + (lane >> 3);
                r = r < tile_end ? r : tile_end - 1;          // rows past the tile re-read its last row (never stored)
                voff[pc] = (unsigned)((r - tile0) * (long long)rowb) + (unsigned)((((lane & 7) ^ (lane >> 3)) & 7) * 16);
            }
            const char *xtile = static_cast<const char *>(p.x) + (size_t)tile0 * rowb;
            // per iteration s the wave issues G(s) = { its x pieces of chunk s+2, its W_in pieces of chunk s+2 }, then multiplies
            // chunk s: everything older than G(s-1) must have landed (x(s), W(s) are in G(s-2)), G(s-1) may stay in flight.
            // Chunks past the end repeat the last one so that the operation counts stay static.
            constexpr int NPB = (KC * FPK) / F_NW, NPX = (KC * FPK) % F_NW;    // W pieces per wave: NPB, one more on waves < NPX
            auto issue = [&](int cc_in, int slot) {
                const int cc = cc_in < nchunks ? cc_in : nchunks - 1;
#ifdef FQ_NOX   /* tools build: every x piece re-reads chunk 0 of the tile (same operation count, cache-hot lines) */
                const void *xb = uniform_ptr(xtile);
#else
                const void *xb = uniform_ptr(xtile + (size_t)cc * 128);
#endif
                if (x_shared) {
#pragma unroll
                    for (int pc = 0; pc < XP; ++pc) dma_piece(xb, lds0 + off_xw + slot * F_XSLOT + pc * 1024, voff[pc]);
                } else {
#pragma unroll
                    for (int pc = 0; pc < XP; ++pc) dma_piece_nt(xb, lds0 + off_xw + slot * F_XSLOT + pc * 1024, voff[pc]);
                }
                for (int q = wave; q < KC * FPK; q += F_NW) {
                    const int k = q / FPK, f = q - k * FPK;
#ifdef FQ_NOW   /* tools build: every W piece re-reads the image's first KiB */
                    const int ks = 0 * (cc * KC + k);
#else
                    const int ks = cc * KC + k;          // < KSA: the host sends only d % (32 KC) == 0 here
#endif
                    int fu = f / 3;                      // groups past the last one repeat it (MODE 1, never stored)
                    fu = (PROJ ? ug0 : 0) + fu;
                    fu = fu < p.NOG ? fu : p.NOG - 1;
                    dma_piece(uniform_ptr(p.win_img + (((size_t)ks * p.NOG + fu) * 3 + (f % 3)) * 512), lds0 + slot * STAGE_A + (k * FPK + f) * 1024, lane16);
                }
            };
            issue(0, 0);
            issue(1, 1);
            int st = 0;                                   // ring slot of chunk s = s % 3
            // operand read offsets inside an x slot: lane (g, n), token tile T, k-step k
            unsigned xro[A_TT][KC];
#pragma unroll
            for (int T = 0; T < A_TT; ++T)
#pragma unroll
                for (int k = 0; k < KC; ++k) xro[T][k] = (unsigned)((2 * T + (n >> 3)) * 1024 + (n & 7) * 128 + (((4 * k + g) ^ (n & 7)) & 7) * 16);
            for (int s = 0; s < nchunks; ++s) {
                F_STAMPA(s, 0);
                if (wave < NPX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP + NPB + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP + NPB) : "memory");
                __builtin_amdgcn_s_barrier();        // every wave's pieces of W(s) landed; every wave has left chunk s-1
                F_STAMPA(s, 2);
                issue(s + 2, st == 0 ? 2 : st - 1);  // slot (s+2) % 3 was read last in chunk s-1
                F_STAMPA(s, 3);
                if (active) {
                    const char *wb = wring + st * STAGE_A + lane * 16;
                    const char *xs = xw + st * F_XSLOT;
                    constexpr int NF = KC * FPK, MF = A_TT * XFrag<XT>::NP, PF = 6;
                    u32x4 xraw[KC][A_TT][1];
#pragma unroll
                    for (int k = 0; k < KC; ++k)
#pragma unroll
                        for (int T = 0; T < A_TT; ++T) xraw[k][T][0] = *reinterpret_cast<const u32x4 *>(xs + xro[T][k]);
                    if constexpr (NORM) {
#pragma unroll
                        for (int k = 0; k < KC; ++k)
#pragma unroll
                            for (int T = 0; T < A_TT; ++T)
#pragma unroll
                                for (int wd = 0; wd < 4; ++wd) {
                                    const unsigned u = xraw[k][T][0][wd];
                                    if constexpr (XT == RQAE_X_BF16) {
                                        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                                        ssq[T] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u), __builtin_bit_cast(bf2, u), ssq[T], false);
                                    } else {
                                        typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
                                        ssq[T] = __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2, u), __builtin_bit_cast(hf2, u), ssq[T], false);
                                    }
                                }
                    }
                    u32x4 fr[NF];
#pragma unroll
                    for (int i = 0; i < NF; ++i) fr[i] = *reinterpret_cast<const u32x4 *>(wb + i * 1024);
                    u32x4 xp[KC][A_TT][3];
#pragma unroll
                    for (int k = 0; k < KC; ++k)
#pragma unroll
                        for (int T = 0; T < A_TT; ++T) xparts<XT>(xraw[k][T], xp[k][T]);
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const int k = i / FPK, u = (i % FPK) / 3;
#pragma unroll
                        for (int T = 0; T < A_TT; ++T)
#pragma unroll
                            for (int xq = 0; xq < XFrag<XT>::NP; ++xq) acc[u][T] = mfma16(fr[i], xp[k][T][xq], acc[u][T]);
                    }
                    // "all reads, then all MFMAs" put in order: the x operands and PF fragments ahead, then one fragment read per
                    // fragment's MFMAs (left to itself hipcc reads two fragments, waits, multiplies: the LDS latency 18 times per chunk)
                    __builtin_amdgcn_sched_group_barrier(0x100, KC * A_TT + PF, 0);
#pragma unroll
                    for (int i = 0; i < NF - PF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, PF * MF, 0);
                }
                st = st == 2 ? 0 : st + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated groups past the last chunk
        }
        F_STAMP(1);
        if constexpr (PROJ) {
            // D layout: lane (g, n) holds outputs 4g..4g+3 of token 16T+n of group u
            if constexpr (NORM) {
                // the four lanes (g) of a token hold a quarter of its squares each; every one of them ends up with the row's sum
#pragma unroll
                for (int T = 0; T < A_TT; ++T) {
                    float v = ssq[T];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    const float rs = 1.0f / __builtin_sqrtf(v / (float)p.d + p.eps);
                    const long long tok = tw0 + 16 * T + n;
                    if (g == 0 && tok < tile_end && ug0 == p.ug0) p.inv_rms[tok] = rs;
#pragma unroll
                    for (int u = 0; u < NG; ++u) acc[u][T] = f32x4{acc[u][T][0] * rs, acc[u][T][1] * rs, acc[u][T][2] * rs, acc[u][T][3] * rs};
                }
            }
#pragma unroll
            for (int u = 0; u < NG; ++u)
#pragma unroll
                for (int T = 0; T < A_TT; ++T) {
                    const long long tok = tw0 + 16 * T + n;
                    if (tok < tile_end && ug0 + u < p.NOG) *reinterpret_cast<f32x4 *>(p.P + (((size_t)(ug0 + u) * p.tok_pad + tok) * 16 + 4 * g)) = acc[u][T];
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // the rings are free for the next tile
            continue;
        }
        u32x4 cvB[NKS][A_TT];
        // cv pieces of the wave's tokens as B operands of the reconstruction: slot of (level li, component j, term) =
        // (li CP + j) 6 + term, dword (li CP + j) 3 + {0: hi|mid, 1: hi|lo, 2: hi|mid}; lane l holds the fp32 values of token
        // l & 31, two k-steps (32 dwords per token) per trip through the wave's LDS buffer `xb` (32 x 144 bytes)
        auto build_cvB = [&](const float (&cvh)[GL][CP], char *xb) {
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
                *reinterpret_cast<u32x4 *>(xb + tl * 144 + 16 * v) = u32x4{dwr[4 * v], dwr[4 * v + 1], dwr[4 * v + 2], dwr[4 * v + 3]};
            wave_sync_lds();
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
                if (2 * R + kl < NKS) {
#pragma unroll
                    for (int T = 0; T < A_TT; ++T) cvB[2 * R + kl][T] = *reinterpret_cast<const u32x4 *>(xb + (16 * T + n) * 144 + kl * 64 + 16 * g);
                }
            wave_sync_lds();
            });
        };
        if constexpr (MODE == 0) {
        __builtin_amdgcn_s_barrier();    // LDS turns from map A into map B
        F_STAMP(2);
        // ---- map B tables: codebook image + fp32 codebook by LDS-DMA, Gram blocks of the level group by the threads
        {
            const int npi = p.KTP + (int)(cbf_bytes >> 10);
            for (int q = wave; q < npi; q += F_NW) {
                const void *src = q < p.KTP ? static_cast<const void *>(p.cb_img + (size_t)q * 512)
                                            : static_cast<const void *>(reinterpret_cast<const char *>(p.cb_f32) + (size_t)(q - p.KTP) * 1024);
                dma_piece(uniform_ptr(src), lds0 + q * 1024, lane16);      // the fp32 codebook follows the image
            }
            for (int i = tid; i < GL * GL * GB; i += F_NW * 64) {
                const int blk = i / GB, li = blk / GL, li2 = blk - li * GL;
                sgram[i] = (li < p.L && li2 < p.L) ? p.gneg[((size_t)li * p.L + li2) * GB + (i - blk * GB)] : 0.0f;
            }
        }
        // ---- D layout (lane (g, n): outputs 4g..4g+3 of token 16T+n) -> one token per lane, through the exchange buffer
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // map B tables staged
        F_STAMP(3);
        // =========================================================================================== phase B
        float cvh[GL][CP];
        int codes[GL];
        static_for<0, GL>([&](auto li_c) {
            constexpr int li = decltype(li_c)::value;
#ifdef FQ_NOB   /* tools build: no level chain (codes and cv are garbage) */
            if (li < levels && p.n_tokens < 0) {
#else
            if (li < levels && active) {
#endif
                float q[CP];
#pragma unroll
                for (int j = 0; j < CP; ++j) q[j] = Q[li][j];
                float n2 = 0.0f;
#pragma unroll
                for (int j = 0; j < CP; ++j) n2 = __builtin_fmaf(q[j], q[j], n2);
                const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
                float xn[CP];
                unit_dir<CP>(q, c, xn);      // the reference's q / ||q|| (rqae/model.py:188): what the exact rescoring multiplies the rows by
                if (li == 1) F_STAMPA(30, 0);
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
                if (li == 1) F_STAMPA(30, 1);
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
                if (li == 1) F_STAMPA(30, 2);
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
                if (li == 1) F_STAMPA(30, 3);
                // ---- the 16 rows behind that maximum, rescored in fp32 (k-ordered fma chain, the reference's order): the first
                // maximum in row order wins.  All 16 scores first (independent chains, every read issued up front), then a
                // pairwise tournament in row order (the left entry wins ties) instead of a 16-step dependent chain.
                int brow;
                {
                    float sc[16];
                    int rw[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r0 = 16 * (4 * bgr + i) + 4 * bg;
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
                            float acc_s = 0.0f;
#pragma unroll
                            for (int j = 0; j < CP; ++j) acc_s = __builtin_fmaf(xn[j], f[r * CP + j], acc_s);
                            // rows past K (copies of row 0 in LDS) lose to everything; a NaN score loses too (okq handles NaN rows)
                            sc[4 * i + r] = (r0 + r < p.K) ? acc_s : -INFINITY;
                            rw[4 * i + r] = r0 + r;
                        }
                    }
#pragma unroll
                    for (int st2 = 1; st2 < 16; st2 *= 2)
#pragma unroll
                        for (int k = 0; k < 16; k += 2 * st2) {
                            const bool right = sc[k + st2] > sc[k];       // strict: the earlier row keeps ties
                            sc[k] = right ? sc[k + st2] : sc[k];
                            rw[k] = right ? rw[k + st2] : rw[k];
                        }
                    brow = rw[0];
                }
                // equal maxima in one 64-row block on several lanes (duplicate rows): rescan the block in row order.  Rare: the
                // wave branches around it
                if (__builtin_amdgcn_ballot_w64(amb) != 0) {
                    if (amb) {
                        float bsc = -INFINITY;
                        brow = 64 * bgr;
#pragma unroll 1
                        for (int r0 = 64 * bgr; r0 < 64 * bgr + 64; r0 += 4) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float acc_s = 0.0f;
#pragma unroll
                                for (int j = 0; j < CP; ++j) acc_s = __builtin_fmaf(xn[j], j < c ? cbf[(size_t)(r0 + r) * c + j] : 0.0f, acc_s);
                                const bool up = (acc_s > bsc) && (r0 + r < p.K);
                                bsc = up ? acc_s : bsc;
                                brow = up ? r0 + r : brow;
                            }
                        }
                    }
                }
                if (!okq) brow = 0;
                codes[li] = brow;
                if (li == 1) F_STAMPA(31, 0);
                // ---- straight-through value (rqae/model.py:218-220)
#pragma unroll
                for (int j = 0; j < CP; ++j) {
                    const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                    const float df = cv - q[j];
                    cvh[li][j] = q[j] + df;
                }
                if (li == 1) F_STAMPA(31, 1);
                // ---- later levels see the level just applied.  Every block of the group is staged (zeros past the depth in use), so
                // no depth checks here: block k+1 is read while block k is multiplied
                if constexpr (li + 1 < GL) {
                    float gbr[2][GB];
                    auto rd = [&](int li2, float (&dst)[GB]) {
                        const float *gb = sgram + (li * GL + li2) * GB;        // broadcast reads
#pragma unroll
                        for (int w = 0; w < GB / 4; ++w) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(gb + 4 * w);
                            dst[4 * w] = v[0]; dst[4 * w + 1] = v[1]; dst[4 * w + 2] = v[2]; dst[4 * w + 3] = v[3];
                        }
                    };
                    rd(li + 1, gbr[0]);
#pragma unroll
                    for (int li2 = li + 1; li2 < GL; ++li2) {
                        const int cur = (li2 - li - 1) & 1;
                        if (li2 + 1 < GL) rd(li2 + 1, gbr[cur ^ 1]);
#pragma unroll
                        for (int j = 0; j < CP; ++j)
#pragma unroll
                            for (int i = 0; i < CP; ++i) Q[li2][j] = __builtin_fmaf(gbr[cur][j * CP + i], cvh[li][i], Q[li2][j]);
                    }
                }
                if (li == 1) F_STAMPA(31, 2);
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
                    if (li < levels) p.idx[t * levels + li] = (long long)codes[li];
            }
        }
        F_STAMP(4);
        if (p.out == nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();     // LDS turns back into map A
            continue;
        }

        // =========================================================================================== phase C
        build_cvB(cvh, xch);
        F_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the code stores: the counted waits below start from zero
        __builtin_amdgcn_s_barrier();     // LDS turns from map B into map C
        F_STAMP(6);
        }   // MODE == 0
        if constexpr (MODE == 2) {
            // operand rows written by gram_levels_kernel: lane (g, n) takes dwords 4g..4g+3 of token 16T+n for every k-step
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int T = 0; T < A_TT; ++T) {
                    long long tok = tw0 + 16 * T + n;
                    tok = tok < tile_end ? tok : tile_end - 1;
                    const int kc = ks < p.nks_live ? ks : p.nks_live - 1;      // k-steps past the depth in use: any row (their weights are zeroed below)
                    cvB[ks][T] = *reinterpret_cast<const u32x4 *>(p.cvimg + ((size_t)kc * p.tok_pad + tok) * 16 + 4 * g);
                    if (ks >= p.nks_live) cvB[ks][T] = u32x4{0u, 0u, 0u, 0u};
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (MODE == 3 || MODE == 6) {
            // fp32 straight-through values written by gram_levels2_kernel, level planes [GL][tok_pad][CP] (zeros past the depth in use): the
            // split into bf16 pieces happens here, on the memory-bound side
            float cvf[GL][CP];
            {
                long long tok = tw0 + tl;
                tok = tok < tile_end ? tok : tile_end - 1;
#pragma unroll
                for (int li = 0; li < GL; ++li) {
                    const float *src = p.cvf + ((size_t)li * p.tok_pad + tok) * CP;
#pragma unroll
                    for (int j = 0; j < CP; ++j) cvf[li][j] = src[j];
                }
            }
            build_cvB(cvf, trb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (MODE == 4) {
            // decode (rqae/model.py:236-252): the values are the codebook rows the codes name -- gathered from the fp32 rows in L2,
            // split here like MODE 3's
            float cvf[GL][CP];
            {
                long long tok = tw0 + tl;
                tok = tok < tile_end ? tok : tile_end - 1;
#pragma unroll
                for (int li = 0; li < GL; ++li) {
                    long long code = 0;
                    if (li < levels) {
                        if constexpr (XT == RQAE_IDX_I64) code = static_cast<const long long *>(p.codes)[tok * levels + li];
                        else if constexpr (XT == RQAE_IDX_I32) code = static_cast<const int *>(p.codes)[tok * levels + li];
                        else code = static_cast<const unsigned short *>(p.codes)[tok * levels + li];
                        // torch indexing (rqae/model.py:234): -K <= code < K; anything else raises IndexError there, here it reads
                        // row 0 and flags the call (include/rqae_hip.h "Status word")
                        code = code < 0 ? code + p.K : code;
                        if ((unsigned long long)code >= (unsigned long long)p.K) {
                            if (p.status) atomicOr(p.status, RQAE_STATUS_CODE_RANGE);
                            code = 0;
                        }
                    }
                    const float *row = p.cb_f32 + (size_t)code * c;
#pragma unroll
                    for (int j = 0; j < CP; ++j) cvf[li][j] = (li < levels && j < c) ? row[j] : 0.0f;
                }
            }
            build_cvB(cvf, trb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        {
            const int nstg = (p.NET + F_ETS - 1) / F_ETS;
            // stage e = the W_out fragments of element tiles 4e .. 4e+3, all NKS k-steps (4 pieces per wave)
            auto issueC = [&](int e, int slot) {
                for (int q = wave; q < F_ETS * NKS; q += F_NW) {
                    int et = e * F_ETS + q / NKS;
                    et = et < p.NET ? et : p.NET - 1;
#ifdef FQ_NOWOUT   /* tools build: every W_out piece re-reads element tile 0 */
                    et = 0;
#endif
                    dma_piece(uniform_ptr(p.wout_img + ((size_t)et * p.NKSC + (q % NKS)) * 512), lds0 + off_wo + slot * STAGE_C + q * 1024, lane16);
                }
            };
            for (int q = wave; q < (int)(bias_bytes >> 10); q += F_NW)
                dma_piece(uniform_ptr(reinterpret_cast<const char *>(p.bias) + (size_t)q * 1024), lds0 + q * 1024, lane16);
            // MODE 6: per token 1 / rs (lane l holds token l & 31 of the wave) and whether it is a BOS row; 1 / (1 + w) into LDS
            float irs_tok = 1.0f;
            unsigned long long bosmask = 0ull;
            if constexpr (MODE == 6) {
                for (int q = wave; q < (int)(bias_bytes >> 10); q += F_NW)
                    dma_piece(uniform_ptr(reinterpret_cast<const char *>(p.inv1pw) + (size_t)q * 1024), lds0 + bias_bytes + q * 1024, lane16);
                long long tk = tw0 + tl;
                tk = tk < tile_end ? tk : tile_end - 1;
                irs_tok = 1.0f / p.inv_rms[tk];
                bosmask = __builtin_amdgcn_ballot_w64(p.skip_bos && ((p.tok0 + tk) % p.S) == 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the counted waits below start from the pieces issued next
            }
            issueC(0, 0);
            if (nstg > 1) issueC(1, 1);
            constexpr int PCS = (F_ETS * NKS) / F_NW;     // W_out pieces per wave and stage
            static_assert((F_ETS * NKS) % F_NW == 0, "whole pieces per wave");
            constexpr int NST = MODE == 6 ? 8 : 16;       // stores per wave and stage pair (MODE 6: 16 bytes = 8 two-byte elements per lane)
            int slot = 0;
            f32x4 accC[2 * F_ETS][A_TT];
            // a wave whose 32 rows and 128-element column blocks all exist issues exactly NST stores per pair (the waits count
            // them); any other wave predicates its stores and waits for everything instead
            const bool fullw = (tw0 + 32 <= tile_end) && (p.d % 128 == 0);
            float *orow = p.out + (size_t)tw0 * p.d;                  // wave-uniform base; lanes add 32-bit offsets
            for (int e0 = 0; e0 < nstg; e0 += 2) {
                static_for<0, 2>([&](auto half_c) {
                    constexpr int HALF = decltype(half_c)::value;
                    const int e = e0 + HALF;
                    if (e < nstg) {
                        // this wave's pieces of stage e landed.  Younger operations: the pieces of stage e+1 (if it was issued) and the
                        // stores of one pair
                        // (D(e) was issued at stage e-2; between then and now exactly one pair stored -- for every e >= 2 --
                        //  and D(e+1) was issued at stage e-1)
                        if (!fullw) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ragged wave: its store count is not static
                        } else if (e + 1 < nstg) {
                            if (e >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS + NST) : "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS) : "memory");
                        } else {
                            if (e >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        __builtin_amdgcn_s_barrier();     // every wave's pieces landed; every wave has left stage e-1
                        if (e + 2 < nstg) issueC(e + 2, slot == 0 ? 2 : slot - 1);      // slot (e+2) % 3 was read last in stage e-1
                        const char *wb = smem + off_wo + slot * STAGE_C + lane * 16;
                        if (active) {
                            // the stage's F_ETS x NKS fragments, each feeding A_TT MFMAs: "all reads, then all MFMAs" put in order by
                            // sched_group_barrier (PF reads ahead, then one read per fragment's MFMAs), as in phase A
                            constexpr int NFC = F_ETS * NKS, PFC = 6;
#pragma unroll
                            for (int t = 0; t < F_ETS; ++t) {
                                int et = e * F_ETS + t;
                                et = et < p.NET ? et : p.NET - 1;
                                const f32x4 bias = *reinterpret_cast<const f32x4 *>(sbias + et * 16 + 4 * g);
#pragma unroll
                                for (int T = 0; T < A_TT; ++T) accC[HALF * F_ETS + t][T] = bias;
                            }
                            u32x4 frc[NFC];
#pragma unroll
                            for (int i = 0; i < NFC; ++i) frc[i] = *reinterpret_cast<const u32x4 *>(wb + i * 1024);
#pragma unroll
                            for (int i = 0; i < NFC; ++i) {
                                const int t = i / NKS, ks = i % NKS;
#pragma unroll
                                for (int T = 0; T < A_TT; ++T) accC[HALF * F_ETS + t][T] = mfma16(frc[i], cvB[ks][T], accC[HALF * F_ETS + t][T]);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x100, F_ETS + PFC, 0);      // the bias reads and PFC fragments
#pragma unroll
                            for (int i = 0; i < NFC - PFC; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, A_TT, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, PFC * A_TT, 0);
                        }
                        slot = slot == 2 ? 0 : slot + 1;
                    }
                    if (HALF == 1 && active) {
                        // ---- the pair's 8 element tiles x 32 tokens: D layout (lane (g, n): token 16T+n, elements 4g..4g+3 of a tile)
                        // -> rows of 128 elements through LDS, 8 tokens at a time; every store instruction writes 2 rows x 512 bytes.
                        // A full wave issues exactly NST stores per pair: the waits count them.
                        const int el0 = e0 * F_ETS * 16 + 4 * (lane & 31);               // this lane's first element
#pragma unroll
                        for (int T = 0; T < A_TT; ++T)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                if ((n >> 3) == hh) {
#pragma unroll
                                    for (int t = 0; t < 2 * F_ETS; ++t) *reinterpret_cast<f32x4 *>(trb + (n & 7) * F_TRS + t * 64 + 16 * g) = accC[t][T];
                                }
                                wave_sync_lds();
                                if constexpr (MODE == 6) {
                                    // RQAE.hook's way back (rqae/model.py:284-289, rqae/llm.py:68-73): new = (q / (1 + w)) / rs, BOS rows keep
                                    // the layer's own values, the result goes over the layer output in its dtype.  Two multiplications by
                                    // reciprocals instead of the reference's two divisions (this arm is Tier 2; the difference is below the
                                    // fp16 / bf16 rounding that follows).  A lane takes 8 consecutive elements (16 bytes out: 8-byte stores run
                                    // at 0.5-0.7 of the 16-byte rate), 16 lanes a 128-element row, an instruction 4 rows; a BOS row's store
                                    // goes to the dump rows: the store count stays static.
                                    const int el8 = e0 * F_ETS * 16 + 8 * (lane & 15);
#pragma unroll
                                    for (int i = 0; i < 2; ++i) {
                                        const int rr = 4 * i + (lane >> 4);
                                        const int row = 16 * T + 8 * hh + rr;
                                        const f32x4 va = *reinterpret_cast<const f32x4 *>(trb + rr * F_TRS + (lane & 15) * 32);
                                        const f32x4 vb = *reinterpret_cast<const f32x4 *>(trb + rr * F_TRS + (lane & 15) * 32 + 16);
                                        if (fullw || (tw0 + row < tile_end && el8 < p.d)) {
                                            const float r = __shfl(irs_tok, row);
                                            const f32x4 wa = *reinterpret_cast<const f32x4 *>(smem + bias_bytes + (size_t)el8 * 4);
                                            const f32x4 wb8 = *reinterpret_cast<const f32x4 *>(smem + bias_bytes + (size_t)el8 * 4 + 16);
                                            float o[8];
#pragma unroll
                                            for (int q = 0; q < 4; ++q) { o[q] = va[q] * wa[q] * r; o[4 + q] = vb[q] * wb8[q] * r; }
                                            u32x4 pk;
                                            if (p.hs_dtype == RQAE_X_F16) {
                                                typedef __attribute__((ext_vector_type(2))) _Float16 h2;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) pk[q] = __builtin_bit_cast(unsigned, h2{(_Float16)o[2 * q], (_Float16)o[2 * q + 1]});
                                            } else {
                                                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
#pragma unroll
                                                for (int q = 0; q < 4; ++q) pk[q] = __builtin_bit_cast(unsigned, b2{(__bf16)o[2 * q], (__bf16)o[2 * q + 1]});
                                            }
                                            const bool bos = (bosmask >> row) & 1ull;
                                            char *dst = bos ? reinterpret_cast<char *>(p.dump) + lane * 16
                                                            : static_cast<char *>(p.hs_out) + ((size_t)(tw0 + row) * p.d + el8) * 2;
                                            *reinterpret_cast<u32x4 *>(dst) = pk;
                                        }
                                    }
                                } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int rr = 2 * i + (lane >> 5);
                                    const f32x4 v = *reinterpret_cast<const f32x4 *>(trb + rr * F_TRS + (lane & 31) * 16);
                                    const int row = 16 * T + 8 * hh + rr;
#ifdef FQ_NOSTORE   /* tools build: the reconstruction is computed and transposed but not stored */
                                    if (p.n_tokens < 0)
#else
                                    if (fullw || (tw0 + row < tile_end && el0 < p.d))
#endif
                                        // non-temporal: 2.4 GB written once.  Left in L2 / MALL as dirty lines they are written back under the
                                        // next kernel's reads (the projection ran 0.27-0.29 ms behind plain stores, 0.25 behind these)
                                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(orow + (unsigned)(row * p.d + el0)));
                                }
                                }
                                wave_sync_lds();
                            }
                    }
                });
            }
        }
        F_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // LDS turns back into map A
    }
}

// ------------------------------------------------------------------------------------------------ level chain, lean form
// The level chain of the three-kernel path for one level group (levels <= 8): the arithmetic of gram_levels_kernel / phase B
// above (same scores, same rescoring, same Gram updates: same codes), with what made that kernel issue-bound taken out of it:
//   * the straight-through values leave as fp32 ([token][8 CP], 160 bytes per token at c = 5 instead of 512 bytes of bf16
//     pieces): the split into pieces moves to the reconstruction kernel (MODE 3), which waits on HBM anyway;
//   * c == CP and the shared codebook are template facts (aligned float4 row quads, no per-component bounds);
//   * Gram blocks come from LDS (broadcast reads, next block read while this one is multiplied);
//   * whole row tiles are scanned unmasked (rows past K repeat row 0 in the image).
// One wave = 64 tokens, one token per lane in c-space; 8 waves per workgroup share the LDS tables; persistent over tiles.
struct Lev2Params {
    const float *P;            // [NOG][tok_pad][16]
    const float *kconst;       // [NOG * 16]
    const float *gneg;         // [L][L][GB]
    const uint16_t *cb_img;    // [KTP][512]
    const float *cb_f32;       // [K + 64][c], whole KiB
    float *cvf;                // [8][tok_pad][CP]: level planes (a wave stores, and the reconstruction loads, along tokens)
    long long *idx;            // [n_tokens][levels]
    long long n_tokens, tok_pad, ntiles;
    int levels, L, K, KTP, c;
    const int *lat_flag;          // the block's lattice word (n >= 2: gram_wide.hpp's lattice form has run, this kernel returns), or NULL
    unsigned long long *stamps;   // tools build (-DRQAE_LEV_TIMELINE) only: 16 s_memtime / s_memrealtime stamps per wave
};

#ifdef RQAE_LEV_TIMELINE
#define L_STAMP(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define L_STAMPR(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define L_STAMP(i) do { } while (0)
#define L_STAMPR(i) do { } while (0)
#endif

__host__ __device__ inline size_t lev2_lds_bytes(int c, int K, int nw = F_NW)
{
    const int CP = cp_of(c);
    const size_t KTP = (((size_t)(K + 15) / 16 + 3) / 4) * 4, gb = (CP * CP + 3) & ~3;
    return KTP * 1024 + pad1k(((size_t)K + 64) * c * 4) + (size_t)nw * 4096 + 64 * gb * 4;
}

template <int CP, bool CEQ, int NW = F_NW>
__global__ __launch_bounds__(NW * 64) void gram_levels2_kernel(Lev2Params p)
{
    if (p.lat_flag && *reinterpret_cast<const volatile int *>(p.lat_flag) >= 2) return;
    constexpr int GL = 8, GP = ((GL * CP + 15) / 16) * 16, NG = GP / 16, GB = (CP * CP + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = CEQ ? CP : p.c;
    const int levels = p.levels;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    L_STAMP(0);
    L_STAMPR(12);
    char *cbl = smem;
    const unsigned off_cbf = (unsigned)p.KTP * 1024u;
    const unsigned cbf_bytes = (unsigned)pad1k(((size_t)p.K + 64) * c * 4);
    const float *cbf = reinterpret_cast<const float *>(smem + off_cbf);
    char *xch = smem + off_cbf + cbf_bytes + wave * 4096;
    float *sgram = reinterpret_cast<float *>(smem + off_cbf + cbf_bytes + NW * 4096);
    {   // tables: codebook image + fp32 rows by LDS-DMA, Gram blocks of the group by the threads
        const int npi = p.KTP + (int)(cbf_bytes >> 10);
        for (int q = wave; q < npi; q += NW) {
            const void *src = q < p.KTP ? static_cast<const void *>(p.cb_img + (size_t)q * 512)
                                        : static_cast<const void *>(reinterpret_cast<const char *>(p.cb_f32) + (size_t)(q - p.KTP) * 1024);
            dma_piece(uniform_ptr(src), lds0 + q * 1024, (unsigned)lane * 16u);
        }
        for (int i = tid; i < GL * GL * GB; i += NW * 64) {
            const int blk = i / GB, li = blk / GL, li2 = blk - li * GL;
            sgram[i] = (li < p.L && li2 < p.L) ? p.gneg[((size_t)li * p.L + li2) * GB + (i - blk * GB)] : 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    L_STAMP(1);
    const int ngrp = p.KTP / 4;
    const bool kragged = (p.K & 15) != 0;
    for (long long tile = (long long)blockIdx.x * NW + wave; tile < p.ntiles; tile += (long long)gridDim.x * NW) {
        const long long t = tile * 64 + lane;          // < tok_pad
        float Q[GL][CP];
        {
            float buf[NG * 16];
#pragma unroll
            for (int u = 0; u < NG; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(p.P + (((size_t)u * p.tok_pad + t) * 16 + 4 * v));
                    const f32x4 kc = *reinterpret_cast<const f32x4 *>(p.kconst + 16 * u + 4 * v);
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
        L_STAMP(2);
        constexpr bool EARLY = NW > 8;
        if constexpr (EARLY) {
            // planes of the levels not run (max_layers < 8) are zeros for the reconstruction: written here, ahead of the chain (a
            // store inside an unrolled level drags a spilled zero vector and its vmcnt waits through the levels that do run)
#pragma unroll 1
            for (int lz = levels; lz < GL; ++lz) {
                float *dst = p.cvf + ((size_t)lz * p.tok_pad + t) * CP;
#pragma unroll
                for (int j = 0; j < CP; ++j) dst[j] = 0.0f;
            }
        }           // 16 waves: 128 registers per lane -- the values leave level by level instead of at the end
        float cvh[EARLY ? 1 : GL][CP];
        int codes[GL];
        static_for<0, GL>([&](auto li_c) {
            constexpr int li = decltype(li_c)::value;
            if (li < levels) {
                // Lane-derived LDS addresses are recomputed in every level from an opaque copy of the lane id: as common
                // subexpressions of the eight unrolled levels hipcc keeps them live across the whole chain -- at 128 registers
                // that meant spills, and every reload of a spilled register is a vector-memory operation whose s_waitcnt also
                // waits for the level's code and value stores (loads and stores share vmcnt): a store round trip per level on
                // the critical path of every wave.
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int g = ln >> 4, n = ln & 15, wq = (ln >> 2) & 3;
                float q[CP];
#pragma unroll
                for (int j = 0; j < CP; ++j) q[j] = Q[li][j];
                float n2 = 0.0f;
#pragma unroll
                for (int j = 0; j < CP; ++j) n2 = __builtin_fmaf(q[j], q[j], n2);
                const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
                {   // B operand rows of the wave's 64 tokens: 32 slots = 64 bytes per token
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
                        *reinterpret_cast<u32x4 *>(xch + ln * 64 + pos * 16) = u32x4{dw[4 * pc], dw[4 * pc + 1], dw[4 * pc + 2], dw[4 * pc + 3]};
                    }
                }
                wave_sync_lds();
                u32x4 B[4];
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    const int pos = (g + 2 * ((n >> 2) & 3)) & 3;
                    B[T] = *reinterpret_cast<const u32x4 *>(xch + (16 * T + n) * 64 + pos * 16);
                }
                // ---- scan: running maximum per (lane, token tile) over groups of 4 row tiles (16 rows per lane)
                float best[4];
                int bgrp[4];
#pragma unroll
                for (int T = 0; T < 4; ++T) { best[T] = -INFINITY; bgrp[T] = 0; }
                // The fragments of a row-tile group (4 KiB: one ds_read_b128 per tile and lane) are read one group ahead into two
                // alternating register sets; __builtin_amdgcn_sched_barrier keeps each group's reads above the other group's MFMAs
                // (left alone hipcc puts a group's reads at the top of its own iteration and the first MFMA waits out the LDS
                // latency 16 times per level).  Plain loads: hipcc counts lgkmcnt itself.
                u32x4 fa[4], fb[4];
                auto ldfrag = [&](u32x4 (&f)[4], int grp) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const u32x4 *>(cbl + (grp * 4 + i) * 1024 + ln * 16);
                };
                auto scan4 = [&](const u32x4 (&a)[4], int grp) {
                    float mx[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // four independent MFMAs back to back (one per token tile), then their maxima (v_maximum3_f32: a NaN score
                        // poisons the maximum, which then never beats `best`; the level takes index 0 through okq)
                        f32x4 s[4];
#pragma unroll
                        for (int T = 0; T < 4; ++T) s[T] = mfma16(a[i], B[T], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int T = 0; T < 4; ++T) {
                            const float m012 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s[T][0], s[T][1]), s[T][2]);
                            if (i == 0) mx[T] = __builtin_elementwise_maximum(m012, s[T][3]);
                            else mx[T] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m012, s[T][3]), mx[T]);
                        }
                    }
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        const bool up = mx[T] > best[T];      // (the clamped reload below is never scanned twice as a different group)
                        best[T] = up ? mx[T] : best[T];
                        bgrp[T] = up ? grp : bgrp[T];
                    }
                };
                ldfrag(fa, 0);
                for (int grp = 0; grp < ngrp; grp += 2) {
                    const int g1 = grp + 1 < ngrp ? grp + 1 : ngrp - 1, g2 = grp + 2 < ngrp ? grp + 2 : ngrp - 1;
                    ldfrag(fb, g1);
                    scan4(fa, grp);
                    __builtin_amdgcn_sched_barrier(0);      // this group's MFMAs stay above the next reads, those reads above the next MFMAs
                    ldfrag(fa, g2);
                    scan4(fb, g1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- the four lanes (g) of a token hand their (maximum, group) to the token's own lane
                wave_sync_lds();
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    *reinterpret_cast<u32x2 *>(xch + ((16 * T + n) * 4 + g) * 8) = u32x2{u32_of(best[T]), (unsigned)bgrp[T]};
                }
                wave_sync_lds();
                const u32x4 h0 = *reinterpret_cast<const u32x4 *>(xch + ln * 32);
                const u32x4 h1 = *reinterpret_cast<const u32x4 *>(xch + ln * 32 + 16);
                wave_sync_lds();
                float xn[CP];
                unit_dir<CP>(q, c, xn);      // the reference's q / ||q|| (rqae/model.py:188): what the exact rescoring multiplies the rows by
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
                // ---- the 16 rows behind that maximum rescored in fp32 (the reference's k-ordered fma chain), first maximum in row
                // order wins: all scores first, then a pairwise tournament in row order (the left entry keeps ties)
                int brow;
                {
                    float sc[16];
                    int rw[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r0 = 16 * (4 * bgr + i) + 4 * bg;
                        float f[4 * CP];
                        if (CEQ) {
                            const f32x4 *qp = reinterpret_cast<const f32x4 *>(cbf + (size_t)r0 * CP);
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
                            float acc_s = 0.0f;
#pragma unroll
                            for (int j = 0; j < CP; ++j) acc_s = __builtin_fmaf(xn[j], f[r * CP + j], acc_s);
                            sc[4 * i + r] = acc_s;
                            rw[4 * i + r] = r0 + r;
                        }
                    }
                    // rows past K exist only in the last row tile of a codebook whose size is no multiple of 16 (a wave-uniform fact):
                    // they never win
                    if (kragged) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) sc[k] = rw[k] < p.K ? sc[k] : -INFINITY;
                    }
#pragma unroll
                    for (int st2 = 1; st2 < 16; st2 *= 2)
#pragma unroll
                        for (int k = 0; k < 16; k += 2 * st2) {
                            const bool right = sc[k + st2] > sc[k];
                            sc[k] = right ? sc[k + st2] : sc[k];
                            rw[k] = right ? rw[k + st2] : rw[k];
                        }
                    brow = rw[0];
                }
                // equal maxima in one 64-row block on several lanes (duplicate rows): rescan the block in row order; rare, so the
                // wave branches around it
                if (__builtin_amdgcn_ballot_w64(amb) != 0) {
                    if (amb) {
                        float bsc = -INFINITY;
                        brow = 64 * bgr;
#pragma unroll 1
                        for (int r0 = 64 * bgr; r0 < 64 * bgr + 64; r0 += 4) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float acc_s = 0.0f;
#pragma unroll
                                for (int j = 0; j < CP; ++j) acc_s = __builtin_fmaf(xn[j], j < c ? cbf[(size_t)(r0 + r) * c + j] : 0.0f, acc_s);
                                const bool up = (acc_s > bsc) && (r0 + r < p.K);
                                bsc = up ? acc_s : bsc;
                                brow = up ? r0 + r : brow;
                            }
                        }
                    }
                }
                if (!okq) brow = 0;
                codes[li] = brow;
                if constexpr (EARLY) { if (t < p.n_tokens) p.idx[t * levels + li] = (long long)brow; }
                // ---- straight-through value (rqae/model.py:218-220)
#pragma unroll
                for (int j = 0; j < CP; ++j) {
                    const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                    const float df = cv - q[j];
                    cvh[EARLY ? 0 : li][j] = q[j] + df;
                }
                if constexpr (EARLY) {
                    float *dst = p.cvf + ((size_t)li * p.tok_pad + t) * CP;      // a wave's 64 lanes: 64 x CP contiguous floats
#pragma unroll
                    for (int j = 0; j < CP; ++j) dst[j] = cvh[0][j];
                }
                // ---- later levels see the level just applied (every block of the group is staged: zeros past the depth in use)
                if constexpr (li + 1 < GL) {
                    float gbr[2][GB];
                    auto rd = [&](int li2, float (&dst)[GB]) {
                        const float *gb = sgram + (li * GL + li2) * GB;        // broadcast reads
#pragma unroll
                        for (int w = 0; w < GB / 4; ++w) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(gb + 4 * w);
                            dst[4 * w] = v[0]; dst[4 * w + 1] = v[1]; dst[4 * w + 2] = v[2]; dst[4 * w + 3] = v[3];
                        }
                    };
                    rd(li + 1, gbr[0]);
#pragma unroll
                    for (int li2 = li + 1; li2 < GL; ++li2) {
                        const int cur = (li2 - li - 1) & 1;
                        if (li2 + 1 < GL) rd(li2 + 1, gbr[cur ^ 1]);
#pragma unroll
                        for (int j = 0; j < CP; ++j)
#pragma unroll
                            for (int i = 0; i < CP; ++i) Q[li2][j] = __builtin_fmaf(gbr[cur][j * CP + i], cvh[EARLY ? 0 : li][i], Q[li2][j]);
                    }
                }
                L_STAMP(3 + li);
            } else {
                codes[li] = 0;
                if constexpr (!EARLY) {
#pragma unroll
                    for (int j = 0; j < CP; ++j) cvh[li][j] = 0.0f;
                }
            }
        });
        // ---- codes and fp32 straight-through values of the tile
        if (!EARLY && t < p.n_tokens) {
#pragma unroll
            for (int li = 0; li < GL; ++li)
                if (li < levels) p.idx[t * levels + li] = (long long)codes[li];
        }
        if constexpr (!EARLY) {
#pragma unroll
            for (int li = 0; li < GL; ++li) {
                float *dst = p.cvf + ((size_t)li * p.tok_pad + t) * CP;
#pragma unroll
                for (int j = 0; j < CP; ++j) dst[j] = cvh[li][j];
            }
        }
        L_STAMP(11);
        L_STAMPR(13);
#ifdef RQAE_LEV_TIMELINE
        if (lane == 0) {
            p.stamps[((size_t)blockIdx.x * NW + wave) * 16 + 14] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
            p.stamps[((size_t)blockIdx.x * NW + wave) * 16 + 15] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
        }
#endif
    }
}
