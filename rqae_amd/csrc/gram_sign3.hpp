// gram_sign3.hpp -- the level chain of a TWO-POINT FSQ lattice on the matrix cores (included by sign3.hip inside its anonymous
// namespace; BASELINE configs[3]: c = 13, K = 2^13 = 8192, depth 16; rqae/model.py:63-72, 204-224).
//
// With n = 2 magnitudes per coordinate the nearest codebook row IS the sign pattern of the projection (lattice_best, gram_wide.hpp:
// one candidate, nothing to score), so a level costs nothing to decide and the whole chain is its Gram updates
//
//     P[l2] += -(W_in[l2] W_out[l]) cv_l          for every later level l2                       (rqae/model.py:221-224 in c-space)
//
// -- 120 blocks of 13 x 13 per token.  gram_levels_wide_kernel<13, 8, .., LAT> does them as fp32 fma chains, one token per lane:
// 20 k dependent vector instructions per lane, one wave per SIMD at C4's stated batch of 65 536 tokens -- 0.28 ms, a third of the
// forward.  Here the update of ALL later columns by one level is one small matrix product on v_mfma_f32_32x32x16_f16:
//
//     acc[column tile T] (32 columns x 32 tokens) += G_l[T] (32 columns x 16 slots) x cv_l (16 slots x 32 tokens)
//
// with the fp16 two-piece operands of gram_pieces.hpp (three products: hi hi, lo hi, hi lo).  A wave owns 32 tokens and keeps the
// projections of every level in the accumulators of the eight column tiles (128 registers) -- the layout gram_proj3_kernel left them
// in: lane (token r, half h) holds columns 32 T + 8 i + 4 h + 0..3.  A level's 13 coordinates therefore lie in the registers of
// BOTH lanes of a token, in runs of four.  Nothing is transposed: the 16 k-slots of the product are dealt to the coordinates so that
// slots 8 h .. 8 h + 7 -- the ones lane (r, h) supplies as B operand -- are exactly the coordinates that lane holds (at most 8 of
// the 13 fall to one half, whatever the level's first column); the image of G is packed with the same deal.  The two lanes of a token
// exchange one word per level (their halves of the sign pattern, v_permlane32_swap) and one float (their halves of sum q^2: a zero or
// non-finite projection takes row 0, rqae/model.py:182,188).  The image (68 fragments pairs of 2 KiB: every (level, later tile))
// sits in LDS (staged once per workgroup); each level's fragment pairs are read two pairs ahead of the matrix instructions that take them
// (plain LDS loads put in order with sched_group_barrier: an asm read whose register hipcc spills between the read and its wait
// reloads stale data -- the first version of this kernel did, from level 6 on).
//
// Same outputs as the lattice form: codes, and the straight-through values q + (cv - q) as fp32 planes [level][tok_pad][16] for
// gram_recon3_kernel.  Arithmetic: the Gram updates are 2^-22-accurate products accumulated in fp32 in the matrix pipe's order, where the
// lattice form chains fp32 fmas: tokens whose next projection has a coordinate within rounding of zero may take the neighbouring row
// (SURVEY 8(c) Tier 2, as everywhere on this arm).  The codebook row of a sign pattern is (+-mag)^c, mag the one magnitude of an FSQ
// lattice with two points per coordinate (1 / sqrt(c) as torch's normalisation rounds it, or 1); sign3.hip's pack checks that every row of
// the codebook is exactly that, bit for bit (`exact`), else the lattice form runs.

template <int CP> struct S3G {
    static constexpr int GL = 4;                                  // levels per level group (WideGeom<CP>::GL for CP > 8)
    static constexpr int GP = ((GL * CP + 15) / 16) * 16;         // projection columns per group
    static constexpr int L = 16, NG = L / GL, NCOL = NG * GP, NT = NCOL / 32;
    static_assert(CP > 8 && GP % 32 == 0 && NT <= 8, "four level groups of whole column tiles");
    static constexpr int col0(int l) { return (l / GL) * GP + (l % GL) * CP; }                     // first column of level l
    static constexpr int t0(int l) { return l + 1 < L ? col0(l + 1) / 32 : NT; }                    // first tile with a later level's column
    static constexpr int ntl(int l) { return NT - t0(l); }
    static constexpr int base(int l) { int b = 0; for (int i = 0; i < l; ++i) b += ntl(i); return b; }    // first fragment pair of level l
    static constexpr int NITEM = base(L);
    static constexpr int item_level(int it) { int l = 0; while (base(l + 1) <= it) ++l; return l; }
    static constexpr int item_tile(int it) { return t0(item_level(it)) + it - base(item_level(it)); }
    // the deal of k-slots: coordinate j of level l (column col0(l) + j) belongs to the lanes of half (column >> 2) & 1 and takes the
    // next free slot of that half
    static constexpr int half_of(int l, int j) { return ((col0(l) + j) >> 2) & 1; }
    static constexpr int slot_of(int l, int j) { int s = 0; for (int i = 0; i < j; ++i) s += half_of(l, i) == half_of(l, j) ? 1 : 0; return s; }
    static constexpr int coord_of(int l, int h, int s) { for (int j = 0; j < CP; ++j) if (half_of(l, j) == h && slot_of(l, j) == s) return j; return -1; }
    static constexpr bool deal_ok() { for (int l = 0; l < L; ++l) for (int j = 0; j < CP; ++j) if (slot_of(l, j) > 7) return false; return true; }
    static_assert(deal_ok(), "at most 8 coordinates of a level per half");
};

struct Sign3Params {
    const float *P;              // [column groups of 16][tok_pad][16]
    const float *kconst;         // [columns]
    const uint16_t *img;         // [NITEM][2][512]
    const unsigned *maxbits;
    const int *lat_flag, *exact_flag;
    const float *cb;             // [K][c]
    float *planes;               // [L][tok_pad][16]
    long long *idx;
    long long n_tokens, tok_pad;
    int K, idx_stride;
};

constexpr int S3_NW = 8;
constexpr int S3_TILE = S3_NW * 32;

// the other lane of the token (r, h) <-> (r, 1 - h)
__device__ __forceinline__ unsigned s3_other(unsigned v, int h)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // [0]: lower half everywhere; [1]: upper half everywhere
    return h ? sw[0] : sw[1];
}

template <int CP>
__global__ __launch_bounds__(S3_NW * 64) void sign_chain_kernel(Sign3Params p)
{
    using G = S3G<CP>;
    constexpr int NT = G::NT, NITEM = G::NITEM;
    if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(p.lat_flag)) != 2 ||
        __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(p.exact_flag)) != 1)
        return;                                       // not a two-point lattice with exact rows: the lattice form behind this launch runs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < NITEM * 128; i += blockDim.x)
        *reinterpret_cast<u32x4 *>(smem + 16 * i) = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.img) + 16 * (size_t)i);
    // ... and the projection's constants behind it (one float per column): read per tile from here, 16 bytes per lane and column block
    float *kc_l = reinterpret_cast<float *>(smem + (size_t)NITEM * 2048);
    for (int i = threadIdx.x; i < G::NCOL; i += blockDim.x) kc_l[i] = p.kconst[i];
    __syncthreads();
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int Tg = r3_shift_of(*p.maxbits);
    const int Tc = (Tg > 90 ? 90 : Tg < -90 ? -90 : Tg) + R3_CV_SHIFT;
    const float scale = r3_pow2(Tc), unscale = r3_pow2(-Tc), cvs = r3_pow2(R3_CV_SHIFT);
    const float mag = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.cb[(size_t)(p.K - 1) * CP])));      // every row is (+-mag)^c (pack's check)
    const long long ntile = (p.n_tokens + S3_TILE - 1) / S3_TILE;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long tile0 = tile * S3_TILE;
        unsigned tw = (unsigned)(wave * 32 + r);              // the lane's token inside the tile: 32-bit offsets against uniform bases
        asm volatile("" : "+v"(tw));                          // (opaque per tile: hipcc otherwise hoists every level's store offsets out of the tile loop -- and spills them)
        unsigned kofs = 4u * (unsigned)h;                     // (opaque too: the constants' 128 floats per lane stay loads inside the loop, not registers across it)
        asm volatile("" : "+v"(kofs));
        const unsigned twl = tile0 + tw < p.tok_pad ? tw : (unsigned)(p.tok_pad - 1 - tile0);      // rows past the workspace read its last row (never stored)
        const unsigned pofs = tw * 16u + 4u * (unsigned)h, lofs = twl * 16u + 4u * (unsigned)h;    // the lane's first float in a 16-float row: its coordinates' offsets are compile-time constants from here
        const bool live = tile0 + tw < p.n_tokens;
        // ---- the projections of every level, scaled by the products' power of two (exact), in the matrix pipe's layout
        f32x16 acc[NT];
        // (in two halves, the constants tile by tile: all 32 loads and 32 reads issued at once -- hipcc's choice -- need more than the
        //  256 registers two waves per SIMD leave a wave)
        static_for<0, 2>([&](auto half_c) {
            constexpr int TA = decltype(half_c)::value * (NT / 2), TB = decltype(half_c)::value ? NT : NT / 2;
#pragma unroll
            for (int T = TA; T < TB; ++T)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *pb = p.P + (((size_t)(2 * T + (i >> 1)) * p.tok_pad + tile0) * 16 + 8 * (i & 1));
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(pb + lofs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[T][4 * i + e] = w[e];
                }
#pragma unroll
            for (int T = TA; T < TB; ++T) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 kc = *reinterpret_cast<const f32x4 *>(kc_l + 32 * T + 8 * i + kofs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[T][4 * i + e] = (acc[T][4 * i + e] + kc[e]) * scale;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        long long *ib = p.idx + (size_t)tile0 * p.idx_stride;
        const unsigned ioff = tw * (unsigned)p.idx_stride;
        static_for<0, G::L>([&](auto l_c) {
            constexpr int l = decltype(l_c)::value, lg = l / G::GL, li = l % G::GL;
            constexpr int C0 = G::col0(l), B0 = C0 >> 3, B1 = (C0 + CP - 1) >> 3, NB = B1 - B0 + 1;
            // ---- this lane's coordinates of the level: elements (block bi, e) of the accumulators hold coordinate j0 in the lower
            // half's lanes and j0 + 4 in the upper half's (j0 a compile-time constant; outside 0 .. CP - 1: not a coordinate).
            // (hh: the lane's half, opaque per level -- as a loop invariant hipcc would compute every level's per-lane selects before
            //  the tile loop and spill them.)
            int hh = h;
            asm volatile("" : "+v"(hh));
            float q[NB * 4];
            bool mine[NB * 4];
            float n2p = 0.0f;
            unsigned raw = 0u;                         // sign bits at 20 - j0: the row index's bit CP - 1 - j after a shift by 8 (lower half) / 12 (upper)
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = B0 + bi, T = b >> 2, i = b & 3, x = bi * 4 + e;
                    const int j0 = 8 * b + e - C0;
                    mine[x] = (unsigned)(j0 + 4 * hh) < (unsigned)CP;
                    q[x] = acc[T][4 * i + e] * unscale;
                    const float qq = mine[x] ? q[x] : 0.0f;
                    n2p = __builtin_fmaf(qq, qq, n2p);
                    raw |= (q[x] > 0.0f ? 1u : 0u) << (CP + 7 - j0);      // (a shift, not a select of the bit: 27 such constants would sit in registers)
                }
            constexpr unsigned M0 = [] { unsigned m = 0; for (int j = 0; j < CP; ++j) if (G::half_of(l, j) == 0) m |= 1u << (CP - 1 - j); return m; }();
            constexpr unsigned M1 = [] { unsigned m = 0; for (int j = 0; j < CP; ++j) if (G::half_of(l, j) == 1) m |= 1u << (CP - 1 - j); return m; }();
            unsigned mk = (raw >> (8 + 4 * hh)) & (hh ? M1 : M0);
            const float n2 = n2p + __builtin_bit_cast(float, s3_other(__builtin_bit_cast(unsigned, n2p), hh));
            mk |= s3_other(mk, hh);
            const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
            mk = okq ? mk : 0u;                        // rqae/model.py:182,188: argmax over NaNs (or of a zero row's equal scores) is row 0
            {
                unsigned io = ioff;                    // (opaque per level, like hh: the row's 64-bit address is cheaper to form again than to keep -- hipcc spills it)
                asm volatile("" : "+v"(io));
                if (live && (l >> 3) == hh) ib[io + l] = (long long)mk;      // lane (r, h) writes levels 8 h .. 8 h + 7 of its token
            }
            // ---- straight-through values (rqae/model.py:218-220) of this lane's coordinates: planes, and the B operand's slots
            float cvp[NB * 4];
            float *pl0 = p.planes + ((size_t)l * p.tok_pad + tile0) * 16;      // uniform; the lane's row at tw * 16
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = B0 + bi, x = bi * 4 + e;
                    const int j0 = 8 * b + e - C0;
                    const float cv = (okq && q[x] > 0.0f) ? mag : -mag;
                    const float df = cv - q[x];
                    cvp[x] = q[x] + df;
                    if (mine[x] && live) (pl0 + pofs)[j0] = cvp[x];      // coordinate j0 (lower half) / j0 + 4 (upper half)
                }
            if (hh == 0 && live) {
                float z = 0.0f;                        // (opaque: three zeros kept in registers across the levels get spilled)
                asm volatile("" : "+v"(z));
#pragma unroll
                for (int j = CP; j < 16; ++j) (pl0 + pofs)[j] = z;      // the reconstruction reads whole 16-float rows
            }
            u32x4 bh, bl;
            {
                unsigned short ph[8], pl[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int ja = G::coord_of(l, 0, s), jb = G::coord_of(l, 1, s);      // the coordinate in slot s of either half (-1: none)
                    // element index of a coordinate: column C0 + j -> block (C0 + j) >> 3, e = (C0 + j) & 3
                    const int xa = ja >= 0 ? (((C0 + ja) >> 3) - B0) * 4 + ((C0 + ja) & 3) : -1;
                    const int xb = jb >= 0 ? (((C0 + jb) >> 3) - B0) * 4 + ((C0 + jb) & 3) : -1;
                    const float va = xa >= 0 ? cvp[xa] : 0.0f, vb = xb >= 0 ? cvp[xb] : 0.0f;
                    split2((hh ? vb : va) * cvs, ph[s], pl[s]);
                }
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    bh[w] = (unsigned)ph[2 * w] | ((unsigned)ph[2 * w + 1] << 16);
                    bl[w] = (unsigned)pl[2 * w] | ((unsigned)pl[2 * w + 1] << 16);
                }
            }
            // ---- every later column sees the level
            static_for<0, G::ntl(l)>([&](auto k_c) {
                constexpr int it = G::base(l) + decltype(k_c)::value, T = G::t0(l) + decltype(k_c)::value;
                const u32x4 ah = *reinterpret_cast<const u32x4 *>(smem + (size_t)it * 2048 + lane * 16);
                const u32x4 al = *reinterpret_cast<const u32x4 *>(smem + (size_t)it * 2048 + 1024 + lane * 16);
                acc[T] = mfma32(ah, bh, acc[T]);
                acc[T] = mfma32(al, bh, acc[T]);
                acc[T] = mfma32(ah, bl, acc[T]);
            });
            // the level's products in order: two fragment pairs ahead, then a pair's reads behind each pair's matrix instructions (left
            // alone hipcc issues all of a level's reads at once -- 100 registers -- and spills accumulators for them)
            {
                constexpr int NI = G::ntl(l), PFP = NI < 2 ? NI : 2;
                if constexpr (NI > 0) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * PFP, 0);
#pragma unroll
                    for (int i = 0; i < NI - PFP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * PFP, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            (void)lg; (void)li;
        });
    }
}

// ------------------------------------------------------------------------------------------------ pack
// exact: every codebook row k is, bit for bit, (+-mag)^c with + where k's digit is 1 (mag: the first value of row K - 1)
template <int CP>
__global__ void sign_pack_exact_kernel(const float *cb, int K, int *exact)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    bool ok = true;
    for (int j = 0; j < CP; ++j) {
        const int digit = (k >> (CP - 1 - j)) & 1;
        const float mag = cb[(size_t)(K - 1) * CP];      // one magnitude for every coordinate ("round_fsq": 1 / sqrt(c); "fsq": 1)
        const float want = digit ? mag : -mag;
        ok = ok && __builtin_bit_cast(unsigned, cb[(size_t)k * CP + j]) == __builtin_bit_cast(unsigned, want);
    }
    if (!ok) *exact = 0;
}

// fragment (item, piece): the A operand of v_mfma_f32_32x32x16_f16 -- lane (m, hh) holds column 32 T + m, k-slots 8 hh + 0..7.
// Column -> (level l2, coordinate i); slot -> coordinate j of level l (the deal above); value gneg[l][l2][i][j], zero unless l2 > l.
template <int CP>
__global__ void sign_pack_image_kernel(const float *gneg, int GB, const unsigned *maxbits, uint16_t *img)
{
    using G = S3G<CP>;
    const int T0 = r3_shift_of(*maxbits);
    const float sc = r3_pow2(T0 > 90 ? 90 : T0 < -90 ? -90 : T0);
    const size_t total = (size_t)G::NITEM * 2 * 512;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int piece = (int)((idx >> 9) & 1), it = (int)(idx >> 10);
        const int l = G::item_level(it), T = G::item_tile(it);
        const int col = 32 * T + (lane & 31), hh = lane >> 5, s = jj;
        const int lg2 = col / G::GP, rem = col % G::GP, li2 = rem / CP, i = rem % CP, l2 = lg2 * G::GL + li2;
        const int j = G::coord_of(l, hh, s);
        float v = 0.0f;
        if (rem < G::GL * CP && l2 > l && l2 < G::L && j >= 0) v = gneg[((size_t)l * G::L + l2) * GB + i * CP + j] * sc;
        unsigned short hi, lo;
        split2(v, hi, lo);
        img[idx] = piece ? lo : hi;
    }
}
