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

#ifdef W_TIMELINE   /* tools build (tools/wide_timeline.py): per wave, cycles spent in the parts of the level chain */
__device__ unsigned long long g_wide_stamps[8192 * 8];
#define W_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define W_ACC(acc, a, b) acc += (unsigned)((b) - (a))
#else
#define W_T(var) do { } while (0)
#define W_ACC(acc, a, b) do { } while (0)
#endif

constexpr int W_STT = 8;        // row tiles per ring stage
constexpr int W_NST = 3;        // ring slots: the stage being multiplied, the one being written, one spare
constexpr int R_NST = 3;        // ring slots of the reconstruction (40 KiB each at C4)
constexpr int W_NWL = 2;        // loader waves per workgroup of the level chain (four measured the same)

template <int CP> struct WideGeom {
    static constexpr int GL = CP <= 8 ? 8 : 4;                     // levels per level group (gl_of)
    static constexpr int GP = ((GL * CP + 15) / 16) * 16;          // projection outputs per group, whole 16-wide tiles
    static constexpr int NKC = (6 * CP + 31) / 32;                 // k-steps of the cosine contraction
    static constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;      // reconstruction slots per group
    static constexpr int NKG = GS / 32;                            // reconstruction k-steps per group
    static constexpr int GB = (CP * CP + 3) & ~3;
    static constexpr int STAGE = W_STT * NKC * 1024;               // bytes per ring stage
    static constexpr int CVS = CP <= 5 ? CP : 16;                  // floats per token and level in the value planes: CP (what the
                                                                   // one-group reconstruction, gram_fused_kernel MODE 3, reads) or 16
};

// Cutting dwords [d0, d1) of a row into aligned stores of 4, 2 or 1 dwords, greedily from d0: the width of the piece that STARTS at
// dword d, or 0 when d lies inside a piece that started earlier.
__host__ __device__ constexpr int wide_piece_width(int d0, int d1, int d)
{
    int pos = d0;
    while (pos < d1) {
        const int w = (pos % 4 == 0 && pos + 4 <= d1) ? 4 : (pos % 2 == 0 && pos + 2 <= d1) ? 2 : 1;
        if (pos == d) return w;
        if (pos > d) return 0;
        pos += w;
    }
    return 0;
}

__host__ __device__ inline size_t wide_levels_lds_bytes(int nkc, int nw) { return (size_t)W_NST * W_STT * nkc * 1024 + (size_t)nw * 4096; }   // nw = computing waves

// ------------------------------------------------------------------------------------------------ FSQ lattice: the argmax without the scan
// An FSQ codebook (rqae/model.py:63-72) is the lattice {linspace(-1, 1, n)}^c, row index = the digits of the coordinates (first
// coordinate most significant), each row scaled to unit length ("round_fsq") or left as it is ("fsq").  For a projection q the row
// that maximises q . row / |row| (and q . row) has, coordinate by coordinate, the sign of q and magnitudes in the order of |q|: the
// larger |q_j|, the larger |row_j| (rearrangement inequality).  Such rows are given by how many coordinates reach each magnitude:
// non-increasing thresholds c >= t_1 >= ... >= t_{nmag-1} >= 0 -- 6 candidates at n = 4, c = 5 (instead of 1024 rows), 15 at n = 5,
// c = 4, 35 at n = 8, c = 4, ONE at n = 2.  Each candidate is scored exactly as the scan kernels' rescoring scores a row -- the
// reference's q / ||q|| against the row as stored, fma chain in coordinate order, first maximum = lowest row among equal scores
// (proportional lattice points of an odd n are bit-identical unit rows: the reference's argmax takes the first) -- so every token
// whose top-1 / top-2 margin exceeds fp32 rounding gets the scan kernels' code.  A coordinate that is exactly zero ties between
// its two signs: the negative digit is the lower row.  gram_pack_lattice_kernel (gram.hip) checks at pack time that the codebook
// IS such a lattice and leaves n in the block's header; the kernels below run only then.
typedef __attribute__((address_space(3))) const float lds_cf32;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
template <typename T> struct RowPtr;                              // rows in global memory (L2) or staged in LDS
template <> struct RowPtr<const float *> {
    static __device__ __forceinline__ f32x4 quad(const float *p, size_t i) { return *reinterpret_cast<const f32x4 *>(p + i); }
};
template <> struct RowPtr<lds_cf32 *> {
    static __device__ __forceinline__ f32x4 quad(lds_cf32 *p, size_t i) { return *(lds_cf32x4 *)(p + i); }
};

// ---- the fast form of the argmax for c = 4 lattices with three or four magnitudes (n = 5: the released shape; n = 8: C5), rows of unit
// length.  The cosine of q with the candidate of thresholds (t_1 >= ... >= t_nth) is, in integer magnitudes M_k = M_0 + 2 k,
//     ( M_0 S[c] + 2 sum_k S[t_k] ) / sqrt( c M_0^2 + sum_k t_k (M_k^2 - M_{k-1}^2) ),      S[t] = the t largest |q_j| summed,
// -- two adds, one fma and one multiplication by a compile-time constant per candidate, no row fetched -- where the exact scoring
// (the reference's q / ||q|| against the stored row, an fma chain in coordinate order) costs a gather and a dozen instructions.  The
// approximate scores pick the candidate; the runner-up's distance says whether they may: within LAT_EPS (relative) of the best the
// reference's fp32 rounding could decide otherwise, and the whole wave takes the exact enumeration below.  Candidates that are
// multiples of one another (n odd: t coordinates at ONE magnitude, the rest zero; n even: every coordinate at the same magnitude) are
// one direction: one of them is enumerated, and when it wins its members are scored exactly as stored, as the exact form does (the
// stored rows of such a family are equal or an ulp apart: the reference's own rounding and its lowest-index rule decide there).
constexpr float LAT_EPS = 8e-6f;
template <int CP, int N, typename SC>
__device__ __forceinline__ bool lattice_fast(const float (&a)[CP], int row0, const int (&PW)[CP + 1], SC &&exact_score, int &brow_out)
{
    constexpr bool EVEN = N % 2 == 0;
    constexpr int NTH = (N + 1) / 2 - 1, M0 = EVEN ? 1 : 0;
    static_assert(CP == 4 && (NTH == 2 || NTH == 3), "instantiated for c = 4, n = 5 / 8");
    float S[CP + 1];
    S[0] = 0.0f;
#pragma unroll
    for (int t = 1; t <= CP; ++t) S[t] = S[t - 1] + a[t - 1];
    const float base = (float)M0 * S[CP];
    float best = -INFINITY, second = -INFINITY;
    int brow = row0, bfam = -1;                        // bfam: the winner is the family of t coordinates at one magnitude (n even: t = CP)
    static_for<0, CP + 1>([&](auto t1_c) {
        constexpr int t1 = decltype(t1_c)::value;
        static_for<0, t1 + 1>([&](auto t2_c) {
            constexpr int t2 = decltype(t2_c)::value;
            static_for<0, (NTH == 3 ? t2 + 1 : 1)>([&](auto t3_c) {
                constexpr int t3 = NTH == 3 ? decltype(t3_c)::value : 0;
                // integer norm^2; family membership
                constexpr int D1 = (M0 + 2) * (M0 + 2) - M0 * M0, D2 = (M0 + 4) * (M0 + 4) - (M0 + 2) * (M0 + 2), D3 = (M0 + 6) * (M0 + 6) - (M0 + 4) * (M0 + 4);
                constexpr int N2 = CP * M0 * M0 + t1 * D1 + t2 * D2 + t3 * D3;
                constexpr bool zero = N2 == 0;                                                    // n odd: the zero row (cosine 0: never the argmax of a non-zero q)
                constexpr bool one_mag = EVEN ? ((t1 == 0 || t1 == CP) && (t2 == 0 || t2 == CP) && (t3 == 0 || t3 == CP))
                                              : (t1 > 0 && (t2 == 0 || t2 == t1) && (t3 == 0 || t3 == t2) && !(t2 == 0 && t3 != 0));
                constexpr bool rep = EVEN ? (t1 == 0) : (t2 == 0 && t3 == 0);                      // the member that is enumerated
                if constexpr (!zero && (!one_mag || rep)) {
                    const float inv = 1.0f / __builtin_sqrtf((float)N2);      // (folded at compile time)
                    const float sc = __builtin_fmaf(2.0f, S[t1] + S[t2] + S[t3], base) * inv;
                    const int row_s = row0 + PW[t1] + PW[t2] + PW[t3];
                    const bool up = sc > best;
                    second = up ? best : __builtin_fmaxf(second, sc);
                    best = up ? sc : best;
                    brow = up ? row_s : brow;
                    bfam = up ? (one_mag ? (EVEN ? CP : t1) : -1) : bfam;
                }
            });
        });
    });
    // a family won somewhere in the wave: its members (k magnitudes up: row0 + k PW[t]) scored as stored, first maximum = lowest row
    if (__builtin_amdgcn_ballot_w64(bfam >= 0) != 0) {
        int pwf = 0;
#pragma unroll
        for (int t = 1; t <= CP; ++t) pwf = bfam == t ? PW[t] : pwf;
        float fs = -INFINITY;
        int frow = row0;
#pragma unroll
        for (int k = (EVEN ? 0 : 1); k <= NTH; ++k) {
            const int rk = row0 + k * pwf;
            const float sk = exact_score(rk);
            const bool up = (sk > fs) | ((sk == fs) & (rk < frow));
            fs = up ? sk : fs;
            frow = up ? rk : frow;
        }
        brow = bfam >= 0 ? frow : brow;
    }
    brow_out = brow;
    return (best - second) > LAT_EPS * best;           // (false for a NaN or non-positive best: the exact form decides)
}

// FN: the lattice (5 or 8 points per coordinate) this instantiation may take lattice_fast for; 0: the exact enumeration only
template <int CP, int FN = 0, typename PT>
__device__ __forceinline__ int lattice_best(const float (&q)[CP], int c, int n, PT cbf, bool fast = false)
{
    const int nth = (n + 1) / 2 - 1;                    // thresholds: magnitudes above the lowest one
    const int base_pos = n / 2, base_neg = (n & 1) ? n / 2 : n / 2 - 1;
    // the row with every coordinate at its lowest magnitude, and what one step up in magnitude of coordinate j adds to the row index
    int row0 = 0, w[CP];
    {
        int pw = 1;
#pragma unroll
        for (int j = CP - 1; j >= 0; --j) {
            if (j < c) {
                const bool pos = q[j] > 0.0f;
                row0 += (pos ? base_pos : base_neg) * pw;
                w[j] = pos ? pw : -pw;
                pw *= n;
            } else {
                w[j] = 0;
            }
        }
    }
    // one magnitude (n = 2: C4): the row IS the sign pattern -- one candidate, nothing to score (a wave-uniform fact: n is a scalar)
    if (nth == 0) return row0;
    float xn[CP];
    unit_dir<CP>(q, c, xn);
    // PW[t] = what raising the t coordinates of largest |q| by one magnitude adds: (|q_j|, step_j) sorted by descending |q_j| through a
    // compare-exchange network (5 exchanges at 4 coordinates, 9 at 5), then prefix sums.  Coordinates of exactly equal magnitude
    // may land either way round: the two rows then score the same in real arithmetic (a tie the reference decides by rounding)
    int PW[CP + 1];
    PW[0] = 0;
    float a[CP];
    if (nth > 0) {
#pragma unroll
        for (int jj = 0; jj < CP; ++jj) a[jj] = jj < c ? __builtin_fabsf(q[jj]) : -1.0f;
        auto cx = [&](auto i_c, auto j_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, j2 = decltype(j_c)::value;
            const bool sw = a[i] < a[j2];
            const float ta = sw ? a[j2] : a[i], tb = sw ? a[i] : a[j2];
            const int wa = sw ? w[j2] : w[i], wb = sw ? w[i] : w[j2];
            a[i] = ta; a[j2] = tb; w[i] = wa; w[j2] = wb;
        };
#define CX(I, J) cx(std::integral_constant<int, I>{}, std::integral_constant<int, J>{})
        if constexpr (CP == 4) {
            CX(0, 1); CX(2, 3); CX(0, 2); CX(1, 3); CX(1, 2);
        } else if constexpr (CP == 5) {
            CX(0, 1); CX(3, 4); CX(2, 4); CX(2, 3); CX(0, 3); CX(0, 2); CX(1, 4); CX(1, 3); CX(1, 2);
        } else {
            // odd-even transposition: CP rounds (wide codebooks with more than two levels per coordinate: K = 3^6 .. 3^8, 4^6)
            static_for<0, CP>([&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                static_for<0, (CP - (r & 1)) / 2>([&](auto k_c) {
                    constexpr int i = 2 * decltype(k_c)::value + (r & 1);
                    cx(std::integral_constant<int, i>{}, std::integral_constant<int, i + 1>{});
                });
            });
        }
#undef CX
#pragma unroll
        for (int r = 0; r < CP; ++r) PW[r + 1] = PW[r] + w[r];
    } else {
#pragma unroll
        for (int r = 0; r < CP; ++r) { PW[r + 1] = 0; a[r] = 0.0f; }
    }
    int brow = 0;
    float bsc = -INFINITY;
    auto score = [&](int row) __attribute__((always_inline)) {
        float sc = 0.0f;
        if (c == CP && CP % 4 == 0) {
#pragma unroll
            for (int w4 = 0; w4 < CP / 4; ++w4) {
                const f32x4 v = RowPtr<PT>::quad(cbf, (size_t)row * CP + 4 * w4);
#pragma unroll
                for (int e = 0; e < 4; ++e) sc = __builtin_fmaf(xn[4 * w4 + e], v[e], sc);
            }
        } else if (c == CP) {
            float f[CP];
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) f[jj] = cbf[row * CP + jj];
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) sc = __builtin_fmaf(xn[jj], f[jj], sc);
        } else {
            float f[CP];
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) f[jj] = cbf[row * c + (jj < c ? jj : 0)];
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) sc = __builtin_fmaf(xn[jj], jj < c ? f[jj] : 0.0f, sc);
        }
        // (bitwise, not short-circuit: hipcc turns `a > b || (a == b && r < s)` into exec-masked branches)
        const bool up = (sc > bsc) | ((sc == bsc) & (row < brow));
        bsc = up ? sc : bsc;
        brow = up ? row : brow;
    };
    // the candidates: thresholds c >= t_1 >= ... >= t_nth >= 0, row = row0 + sum_k PW[t_k]; an odometer of wave-uniform nibbles
    // (scalar registers) walks them, PW is picked by compare-and-select so that it stays in registers
    if (nth <= 1) {
        // one magnitude (n = 2: C4 -- the row is the sign pattern) or two (n = 3, 4: C1, C2 -- c + 1 candidates, unrolled, PW
        // indexed at compile time)
        const int tmax = nth ? c : 0;
        static_for<0, CP + 1>([&](auto t_c) {
            constexpr int t1 = decltype(t_c)::value;
            if (t1 <= tmax) score(row0 + PW[t1]);
        });
        return brow;
    }
    // three and more magnitudes, c = 4, rows of unit length: the approximate scores choose, unless some lane's runner-up is too close
    if constexpr (CP == 4 && (FN == 5 || FN == 8)) {
        if (fast && c == CP && n == FN) {
            auto exact_score = [&](int row) __attribute__((always_inline)) -> float {
                const f32x4 v = RowPtr<PT>::quad(cbf, (size_t)row * CP);
                float sc = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) sc = __builtin_fmaf(xn[e], v[e], sc);
                return sc;
            };
            int fr = row0;
            const bool clear = lattice_fast<CP, FN>(a, row0, PW, exact_score, fr);
            if (__builtin_amdgcn_ballot_w64(!clear) == 0) return fr;
        }
    }
    // three and more magnitudes (n >= 5: the pretrained shape, C5): the odometer walks the first nth - 1 thresholds, the last one
    // (0 .. its predecessor) is unrolled with PW indexed at compile time -- 12 vector instructions per candidate instead of 30
    const int no = nth - 1;                             // thresholds in the odometer
    unsigned tp = 0u;                                   // nibble k: threshold t_{k+1}
    for (;;) {
        int rowb = row0;
        int tlast = c;                                  // the predecessor of the unrolled threshold
        for (int k = 0; k < no; ++k) {
            const int tk = (int)((tp >> (4 * k)) & 15u);
#pragma unroll
            for (int t = 1; t <= CP; ++t) rowb += tk == t ? PW[t] : 0;
            tlast = tk;
        }
        static_for<0, CP + 1>([&](auto t_c) {
            constexpr int tl = decltype(t_c)::value;
            if (tl <= tlast) score(rowb + PW[tl]);
        });
        int k = no - 1;
        for (; k >= 0; --k) {
            const unsigned tk = (tp >> (4 * k)) & 15u;
            const unsigned lim = k == 0 ? (unsigned)c : (tp >> (4 * (k - 1))) & 15u;
            if (tk < lim) { tp += 1u << (4 * k); break; }
            tp &= ~(15u << (4 * k));
        }
        if (k < 0) break;
    }
    return brow;
}

// p.cvimg is used as float planes [levels][tok_pad][CVS] (or as operand rows: IMG below)
// (second launch bound: a budget of 256 registers, so the MFMA accumulators stay in VGPRs.  With the 512 registers one wave per SIMD
//  may take, hipcc parks them in AGPRs and pays four v_accvgpr_read + s_nop per accumulator before the maxima can read them.)
// Waves 0 .. NW-1 compute; waves NW, NW+1 only move the image: a global_load_lds occupies its wave's issue for 100-185 cycles
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), and with one computing wave per SIMD a wave that waits there is an idle matrix
// pipe -- measured: 41.5 cycles per matrix instruction with the pieces issued by the computing waves, 16.3 with no refill at all.
// MULTI: more than one level group (later groups' projections are corrected through the workspace).  NW = 4: one computing wave
// per SIMD (C4 at its stated batch); NW = 8: two, so that one wave's serial part (rescoring, Gram updates) runs behind the other's scan.
// LAT: the lattice form -- no image, no scan, no loader waves, no LDS: lattice_best above picks the row; everything after it (the
// straight-through values in the format the reconstruction reads, Gram updates, later groups' corrections, codes) is the same code.
// CEQ: c == CP as a compile-time fact (no padded coordinate: the selects that mask one disappear).
// FN: see lattice_best (the host names the lattice it found from K and c; measured: n = 8 gains a third of the chain, n = 5 loses a tenth)
template <int CP, int NW, bool MULTI = true, bool IMG = (MULTI && CP <= 5), bool LAT = false, bool CEQ = false, int FN = 0>
__global__ __launch_bounds__((LAT ? NW : NW + W_NWL) * 64, LAT ? 2 : (NW == 8 ? 3 : 2)) void gram_levels_wide_kernel(LevParams p)
{
    // the lattice form runs when the block's header says the codebook is a lattice; the scan forms launched behind it then return
    // (readfirstlane: the word is the same for every lane, and the odometer of lattice_best must stay in scalar registers -- as a
    //  per-lane value hipcc turns its loops into exec-masked ones)
    const int nlat = p.lat_flag ? __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(p.lat_flag)) : 0;
    if (LAT ? nlat < 2 : nlat >= 2) return;
    if (LAT && nlat == 2 && p.sign_flag && __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(p.sign_flag)) == 1) return;      // the sign chain ran (gram_sign3.hpp)
    using G = WideGeom<CP>;
    constexpr int GL = G::GL, GP = G::GP, NKC = G::NKC, GB = G::GB, STAGE = G::STAGE, CVS = G::CVS;
    static_assert(NKC == 1 || NKC == 3, "one or three k-steps per cosine contraction");
    // IMG: the straight-through values leave as the bf16-piece operand rows gram_reconstruct_kernel / MODE 2 read (narrow codebooks
    // outside the lean path), not as fp32 planes (gram_fused_kernel MODE 3 / gram_recon_wide_kernel)
    static_assert(!IMG || CP <= 5, "operand rows are written for narrow codebooks only");
    static_assert((W_STT * NKC) % W_NWL == 0, "every loader wave issues the same number of pieces per stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int c = CEQ ? CP : p.c;
    char *xch = smem + W_NST * STAGE + wave * 4096;
    const float *cbf0 = p.cb_f32;                      // global (L2): [1 or L] x K x c fp32, + 64 zero rows at the end
    // lattice form, narrow codebooks: the fp32 rows (K c 4 bytes: 20 KB at C2, 64 KB at C5) staged in LDS once per workgroup -- the
    // candidates' rows are gathers one after the other, and from L2 each costs a round trip (measured: 9.5 k cycles per wave and level)
    constexpr bool TAB = LAT && CP <= 5;
    lds_cf32 *ltab = (lds_cf32 *)smem;
    if constexpr (TAB) {
        const int nfl = p.K * p.c;
        __attribute__((address_space(3))) float *lw = (__attribute__((address_space(3))) float *)smem;
        for (int i = threadIdx.x * 4; i < nfl; i += blockDim.x * 4) {
            if (i + 4 <= nfl) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cbf0 + i);
                *(__attribute__((address_space(3))) f32x4 *)(lw + i) = v;
            } else {
                for (int e = 0; i + e < nfl; ++e) lw[i + e] = cbf0[i + e];
            }
        }
        // ... and behind them the Gram blocks between levels of one level group ([group][li][li2][GB]: 7 KB at C2, 16 KB at C5)
        __attribute__((address_space(3))) float *lg_w = lw + ((nfl + 3) & ~3);
        const int ngr = ((p.levels + GL - 1) / GL) * GL * GL * (GB / 4);
        for (int i = threadIdx.x; i < ngr; i += blockDim.x) {
            const int wq4 = i % (GB / 4), li2 = (i / (GB / 4)) % GL, li1 = (i / (GB / 4) / GL) % GL, gr = i / (GB / 4) / GL / GL;
            const int l1 = gr * GL + li1, l2 = gr * GL + li2;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (l1 < p.L && l2 < p.L) v = *reinterpret_cast<const f32x4 *>(p.gneg + ((size_t)l1 * p.L + l2) * GB + 4 * wq4);
            *(__attribute__((address_space(3))) f32x4 *)(lg_w + 4 * i) = v;
        }
        __syncthreads();
    }
    lds_cf32 *lgram = ltab + ((p.K * p.c + 3) & ~3);
    // rows of unit length ("round_fsq": the last row, every coordinate +1, is 1 / sqrt(c) each) let the lattice form choose by approximate
    // scores (lattice_fast); plain rows ("fsq": 1.0) keep the exact enumeration.  LAT_EXACT: tools builds, the exact form everywhere
#ifdef LAT_EXACT
    const bool lat_fast = false;
#else
    const bool lat_fast = LAT && FN != 0 && __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cbf0[(size_t)(p.K - 1) * p.c]))) < 0.999f;
#endif
    // lattice form, wide codebooks (C4): every Gram block a chain can need -- level l onto a later level l2, the upper triangle of
    // [L][L], 120 blocks of 172 floats = 82 KB at depth 16 -- staged in LDS once per workgroup.  Read from L2 by every wave for
    // itself (43 dwordx4 loads of one address per block, ~100 blocks per wave and level group) they bound the kernel on the
    // vector-memory pipe: 1.06 ms per 262 144 tokens.
    constexpr bool TABW = LAT && CP > 5;
    lds_cf32 *lgw = (lds_cf32 *)smem;
    auto tri = [&](int l1, int l2) { return l1 * p.L - l1 * (l1 + 1) / 2 + (l2 - l1 - 1); };      // block (l1 < l2) of the triangle
    if constexpr (TABW) {
        __attribute__((address_space(3))) float *gw = (__attribute__((address_space(3))) float *)smem;
        const int nblk = p.L * (p.L - 1) / 2;
        for (int i = threadIdx.x; i < nblk * (GB / 4); i += blockDim.x) {
            const int b = i / (GB / 4), w4 = i % (GB / 4);
            int l1 = 0, rem = b;                          // b -> (l1, l2): rows of the triangle have L - 1 - l1 blocks
            while (rem >= p.L - 1 - l1) { rem -= p.L - 1 - l1; ++l1; }
            const int l2 = l1 + 1 + rem;
            *(__attribute__((address_space(3))) f32x4 *)(gw + (size_t)b * GB + 4 * w4) =
                *reinterpret_cast<const f32x4 *>(p.gneg + ((size_t)l1 * p.L + l2) * GB + 4 * w4);
        }
        __syncthreads();
    }
    float *cvf = reinterpret_cast<float *>(p.cvimg);
    const int nstage = (p.KTP + W_STT - 1) / W_STT;    // KTP is a multiple of 4; W_STT of 4 too: stages hold whole groups of 4 row tiles
    constexpr int PW = (W_STT * NKC) / W_NWL;          // pieces per loader wave and stage
    const char *img = reinterpret_cast<const char *>(p.cb_img);
    const size_t img_bytes = (size_t)p.KTP * NKC * 1024;
    // piece i (< PW) of this wave's share of stage s
    // (Walking the stages in a different order per workgroup -- so that the CUs of an XCD do not all ask their L2 for the same 24 KiB at
    //  the same moment -- was measured and changes nothing; the walk is in row order, which also keeps the tie rule a plain `>`.)
    auto stage_of = [&](int s) { return s; };
    const int nlg = (p.levels + GL - 1) / GL;
    const int wq = (lane >> 2) & 3;
    // The Gram blocks (172 floats each, 120 of them per token tile: 66 KB) are the same for every lane.  Read through a pointer
    // hipcc can prove uniform they become scalar loads, which miss the 16 KB scalar cache and return one block per L2 round trip:
    // 19 % of the kernel (tools/wide_timeline.py).  An opaque per-lane zero makes them vector loads of one address -- one cache
    // line per instruction, dozens in flight.
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    const long long tiles_per_round = (long long)gridDim.x * NW;
#ifdef W_TIMELINE
    unsigned a_brow = 0, a_scan = 0, a_hand = 0, a_resc = 0, a_gram = 0, a_multi = 0, a_load = 0;
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    if (!LAT && wave >= NW) {
        // ---- loader waves: the barriers of the computing waves, stage for stage and level for level, and the image between them.
        // REGISTER-STAGED: plain global loads into registers, ds_write into the ring one stage ahead of the stage being multiplied
        // (tools build -DW_LOADER_DMA keeps the LDS-DMA form for comparison).  Stage k is loaded after barrier(k - 3), written into
        // slot k % W_NST before barrier(k - 1) and read by the computing waves after barrier(k).
        const int lw = wave - NW;
        constexpr int NR = 3;                          // register sets: loads of up to two stages in flight behind the one being written
        u32x4 R[NR][PW];
        const char *img_l = img;                      // the image of the level being streamed (per-level codebooks: one each)
        auto gload = [&](int s, u32x4 (&r)[PW]) {
#ifdef W_NO_LOAD   /* tools build, timing only (wrong results): the loaders keep the barriers and move nothing */
            return;
#endif
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int pc = lw * PW + i;
                size_t off = ((size_t)stage_of(s) * W_STT * NKC + pc) * 1024;
                off = off < img_bytes ? off : img_bytes - 1024;
                r[i] = *reinterpret_cast<const u32x4 *>(img_l + off + lane * 16);
            }
        };
        auto lwrite = [&](int s, const u32x4 (&r)[PW]) {
#ifdef W_NO_LOAD
            return;
#endif
            char *dst = smem + (s % W_NST) * STAGE + lw * PW * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < PW; ++i) *reinterpret_cast<u32x4 *>(dst + i * 1024) = r[i];
        };
        // per level: stages 0 and 1 are written before the level's first barrier; loads run two stages ahead of the writes
        auto prologue = [&]() {
            gload(0, R[0]);
            if (nstage > 1) gload(1, R[1]);
            if (nstage > 2) gload(2, R[2]);
            if (nstage > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
            else if (nstage > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lwrite(0, R[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        prologue();
        for (long long tbase = (long long)blockIdx.x * NW; tbase < p.ntiles; tbase += tiles_per_round)
            for (int l = 0; l < p.levels; ++l) {
                // invariant at the top of stage s: stage s is written; loads of stages s + 1 and s + 2 are in flight (sets
                // (s + 1) % 3 and (s + 2) % 3)
                for (int s = 0; s < nstage; s += 3) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int ss = s + u;
                        if (ss < nstage) {
                            __builtin_amdgcn_s_barrier();      // stage ss is in LDS for everyone; everyone has left stage ss - 1
                            if (ss + 3 < nstage) gload(ss + 3, R[u]);                     // set ss % 3 == u: stage ss was written from it
                            if (ss + 1 < nstage) {
                                // stage ss + 1's loads (set (u + 1) % 3) have landed; younger: those of ss + 2 and ss + 3
                                const int younger = (ss + 3 < nstage ? 2 : ss + 2 < nstage ? 1 : 0);
                                if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
                                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                lwrite(ss + 1, R[(u + 1) % 3]);
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();          // every wave has left the last stages: the ring is free
                if (!p.shared_cb) img_l = img + (size_t)(l + 1 < p.levels ? l + 1 : 0) * img_bytes;
                prologue();                            // the next level's (or tile's) first stages, behind the serial part
            }
        return;
    }
    for (long long tbase = (long long)blockIdx.x * NW; tbase < p.ntiles; tbase += tiles_per_round) {
        const bool live = tbase + wave < p.ntiles;                        // a wave without tokens keeps the workgroup's schedule
        const long long tile = live ? tbase + wave : p.ntiles - 1;
        const long long t = tile * 64 + lane;                             // < tok_pad
        for (int lg = 0; lg < nlg; ++lg) {
            W_T(t_l0);
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
            float cvh[MULTI ? GL : 1][CP];
            int codes[GL];
            if constexpr (!MULTI && !IMG) {
                // planes of the levels not run (max_layers < GL) are zeros for the reconstruction
                if (live)
                    for (int lz = p.levels; lz < GL; ++lz) {
                        float *dst = cvf + ((size_t)lz * p.tok_pad + t) * CVS;
#pragma unroll
                        for (int j = 0; j < CVS; ++j) dst[j] = 0.0f;
                    }
            }
            W_T(t_l1);
            W_ACC(a_load, t_l0, t_l1);
            static_for<0, GL>([&](auto li_c) {
                constexpr int li = decltype(li_c)::value;
                const int l = lg * GL + li;
                if (l < p.levels) {
                    W_T(t_a);
                    const float *cbf = p.shared_cb ? cbf0 : cbf0 + (size_t)l * p.K * c;      // this level's rows
                    float q[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) q[j] = Q[li][j];
                    float n2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < CP; ++j) n2 = __builtin_fmaf(q[j], q[j], n2);
                    const bool okq = (n2 > 0.0f) && (n2 < INFINITY);
                    int brow;
                    if constexpr (LAT) {
                        if constexpr (TAB) brow = lattice_best<CP, FN>(q, c, nlat, ltab, lat_fast);
                        else brow = lattice_best<CP, FN>(q, c, nlat, cbf, lat_fast);
                    } else {
                    brow = [&]() __attribute__((always_inline)) -> int {
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
                    W_T(t_b);
                    // The scores of a row tile are reduced one tile LATE: while tile t's 12 matrix instructions issue, the vector
                    // pipe takes the maxima of tile t - 1 out of the other accumulator set (an MFMA leaves half of its 16 issue
                    // cycles to other instructions; maxima placed behind a tile's own MFMAs wait out the last result's latency and
                    // then hold the issue port -- with one computing wave per SIMD that is an idle matrix pipe).
                    f32x4 scA[4], scB[4];
#pragma unroll
                    for (int T = 0; T < 4; ++T) scB[T] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    int pgrp = 0x7fffffff;             // row group of the pending tile (none yet: never taken)
                    auto mm = [&](const u32x4 (&a)[NKC], f32x4 (&sc)[4]) {
#ifdef W_NO_MFMA   /* tools build, timing only (wrong results): no matrix instructions -- what the refill alone takes */
                        sc[0] = f32x4{f32_of(a[0][0]), f32_of(a[1][1]), f32_of(a[2][2]), 0.f};
                        return;
#endif
                        // k-step outermost: four independent accumulators (one per token tile) between dependent instructions
#pragma unroll
                        for (int T = 0; T < 4; ++T) sc[T] = mfma16(a[0], B[T][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int ks = 1; ks < NKC; ++ks)
#pragma unroll
                            for (int T = 0; T < 4; ++T) sc[T] = mfma16(a[ks], B[T][ks], sc[T]);
                    };
                    // maxima of the pending tile (position pi = 0..3 inside its group of 4 row tiles); after the group's last tile the
                    // running (maximum, group) per lane and token tile
                    auto red = [&](const f32x4 (&sc)[4], int pi) {
#pragma unroll
                        for (int T = 0; T < 4; ++T) {
                            const float m012 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(sc[T][0], sc[T][1]), sc[T][2]);
                            if (pi == 0) mx[T] = __builtin_elementwise_maximum(m012, sc[T][3]);
                            else mx[T] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m012, sc[T][3]), mx[T]);
                        }
                        if (pi == 3) {
#pragma unroll
                            for (int T = 0; T < 4; ++T) {
                                const bool up = mx[T] > best[T];          // groups arrive in row order: the first maximum keeps ties
                                best[T] = up ? mx[T] : best[T];
                                bgrp[T] = up ? pgrp : bgrp[T];
                            }
                        }
                    };
                    for (int s = 0; s < nstage; ++s) {
#ifndef W_NO_BAR   /* tools build, timing only (races): no barrier per stage */
                        __builtin_amdgcn_s_barrier();      // stage s is in LDS (the loaders waited for it); everyone has left stage s - 1
#endif
                        const char *sb = smem + (s % W_NST) * STAGE + lane * 16;
                        const int sx = stage_of(s);            // the image stage in this ring slot
                        int gcount = (p.KTP - sx * W_STT) / 4;
                        gcount = gcount < W_STT / 4 ? gcount : W_STT / 4;
                        // The fragments of a row tile (3 KiB: NKC ds_read_b128 per lane) are read one tile ahead into two alternating
                        // register sets; the sched_group_barriers pin the order: next tile's reads, then the matrix instructions with
                        // the pending tile's maxima between them
                        u32x4 fa[NKC], fb[NKC];
                        auto ldf = [&](u32x4 (&f)[NKC], int tl) {
#ifdef W_NO_LDS   /* tools build, timing only (wrong results): the stage's first two row tiles' fragments for every tile */
                            if (tl > 1) return;
#endif
#pragma unroll
                            for (int ks = 0; ks < NKC; ++ks) f[ks] = *reinterpret_cast<const u32x4 *>(sb + (tl * NKC + ks) * 1024);
                        };
                        auto order = [&]() {
                            __builtin_amdgcn_sched_group_barrier(0x100, NKC, 0);
                            if constexpr (NKC == 1) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                                }
                                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NKC - 8, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        const int ntl = gcount * 4;
                        ldf(fa, 0);
                        for (int gi = 0; gi < gcount; ++gi) {
                            const int grp = sx * (W_STT / 4) + gi;
                            // tile 0 of the group: pending = the last tile (3) of the group before (or nothing)
                            ldf(fb, gi * 4 + 1);
                            mm(fa, scA);
                            red(scB, 3);
                            order();
                            pgrp = grp;
                            ldf(fa, gi * 4 + 2);
                            mm(fb, scB);
                            red(scA, 0);
                            order();
                            ldf(fb, gi * 4 + 3);
                            mm(fa, scA);
                            red(scB, 1);
                            order();
                            ldf(fa, gi * 4 + 4 < ntl ? gi * 4 + 4 : ntl - 1);
                            mm(fb, scB);
                            red(scA, 2);
                            order();
                        }
                    }
                    red(scB, 3);                           // the last tile of the level
                    W_T(t_c);
                    W_ACC(a_scan, t_b, t_c);
                    __builtin_amdgcn_s_barrier();          // every wave has left the last stages: the ring is free (the loaders refill it)
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
                    W_T(t_d);
                    W_ACC(a_hand, t_c, t_d);
                    // ---- the 16 rows behind that maximum, rescored in fp32 on the reference's q / ||q|| (k-ordered fma chain): first
                    // maximum wins
                    float xn[CP];
                    unit_dir<CP>(q, c, xn);
                    int brow = 64 * bgr + 4 * bg;     // (shadows nothing: the caller takes the lambda's result)
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
                    W_T(t_e0);
                    W_ACC(a_resc, t_d, t_e0);
                    return brow;
                    }();
                    }
                    if (!okq) brow = 0;
                    codes[li] = brow;
                    W_T(t_e);
                    if constexpr (LAT) W_ACC(a_resc, t_a, t_e);
                    // ---- straight-through value (rqae/model.py:218-220), as fp32 for the reconstruction
                    float cvp[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) {
                        float cv;
                        if constexpr (TAB) cv = j < c ? ltab[brow * c + j] : 0.0f;
                        else cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                        const float df = cv - q[j];
                        cvp[j] = q[j] + df;
                        if constexpr (MULTI) cvh[li][j] = cvp[j];
                    }
                    if constexpr (IMG) {
                        // operand rows of gram_reconstruct_kernel ([k-step][tok_pad][16 dwords]; slot (li CP + j) 6 + term, two bf16 per
                        // dword: hi|mid, hi|lo, hi|mid): a level's 12 dwords are three 16-byte pieces, none of which straddles a row
                        if (live) {
                            unsigned dw[3 * CP];
#pragma unroll
                            for (int j = 0; j < CP; ++j) {
                                unsigned h, m, lo;
                                split3(cvp[j], h, m, lo);
                                dw[3 * j] = pack_hi16(h, m);
                                dw[3 * j + 1] = pack_hi16(h, lo);
                                dw[3 * j + 2] = dw[3 * j];
                            }
                            // the level's 3 CP dwords start at dword 3 CP li of the group row: the widest aligned stores that fit (12
                            // dwords at CP 4 are three 16-byte pieces; 15 at CP 5 fall as the level's position has it), none across a
                            // k-step row (16 dwords); then zeros up to the end of the k-step after the chain's or the group's last level
                            auto put = [&](int dpos, auto w_c, unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
                                constexpr int W = decltype(w_c)::value;
                                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                                unsigned *dst = p.cvimg + ((size_t)(lg * G::NKG + dpos / 16) * p.tok_pad + t) * 16 + dpos % 16;
                                if constexpr (W == 4) *reinterpret_cast<u32x4 *>(dst) = u32x4{a0, a1, a2, a3};
                                else if constexpr (W == 2) *reinterpret_cast<u32x2 *>(dst) = u32x2{a0, a1};
                                else *dst = a0;
                            };
                            constexpr int D0 = 3 * CP * li, D1 = 3 * CP * (li + 1);
                            static_for<0, 3 * CP>([&](auto i_c) {
                                constexpr int i = decltype(i_c)::value, dpos = D0 + i;
                                // greedy cut: this dword starts a piece iff no wider piece that began earlier covers it
                                constexpr int w = wide_piece_width(D0, D1, dpos);
                                if constexpr (w == 4) put(dpos, std::integral_constant<int, 4>{}, dw[i], dw[i + 1], dw[i + 2], dw[i + 3]);
                                else if constexpr (w == 2) put(dpos, std::integral_constant<int, 2>{}, dw[i], dw[i + 1], 0u, 0u);
                                else if constexpr (w == 1) put(dpos, std::integral_constant<int, 1>{}, dw[i], 0u, 0u, 0u);
                            });
                            constexpr int DE = (D1 + 15) / 16 * 16;
                            if (l == p.levels - 1 || li == GL - 1) {
                                static_for<0, 16>([&](auto i_c) {
                                    constexpr int dpos = D1 + decltype(i_c)::value;
                                    if constexpr (dpos < DE) {
                                        constexpr int w = wide_piece_width(D1, DE, dpos);
                                        if constexpr (w == 4) put(dpos, std::integral_constant<int, 4>{}, 0u, 0u, 0u, 0u);
                                        else if constexpr (w == 2) put(dpos, std::integral_constant<int, 2>{}, 0u, 0u, 0u, 0u);
                                        else if constexpr (w == 1) put(dpos, std::integral_constant<int, 1>{}, 0u, 0u, 0u, 0u);
                                    }
                                });
                            }
                        }
                    } else if (live) {
                        float *dst = cvf + ((size_t)l * p.tok_pad + t) * CVS;
                        if constexpr (CVS == CP) {
#pragma unroll
                            for (int j = 0; j < CP; ++j) dst[j] = cvp[j];      // a wave's 64 lanes: 64 x CP contiguous floats
                        } else {
#pragma unroll
                            for (int v = 0; v < CVS / 4; ++v)
                                *reinterpret_cast<f32x4 *>(dst + 4 * v) = f32x4{4 * v < CP ? cvp[4 * v < CP ? 4 * v : 0] : 0.0f,
                                                                                 4 * v + 1 < CP ? cvp[4 * v + 1 < CP ? 4 * v + 1 : 0] : 0.0f,
                                                                                 4 * v + 2 < CP ? cvp[4 * v + 2 < CP ? 4 * v + 2 : 0] : 0.0f,
                                                                                 4 * v + 3 < CP ? cvp[4 * v + 3 < CP ? 4 * v + 3 : 0] : 0.0f};
                        }
                    }
                    // ---- later levels of this group see the level just applied
#pragma unroll
                    for (int li2 = li + 1; li2 < GL; ++li2) {
                        const int l2 = lg * GL + li2;
                        if (l2 < p.levels) {
                            const f32x4 *gb4 = reinterpret_cast<const f32x4 *>(p.gneg + ((size_t)l * p.L + l2) * GB + vzero);
                            // eight 16-byte pieces (32 values) at a time: the whole block at once would take 172 registers
#pragma unroll
                            for (int w0 = 0; w0 < GB / 4; w0 += 8) {
                                f32x4 gv[8];
#pragma unroll
                                for (int w = 0; w < 8; ++w)
                                    if (w0 + w < GB / 4) {
                                        if constexpr (TAB) gv[w] = *(lds_cf32x4 *)(lgram + ((lg * GL + li) * GL + li2) * GB + 4 * (w0 + w));      // broadcast read
                                        else if constexpr (TABW) gv[w] = *(lds_cf32x4 *)(lgw + tri(l, l2) * GB + 4 * (w0 + w));
                                        else gv[w] = gb4[w0 + w];
                                    }
#pragma unroll
                                for (int w = 0; w < 8; ++w)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const int f = 4 * (w0 + w) + e;
                                        if (w0 + w < GB / 4 && f < CP * CP) Q[li2][f / CP] = __builtin_fmaf(gv[w][e], cvp[f % CP], Q[li2][f / CP]);
                                    }
                            }
                        }
                    }
                    W_T(t_f);
                    W_ACC(a_gram, t_e, t_f);
                } else {
                    codes[li] = 0;
                    if constexpr (MULTI) {
#pragma unroll
                        for (int j = 0; j < CP; ++j) cvh[li][j] = 0.0f;
                    }
                }
            });
            if (live && t < p.n_tokens) {
#pragma unroll
                for (int li = 0; li < GL; ++li)
                    if (lg * GL + li < p.levels) p.idx[t * p.idx_stride + lg * GL + li] = (long long)codes[li];
            }
            // ---- projections of the later level groups see this group's levels
            W_T(t_m0);
            if constexpr (MULTI)
            if (live)
                for (int l2 = (lg + 1) * GL; l2 < p.levels; ++l2) {
                    const int o0 = (l2 / GL) * GP + (l2 % GL) * CP;
                    float v[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) v[j] = p.P[((size_t)((o0 + j) >> 4) * p.tok_pad + t) * 16 + ((o0 + j) & 15)];
#pragma unroll
                    for (int li = 0; li < GL; ++li) {
                        const f32x4 *gb4 = reinterpret_cast<const f32x4 *>(p.gneg + ((size_t)(lg * GL + li) * p.L + l2) * GB + vzero);
                        // half a block (22 pieces, 88 registers) per round trip: the chain is over (Q is dead), the registers are free, and
                        // what this loop pays for is L2 round trips, not instructions
                        constexpr int HB = (GB / 4 + 1) / 2;
#pragma unroll
                        for (int w0 = 0; w0 < GB / 4; w0 += HB) {
                            f32x4 gv[HB];
#pragma unroll
                            for (int w = 0; w < HB; ++w)
                                if (w0 + w < GB / 4) {
                                    if constexpr (TABW) gv[w] = *(lds_cf32x4 *)(lgw + tri(lg * GL + li, l2) * GB + 4 * (w0 + w));
                                    else gv[w] = gb4[w0 + w];
                                }
#pragma unroll
                            for (int w = 0; w < HB; ++w)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int f = 4 * (w0 + w) + e;
                                    if (w0 + w < GB / 4 && f < CP * CP) v[f / CP] = __builtin_fmaf(gv[w][e], cvh[li][f % CP], v[f / CP]);
                                }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < CP; ++j) p.P[((size_t)((o0 + j) >> 4) * p.tok_pad + t) * 16 + ((o0 + j) & 15)] = v[j];
                }
            W_T(t_m1);
            W_ACC(a_multi, t_m0, t_m1);
        }
    }
#ifdef W_TIMELINE
    if (lane == 0) {
        unsigned long long *o = g_wide_stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - t_entry;
        o[1] = a_load; o[2] = a_brow; o[3] = a_scan; o[4] = a_hand; o[5] = a_resc; o[6] = a_gram; o[7] = a_multi;
    }
#endif
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
__host__ __device__ constexpr size_t wide_recon_lds_bytes() { return (size_t)R_NST * NLG * WideGeom<CP>::NKG * 1024 + 8 * 16 * 208; }

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
    char *xb = smem + R_NST * STAGE + wave * (16 * XROW);
    constexpr int PW = (NKS + NW - 1) / NW;           // pieces per wave and stage (waves past the last piece repeat it)
    auto issue1 = [&](int et_in, int slot, int i) {
        const int et = et_in < p.NET ? et_in : p.NET - 1;
        int q = wave + i * NW;
        q = q < NKS ? q : NKS - 1;
        dma_piece(uniform_ptr(p.wout_img + ((size_t)et * p.NKSC + q) * 512), lds0 + slot * STAGE + q * 1024, (unsigned)lane * 16u);
    };
    auto issue = [&](int et_in, int slot) {
#pragma unroll
        for (int i = 0; i < PW; ++i) issue1(et_in, slot, i);
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
            const int nslot = st == 0 ? 2 : st - 1;    // slot (e + 2) % 3 was read last in stage e - 1: stage e + 2 goes there, one piece
                                                       // behind each group of matrix instructions (a burst waits on the CU's address queue)
            const char *wb = smem + st * STAGE + lane * 16;
            f32x4 acc = *reinterpret_cast<const f32x4 *>(p.bias + 16 * e + 4 * g);
            f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            // four fragments at a time, read one group ahead into two alternating register sets (sched_barrier pins the order: left
            // alone hipcc either hoists all NKS reads above the first MFMA -- 160 more registers -- or reads, waits, multiplies)
            u32x4 fr0[4], fr1[4];
            auto ldg = [&](u32x4 (&f)[4], int k0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k0 + i < NKS) f[i] = *reinterpret_cast<const u32x4 *>(wb + (k0 + i) * 1024);
            };
            auto mmg = [&](const u32x4 (&f)[4], int k0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k0 + i < NKS) {
                        if (i & 1) acc2 = mfma16(f[i], cvB[k0 + i], acc2);
                        else acc = mfma16(f[i], cvB[k0 + i], acc);
                    }
            };
            ldg(fr0, 0);
#pragma unroll
            for (int k0 = 0; k0 < NKS; k0 += 8) {
                if (k0 + 4 < NKS) ldg(fr1, k0 + 4);
                mmg(fr0, k0);
                __builtin_amdgcn_sched_barrier(0);
                if (k0 / 8 < PW) issue1(e + 2, nslot, k0 / 8);
                if (k0 + 8 < NKS) ldg(fr0, k0 + 8);
                if (k0 + 4 < NKS) mmg(fr1, k0 + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = (NKS + 7) / 8; i < PW; ++i) issue1(e + 2, nslot, i);
            acc = f32x4{acc[0] + acc2[0], acc[1] + acc2[1], acc[2] + acc2[2], acc[3] + acc2[3]};
            if (tok_ok) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(p.out + (size_t)tok * p.d + 16 * e + 4 * g));
            st = st == 2 ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated stages past the last element tile
        __builtin_amdgcn_s_barrier();                       // the ring is free for the next tile
    }
}
