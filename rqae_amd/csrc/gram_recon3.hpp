// gram_recon3.hpp -- round 4's reconstruction of the matrix-core arm (included by recon3.hip inside its anonymous namespace).
//
//   out[tok][e] = sum_l ( W_out[l][e][:] . cv_l[tok][:] + b_out[l][e] )          (rqae/model.py:221,224 as the decode sum, :236-248)
//
// What changed against gram_fused_kernel<.., MODE 3> / gram_reconstruct_kernel / gram_recon_wide_kernel:
//   * operands are TWO fp16 pieces per fp32 value instead of three bf16 pieces: hi = rn16(v), lo = rn16(v - hi) leave a remainder
//     below 2^-24 |v| (fp16 carries 11 significant bits, bf16 8), so the three products hi*hi + hi*lo + lo*hi reproduce the fp32
//     product to ~2^-23 where the bf16 form needed six -- HALF the matrix instructions at the same fp32-level accuracy.  fp16's
//     narrow exponent is met by exact power-of-two scalings: the W_out image is scaled at pack time so that its largest magnitude
//     sits in [2^14, 2^15) (gram_pack_absmax_kernel leaves the exponent in the block's header), the straight-through values by
//     2^R3_CV_SHIFT (unit codebook rows: |cv| <= 1 up to one ulp of the projection); the accumulator starts at the scaled bias sum
//     and is multiplied by the inverse scale once at the end -- powers of two, so no rounding is added;
//   * v_mfma_f32_32x32x16_f16: a 1-KiB operand fragment feeds 32 x 32 x 16 products (twice what a 16x16x32 fragment feeds), so the
//     W_out fragments can come from LDS once per 32-token wave at a third of the LDS array's rate, and one accumulator chain needs
//     no interleaving (MI355X_MICROARCH.md, cycle constants);
//   * the pieces of a wave's 32 tokens for ALL k-steps stay in registers (B operand: KS x 8 registers), built once per token tile
//     from the level chain's fp32 planes through the wave's LDS buffer; any depth up to 16 k-steps (256 values) -- C2, C4, C5 and
//     the released shape at depth <= 64 take the same kernel (the three earlier kernels split these shapes between them);
//   * a finished element tile (32 tokens x 32 elements) goes through the wave's LDS buffer with every lane writing and leaves as 8
//     token rows x 128 bytes per store instruction, non-temporal (see the kernel's comment for why not wider).
// One workgroup = 8 waves x 32 tokens (one workgroup per CU, two waves per SIMD), persistent over 256-token tiles.  The W_out
// image streams L2 -> LDS by LDS-DMA through a ring of r3_ring(KS) slots, G element tiles (all k-steps, both pieces) per stage; counted
// vmcnt waits and one raw barrier per stage, as in gram_fused_kernel.  Tier-2 parity like the rest of the arm.

constexpr int R3_NW = 8;                 // waves per workgroup
constexpr int R3_TILE = R3_NW * 32;      // tokens per tile
constexpr int R3_WBUF = 4608;            // bytes of LDS per wave: piece build / transposition of one element tile (32 rows x 144)
constexpr int R3_BK = 2;                 // k-steps per trip of the piece build through the wave's buffer

// k-steps (of 16 values) the reconstruction of `levels` levels runs over, and the instantiated depths
__host__ __device__ inline int r3_ksteps(int c, int levels) { return (c * levels + 15) / 16; }
inline int r3_ks_template(int ks)
{
    const int cand[] = {2, 3, 4, 8, 13, 16};
    for (int v : cand)
        if (ks <= v) return v;
    return 0;
}
// element tiles per ring stage: stages of 16-32 KiB (three of them beside 36 KiB of wave buffers and the bias sums)
__host__ __device__ constexpr int r3_group(int KS, int TS = 1) { return TS == 2 ? (KS <= 4 ? 2 : 1) : KS <= 4 ? 4 : KS <= 8 ? 2 : 1; }
// ring slots: the stage being multiplied and the ones on their way.  (Four slots at C4's depth -- a stage there is ONE element tile --
// were measured and change nothing: 0.385 / 0.383 ms, tools/r04_run_n.sh; what a stage waits for is not the image.)
#ifndef R3_RING13
#define R3_RING13 3
#endif
__host__ __device__ constexpr int r3_ring(int KS) { return KS == 13 ? R3_RING13 : 3; }
__host__ __device__ inline size_t r3_lds_bytes(int d, int KS, int TS = 1) { return pad1k((size_t)d * 4) + (size_t)r3_ring(KS) * r3_group(KS, TS) * KS * 2048 + (size_t)R3_NW * R3_WBUF; }

// ------------------------------------------------------------------------------------------------ pack
// W_out image for v_mfma_f32_32x32x16_f16: fragment (mt, ks, piece) = 1 KiB = the A operand: lane l holds element 32 mt + (l & 31),
// values k = 16 ks + 8 (l >> 5) + j, j = 0..7; value k = level k / c, component k % c (tight: no padding between levels), zero past
// L c; every weight scaled by 2^T (T from the header word gram_pack_absmax_kernel left).
__global__ void gram_pack_wout3_kernel(int d, int c, int L, const float *w_out, const unsigned *maxbits, uint16_t *img, int KS3)
{
    const int T = r3_shift_of(*maxbits);
    const float sc = r3_pow2(T > 100 ? 100 : T < -100 ? -100 : T);
    const int NMT = d / 32;
    const size_t total = (size_t)NMT * KS3 * 2 * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t f = i >> 9;
        const int piece = (int)(f & 1); f >>= 1;
        const int ks = (int)(f % KS3);
        const int mt = (int)(f / KS3);
        const int e = 32 * mt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
        float v = 0.0f;
        if (k < L * c) v = w_out[((size_t)(k / c) * d + e) * c + (k % c)] * sc;
        unsigned short hi, lo;
        split2(v, hi, lo);
        img[i] = piece ? lo : hi;
    }
}

// ------------------------------------------------------------------------------------------------ kernel
struct Rec3Params {
    const uint16_t *wimg;        // [d / 32][KS3][2][512]
    const unsigned *maxbits;     // the header word of gram_pack_absmax_kernel (the image's scale)
    const float *bias;           // bias_prefix[levels - 1][d]
    const float *cvf;            // fp32 planes [level][tok_pad][cvs]
    float *out;                  // [n_tokens][d]
    long long n_tokens, tok_pad;
    int d, c, levels, KS3;       // KS3: k-steps per element tile in the image (the block's full depth)
    int cvs;                     // floats per token and level in the planes (c <= cvs)
};

#ifndef R3_ABL     /* tools builds: timing-only ablations (wrong results): 1 = no matrix instructions, 2 = no global stores, 4 = no W_out DMA */
#define R3_ABL 0
#endif

// KS: k-steps resident in registers (>= r3_ksteps(c, levels)).
// CPF: 4 / 5 = the planes hold exactly CPF floats per token and level (the lean chains) and the NEXT tile's values are fetched into
// registers before the current tile's last stores are issued -- so no load ever waits behind a tile's stores (loads and stores
// retire through one in-order counter); 0 = any plane width (p.cvs), values loaded at the tile's start.
// The W_out stages form ONE stream over all tiles of the workgroup (the stage after a tile's last one is the next tile's first --
// the same image), so nothing is drained or re-primed at a tile boundary.
// Output: ONE element tile (32 tokens x 32 elements) at a time -- the accumulator goes through the wave's LDS buffer with every lane
// writing (a ds_write_b128 occupies the LDS for its full 1-KiB slot however few lanes are active: the first form of this kernel,
// with quarter-active writes of eight tiles at once, spent 0.33 of its 0.49 ms there), and leaves as 8 token rows x 128 bytes per
// store instruction (5.82 TB/s on the full chip against 6.07 for 1-KiB rows, profiles/r02_store_patterns.txt).
// TS: sets of 32 tokens per wave (1 or 2).  With two, a weight fragment read from LDS feeds six matrix instructions instead of three and a
// 512-token tile passes the image, its LDS-DMA and its barriers once: the stage's fixed work per token halves (the pieces of both sets
// must fit the registers: KS <= 8).
template <int KS, int CPF, int TS = 1>
__global__ __launch_bounds__(R3_NW * 64) void gram_recon3_kernel(Rec3Params p)
{
    constexpr int G = r3_group(KS, TS);                           // element tiles per stage: 4, 2 or 1
    constexpr int TILE = R3_TILE * TS;                            // tokens per tile
    constexpr int RD = r3_ring(KS);                               // ring slots
    constexpr int STAGE = G * KS * 2048;
    constexpr int NPC = G * KS * 2;                               // 1-KiB pieces per stage
    constexpr int NPB = NPC / R3_NW, NPX = NPC % R3_NW;           // per wave: NPB, one more on waves < NPX
    constexpr int NSS = 4 * G * TS;                               // store instructions per wave and stage (4 per element tile and token set)
    constexpr bool PREF = CPF != 0;
    constexpr int LM = PREF ? (16 * KS) / CPF : 1;                // levels a prefetching instantiation can hold (c == CPF)
    constexpr int NL = PREF ? (32 * CPF + 63) / 64 : 1;           // plane floats per lane and level (a wave's 32 tokens x CPF, lane-contiguous)
    constexpr int TRS = 144;                                      // transposition rows: 32 elements x 4 bytes + 16
    static_assert(32 * TRS <= R3_WBUF, "the tile's rows fit the wave's buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned bias_bytes = (unsigned)pad1k((size_t)p.d * 4);
    const unsigned off_ring = bias_bytes;
    char *wbuf = smem + off_ring + RD * STAGE + wave * R3_WBUF;
    const int Tsh = r3_shift_of(*p.maxbits);
    const int Tc = Tsh > 100 ? 100 : Tsh < -100 ? -100 : Tsh;
    const float bsc = r3_pow2(Tc) * (float)(1 << R3_CV_SHIFT);      // accumulators live at scale 2^(T + shift) ...
    const float isc = 1.0f / bsc;                                    // ... an exact power of two either way
    const int nstg = p.d / (32 * G);                                // stages per tile (d % 128 == 0: whole stages, whole pairs)
    const int kval = p.c * p.levels;                                // values in use
    const long long ntile = (p.n_tokens + TILE - 1) / TILE;
    if ((long long)blockIdx.x >= ntile) return;
    // bias sums, scaled, once per workgroup
    for (int i = tid; i < (int)(bias_bytes >> 2); i += R3_NW * 64) reinterpret_cast<float *>(smem)[i] = i < p.d ? p.bias[i] * bsc : 0.0f;
    auto issue = [&](int s_in, int slot) {
        if (R3_ABL & 4) return;
        const int s = s_in < nstg ? s_in : s_in - nstg;            // the stream wraps round: the next tile starts with the same image
        const int mt0 = s * G;
        for (int q = wave; q < NPC; q += R3_NW) {
            const int gi = q / (KS * 2), rem = q - gi * (KS * 2);
            int ks = rem >> 1;
            ks = ks < p.KS3 ? ks : p.KS3 - 1;                      // k-steps past the image's depth (KS > KS3): any fragment -- their token pieces are zero
            dma_piece(uniform_ptr(p.wimg + (((size_t)(mt0 + gi) * p.KS3 + ks) * 2 + (rem & 1)) * 512), lds0 + off_ring + slot * STAGE + q * 1024, lane16);
        }
    };
    // the wave's rows of a tile: first row, how many exist (rows past the batch re-read the last one; never stored)
    auto rows_of = [&](long long tile, int ts, long long &trow0, int &rows_ok) {
        const long long tile0 = tile * TILE;
        const long long tile_end = tile0 + TILE < p.n_tokens ? tile0 + TILE : p.n_tokens;
        const long long tw0 = tile0 + (long long)(wave * TS + ts) * 32;
        const bool act = tw0 < tile_end;
        rows_ok = act ? (int)(tile_end - tw0 < 32 ? tile_end - tw0 : 32) : 1;
        trow0 = act ? tw0 : tile_end - 1;
    };
    // PREF: the raw values of a tile, lane-contiguous per level (idx = it 64 + lane over the wave's 32 x CPF floats)
    float raw[TS * LM * NL];
    auto load_raw = [&](long long tile) {
        if constexpr (PREF) {
          static_for<0, TS>([&](auto ts_c) {
            constexpr int ts = decltype(ts_c)::value;
            long long trow0; int rows_ok;
            rows_of(tile, ts, trow0, rows_ok);
            static_for<0, LM>([&](auto l_c) {
                constexpr int l = decltype(l_c)::value;
                if (l < p.levels) {
                    const float *pl = p.cvf + ((size_t)l * p.tok_pad + trow0) * CPF;
#pragma unroll
                    for (int it = 0; it < NL; ++it) {
                        const int idx = it * 64 + lane;
                        const int tk = idx / CPF, comp = idx - tk * CPF;
                        const int tsrc = tk < rows_ok ? tk : rows_ok - 1;
                        raw[(ts * LM + l) * NL + it] = idx < 32 * CPF ? pl[tsrc * CPF + comp] : 0.0f;
                    }
                }
            });
          });
        }
    };
    load_raw(blockIdx.x);
    __syncthreads();
    static_for<0, RD - 1>([&](auto i_c) { issue(decltype(i_c)::value, decltype(i_c)::value); });
    int slot = 0;
    int sdone = 0;                // stages done so far (saturating at RD - 1): the first ones of the kernel have fewer stores behind them
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long tile0 = tile * TILE;
        const long long tile_end = tile0 + TILE < p.n_tokens ? tile0 + TILE : p.n_tokens;
        const long long tw0 = tile0 + (long long)wave * 32 * TS;       // the wave's first token (set ts: + 32 ts)
        const bool active = tw0 < tile_end;
        const bool fullw = tw0 + 32 * TS <= tile_end;
        // ---- this wave's token pieces (B operands), R3_BK k-steps per trip through the wave's buffer: rows of 32 BK hi + 32 BK lo bytes
        u32x4 Ahi[TS][KS], Alo[TS][KS];
        static_for<0, TS>([&](auto ts_c) {
            constexpr int ts = decltype(ts_c)::value;
            constexpr int VT = 16 * R3_BK;                          // values per trip
            constexpr int HB = 2 * VT;                              // bytes of a row's hi pieces; the lo pieces follow
            constexpr int RS = 2 * HB + 16;                         // row stride: 16 bytes x an odd number (conflict-free b128 reads)
            static_assert(32 * RS <= R3_WBUF, "a trip's rows fit the wave's buffer");
            long long trow0; int rows_ok;
            rows_of(tile, ts, trow0, rows_ok);
            static_for<0, (KS + R3_BK - 1) / R3_BK>([&](auto t_c) {
                constexpr int TR = decltype(t_c)::value;
                const int k0 = VT * TR;                             // first value of this trip
                // zero the trip's rows first where values past the depth in use fall into it (stale bytes must not reach an MFMA)
                if (k0 + VT > kval) {
#pragma unroll
                    for (int i = 0; i < (32 * RS / 16 + 63) / 64; ++i) {
                        const int o = (i * 64 + lane) * 16;
                        if (o < 32 * RS) *reinterpret_cast<u32x4 *>(wbuf + o) = u32x4{0u, 0u, 0u, 0u};
                    }
                    wave_sync_lds();
                }
                auto put = [&](float v, int tk, int k) {
                    unsigned short hi, lo;
                    split2(v * (float)(1 << R3_CV_SHIFT), hi, lo);
                    *reinterpret_cast<unsigned short *>(wbuf + tk * RS + 2 * k) = hi;
                    *reinterpret_cast<unsigned short *>(wbuf + tk * RS + HB + 2 * k) = lo;
                };
                if constexpr (PREF) {
                    static_for<0, LM>([&](auto l_c) {
                        constexpr int l = decltype(l_c)::value;
                        if (l < p.levels) {
#pragma unroll
                            for (int it = 0; it < NL; ++it) {
                                const int idx = it * 64 + lane;
                                const int tk = idx / CPF, comp = idx - tk * CPF;
                                const int k = l * p.c + comp - k0;
                                if (idx < 32 * CPF && comp < p.c && k >= 0 && k < VT) put(raw[(ts * LM + l) * NL + it], tk, k);
                            }
                        }
                    });
                } else {
                    // any plane width: the planes of the levels with a value in this trip are rows of 64 floats (NR per level, one float per
                    // lane and row); up to 32 rows are fetched as ONE batch, then cut into pieces -- one load, one wait, one value at a
                    // time (the first form of this branch) paid a memory round trip per value: 128 of them per tile at C4, 0.1 ms of 0.33
                    const int l_lo = k0 / p.c;
                    int l_hi = (k0 + VT - 1) / p.c;
                    l_hi = l_hi < p.levels ? l_hi : p.levels - 1;
                    const int NR = (32 * p.cvs + 63) >> 6;
                    const int nrow = (l_hi - l_lo + 1) * NR;
                    const unsigned rcp = ((1u << 20) + (unsigned)p.cvs - 1u) / (unsigned)p.cvs;      // idx / cvs = idx rcp >> 20 (idx < 512, cvs <= 16)
                    for (int r0 = 0; r0 < nrow; r0 += 32) {
                        float rawt[32];
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            const int rr = r0 + i, lq = rr / NR, it = rr - lq * NR;            // (uniform: scalar arithmetic)
                            const unsigned idx = (unsigned)(it * 64 + lane), tk = (idx * rcp) >> 20, comp = idx - tk * (unsigned)p.cvs;
                            // (no branch round the load: hipcc waits for ALL loads at a branch's end; a row the lane has nothing in reads the
                            //  trip's first float instead)
                            const bool ok = rr < nrow && idx < 32u * (unsigned)p.cvs;
                            const float *pl = p.cvf + ((size_t)(l_lo + (rr < nrow ? lq : 0)) * p.tok_pad + trow0) * p.cvs;
                            const float v = pl[ok ? ((int)tk < rows_ok ? tk : (unsigned)(rows_ok - 1)) * (unsigned)p.cvs + comp : 0u];
                            rawt[i] = ok ? v : 0.0f;
                        }
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            const int rr = r0 + i, lq = rr / NR, it = rr - lq * NR;
                            const unsigned idx = (unsigned)(it * 64 + lane), tk = (idx * rcp) >> 20, comp = idx - tk * (unsigned)p.cvs;
                            const int k = (l_lo + lq) * p.c + (int)comp - k0;
                            if (rr < nrow && idx < 32u * (unsigned)p.cvs && (int)comp < p.c && k >= 0 && k < VT) put(rawt[i], (int)tk, k);
                        }
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int kl = 0; kl < R3_BK; ++kl)
                    if (R3_BK * TR + kl < KS) {
                        Ahi[ts][R3_BK * TR + kl] = *reinterpret_cast<const u32x4 *>(wbuf + r * RS + (16 * kl + 8 * h) * 2);
                        Alo[ts][R3_BK * TR + kl] = *reinterpret_cast<const u32x4 *>(wbuf + r * RS + HB + (16 * kl + 8 * h) * 2);
                    }
                wave_sync_lds();
            });
        });
        float *orow = p.out + (size_t)tw0 * p.d;
        // one element tile: 3 KS matrix instructions on one accumulator that starts at the scaled bias sum; back to scale 1 at the end
        auto tile_mm = [&](const char *wb, int gi, int mt_abs, f32x16 (&a)[TS]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // lane (token r, half h) holds elements 32 mt + 8 i + 4 h + 0..3
                const f32x4 b = *reinterpret_cast<const f32x4 *>(smem + ((size_t)mt_abs * 32 + 8 * i + 4 * h) * 4);
#pragma unroll
                for (int ts = 0; ts < TS; ++ts) { a[ts][4 * i] = b[0]; a[ts][4 * i + 1] = b[1]; a[ts][4 * i + 2] = b[2]; a[ts][4 * i + 3] = b[3]; }
            }
            u32x4 fh[KS], fl[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                fh[ks] = *reinterpret_cast<const u32x4 *>(wb + ((gi * KS + ks) * 2) * 1024);
                fl[ks] = *reinterpret_cast<const u32x4 *>(wb + ((gi * KS + ks) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int ts = 0; ts < TS; ++ts) {
                    if (R3_ABL & 1) {
                        a[ts][0] += f32_bits(fh[ks][0] ^ Ahi[ts][ks][0] ^ fl[ks][1] ^ Alo[ts][ks][1]);
                    } else {
                        a[ts] = mfma32(fh[ks], Ahi[ts][ks], a[ts]);
                        a[ts] = mfma32(fh[ks], Alo[ts][ks], a[ts]);
                        a[ts] = mfma32(fl[ks], Ahi[ts][ks], a[ts]);
                    }
                }
            }
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) a[ts] = a[ts] * isc;
        };
        // "all reads, then all MFMAs" of one element tile put in order: the bias reads and two k-steps' fragments ahead, then a k-step's
        // fragments behind each k-step's three matrix instructions
        auto tile_order = [&]() {
            if (!(R3_ABL & 1)) {
                constexpr int PFK = KS < 2 ? KS : 2;
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * PFK, 0);
#pragma unroll
                for (int ks = 0; ks < KS - PFK; ++ks) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * TS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * TS * PFK, 0);
            }
        };
        // a finished element tile (elements el0 .. el0 + 31 of the wave's 32 tokens): every lane writes its token's 16 values, then
        // 4 x (8 rows x 128 bytes) come back and leave with non-temporal stores
        auto tile_out = [&](const f32x16 &a0, int el0, int ts) {
            float *orow_s = orow + (size_t)ts * 32 * p.d;
            const long long tws = tw0 + 32 * ts;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4 *>(wbuf + r * TRS + (8 * i + 4 * h) * 4) = f32x4{a0[4 * i], a0[4 * i + 1], a0[4 * i + 2], a0[4 * i + 3]};
            wave_sync_lds();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(wbuf + row * TRS + (lane & 7) * 16);
                if ((R3_ABL & 2) ? p.n_tokens < 0 : (fullw || tws + row < tile_end))
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(orow_s + (size_t)row * p.d + el0 + (lane & 7) * 4));
            }
            wave_sync_lds();
        };
        for (int s = 0; s < nstg; ++s) {
            // This stage landed when only the pieces issued one stage ago are younger -- and the stores of the two stages between its
            // issue and now (none before the kernel's first stage, one stage's worth before its second).  (The values prefetched for the
            // next tile lie there too at a tile boundary; not counting them waits for a few operations more than needed, never fewer.)
            if (!fullw) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ragged wave: its store count is not static
            } else {
                // younger than this stage's pieces: those of the RD - 2 stages issued after it, and the stores of the min(sdone, RD - 1) stages between
                static_for<0, RD>([&](auto n_c) {
                    constexpr int n = decltype(n_c)::value;
                    if (sdone == n) {
                        if (wave < NPX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * (NPB + 1) + n * NSS) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * NPB + n * NSS) : "memory");
                    }
                });
            }
            __builtin_amdgcn_s_barrier();      // every wave's pieces of this stage landed; every wave has left the stage before
            issue(s + RD - 1, slot == 0 ? RD - 1 : slot - 1);      // slot (s + RD - 1) % RD was read last in the stage before
            const char *wb = smem + off_ring + slot * STAGE + lane * 16;
            if (active) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    f32x16 a0[TS];
                    tile_mm(wb, gi, s * G + gi, a0);
                    tile_order();
                    // (the next tile's values before this tile's last stores: see CPF)
                    if (PREF && gi == G - 1 && s == nstg - 1 && tile + gridDim.x < ntile) load_raw(tile + gridDim.x);
#pragma unroll
                    for (int ts = 0; ts < TS; ++ts) tile_out(a0[ts], (s * G + gi) * 32, ts);
                }
            }
            slot = slot == RD - 1 ? 0 : slot + 1;
            sdone = sdone < RD - 1 ? sdone + 1 : RD - 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stages issued past the last tile must have landed before the workgroup's LDS is handed on
}
