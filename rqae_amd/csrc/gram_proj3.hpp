// gram_proj3.hpp -- round 4's all-levels projection of the matrix-core arm (included by proj3.hip inside its anonymous namespace).
//
//   P[tok][o] = sum_e x[tok][e] W_in[l][j][e]        o = (level group, level in group, component)      (rqae/model.py:211, every level)
//
// Against gram_fused_kernel<.., MODE 1 / 5> and gram_project2_kernel:
//   * W_in goes in as TWO fp16 pieces (hi = rn16(w 2^T), lo = rn16(w 2^T - hi): remainder below 2^-24 |w|) instead of three bf16
//     pieces; x in bf16 / fp16 is ONE fp16 piece (a bf16 value is an fp16 value wherever |x| lies in fp16's normal range -- the
//     RMS-normalised residual stream does; below 6.1e-5 the conversion truncates at 2^-24 absolute, beyond 65504 it gives inf and the
//     row goes NaN -- index 0, loudly: such inputs take proj = 2), so two matrix instructions per fragment pair instead of three; fp32 x is two pieces, three products
//     instead of six.  T: the power of two that brings max |W_in| into [2^14, 2^15) (found on the device at pack time);
//   * v_mfma_f32_32x32x16_f16, token-stationary: a wave keeps ALL output tiles of its 32 tokens (up to 8 x 32 outputs: 128
//     accumulator registers) and streams W_in through LDS once -- a W fragment (1 KiB) feeds a 32 x 32 x 16 product, so even C4's
//     256 outputs need 125 B/clk of the LDS array's 256;
//   * one launch covers every output tile of the shape (gram_project2_kernel re-read x from L2 once per block of 48 outputs:
//     C4, 0.55 ms against 0.08 ms for the bare x stream).
// x arrives by LDS-DMA (pieces of 8 rows x 128 bytes, XOR-swizzled through the per-lane SOURCE address so that the operand reads
// are conflict-free), the W_in image through a 3-slot ring; counted vmcnt + one raw barrier per chunk, as in gram_fused_kernel.
// NORM: Gemma's RMSNorm folded in (sum of x^2 from the raw fragments, projections scaled by 1 / rms; MODE 5 of the fused kernel).

struct Proj3Params {
    const void *x;               // [n_tokens][d] bf16 / fp16 / fp32
    const uint16_t *wimg;        // [d / 16][OTT][2][512]: fragment (k-step, output tile, piece)
    const unsigned *maxbits;     // bits of max |W_in (x in_scale)|: the image's scale
    float *P;                    // [NOG][tok_pad][16]
    float *inv_rms;              // NORM: 1 / sqrt(mean(x^2) + eps) per token
    float eps;
    long long n_tokens, tok_pad;
    int d, OTT, NOG;             // OTT: output tiles of 32 in the image (NOG 16-wide groups, rounded up to whole tiles)
    int ny;                      // column blocks of OT output tiles (deep shapes: OTT > 8); the ny workgroups of a token tile share an XCD
};

constexpr int P3_NW = 8;
constexpr int P3_TILE = P3_NW * 32;

template <int XT> struct P3X;            // per x type: bytes per element, k-steps per chunk (a chunk = 128 bytes of every row)
template <> struct P3X<RQAE_X_BF16> { static constexpr int ESZ = 2, KCH = 4, NP = 1; };
template <> struct P3X<RQAE_X_F16> { static constexpr int ESZ = 2, KCH = 4, NP = 1; };
template <> struct P3X<RQAE_X_F32> { static constexpr int ESZ = 4, KCH = 2, NP = 2; };

// k-steps per W stage (an x chunk -- 128 bytes of every row -- is 4 k-steps of 2-byte x, 2 of fp32 x: one or two W stages)
__host__ __device__ constexpr int p3_kch(int xt, int ot) { return xt == RQAE_X_F32 ? (ot <= 4 ? 2 : 1) : (ot <= 2 ? 4 : 2); }
__host__ __device__ constexpr int p3_xslots(int xt, int ot) { return p3_kch(xt, ot) == (xt == RQAE_X_F32 ? 2 : 4) ? 3 : 2; }
__host__ __device__ constexpr size_t p3_lds_bytes(int xt, int ot) { return 3 * (size_t)p3_kch(xt, ot) * ot * 2048 + (size_t)P3_NW * p3_xslots(xt, ot) * 4096; }

// the x fragment of one k-step (B operand: lane (r, h) holds token r, values 16 ks + 8 h + 0..7) as fp16 pieces
template <int XT>
__device__ __forceinline__ void p3_xpieces(const u32x4 (&raw)[P3X<XT>::NP], u32x4 &hi, u32x4 &lo)
{
    if constexpr (XT == RQAE_X_F16) {
        hi = raw[0];
    } else if constexpr (XT == RQAE_X_BF16) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            // two bf16 -> two fp16: exact inside fp16's normal range (8 significant bits into 11); beyond 65504 -> inf (the row goes NaN
            // and takes index 0: loud, not saturated); three vector instructions per pair
            const unsigned u = raw[0][w];
            const float a = __builtin_bit_cast(float, u << 16), b = __builtin_bit_cast(float, u & 0xffff0000u);
            hi[w] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
        }
    } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __builtin_bit_cast(float, w < 2 ? raw[0][2 * w] : raw[1][2 * w - 4]);
            const float b = __builtin_bit_cast(float, w < 2 ? raw[0][2 * w + 1] : raw[1][2 * w - 3]);
            unsigned short ah, al, bh, bl;
            split2(a, ah, al);
            split2(b, bh, bl);
            hi[w] = (unsigned)ah | ((unsigned)bh << 16);
            lo[w] = (unsigned)al | ((unsigned)bl << 16);
        }
    }
}

template <int XT, int OT, bool NORM>
__global__ __launch_bounds__(P3_NW * 64) void gram_proj3_kernel(Proj3Params p)
{
    using X = P3X<XT>;
    constexpr int ESZ = X::ESZ, NP = X::NP;
    constexpr int KX = X::KCH;                       // k-steps per x chunk (128 bytes of a row)
    constexpr int KW = p3_kch(XT, OT);               // k-steps per W stage
    static_assert(KX % KW == 0, "whole W stages per x chunk");
    constexpr int WPX = KX / KW;                     // W stages per x chunk
    constexpr int STAGE = KW * OT * 2048;
    constexpr int NPC = KW * OT * 2;                 // W pieces per stage
    constexpr int NPB = NPC / P3_NW, NPX = NPC % P3_NW;
    constexpr int XP = 4;                            // x pieces per chunk and wave (32 rows x 128 bytes)
    static_assert(!NORM || XT != RQAE_X_F32, "the folded norm takes 2-byte hidden states");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr int NXS = p3_xslots(XT, OT);          // x slots per wave: 3 when a chunk is one W stage (issued two chunks ahead), else 2
    const unsigned off_xw = 3u * STAGE + (unsigned)wave * (unsigned)NXS * 4096u;
    const char *xw = smem + off_xw;
    const size_t rowb = (size_t)p.d * ESZ;
    const int nxc = (p.d * ESZ) / 128;               // x chunks per tile
    const int nws = nxc * WPX;                       // W stages per tile
    const int Tsh = r3_shift_of(*p.maxbits);
    const int Tc = Tsh > 100 ? 100 : Tsh < -100 ? -100 : Tsh;
    const float isc = 1.0f / r3_pow2(Tc);
    const long long ntile = (p.n_tokens + P3_TILE - 1) / P3_TILE;
    // XCD-aware mapping for several column blocks: consecutive workgroup ids go round the 8 XCDs, so ids that are equal mod 8 share an L2.
    // The ny workgroups that take the same token tile through different output tiles get such ids and run at the same time: the
    // tile's x comes from HBM once and from that XCD's L2 ny - 1 times (no non-temporal hint on it then).
    long long tile_first = blockIdx.x, tile_step = gridDim.x;
    int ot0 = 0;
    if (p.ny > 1) {
        const unsigned bid = blockIdx.x, xcd = bid & 7u, within = bid >> 3;
        ot0 = (int)(within % (unsigned)p.ny) * OT;
        tile_first = (long long)(within / (unsigned)p.ny) * 8 + xcd;
        tile_step = gridDim.x / (unsigned)p.ny;
    }
    if (tile_first >= ntile) return;
    constexpr int NPS = 4 * OT + (NORM ? 1 : 0);     // P (and 1 / rms) stores per wave and tile (the workspace holds 2 OT groups: every store lands)
    // x piece pc of a chunk = rows 8 pc .. 8 pc + 7 of the wave x 128 bytes; DMA lane i = (row i >> 3, slot i & 7) lands at piece +
    // 16 i and fetches the row's 16-byte chunk (i & 7) ^ swz(row), swz(row) = (row >> 1) & 7: LDS slot j of a row holds source chunk
    // j ^ swz, and the operand reads (lane (r, h): row r, chunk c) touch 16 distinct 16-byte slots of the 256-byte bank span per
    // 16 lanes (rows of equal parity differ in swz)
    auto offsets_of = [&](long long tile, unsigned (&vo)[XP]) {
        const long long tile0 = tile * P3_TILE;
        const long long tile_end = tile0 + P3_TILE < p.n_tokens ? tile0 + P3_TILE : p.n_tokens;
#pragma unroll
        for (int pc = 0; pc < XP; ++pc) {
            const int row = 8 * pc + (lane >> 3);
            long long rr = tile0 + (long long)wave * 32 + row;
            rr = rr < tile_end ? rr : tile_end - 1;          // rows past the batch re-read its last row (never stored)
            vo[pc] = (unsigned)((rr - tile0) * (long long)rowb) + (unsigned)((((lane & 7) ^ (row >> 1)) & 7) * 16);
        }
    };
    // The x chunks and the W stages form ONE stream over all tiles of the workgroup: the chunk after a tile's last is the next tile's
    // first, the W stage after the image's last is its first -- nothing is drained or re-primed at a tile boundary.
    unsigned voff_i[XP];                             // offsets of the tile the x ISSUE stream is in
    long long itile = tile_first;                    // ... that tile
    int ixc = 0;                                     // ... and its next chunk
    offsets_of(itile, voff_i);
    auto issue_x = [&](int slot) {
        const void *xb = uniform_ptr(static_cast<const char *>(p.x) + (size_t)itile * P3_TILE * rowb + (size_t)ixc * 128);
        if (p.ny > 1) {
#pragma unroll
            for (int pc = 0; pc < XP; ++pc) dma_piece(xb, lds0 + off_xw + slot * 4096 + pc * 1024, voff_i[pc]);
        } else {
#pragma unroll
            for (int pc = 0; pc < XP; ++pc) dma_piece_nt(xb, lds0 + off_xw + slot * 4096 + pc * 1024, voff_i[pc]);
        }
        if (++ixc == nxc) {                          // on to the next tile (past the last one: its last chunk again, static counts)
            if (itile + tile_step < ntile) { itile += tile_step; ixc = 0; offsets_of(itile, voff_i); }
            else ixc = nxc - 1;
        }
    };
    auto issue_w = [&](int ws_in, int slot) {
        const int ws = ws_in < nws ? ws_in : ws_in - nws;
        for (int q = wave; q < NPC; q += P3_NW) {
            const int k = q / (OT * 2), rem = q - k * (OT * 2);
            int ot = ot0 + (rem >> 1);
            ot = ot < p.OTT ? ot : p.OTT - 1;         // output tiles past the image (OT > OTT): any fragment, never used
            dma_piece(uniform_ptr(p.wimg + (((size_t)(ws * KW + k) * p.OTT + ot) * 2 + (rem & 1)) * 512), lds0 + slot * STAGE + q * 1024, lane16);
        }
    };
    // group s (a W stage) = { the x pieces of a chunk when s starts one, the W pieces of stage s + 2 }; the wave multiplies stage s when
    // everything older than the newest group has landed
    issue_x(0);
    issue_w(0, 0);
    if (WPX == 1) issue_x(1);
    issue_w(1, 1);
    int wslot = 0, xslot = 0;
    unsigned xro[KX];        // operand read offsets inside an x slot: lane (r, h), k-step k -> chunk 2 k + h (2-byte x), 4 k + 2 h .. (fp32)
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int cch = ESZ == 2 ? 2 * k + h : 4 * k + 2 * h;
        xro[k] = (unsigned)(r * 128 + ((cch ^ (r >> 1)) & 7) * 16);
    }
    bool first_tile = true;
    for (long long tile = tile_first; tile < ntile; tile += tile_step) {
        const long long tile0 = tile * P3_TILE;
        const long long tile_end = tile0 + P3_TILE < p.n_tokens ? tile0 + P3_TILE : p.n_tokens;
        const long long tw0 = tile0 + (long long)wave * 32;
        const bool active = tw0 < tile_end;
        const bool fullw = tw0 + 32 <= tile_end;
        f32x16 acc[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[o][i] = 0.0f;
        float ssq = 0.0f;
        for (int xc = 0; xc < nxc; ++xc) {
            static_for<0, WPX>([&](auto w_c) {
                constexpr int WS = decltype(w_c)::value;
                const int s = xc * WPX + WS;
                // younger than what stage s needs: the group issued one stage ago -- its W pieces, and its x pieces where it carried
                // some -- and the P stores of the tile before where they lie between (the tile's first two stages)
                constexpr bool prev_had_x = WPX == 1 || WS == 1;       // the group before this one started an x chunk (WPX = 2: groups of even s do)
                const bool stores_between = !first_tile && s < 2;
                if (!fullw) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ragged wave: its store count is not static
                } else if (wave < NPX) {
                    if (stores_between) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + 1 + (prev_had_x ? XP : 0) + NPS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + 1 + (prev_had_x ? XP : 0)) : "memory");
                } else {
                    if (stores_between) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + (prev_had_x ? XP : 0) + NPS) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + (prev_had_x ? XP : 0)) : "memory");
                }
                __builtin_amdgcn_s_barrier();      // every wave's W pieces of stage s landed; every wave has left stage s - 1
                // group s: x of the chunk two ahead (its slot was read last in the chunk before this one), W of stage s + 2
                if (WS == 0) issue_x(WPX == 1 ? (xslot == 0 ? 2 : xslot - 1) : (xslot ^ 1));
                issue_w(s + 2, wslot == 0 ? 2 : wslot - 1);
                if (active) {
                    const char *wb = smem + wslot * STAGE + lane * 16;
                    const char *xs = xw + xslot * 4096;
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        const int kx = WS * KW + k;              // k-step inside the x chunk
                        u32x4 raw[NP];
                        raw[0] = *reinterpret_cast<const u32x4 *>(xs + xro[kx]);
                        if constexpr (NP == 2) raw[1] = *reinterpret_cast<const u32x4 *>(xs + (xro[kx] ^ 16u));
                        if constexpr (NORM) {
#pragma unroll
                            for (int wd = 0; wd < 4; ++wd) {
                                const unsigned u = raw[0][wd];
                                if constexpr (XT == RQAE_X_BF16) {
                                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                                    ssq = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u), __builtin_bit_cast(bf2, u), ssq, false);
                                } else {
                                    typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
                                    ssq = __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2, u), __builtin_bit_cast(hf2, u), ssq, false);
                                }
                            }
                        }
                        u32x4 bh, bl;
                        p3_xpieces<XT>(raw, bh, bl);
#pragma unroll
                        for (int o = 0; o < OT; ++o) {
                            const u32x4 wh = *reinterpret_cast<const u32x4 *>(wb + ((k * OT + o) * 2) * 1024);
                            const u32x4 wl = *reinterpret_cast<const u32x4 *>(wb + ((k * OT + o) * 2 + 1) * 1024);
                            acc[o] = mfma32(wh, bh, acc[o]);
                            acc[o] = mfma32(wl, bh, acc[o]);
                            if constexpr (NP == 2) acc[o] = mfma32(wh, bl, acc[o]);
                        }
                    }
                    // the stage's fragment pairs put in order: the x fragments and two W pairs ahead, then a pair's reads behind each
                    // pair's matrix instructions (left alone hipcc reads a pair, waits out the LDS latency, multiplies)
                    if constexpr (!NORM && XT != RQAE_X_BF16 && XT != RQAE_X_F32) {
                        constexpr int NPAIR = KW * OT, PFP = NPAIR < 2 ? NPAIR : 2, MPP = NP == 2 ? 3 : 2;
                        __builtin_amdgcn_sched_group_barrier(0x100, KW * NP + 2 * PFP, 0);
#pragma unroll
                        for (int i = 0; i < NPAIR - PFP; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, MPP, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, MPP * PFP, 0);
                    }
                }
                wslot = wslot == 2 ? 0 : wslot + 1;
                if (WS == WPX - 1) xslot = NXS == 3 ? (xslot == 2 ? 0 : xslot + 1) : (xslot ^ 1);
            });
        }
        // ---- projections out: lane (token r, half h) holds outputs 32 o + 8 i + 4 h + 0..3 -> P[group][tok][16]; NPS stores per wave
        float rs = 1.0f;
        if constexpr (NORM) {
            float v = ssq + __shfl_xor(ssq, 32);              // the two halves of a token's row
            rs = 1.0f / __builtin_sqrtf(v / (float)p.d + p.eps);
            if (h == 0 && (fullw || tw0 + r < tile_end)) p.inv_rms[tw0 + r] = rs;      // (every column block writes the same value: one store either way, the waits count it)
        }
        const float osc = isc * rs;
        if (active && (fullw || tw0 + r < tile_end)) {
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = 2 * (ot0 + o) + (i >> 1);  // < the workspace's groups (rounded up to whole tiles)
                    *reinterpret_cast<f32x4 *>(p.P + (((size_t)u * p.tok_pad + tw0 + r) * 16 + 8 * (i & 1) + 4 * h)) =
                        f32x4{acc[o][4 * i] * osc, acc[o][4 * i + 1] * osc, acc[o][4 * i + 2] * osc, acc[o][4 * i + 3] * osc};
                }
        }
        first_tile = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the groups issued past the last tile must have landed before the workgroup's LDS is handed on
}

// ------------------------------------------------------------------------------------------------ pack
// max | W_in[l][j][e] (x in_scale[e]) | as float bits
__global__ void gram_pack_absmax_in_kernel(const float *w_in, const float *in_scale, size_t n, int d, unsigned *out_bits)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = w_in[i];
        if (in_scale) v = v * in_scale[i % d];
        const float a = __builtin_fabsf(v);
        m = a > m ? a : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));
}

// W_in image: fragment (ks, ot, piece) = 1 KiB = the A operand of v_mfma_f32_32x32x16_f16: lane l holds output 32 ot + (l & 31), element
// 16 ks + 8 (l >> 5) + j.  Output o = (level group o / GP, level (o % GP) / CP, component (o % GP) % CP) as in the bf16 image
// (gram_pack_win_kernel): the chain kernels read the same P.
__global__ void gram_pack_win3_kernel(int d, int c, int L, int CP, int GL, int GP, int OTT, const float *w_in, const float *in_scale,
                                      const unsigned *maxbits, uint16_t *img)
{
    const int T = r3_shift_of(*maxbits);
    const float sc = r3_pow2(T > 100 ? 100 : T < -100 ? -100 : T);
    const size_t total = (size_t)(d / 16) * OTT * 2 * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t f = i >> 9;
        const int piece = (int)(f & 1); f >>= 1;
        const int ot = (int)(f % OTT);
        const int ks = (int)(f / OTT);
        const int o = 32 * ot + (lane & 31), e = 16 * ks + 8 * (lane >> 5) + j;
        const int lg = o / GP, rem = o % GP, li = rem / CP, jj = rem % CP, l = lg * GL + li;
        float v = 0.0f;
        if (rem < GL * CP && l < L && jj < c) {
            v = w_in[((size_t)l * c + jj) * d + e];
            if (in_scale) v = v * in_scale[e];
            v = v * sc;
        }
        unsigned short hi, lo;
        split2(v, hi, lo);
        img[i] = piece ? lo : hi;
    }
}
