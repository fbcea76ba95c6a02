// gram_deep3.hpp -- the matrix-core arm beyond 64 levels: the released model runs 1024 (rqae/model.py:18-24; included by deep3.hip
// inside its anonymous namespace).
//
// The level chain's corrections  q_l -= sum_{m<l} (W_in[l] W_out[m]) cv_m  grow as L^2 c^2 / 2 multiply-adds per token (8.4 M at
// L = 1024, c = 4): as vector instructions they bound the arm.  Here the levels are cut into BLOCKS of 32
// (128 values at c = 4): inside a block the lattice form of the level chain runs as before (its own Gram updates in registers and,
// between its four level groups, through the projection buffer); ACROSS blocks the corrections are one matrix product per block,
//     P[tok][block b's 128 columns] += CV[tok][values of levels < 32 b] x G[those values][those columns],        G = -(W_in W_out)
// on v_mfma_f32_32x32x16_f16 with both operands as two fp16 pieces (three products: gram_pieces.hpp) -- "left-looking": every column
// is corrected once, right before its block runs, from the straight-through values written so far.  The same kernel, with the W_out
// image as its weights and all L c values as its K, is the reconstruction at this depth.
//   deep_pieces_kernel   fp32 planes of a block's 32 levels -> fp16 two-piece operand fragments [32-token tile][k-step][piece][1 KiB]
//                        (B operand layout: lane (r, h) holds token r, values 16 ks + 8 h + 0..7), scaled by 2^R3_CV_SHIFT
//   deep_gemm_kernel     C[tok][n] = sum_k A[tok][k] W[n][k] for k-steps [0, nks): A fragments straight from global memory into
//                        registers (coalesced 1-KiB loads, double-buffered), W fragments through a double-buffered LDS stage (4
//                        output tiles x 4 k-steps x 2 pieces = 32 KiB) by LDS-DMA; 8 waves x 32 tokens, 48 matrix instructions per
//                        wave between barriers.  MODE 0: C is added into the projection buffer P (the corrections); MODE 1: C + bias
//                        is the reconstruction, written like gram_recon3_kernel writes it.
//   deep_pack_g3_kernel  the fp16 two-piece image of G for the A operand (rows = output columns o = 4 l + j, k = 4 m + i; zero for m >= l)

constexpr int D3_NW = 8;
constexpr int D3_TILE = D3_NW * 32;
constexpr int D3_NTB = 4;                // output tiles (of 32) per pass: 4 (a block's 128 columns) or 8 (template argument NTB)
constexpr int D3_KC = 4;                 // k-steps per chunk
__host__ __device__ constexpr int d3_stage(int ntb) { return ntb * D3_KC * 2048; }
constexpr int D3_WBUF = 4608;            // transposition bytes per wave (MODE 1)

// (MODE 1's transposition buffers lie over the stages: the K loop is over when they are used)
__host__ __device__ inline size_t d3_lds_bytes(int mode, int d, int ntb) { return 2 * (size_t)d3_stage(ntb) + (mode == 1 ? pad1k((size_t)d * 4) : 0); }
static_assert(D3_NW * D3_WBUF <= 2 * d3_stage(4), "the waves' transposition buffers fit the two stages");

// ------------------------------------------------------------------------------------------------ planes -> pieces
// planes [nlev][tok_pad][4] fp32 (levels l0 .. l0 + nlev - 1 of the block, c = 4) -> fragments of k-steps ks0 .. ks0 + nlev / 4 - 1:
// k-step ks holds values 16 ks .. 16 ks + 15 = levels 4 ks .. 4 ks + 3; lane (r, h) holds levels 4 ks + 2 h and 4 ks + 2 h + 1 of token r
__global__ __launch_bounds__(256) void deep_pieces_kernel(const float *planes, long long tok_pad, int nlev, uint16_t *pieces, int KS_total, int ks0)
{
    const long long ntile = tok_pad / 32;
    const int nks = nlev / 4;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    for (long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); w < ntile * nks; w += (long long)gridDim.x * 4) {
        const long long tile = w / nks;
        const int ks = (int)(w - tile * nks);
        const long long tok = tile * 32 + r;
        const int la = 4 * ks + 2 * h;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(planes + ((size_t)la * tok_pad + tok) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(planes + ((size_t)(la + 1) * tok_pad + tok) * 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h0, l0, h1, l1;
            split2(v[2 * q] * (float)(1 << R3_CV_SHIFT), h0, l0);
            split2(v[2 * q + 1] * (float)(1 << R3_CV_SHIFT), h1, l1);
            hi[q] = (unsigned)h0 | ((unsigned)h1 << 16);
            lo[q] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        uint16_t *dst = pieces + (((size_t)tile * KS_total + ks0 + ks) * 2) * 512;
        *reinterpret_cast<u32x4 *>(dst + lane * 8) = hi;
        *reinterpret_cast<u32x4 *>(dst + 512 + lane * 8) = lo;
    }
}

// ------------------------------------------------------------------------------------------------ G image
// fragment (nt, ks, piece): lane l holds output column o = 32 nt + (l & 31) (level o / 4, component o % 4), values k = 16 ks + 8 (l >> 5) + j
// (level k / 4, component k % 4): G[o][k] = gneg[m = k / 4][l = o / 4][j = o % 4][i = k % 4], zero unless m < l; scaled by 2^T(max |gneg|)
__global__ void deep_pack_g3_kernel(const float *gneg, int L, int GB, const unsigned *maxbits, uint16_t *img)
{
    const int T = r3_shift_of(*maxbits);
    const float sc = r3_pow2(T > 100 ? 100 : T < -100 ? -100 : T);
    const int NT = L * 4 / 32, KS = L * 4 / 16;
    const size_t total = (size_t)NT * KS * 2 * 512;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j8 = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        size_t f = idx >> 9;
        const int piece = (int)(f & 1); f >>= 1;
        const int ks = (int)(f % KS);
        const int nt = (int)(f / KS);
        const int o = 32 * nt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j8;
        const int l = o >> 2, j = o & 3, m = k >> 2, i = k & 3;
        float v = 0.0f;
        if (m < l) v = gneg[((size_t)m * L + l) * GB + j * 4 + i] * sc;
        unsigned short hi, lo;
        split2(v, hi, lo);
        img[idx] = piece ? lo : hi;
    }
}

// ------------------------------------------------------------------------------------------------ the product
struct Deep3Params {
    const uint16_t *apieces;     // [tok_pad / 32][KSA][2][512]: the straight-through values' fragments
    const uint16_t *wimg;        // [n tiles][KSW][2][512]: the weights' fragments (G image / W_out image)
    const unsigned *maxbits;     // the weight image's scale word
    float *P;                    // MODE 0: [groups][tok_pad][16], columns nt0 * 32 .. are added to
    const float *bias;           // MODE 1: [d]
    float *out;                  // MODE 1: [n_tokens][d]
    long long n_tokens, tok_pad;
    int KSA, KSW;                // k-steps per tile in the two images
    int ks0, nks;                // k-steps of this product: [ks0, ks0 + nks)
    int nt0, npass;              // first output tile, passes of D3_NTB tiles
    int d;                       // MODE 1: row length of out
};

// NTB: output tiles per pass.  The products are bound by their operands' way from L2 / the Infinity Cache (per chunk and wave 8 KiB of token
// fragments + the wave's share of the weights' stage against NTB x 12 matrix instructions: 256 bytes per instruction at NTB = 4, 170 at
// 8), so the wide products -- the reconstruction, the right-looking corrections -- take 8 tiles; a block's own 128 columns are 4.
template <int MODE, int NTB>
__global__ __launch_bounds__(D3_NW * 64) void deep_gemm_kernel(Deep3Params p)
{
    constexpr int D3_NTB = NTB, D3_STAGE = d3_stage(NTB);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned off_bias = 2u * D3_STAGE;
    const unsigned bias_bytes = MODE == 1 ? (unsigned)pad1k((size_t)p.d * 4) : 0u;
    char *wbuf = smem + wave * D3_WBUF;          // over the stages, after the K loop (a barrier in between)
    const int Tsh = r3_shift_of(*p.maxbits);
    const int Tc = Tsh > 100 ? 100 : Tsh < -100 ? -100 : Tsh;
    const float bsc = r3_pow2(Tc) * (float)(1 << R3_CV_SHIFT);
    const float isc = 1.0f / bsc;
    if constexpr (MODE == 1) {
        for (int i = tid; i < (int)(bias_bytes >> 2); i += D3_NW * 64) reinterpret_cast<float *>(smem + off_bias)[i] = i < p.d ? p.bias[i] * bsc : 0.0f;
        __syncthreads();      // the accumulators of the first item start from these sums
    }
    const int nch = (p.nks + D3_KC - 1) / D3_KC;
    const long long ntile = (p.n_tokens + D3_TILE - 1) / D3_TILE;
    // work items = (token tile, pass), the PASS fastest-varying: the workgroups in flight at any moment share a few token tiles, so a
    // tile's operand fragments (16 KB per token at L = 1024) come from HBM once and from the Infinity Cache for its other passes --
    // with the tile fastest-varying every pass re-read all of them from HBM (18 x 2.1 GB per 131072 tokens: the reconstruction ran
    // at the speed of those reads, 8.9 ms)
    const long long nitem = ntile * p.npass;
    for (long long item = blockIdx.x; item < nitem; item += gridDim.x) {
        const long long tile = item / p.npass;
        const int pass = (int)(item - tile * p.npass);
        const long long tile0 = tile * D3_TILE;
        const long long tile_end = tile0 + D3_TILE < p.n_tokens ? tile0 + D3_TILE : p.n_tokens;
        const long long tw0 = tile0 + (long long)wave * 32;
        const bool active = tw0 < tile_end;
        const long long t32 = (active ? tw0 : tile0) / 32;           // the wave's 32-token tile in the pieces image (a wave without tokens reads the tile's first)
        const int nt_first = p.nt0 + pass * D3_NTB;
        f32x16 acc[D3_NTB];
#pragma unroll
        for (int o = 0; o < D3_NTB; ++o) {
            if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(smem + off_bias + ((size_t)(nt_first + o) * 32 + 8 * i + 4 * h) * 4);
                    acc[o][4 * i] = b[0]; acc[o][4 * i + 1] = b[1]; acc[o][4 * i + 2] = b[2]; acc[o][4 * i + 3] = b[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[o][i] = 0.0f;
            }
        }
        auto issue_w = [&](int ch, int buf) {
            // stage = [k][o][piece] x 1 KiB; k-steps past nks repeat the last one (their A fragments are zeroed below)
            for (int q = wave; q < D3_NTB * D3_KC * 2; q += D3_NW) {
                const int k = q / (D3_NTB * 2), rem = q - k * (D3_NTB * 2);
                int ks = ch * D3_KC + k;
                ks = p.ks0 + (ks < p.nks ? ks : p.nks - 1);
                dma_piece(uniform_ptr(p.wimg + (((size_t)(nt_first + (rem >> 1)) * p.KSW + ks) * 2 + (rem & 1)) * 512), lds0 + buf * D3_STAGE + q * 1024, lane16);
            }
        };
        u32x4 A0[D3_KC][2], A1[D3_KC][2];
        auto load_a = [&](int ch, u32x4 (&A)[D3_KC][2]) {
#pragma unroll
            for (int k = 0; k < D3_KC; ++k) {
                int ks = ch * D3_KC + k;
                const bool live = ks < p.nks;
                ks = p.ks0 + (live ? ks : p.nks - 1);
                const uint16_t *src = p.apieces + (((size_t)t32 * p.KSA + ks) * 2) * 512 + lane * 8;
                A[k][0] = *reinterpret_cast<const u32x4 *>(src);
                A[k][1] = *reinterpret_cast<const u32x4 *>(src + 512);
                if (!live) { A[k][0] = u32x4{0u, 0u, 0u, 0u}; A[k][1] = u32x4{0u, 0u, 0u, 0u}; }
            }
        };
        auto mm = [&](int buf, const u32x4 (&A)[D3_KC][2]) {
            const char *wb = smem + buf * D3_STAGE + lane * 16;
#pragma unroll
            for (int k = 0; k < D3_KC; ++k)
#pragma unroll
                for (int o = 0; o < D3_NTB; ++o) {
                    const u32x4 wh = *reinterpret_cast<const u32x4 *>(wb + ((k * D3_NTB + o) * 2) * 1024);
                    const u32x4 wl = *reinterpret_cast<const u32x4 *>(wb + ((k * D3_NTB + o) * 2 + 1) * 1024);
                    acc[o] = mfma32(wh, A[k][0], acc[o]);
                    acc[o] = mfma32(wh, A[k][1], acc[o]);
                    acc[o] = mfma32(wl, A[k][0], acc[o]);
                }
            // "all reads, then all MFMAs" put in order: two fragment pairs ahead, then a pair's reads behind each pair's three matrix
            // instructions (left alone hipcc reads a pair, waits out the LDS latency, multiplies: 16 exposed latencies per chunk)
            constexpr int NPAIR = D3_KC * D3_NTB, PFP = 2;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PFP, 0);
#pragma unroll
            for (int i = 0; i < NPAIR - PFP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * PFP, 0);
        };
        __syncthreads();              // every wave has left the item before (its stage reads, its transposition buffer)
        issue_w(0, 0);
        load_a(0, A0);
        for (int ch = 0; ch < nch; ch += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // stage ch landed for everyone; everyone has left stage ch - 1
            if (ch + 1 < nch) { issue_w(ch + 1, 1); load_a(ch + 1, A1); }
            if (active) mm(0, A0);
            if (ch + 1 < nch) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (ch + 2 < nch) { issue_w(ch + 2, 0); load_a(ch + 2, A0); }
                if (active) mm(1, A1);
            }
        }
        if constexpr (MODE == 1) __syncthreads();      // every wave has left the last stage: the stages now hold the waves' transposition buffers
        if (!active) continue;
        if constexpr (MODE == 0) {
            // lane (token r, half h) holds columns 32 (nt_first + o) + 8 i + 4 h + 0..3: group u = 2 nt + (i >> 1) of P[group][tok][16]
            const long long tok = tw0 + r;
            if (tok < tile_end) {
#pragma unroll
                for (int o = 0; o < D3_NTB; ++o)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float *dst = p.P + (((size_t)(2 * (nt_first + o) + (i >> 1)) * p.tok_pad + tok) * 16 + 8 * (i & 1) + 4 * h);
                        f32x4 v = *reinterpret_cast<const f32x4 *>(dst);
                        v[0] += acc[o][4 * i] * isc; v[1] += acc[o][4 * i + 1] * isc; v[2] += acc[o][4 * i + 2] * isc; v[3] += acc[o][4 * i + 3] * isc;
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
            }
        } else {
            float *orow = p.out + (size_t)tw0 * p.d;
            constexpr int TRS = 144;
#pragma unroll
            for (int o = 0; o < D3_NTB; ++o) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4 *>(wbuf + r * TRS + (8 * i + 4 * h) * 4) =
                        f32x4{acc[o][4 * i] * isc, acc[o][4 * i + 1] * isc, acc[o][4 * i + 2] * isc, acc[o][4 * i + 3] * isc};
                wave_sync_lds();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + (lane >> 3);
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(wbuf + row * TRS + (lane & 7) * 16);
                    if (tw0 + row < tile_end)
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(orow + (size_t)row * p.d + (nt_first + o) * 32 + (lane & 7) * 4));
                }
                wave_sync_lds();
            }
        }
    }
}
