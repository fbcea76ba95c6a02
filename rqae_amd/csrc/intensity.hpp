// intensity.hpp -- RQAEFeature.intensity: prepare and main kernels; included by rqae_hip.hip.
#pragma once
#include <type_traits>

#include "rqae_common.hpp"

namespace {

// =============================================================== feature intensity
// RQAEFeature.intensity (rqae/feature.py:102-129): for every token n and every requested level l
//     sims[n][m] = codebook_sims[center[m], code[n][m]] * w[m]            (fp16 product)
//     out[n][l]  = fp16( fp16(sum_{m<=l} sims[n][m]) / wcum[l] )           (fp32 running sum, sequential)
// with torch-CPU's fp16 semantics written out (oracle/oracle.py:intensity restates them; bit-equal to the
// reference's output).  One lane = one token, levels strictly in order (the sum is a sequential fp32 chain, so
// there is no parallel scan).  HBM-bound on the code matrix: a wave reads its 64 tokens x TL levels tile with
// fully coalesced 16-byte loads (one tile ahead, in registers), transposes it through LDS and walks its own
// row; the per-level table rows (codebook_sims[center[m], :], K fp16 each) sit in LDS a chunk of levels at a time.
struct IntensityParams {
    const _Float16 *table;    // [M][K]  = codebook_sims[center[m], :]
    const _Float16 *w, *wcum; // [M]
    const void *idx;
    const int *layers;        // [n_layers] device
    _Float16 *out;            // [n_tokens][n_layers]
    long long n_tokens;
    int nq, M, K, LC, n_layers, vec_ok;
    int *status;              // include/rqae_hip.h "Status word" (a code outside 0..K-1), or NULL
};

__global__ void intensity_prepare_kernel(const _Float16 *sims, const int *center, const _Float16 *w, int M, int K,
                                         _Float16 *table, _Float16 *wcum)
{
    const size_t n = (size_t)M * K;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(u / K), k = (int)(u % K);
        table[u] = sims[(size_t)center[m] * K + k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // layer_weights[:M].cumsum(): fp32 accumulator, fp16 prefixes
        float acc = 0.0f;
        for (int m = 0; m < M; ++m) {
            acc += (float)w[m];
            wcum[m] = (_Float16)acc;
        }
    }
}

template <typename IT>
__global__ void __launch_bounds__(256)
intensity_kernel(const IntensityParams p)
{
    constexpr int TL = 16;                          // levels per index tile
    constexpr int EPC = 16 / sizeof(IT);            // index elements per 16-byte chunk
    constexpr int CPR = TL / EPC;                   // chunks per tile row
    constexpr int RS = CPR + 1;                     // row stride in chunks (odd: conflict-free row reads)
    typedef IT ITV __attribute__((ext_vector_type(EPC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tab_bytes = ((p.LC * p.K * 2 + 15) & ~15);
    _Float16 *tab = reinterpret_cast<_Float16 *>(smem_raw);                       // [LC][K]
    float *wf = reinterpret_cast<float *>(smem_raw + tab_bytes);                   // [32] w, [32] wcum
    unsigned *omask = reinterpret_cast<unsigned *>(wf + 64);                       // output levels of the tile
    ITV *tiles = reinterpret_cast<ITV *>(smem_raw + tab_bytes + 64 * 4 + 16);      // [4 waves][64 rows][RS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tokw = (long long)blockIdx.x * 256 + wave * 64;
    const long long tok = tokw + lane;
    ITV *tile = tiles + (size_t)wave * 64 * RS;
    const IT *idx = reinterpret_cast<const IT *>(p.idx);

    // this lane's share of the wave's tile: chunk c of row r for the CPR load steps
    auto load_tile = [&](int m0, ITV (&regs)[CPR]) {
#pragma unroll
        for (int j = 0; j < CPR; ++j) {
            const int u = j * 64 + lane, r = u / CPR, c = u % CPR;
            long long t = tokw + r;
            if (t >= p.n_tokens) t = p.n_tokens - 1;
            const IT *src = idx + (size_t)t * p.nq + m0 + c * EPC;
            if (p.vec_ok && m0 + TL <= p.nq) {
                regs[j] = *reinterpret_cast<const ITV *>(src);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) regs[j][e] = (m0 + c * EPC + e < p.nq) ? src[e] : (IT)0;
            }
        }
    };
    // table chunk [c0, c0+LC) -> registers (whole 16-byte pieces; at most 12 per thread for a 48-KiB chunk)
    constexpr int TPT = 12;
    auto load_table = [&](int c0, f4 (&tregs)[TPT]) {
        const int nlev = (p.M - c0) < p.LC ? (p.M - c0) : p.LC;
        const int n16 = (nlev * p.K * 2 + 15) / 16;
        const f4 *src = reinterpret_cast<const f4 *>(p.table + (size_t)c0 * p.K);
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int u = j * 256 + threadIdx.x;
            if (u < n16) tregs[j] = src[u];
        }
    };
    ITV regs[CPR];
    f4 tregs[TPT];
    load_tile(0, regs);
    load_table(0, tregs);
    float acc = 0.0f;
    // largest code this lane met, as unsigned (a negative code is huge): the gathers below are LDS reads -- a code past the table
    // reads other LDS bytes or, past the allocation, zeros, never a fault -- so the range check is one max per code and one
    // compare at the end (the reference raises IndexError, rqae/feature.py:120)
    typedef typename std::conditional<sizeof(IT) == 8, unsigned long long, unsigned>::type UT;
    UT cmax = 0;

    for (int c0 = 0; c0 < p.M; c0 += p.LC) {
        const int m0 = c0 & ~(TL - 1);
        const bool new_tile = c0 == m0;
        __syncthreads();                                  // previous table chunk (and tile) no longer read
        if (new_tile) {
#pragma unroll
            for (int j = 0; j < CPR; ++j) {
                const int u = j * 64 + lane, r = u / CPR, c = u % CPR;
                tile[r * RS + c] = regs[j];
            }
            if (threadIdx.x < 64) {                       // weights of the tile's levels; which of them are outputs
                const int m = m0 + threadIdx.x;
                bool hit = false;
                if (threadIdx.x < TL) {
                    wf[threadIdx.x] = m < p.M ? (float)p.w[m] : 0.0f;
                    wf[32 + threadIdx.x] = m < p.M ? (float)p.wcum[m] : 1.0f;
                    for (int j = 0; j < p.n_layers; ++j) hit |= p.layers[j] == m;
                }
                const unsigned long long b = __ballot(hit);
                if (threadIdx.x == 0) *omask = (unsigned)b;
            }
        }
        {
            const int nlev = (p.M - c0) < p.LC ? (p.M - c0) : p.LC;
            const int n16 = (nlev * p.K * 2 + 15) / 16;
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const int u = j * 256 + threadIdx.x;
                if (u < n16) reinterpret_cast<f4 *>(tab)[u] = tregs[j];
            }
        }
        __syncthreads();
        if (c0 + p.LC < p.M) {                            // next chunk (and tile): in flight behind this one
            load_table(c0 + p.LC, tregs);
            if (((c0 + p.LC) & (TL - 1)) == 0) load_tile(c0 + p.LC, regs);
        }
        const unsigned mask = *omask;
        int hi = c0 + p.LC;
        if (hi > p.M) hi = p.M;
        auto emit = [&](int m, int rel) __attribute__((always_inline)) {   // output level: wave-uniform branch
            const _Float16 pref = (_Float16)acc;                              // cumsum's prefixes are fp16
            const _Float16 o = (_Float16)((float)pref / wf[32 + rel]);
            if (tok < p.n_tokens)
                for (int j = 0; j < p.n_layers; ++j)
                    if (p.layers[j] == m) p.out[(size_t)tok * p.n_layers + j] = o;
        };
        int m = c0;
        // whole 16-byte groups of codes: one row read, EPC gathers in flight together, then the sequential sum
        for (; m + EPC <= hi && ((m - m0) % EPC) == 0; m += EPC) {
            const int rel = m - m0;
            const ITV codes = tile[lane * RS + rel / EPC];
            float sv[EPC], wv[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                cmax = cmax > (UT)codes[e] ? cmax : (UT)codes[e];
                sv[e] = (float)tab[(m + e - c0) * p.K + (int)codes[e]];
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) wv[e] = wf[rel + e];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const _Float16 prod = (_Float16)(sv[e] * wv[e]);             // sims *= layer_weights
                acc += (float)prod;                                            // cumsum: fp32 accumulator
                if ((mask >> (rel + e)) & 1u) emit(m + e, rel + e);
            }
        }
        for (; m < hi; ++m) {
            const int rel = m - m0;
            const IT code = reinterpret_cast<const IT *>(tile + lane * RS)[rel];
            cmax = cmax > (UT)code ? cmax : (UT)code;
            const float sv = (float)tab[(m - c0) * p.K + (int)code];
            const _Float16 prod = (_Float16)(sv * wf[rel]);
            acc += (float)prod;
            if ((mask >> rel) & 1u) emit(m, rel);
        }
    }
    if (tok < p.n_tokens && cmax >= (UT)p.K && p.status) atomicOr(p.status, RQAE_STATUS_CODE_RANGE);
}

}  // namespace
