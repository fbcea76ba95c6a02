// search.hip -- the demo server's similarity search for gfx950.
//
// /root/reference/demo/server/server.py, IntensityEngine.find_examples (:160-325): for a query sequence (Q positions x L codes)
// and a dataset of N sequences (S positions x L codes),
//     intensity[n, s, q] = sum over levels l of  sims[l][ query[q][l] ][ codes[n][s][l] ]          (get_intensities, :41-68)
// accumulated level range by level range in fp16 (:214-260), then per query position the sequences ordered by their maximum
// over s (:263-266) and the top / middle / bottom examples' intensity rows gathered (:269-311).
//
// This is gather-and-add work, HBM- and LDS-bound; nothing here is a matrix product.  Layout for the MI355X:
//   * the dataset's codes stay resident in HBM ([N][S][L] uint16 or int32 -- 266 MB per 1024-sequence shard as uint16; the
//     reference copies int32 slices host -> device per range and shard);
//   * search_query_table_kernel turns the query's table rows around: tq[l][k][q] = sims[l][query[q][l]][k], so everything one
//     dataset code needs at one level is ONE contiguous row (Q halves, padded to 128: 256 bytes);
//   * search_accumulate_kernel: a workgroup = 1024 dataset tokens x 32 query positions.  The level's table slice (K rows x
//     64 bytes = 40 KB at K = 625) streams through a ring in LDS by LDS-DMA (global_load_lds_dwordx4) two levels ahead; a lane
//     gathers 16 bytes (8 query positions) of its token's row with one ds_read_b128 and adds them into fp32 registers
//     (v_fma_mix_f32: fp16 in, fp32 accumulate); 64-level chunks fold to fp16 exactly as the reference's chunked sums do; the
//     running fp16 accumulation [N][S][128] is read and written once per level range;
//   * search_rowmax_kernel + a hand-written bitonic sort in LDS (search_sort_kernel: one workgroup per query position and
//     chunk of 8192 sequences, 64-bit keys (descending maximum, ascending sequence number) -- so equal maxima keep ascending
//     sequence order; longer datasets merge chunks through global memory) rank the sequences; search_gather_kernel fetches the
//     chosen examples' rows.
//
// Arithmetic (oracle/search_oracle.py restates it; tests/golden/search.npz is the reference's own output): per (n, s, q) and
// chunk of <= 64 levels an fp32 sum in ascending level order, rounded once to fp16; fp16 additions of chunks and of ranges.
//
// Translation unit of librqae_hip.so; C ABI at the bottom, declared in include/rqae_hip.h.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/rqae_hip.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];

namespace {

int sfail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int sfail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define S_TRY(expr)                                                                      \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return sfail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;

constexpr int S_QP = 128;                  // query positions per table / accumulator row (padded)
constexpr int S_QW = 32;                   // query positions per workgroup: 64-byte row segments
constexpr int S_NW = 8;                    // waves per workgroup
constexpr int S_TS = 8;                    // tokens per lane (16 tokens per wave and slot: 4 lanes x 16 bytes per token)
constexpr int S_TOK = S_NW * 16 * S_TS;    // 1024 tokens per workgroup
constexpr int S_LB = 8;                    // levels per block of codes (one 16-byte load of uint16 codes)
constexpr int S_CHUNK = 64;                // server.py:216-218

__device__ __forceinline__ const void *uniform_ptr(const void *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned long long u = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
    return reinterpret_cast<const void *>(u);
}

// one 1-KiB LDS-DMA piece: lane i moves 16 bytes from g_uniform + lane_byte_off to LDS byte address lds_addr + 16 i
__device__ __forceinline__ void dma_piece(const void *g_uniform, unsigned lds_addr, unsigned lane_byte_off)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(g_uniform), "s"(lds_addr)
                 : "memory");
}

// ------------------------------------------------------------------------------------------------ query table
// tq[li][k][q] = sims[l0 + li][ query[q][l0 + li] ][k]  (server.py:186-200 turned around; rows k >= K and columns q >= Q are zero)
// one workgroup = one level x 64 table columns k, through an LDS tile (reads run along k, writes along q)
__global__ __launch_bounds__(256) void search_query_table_kernel(const __half *sims, int K, int KP, const int *qcodes, int Q,
                                                                  long long q_stride, int l0, __half *tq)
{
    __shared__ __half tile[S_QP][66];
    const int li = blockIdx.x, k0 = blockIdx.y * 64, tid = threadIdx.x;
    const int l = l0 + li;
    for (int q = tid >> 6; q < S_QP; q += 4) {
        const int k = k0 + (tid & 63);
        __half v = __float2half(0.0f);
        if (q < Q && k < K) {
            int row = qcodes[(long long)q * q_stride + l];
            row = row < 0 ? 0 : (row >= K ? K - 1 : row);          // memory-safe; the host wrapper rejects such a query
            v = sims[((size_t)l * K + row) * K + k];
        }
        tile[q][tid & 63] = v;
    }
    __syncthreads();
    for (int kk = tid >> 7; kk < 64; kk += 2) {
        const int k = k0 + kk;
        if (k < KP) tq[((size_t)li * KP + k) * S_QP + (tid & 127)] = tile[tid & 127][kk];
    }
}

// ------------------------------------------------------------------------------------------------ accumulate
struct AccParams {
    const void *codes;       // [NT][Ltot] uint16 / int32
    const __half *tq;        // [l1 - l0][KP][128]
    __half *acc;             // [NT][128]
    long long NT;
    int Ltot, KP, l0, l1, first, nbuf, dbg;
    int K;                   // rows of the table that exist (codes >= K are flagged)
    int *status;             // include/rqae_hip.h "Status word", or NULL
};

template <typename CT> struct CodeBlock;
template <> struct CodeBlock<uint16_t> {
    static constexpr int NDW = 4;
    static __device__ __forceinline__ unsigned code(const unsigned (&w)[NDW], int j) { return (w[j >> 1] >> (16 * (j & 1))) & 0xffffu; }
};
template <> struct CodeBlock<int32_t> {
    static constexpr int NDW = 8;
    static __device__ __forceinline__ unsigned code(const unsigned (&w)[NDW], int j) { return w[j]; }
};

template <int SEL>
__device__ __forceinline__ unsigned quad_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SEL | (SEL << 2) | (SEL << 4) | (SEL << 6), 0xf, 0xf, true);
}

template <typename CT>
__global__ __launch_bounds__(S_NW * 64) void search_accumulate_kernel(AccParams p)
{
    using CB = CodeBlock<CT>;
    constexpr int NDW = CB::NDW, NSET = S_TS / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, grp = lane >> 2, sub = lane & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned slotb = (unsigned)p.KP * 64u;
    const int npieces = p.KP / 16;
    const int nl = p.l1 - p.l0;
    const unsigned qseg = blockIdx.y * 64u;                               // this workgroup's 64 bytes of every 256-byte row
    const unsigned voff = (unsigned)(lane >> 2) * 256u + (unsigned)sub * 16u;
    const long long tok0 = (long long)blockIdx.x * S_TOK + (long long)wave * 16 * S_TS;

    // every wave issues the same number of pieces per slice (the waits below count them): a wave past the last piece repeats it
    const int PW = (npieces + S_NW - 1) / S_NW;
    auto stage = [&](int li, int slot) {
#ifdef RQAE_SEARCH_TOOLS   /* tools build, measurement only (RQAE_SEARCH_DBG=1): no slice after the first ones -- wrong results */
        if (p.dbg & 1) { if (li >= p.nbuf - 1) return; }
#endif
        for (int i = 0; i < PW; ++i) {
            int pc = wave + i * S_NW;
            pc = pc < npieces ? pc : npieces - 1;
            const char *src = reinterpret_cast<const char *>(p.tq) + ((size_t)li * p.KP + 16 * pc) * (S_QP * 2) + qseg;
            dma_piece(uniform_ptr(src), lds0 + slot * slotb + pc * 1024, voff);
        }
    };
    // s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
    auto wait_vmcnt = [&](int n) {
        switch (n) {
#define W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16)
#undef W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    constexpr int NC = (S_TS / 4) * (CodeBlock<CT>::NDW / 4);     // fewest load instructions one block of codes can be (more only over-waits)
    // codes of level block [lb, lb + 8) for the two tokens this lane fetches (token slots sub and sub + 4 of its quad)
    auto load_codes = [&](int lb, unsigned (&cw)[NSET][NDW]) {
#pragma unroll
        for (int st = 0; st < NSET; ++st) {
            long long tok = tok0 + (long long)(4 * st + sub) * 16 + grp;
            tok = tok < p.NT ? tok : p.NT - 1;
            const CT *src = static_cast<const CT *>(p.codes) + (size_t)tok * p.Ltot + lb;
            if (lb + S_LB <= p.Ltot) {
                __builtin_memcpy(cw[st], src, NDW * 4);                    // 2-byte aligned rows: the hardware takes unaligned vector loads
            } else {
                // the last block of a row: never read past it (the row after may be past the array)
#pragma unroll
                for (int i = 0; i < NDW; ++i) cw[st][i] = 0u;
#pragma unroll
                for (int j = 0; j < S_LB; ++j)
                    if (lb + j < p.Ltot) {
                        const unsigned v = (unsigned)src[j];
                        if (sizeof(CT) == 2) cw[st][j >> 1] |= (v & 0xffffu) << (16 * (j & 1));
                        else cw[st][j] = v;
                    }
            }
        }
    };

    float acc[S_TS][8];
    h16x2 rng[S_TS][4];
    unsigned cmax = 0u;      // largest code this lane fetched (as unsigned: a negative int32 code is huge), checked once at the end
#pragma unroll
    for (int ts = 0; ts < S_TS; ++ts)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[ts][e] = 0.0f;
#pragma unroll
    for (int ts = 0; ts < S_TS; ++ts)
#pragma unroll
        for (int e = 0; e < 4; ++e) rng[ts][e] = h16x2{(_Float16)0.0f, (_Float16)0.0f};

    for (int i = 0; i < p.nbuf - 1 && i < nl; ++i) stage(i, i);
    unsigned cw[NSET][NDW], cwn[NSET][NDW];
    load_codes(p.l0, cw);
    int slot = 0;
    bool first_chunk = true;
    for (int lb = p.l0; lb < p.l1; lb += S_LB) {
        if (lb + S_LB < p.l1) load_codes(lb + S_LB, cwn);                  // one block ahead
#pragma unroll
        for (int j = 0; j < S_LB; ++j) {
            const int l = lb + j;
            if (l < p.l1) {
                const int li = l - p.l0;
                // this level's slice has landed (it was issued nbuf - 1 levels ago).  Younger operations that may stay in flight:
                // the nbuf - 2 slices after it and, in the first level of a block, the next block's codes issued just above
                // (from the second level on those codes are older than the youngest slice, so they have landed too).  Then
                // every wave is done with the slot about to be refilled.
                {
                    int ahead = nl - 1 - li;                       // slices issued after this one
                    ahead = ahead < p.nbuf - 2 ? ahead : p.nbuf - 2;
                    wait_vmcnt(ahead * PW + ((j == 0 && lb + S_LB < p.l1) ? NC : 0));
                }
                __builtin_amdgcn_s_barrier();
                if (li + p.nbuf - 1 < nl) {
                    int sl = slot + p.nbuf - 1;
                    sl = sl >= p.nbuf ? sl - p.nbuf : sl;
                    stage(li + p.nbuf - 1, sl);
                }
                const char *base = smem + slot * slotb + sub * 16;
                u32x4 row[S_TS];
#pragma unroll
                for (int ts = 0; ts < S_TS; ++ts) {
                    unsigned w;
                    const unsigned mine = CB::code(cw[ts >> 2], j);
                    if ((ts & 3) == 0) cmax = cmax > mine ? cmax : mine;       // each lane's own codes once (NSET per level)
                    switch (ts & 3) {
                    case 0: w = quad_bcast<0>(mine); break;
                    case 1: w = quad_bcast<1>(mine); break;
                    case 2: w = quad_bcast<2>(mine); break;
                    default: w = quad_bcast<3>(mine); break;
                    }
                    row[ts] = *reinterpret_cast<const u32x4 *>(base + (size_t)w * 64);     // a code past the slice reads other LDS bytes or, past the allocation, zeros: never a fault (flagged below)
                }
#pragma unroll
                for (int ts = 0; ts < S_TS; ++ts)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // fp16 -> fp32 and add in one operation each (v * 1.0 + acc, one rounding: the fp32 sum); the cheapest form
                        // measured (tools/ubench/mix_rate.hip)
                        asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(acc[ts][2 * e]) : "v"(row[ts][e]));
                        asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[ts][2 * e + 1]) : "v"(row[ts][e]));
                    }
                slot = slot + 1 == p.nbuf ? 0 : slot + 1;
            }
        }
        // a 64-level chunk (server.py:216-231) or the range ends with this block: round the fp32 sums to fp16, add chunks in fp16
        if ((((lb - p.l0) + S_LB) & (S_CHUNK - 1)) == 0 || lb + S_LB >= p.l1) {
#pragma unroll
            for (int ts = 0; ts < S_TS; ++ts)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16x2 h = h16x2{(_Float16)acc[ts][2 * e], (_Float16)acc[ts][2 * e + 1]};
                    rng[ts][e] = first_chunk ? h : rng[ts][e] + h;
                    acc[ts][2 * e] = 0.0f;
                    acc[ts][2 * e + 1] = 0.0f;
                }
            first_chunk = false;
        }
#pragma unroll
        for (int st = 0; st < NSET; ++st)
#pragma unroll
            for (int i = 0; i < NDW; ++i) cw[st][i] = cwn[st][i];
    }
    // a code outside the table: the reference raises IndexError (server.py:55); here the call is flagged (lanes of rows past the
    // dataset re-read its last row, so they add nothing new)
    if (cmax >= (unsigned)p.K && p.status) atomicOr(p.status, RQAE_STATUS_CODE_RANGE);
    // ---- the running accumulation over ranges (server.py:257-260)
#pragma unroll
    for (int ts = 0; ts < S_TS; ++ts) {
        const long long tok = tok0 + ts * 16 + grp;
        if (tok < p.NT) {
            h16x8 *dst = reinterpret_cast<h16x8 *>(reinterpret_cast<char *>(p.acc) + (size_t)tok * (S_QP * 2) + qseg + sub * 16);
            const h16x8 r = h16x8{rng[ts][0][0], rng[ts][0][1], rng[ts][1][0], rng[ts][1][1], rng[ts][2][0], rng[ts][2][1], rng[ts][3][0], rng[ts][3][1]};
            if (p.first) {
                *dst = r;
            } else {
                const h16x8 old = *dst;
                *dst = old + r;                 // four v_pk_add_f16: fp16 additions, round to nearest even
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ ranking
// keys[q][n] = max over s of acc[n][s][q] (server.py:263).  One workgroup per sequence: thread (row group
// tid >> 4, column block tid & 15) runs over every 16th position with 16-byte loads (8 query positions), then the 16 row
// groups meet in LDS.  A NaN takes the maximum and keeps it, like torch.max.
__device__ __forceinline__ float max_nan(float m, float v) { return (v > m || (v != v && m == m)) ? v : m; }

__global__ __launch_bounds__(256) void search_rowmax_kernel(const __half *acc, long long N, int S, int Q, __half *keys)
{
    __shared__ float part[16][S_QP + 4];
    const long long n = blockIdx.x;
    const int rg = threadIdx.x >> 4, cb = threadIdx.x & 15;
    const char *src = reinterpret_cast<const char *>(acc) + (size_t)n * S * (S_QP * 2) + cb * 16;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int s = rg; s < S; s += 16) {
        const h16x8 v = *reinterpret_cast<const h16x8 *>(src + (size_t)s * (S_QP * 2));
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = max_nan(m[e], (float)v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rg][cb * 8 + e] = m[e];
    __syncthreads();
    if (threadIdx.x < S_QP) {
        const int q = threadIdx.x;
        float r = part[0][q];
        for (int i = 1; i < 16; ++i) r = max_nan(r, part[i][q]);
        if (q < Q) {
            r = r == 0.0f ? 0.0f : r;                            // -0 ranks with +0 (their bit patterns would order them)
            keys[(size_t)q * N + n] = __float2half(r);          // exact: r is one of the fp16 inputs (or -inf for S < 1, excluded by the host)
        }
    }
}

// ---- ranking: per query position the sequences by descending maximum, equal maxima in ascending sequence order
// (server.py:262-266 argsorts the maxima; torch promises no order among equals, this one is the stable order).
// Sort key of sequence n: ((0xFFFF - m16(max)) << 32) | n, ascending, with m16 the usual order-preserving map of fp16 bits to
// unsigned (NaN on top, as torch.sort places it).  A bitonic network: search_sort_kernel sorts chunks of up to SORT_CH keys in
// LDS (and finishes every later stage's strides below the chunk length there), search_merge_kernel does the strides that cross
// chunks in global memory.  Pad keys are all ones: they end up behind every sequence.
constexpr int SORT_CH = 8192;            // keys per chunk: 64 KiB of LDS
constexpr int SORT_NT = 1024;            // threads per workgroup

__device__ __forceinline__ unsigned long long sort_key(__half v, unsigned n)
{
    const unsigned u = __half_as_ushort(v);
    unsigned m = (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
    if ((u & 0x7fffu) > 0x7c00u) m = 0xffffu;                       // NaN of either sign: the maximum
    return ((unsigned long long)(0xffffu - m) << 32) | n;
}

// keys64 [Q][NP] (NP = padded sequence count, a power of two >= ch).  MODE 0: build the keys from the maxima and sort every
// chunk of `ch` (stages k = 2 .. ch); MODE 1: stage `kstage` > ch, strides ch/2 .. 1.  last != 0: this launch completes the
// sort -- the first N keys of the row leave as (order, max) instead of going back to keys64.
template <int MODE>
__global__ __launch_bounds__(SORT_NT) void search_sort_kernel(const __half *maxima, unsigned long long *keys64, long long N, long long NP,
                                                               int ch, long long kstage, int last, __half *max_out, int *order_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    const int q = blockIdx.y;
    const long long base = (long long)blockIdx.x * ch;
    unsigned long long *row = keys64 + (size_t)q * NP;
    for (int i = threadIdx.x; i < ch; i += SORT_NT) {
        const long long n = base + i;
        if (MODE == 0) sk[i] = n < N ? sort_key(maxima[(size_t)q * N + n], (unsigned)n) : ~0ull;
        else sk[i] = row[n];
    }
    __syncthreads();
    auto strides = [&](long long k, int jfirst) {
        for (int j = jfirst; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < ch / 2; t += SORT_NT) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = ((base + lo) & k) == 0;             // ascending block of this stage
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a > b) == up) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    };
    if (MODE == 0) {
        for (int k = 2; k <= ch; k <<= 1) strides(k, k >> 1);
    } else {
        strides(kstage, ch >> 1);
    }
    for (int i = threadIdx.x; i < ch; i += SORT_NT) {
        const long long r = base + i;
        if (!last) {
            row[r] = sk[i];
        } else if (r < N) {
            const unsigned n = (unsigned)(sk[i] & 0xffffffffull);      // ranks < N hold sequences: the pads sort behind them
            order_out[(size_t)q * N + r] = (int)n;
            max_out[(size_t)q * N + r] = maxima[(size_t)q * N + n];
        }
    }
}

// one stride j >= ch of stage k, in global memory
__global__ __launch_bounds__(256) void search_merge_kernel(unsigned long long *keys64, long long NP, long long k, long long j)
{
    unsigned long long *row = keys64 + (size_t)blockIdx.y * NP;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= NP / 2) return;
    const long long lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
    const bool up = (lo & k) == 0;
    const unsigned long long a = row[lo], b = row[hi];
    if ((a > b) == up) { row[lo] = b; row[hi] = a; }
}

inline long long sort_pad(long long n)
{
    long long p = 2;
    while (p < n) p <<= 1;
    return p;
}

// out[q][e][s] = acc[idx[q][e]][s][q]  (server.py:293-311); an example outside the dataset (the reference raises IndexError)
// yields zeros and flags the call
__global__ __launch_bounds__(128) void search_gather_kernel(const __half *acc, long long N, int S, const int *idx, int E, __half *out, int *status)
{
    const int q = blockIdx.x, e = blockIdx.y;
    const long long n = idx[(size_t)q * E + e];
    const bool ok = n >= 0 && n < N;
    if (!ok && status && threadIdx.x == 0) atomicOr(status, RQAE_STATUS_CODE_RANGE);
    for (int s = threadIdx.x; s < S; s += blockDim.x)
        out[((size_t)q * E + e) * S + s] = ok ? acc[((size_t)n * S + s) * S_QP + q] : __float2half(0.0f);
}

int pick_nbuf(int KP)
{
    if (3 * (size_t)KP * 64 <= 160 * 1024) return 3;
    if (2 * (size_t)KP * 64 <= 160 * 1024) return 2;
    return 0;
}

}  // namespace

extern "C" {

int rqae_hip_search_supported(int K, int Q) { return K >= 1 && Q >= 1 && Q <= S_QP && pick_nbuf(((K + 15) / 16) * 16) > 0; }

size_t rqae_hip_search_table_bytes(int K, int n_levels)
{
    if (K < 1 || n_levels < 0) return 0;
    return (size_t)n_levels * (((K + 15) / 16) * 16) * S_QP * 2;
}

int rqae_hip_search_query_table(const void *sims16, int L, int K, const int32_t *query_codes, int Q, int64_t query_stride, int l0,
                                int l1, void *table_out, void *stream)
{
    if (!sims16 || !query_codes || !table_out) return sfail("search_query_table: null pointer");
    if (!rqae_hip_search_supported(K, Q)) return sfail("search_query_table: unsupported K=%d Q=%d (Q <= 128, K <= 1280)", K, Q);
    if (l0 < 0 || l1 > L || l0 > l1) return sfail("search_query_table: levels [%d, %d) outside 0..%d", l0, l1, L);
    if (l0 == l1) return 0;
    const int KP = ((K + 15) / 16) * 16;
    hipLaunchKernelGGL(search_query_table_kernel, dim3((unsigned)(l1 - l0), (unsigned)((KP + 63) / 64)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const __half *>(sims16), K, KP, query_codes, Q,
                       (long long)query_stride, l0, static_cast<__half *>(table_out));
    S_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_search_accumulate(const void *codes, int code_dtype, int64_t n_tokens, int L, const void *table, int K, int Q, int l0,
                               int l1, void *acc, int first, int32_t *status, void *stream)
{
    if (!codes || !table || !acc) return sfail("search_accumulate: null pointer");
    if (!rqae_hip_search_supported(K, Q)) return sfail("search_accumulate: unsupported K=%d Q=%d (Q <= 128, K <= 1280)", K, Q);
    if (l0 < 0 || l1 > L || l0 >= l1) return sfail("search_accumulate: levels [%d, %d) outside 0..%d or empty", l0, l1, L);
    if (code_dtype != RQAE_IDX_I32 && code_dtype != RQAE_IDX_U16) return sfail("search_accumulate: codes must be int32 or uint16");
    if (n_tokens < 0) return sfail("search_accumulate: negative n_tokens");
    if ((reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(table)) & 15) return sfail("search_accumulate: acc and table must be 16-byte aligned");
    if (n_tokens == 0) return 0;
    AccParams p;
    p.codes = codes; p.tq = static_cast<const __half *>(table); p.acc = static_cast<__half *>(acc); p.NT = n_tokens; p.Ltot = L;
    p.KP = ((K + 15) / 16) * 16; p.l0 = l0; p.l1 = l1; p.first = first ? 1 : 0; p.nbuf = pick_nbuf(p.KP);
    p.K = K; p.status = status;
    p.dbg = 0;
#ifdef RQAE_SEARCH_TOOLS   /* tools build: ring depth and the no-refill ablation (profiles/r02_search_ring_and_dma_ablation.txt) */
    static const int dbg = getenv("RQAE_SEARCH_DBG") ? atoi(getenv("RQAE_SEARCH_DBG")) : 0;
    static const int nbuf_env = getenv("RQAE_SEARCH_NBUF") ? atoi(getenv("RQAE_SEARCH_NBUF")) : 0;
    p.dbg = dbg;
    if (nbuf_env >= 2 && (size_t)nbuf_env * p.KP * 64 <= 160 * 1024) p.nbuf = nbuf_env;
#endif
    const size_t lds = (size_t)p.nbuf * p.KP * 64;
    const dim3 grid((unsigned)((n_tokens + S_TOK - 1) / S_TOK), (unsigned)((Q + S_QW - 1) / S_QW));
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        S_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(S_NW * 64), lds, s, p);
        S_TRY(hipGetLastError());
        return 0;
    };
    return code_dtype == RQAE_IDX_U16 ? go(search_accumulate_kernel<uint16_t>) : go(search_accumulate_kernel<int32_t>);
}

size_t rqae_hip_search_rank_workspace_bytes(int64_t n_seqs, int Q)
{
    if (n_seqs < 1 || Q < 1 || Q > S_QP || n_seqs * Q > 0x7fffffffLL) return 0;
    // the maxima [Q][N] fp16 and the 64-bit sort keys [Q][NP]
    const size_t maxima = ((size_t)n_seqs * Q * 2 + 255) & ~(size_t)255;
    return maxima + (size_t)Q * sort_pad(n_seqs) * 8 + 256;
}

int rqae_hip_search_rank(const void *acc, int64_t n_seqs, int S, int Q, void *max_out, int32_t *order_out, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    if (!acc || !max_out || !order_out || !workspace) return sfail("search_rank: null pointer");
    if (n_seqs < 1 || S < 1 || Q < 1 || Q > S_QP || n_seqs * Q > 0x7fffffffLL) return sfail("search_rank: bad shape");
    const size_t need = rqae_hip_search_rank_workspace_bytes(n_seqs, Q);
    if (!need || workspace_bytes < need) return sfail("search_rank: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 7) return sfail("search_rank: workspace must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t maxima_b = ((size_t)n_seqs * Q * 2 + 255) & ~(size_t)255;
    char *w = static_cast<char *>(workspace);
    __half *maxima = reinterpret_cast<__half *>(w);
    unsigned long long *keys64 = reinterpret_cast<unsigned long long *>(w + maxima_b);
    hipLaunchKernelGGL(search_rowmax_kernel, dim3((unsigned)n_seqs), dim3(256), 0, s, static_cast<const __half *>(acc), (long long)n_seqs, S, Q,
                       maxima);
    S_TRY(hipGetLastError());
    const long long NP = sort_pad(n_seqs);
    const int ch = NP < SORT_CH ? (int)NP : SORT_CH;
    const size_t lds = (size_t)ch * 8;
    const dim3 grid((unsigned)(NP / ch), (unsigned)Q);
    if (lds > 48 * 1024) {
        S_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(search_sort_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        S_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(search_sort_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(search_sort_kernel<0>, grid, dim3(SORT_NT), lds, s, maxima, keys64, (long long)n_seqs, NP, ch, 0LL, NP == ch ? 1 : 0,
                       static_cast<__half *>(max_out), order_out);
    S_TRY(hipGetLastError());
    for (long long k = 2LL * ch; k <= NP; k <<= 1) {
        for (long long j = k >> 1; j >= ch; j >>= 1) {
            hipLaunchKernelGGL(search_merge_kernel, dim3((unsigned)((NP / 2 + 255) / 256), (unsigned)Q), dim3(256), 0, s, keys64, NP, k, j);
            S_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(search_sort_kernel<1>, grid, dim3(SORT_NT), lds, s, maxima, keys64, (long long)n_seqs, NP, ch, k, k == NP ? 1 : 0,
                           static_cast<__half *>(max_out), order_out);
        S_TRY(hipGetLastError());
    }
    return 0;
}

int rqae_hip_search_gather(const void *acc, int64_t n_seqs, int S, int Q, const int32_t *examples, int E, void *out, int32_t *status,
                           void *stream)
{
    if (!acc || !examples || !out) return sfail("search_gather: null pointer");
    if (n_seqs < 1 || S < 1 || Q < 1 || Q > S_QP || E < 0) return sfail("search_gather: bad shape");
    if (E == 0) return 0;
    hipLaunchKernelGGL(search_gather_kernel, dim3((unsigned)Q, (unsigned)E), dim3(128), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __half *>(acc), (long long)n_seqs, S, examples, E, static_cast<__half *>(out), status);
    S_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
