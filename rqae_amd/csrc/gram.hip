// gram.hip -- the matrix-core arm of RQAE.forward for gfx950 ("gram arm").
//
// rqae/model.py:199-230 runs, per level l,  q = lin_in_l(r); idx = argmax cos(q, codebook_l);
// cv = q + (codebook_l[idx] - q); y = lin_out_l(cv); r -= y; out += y.  lin_in is linear in r, so
//     q_l = W_in[l] x + b_in[l] - sum_{m<l} ( (W_in[l] W_out[m]) cv_m + W_in[l] b_out[m] )
// ("all-levels-at-once"): the d-long contractions of every level become ONE dense
// [tokens, d] x [d, L*c] product, the level chain runs in c-space on c x c Gram blocks, and the
// reconstruction is the decode sum  out = sum_l (W_out[l] cv_l + b_out[l])  -- a second dense product.
// The residual is never formed.  The three products run on v_mfma_f32_16x16x32_bf16 with every fp32
// operand cut into three bf16 pieces (hi + mid + lo == the fp32 value exactly; a bf16 x bf16 product is
// exact in fp32; fp32 accumulation), so the results carry fp32-level rounding only.  They are NOT the
// bits of the canonical order: this arm is held to SURVEY 8(c)'s Tier-2 policy
// (codes identical wherever the top-1/top-2 margin is >= 1e-5, reconstruction within 1e-5 relative),
// the exact arm (rqae_hip_encode) stays bit-exact.
//
//   gram_project_kernel      P[tok][l*c+j] = sum_e x[tok][e] W_in[l][j][e]                (rqae/model.py:211, all l)
//   gram_levels_kernel       the level chain in c-space + cosine argmax on MFMA           (rqae/model.py:187-224)
//   gram_reconstruct_kernel  out[tok][e] = sum_l W_out[l][e][:] cv_l + b_out[l][e]        (rqae/model.py:221,224)
//
// Translation unit of librqae_hip.so (built by __graft_entry__.py); C ABI at the bottom, declared in
// include/rqae_hip.h.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "../../include/rqae_hip.h"
#include "recon3.h"
#include "proj3.h"
#include "deep3.h"
#include "sign3.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];   // __thread, not thread_local: no TLS init wrapper (a hidden weak _ZTH call crashes)

namespace {

int gfail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int gfail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define G_TRY(expr)                                                                      \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return gfail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#include "gram_small.hpp"

constexpr int GRAM_MAGIC = 0x4752414d;  // "GRAM"
constexpr int GRAM_MAX_L = 1024;       // calls of more than GRAM_DEEP_L levels take the blocked path of gram_deep3.hpp (c = 4, FSQ lattice)
constexpr int GRAM_DEEP_L = 64;
constexpr int GRAM_MAX_C = 13;          // instantiated: CP = 4, 5 (resident / streamed codebook), 13 (gram_wide.hpp: 6 <= c <= 13)

// ------------------------------------------------------------------------------------------------ geometry
// CP   padded codebook_dim the kernels are instantiated for (c <= CP, padding is zero weights)
// GL   levels per level group (the c-space state of one group lives in registers)
// GP   outputs per level group in P, padded to whole 16-wide MFMA tiles
// GS   reconstruction slots per level group, padded to whole 32-deep k-steps (6 bf16 x bf16 terms per component)
// NKC  k-steps of the cosine contraction (6*CP slots)
struct Geom {
    int d, c, K, L, CP, GL, GP, GS, NLG, NOG, KSA, NKC, KT, KTP, NET, NKSC, GB, shared_cb;
    int KS3;                  // k-steps (16 values) of the fp16 reconstruction image (gram_recon3.hpp): L c values, tight
};

__host__ __device__ inline int cp_of(int c) { return c <= 4 ? 4 : c == 5 ? 5 : 13; }
__host__ __device__ inline int gl_of(int CP) { return CP <= 8 ? 8 : 4; }

__host__ __device__ inline Geom make_geom(int d, int c, int K, int L, int shared_cb)
{
    Geom g;
    g.d = d; g.c = c; g.K = K; g.L = L; g.shared_cb = shared_cb;
    g.CP = cp_of(c);
    g.GL = gl_of(g.CP);
    g.GP = ((g.GL * g.CP + 15) / 16) * 16;
    g.GS = ((6 * g.CP * g.GL + 31) / 32) * 32;
    g.NLG = (L + g.GL - 1) / g.GL;
    g.NOG = g.NLG * (g.GP / 16);
    g.KSA = d / 32;
    g.NKC = (6 * g.CP + 31) / 32;
    g.KT = (K + 15) / 16;
    g.KTP = ((g.KT + 3) / 4) * 4;        // whole groups of 4 row tiles (pad rows are masked out in the scan)
    g.NET = d / 16;
    g.NKSC = g.NLG * (g.GS / 32);        // k-steps of the reconstruction at full depth
    g.GB = (g.CP * g.CP + 3) & ~3;       // floats per Gram block
    g.KS3 = (L * c + 15) / 16;
    return g;
}

__host__ __device__ inline bool gram_shape_ok(int d, int c, int K, int L)
{
    return d >= 128 && d % 128 == 0 && d <= 8192 && c >= 1 && c <= GRAM_MAX_C && K >= 1 && K <= 65536 && L >= 1 &&
           L <= GRAM_MAX_L;
}

// k-steps of the reconstruction for `levels` levels
__host__ __device__ inline int nks_recon(const Geom &g, int levels)
{
    const int full = levels / g.GL, rem = levels % g.GL;
    return full * (g.GS / 32) + (6 * g.CP * rem + 31) / 32;
}

// n with n^c == K (2 <= n <= 16), else 0: the lattice rqae_hip_gram_pack looks for (the device-side check confirms it or clears the header's word)
inline int lattice_points(int K, int c)
{
    for (int n = 2; n <= 16; ++n) {
        long long pw = 1;
        for (int j = 0; j < c && pw <= K; ++j) pw *= n;
        if (pw == K) return n;
    }
    return 0;
}

// a block that can run more than GRAM_DEEP_L levels (gram_deep3.hpp): c = 4, whole blocks of 32 levels and whole column blocks of the
// projection (L % 64 == 0), one shared codebook whose fp32 rows fit the lattice form's LDS
inline bool deep_block_ok(const Geom &g) { return g.L > GRAM_DEEP_L && g.c == 4 && g.L % 64 == 0 && g.shared_cb && (size_t)g.K * g.c * 4 <= 150 * 1024 && g.d % 128 == 0; }

struct GLayout {
    size_t win_img, kconst, gneg, cb_img, cb_f32, wout_img, bias_prefix, wout3_img, win3_img, g3_img, sign_img, total;
};

GLayout make_glayout(const Geom &g)
{
    GLayout lo;
    size_t off = 256;
    lo.win_img = off;     off += al256((size_t)g.KSA * g.NOG * 3 * 1024);
    lo.kconst = off;      off += al256(sizeof(float) * (size_t)g.NOG * 16);
    lo.gneg = off;        off += al256(sizeof(float) * (size_t)g.L * g.L * g.GB);
    const int lcb = g.shared_cb ? 1 : g.L;
    lo.cb_img = off;      off += al256((size_t)lcb * g.KTP * g.NKC * 1024);
    lo.cb_f32 = off;      off += al256(pad1k(sizeof(float) * ((size_t)lcb * g.K * g.c + 64 * g.c)));   // + slack: quad-row reads of pad rows; whole KiB: staged by 1-KiB LDS-DMA pieces
    lo.wout_img = off;    off += al256((size_t)g.NET * g.NKSC * 1024);
    lo.bias_prefix = off; off += al256(sizeof(float) * (size_t)g.L * g.d + 1024);                        // + 1 KiB: the fused kernel stages a row in whole 1-KiB pieces
    lo.wout3_img = off;   off += al256(rqae_r3::image_bytes(g.d, g.c, g.L));                              // fp16 two-piece W_out image (gram_recon3.hpp)
    lo.win3_img = off;    off += al256(rqae_p3::image_bytes(g.d, g.NOG));                                   // fp16 two-piece W_in image (gram_proj3.hpp)
    lo.g3_img = off;      off += deep_block_ok(g) ? al256(rqae_d3::g3_bytes(g.L)) : 0;                              // fp16 two-piece image of -(W_in W_out) (gram_deep3.hpp)
    lo.sign_img = off;    off += al256(rqae_s3::image_bytes(g.c, g.L));                                        // Gram blocks as the sign chain multiplies them (gram_sign3.hpp: c = 13, depth 16)
    lo.total = off;
    return lo;
}

struct GHeader {
    int magic, d, c, K, L, shared_cb;
    int lattice_n;            // n >= 2: the shared codebook is the FSQ lattice {linspace(-1, 1, n)}^c (gram_pack_lattice_kernel); else 0
    int lattice_miss[2];      // scratch of that check: rows that are not the unit-length lattice row / not the plain lattice row
    unsigned wout_maxbits;    // bits of max |W_out| (gram_pack_absmax_kernel): fixes the power-of-two scale of the fp16 image
    unsigned win_maxbits;     // the same for W_in (x in_scale)
    unsigned g3_maxbits;      // the same for the Gram blocks (gram_deep3.hpp)
    unsigned sign_maxbits;    // ... and for the sign chain's image of them (gram_sign3.hpp)
    int sign_exact;           // 1: every codebook row is exactly (+-mag)^c -- the sign chain may run
    int pad[50];
};
static_assert(sizeof(GHeader) == 256, "header");

// ------------------------------------------------------------------------------------------------ bf16 pieces
// fp32 = hi + mid + lo exactly, each piece a bf16 (truncation: hi keeps the top 8 significant bits, the
// remainders are exact in fp32 and hold <= 16 and <= 8 significant bits).
__host__ __device__ inline void split3(float v, unsigned &hi, unsigned &mid, unsigned &lo)
{
    const unsigned b = __builtin_bit_cast(unsigned, v);
    hi = b & 0xffff0000u;
    const float r1 = v - __builtin_bit_cast(float, hi);
    mid = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mid);
    lo = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
}
__device__ inline uint16_t piece_of(float v, int part)
{
    unsigned h, m, l;
    split3(v, h, m, l);
    return (uint16_t)((part == 0 ? h : part == 1 ? m : l) >> 16);
}
// q / ||q|| as the reference forms it before the cosine product (rqae/model.py:188: `x / x.norm(dim=-1, keepdim=True)`), with
// torch-CPU's order of the 2-norm written out (oracle/rqae_oracle.c: aten_norm2 -- whole groups of 8 as fma lanes summed in lane
// order, one group of 4 as rounded squares, an fma tail), IEEE square root and divisions.  The exact rescoring dots THIS vector
// with the candidate rows, so two distinct rows whose fp32 cosines tie or differ in the last bit resolve as they do in the
// reference (a zero projection gives NaN here as there; the callers' okq sends it to index 0).
template <int CP>
__device__ __forceinline__ void unit_dir(const float (&q)[CP], int c, float (&xn)[CP])
{
    float acc = 0.0f;
    int d = 0;
    if (c >= 8) {
        float buf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d0 = 0; d0 + 8 <= CP; d0 += 8)
            if (d0 + 8 <= c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) buf[j] = __builtin_fmaf(q[d0 + j], q[d0 + j], buf[j]);
                d = d0 + 8;
            }
        acc = buf[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc = acc + buf[j];
    }
#pragma unroll
    for (int j = 0; j < CP; ++j) {
        // the group of 4 (elements d .. d+3) as rounded squares, the rest as an fma tail; d is 0 or 8 for CP <= 16
        const bool in4 = (c - d >= 4) && j >= d && j < d + 4;
        const bool tail = j >= d + ((c - d >= 4) ? 4 : 0) && j < c;
        if (in4) { const float sq = q[j] * q[j]; acc = acc + sq; }
        else if (tail) acc = __builtin_fmaf(q[j], q[j], acc);
    }
    const float nrm = __builtin_sqrtf(acc);
#pragma unroll
    for (int j = 0; j < CP; ++j) xn[j] = j < c ? q[j] / nrm : 0.0f;
}

// the six products kept per component: (A piece, B piece); dropped: mid*lo, lo*mid, lo*lo (<= 2^-24 relative)
__device__ constexpr int APART[6] = {0, 0, 1, 0, 2, 1};
// B side per component, slot order: hi mid hi lo hi mid  -> dwords {hi|mid, hi|lo, hi|mid}

// bit casts by value: __builtin_bit_cast applied directly to an ext-vector ELEMENT lvalue (v[2]) reads element 0
// with this hipcc (seen in the ISA), so every reinterpretation goes through a by-value helper
__device__ __forceinline__ float f32_of(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned u32_of(float f) { return __builtin_bit_cast(unsigned, f); }

// low half = top 16 bits of a, high half = top 16 bits of b
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 1 KiB global -> LDS, lane i moves 16 bytes from its own address to lds_base + 16 i
__device__ __forceinline__ void glds16(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ pack kernels
// W_in image: fragment (ks, u, part) = 1 KiB = A operand of v_mfma_f32_16x16x32_bf16: lane 16g+r holds row r
// (output 16u+r), k = 8g+j (element 32ks+8g+j).
__global__ void gram_pack_win_kernel(Geom g, const float *w_in, const float *in_scale, uint16_t *img)
{
    const size_t total = (size_t)g.KSA * g.NOG * 3 * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t f = i >> 9;
        const int part = (int)(f % 3); f /= 3;
        const int u = (int)(f % g.NOG);
        const int ks = (int)(f / g.NOG);
        const int o = 16 * u + (lane & 15), e = 32 * ks + 8 * (lane >> 4) + j;
        const int lg = o / g.GP, rem = o % g.GP, li = rem / g.CP, jj = rem % g.CP, l = lg * g.GL + li;
        float v = 0.0f;
        if (rem < g.GL * g.CP && l < g.L && jj < g.c) {
            v = w_in[((size_t)l * g.c + jj) * g.d + e];
            if (in_scale) v = v * in_scale[e];     // a diagonal map in front of lin_in (Gemma's 1 + w) folded into the image: one fp32 product
        }
        img[i] = piece_of(v, part);
    }
}

// Gram blocks, negated:  gneg[m][l2][j][i] = -(W_in[l2] W_out[m])[j][i]   (fp64 sum, rounded once), l2 > m
// kconst[o] = b_in[l][j] - sum_{m<l} W_in[l][j][:] . b_out[m][:]
// grid = (L, L + 1): y < L -> block (m = x, l2 = y);  y == L -> kconst of level x
__global__ void gram_pack_gram_kernel(Geom g, const float *w_in, const float *b_in, const float *w_out, const float *b_out,
                                      float *gneg, float *kconst)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    if ((int)blockIdx.y < g.L) {
        const int m = blockIdx.x, l2 = blockIdx.y;
        float *blk = gneg + ((size_t)m * g.L + l2) * g.GB;
        if (l2 <= m) {
            if (t < g.GB) blk[t] = 0.0f;
            return;
        }
        for (int j = 0; j < g.CP; ++j)
            for (int i = 0; i < g.CP; ++i) {
                double s = 0.0;
                if (j < g.c && i < g.c)
                    for (int e = t; e < g.d; e += 256)
                        s += (double)w_in[((size_t)l2 * g.c + j) * g.d + e] * (double)w_out[((size_t)m * g.d + e) * g.c + i];
                red[t] = s;
                __syncthreads();
                for (int st = 128; st > 0; st >>= 1) {
                    if (t < st) red[t] += red[t + st];
                    __syncthreads();
                }
                if (t == 0) blk[j * g.CP + i] = (float)(-red[0]);
                __syncthreads();
            }
        if (t >= g.CP * g.CP && t < g.GB) blk[t] = 0.0f;
    } else {
        const int l = blockIdx.x;
        for (int j = 0; j < g.CP; ++j) {
            double s = 0.0;
            if (j < g.c)
                for (int e = t; e < g.d; e += 256) {
                    double bp = 0.0;
                    for (int m = 0; m < l; ++m) bp += (double)b_out[(size_t)m * g.d + e];
                    s += (double)w_in[((size_t)l * g.c + j) * g.d + e] * bp;
                }
            red[t] = s;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (t < st) red[t] += red[t + st];
                __syncthreads();
            }
            if (t == 0) {
                const int o = (l / g.GL) * g.GP + (l % g.GL) * g.CP + j;
                kconst[o] = j < g.c ? (float)((double)b_in[(size_t)l * g.c + j] - red[0]) : 0.0f;
            }
            __syncthreads();
        }
    }
}

// Is the shared codebook the FSQ lattice (gram_wide.hpp, "FSQ lattice")?  Row k has digits k / n^(c-1-j) % n, coordinate values
// -1 + 2 digit / (n - 1), every row scaled to unit length (the zero row of an odd n stays zero) or every row left as it is.  One
// thread per row, values in double, 1e-6 absolute (the stored rows are those doubles rounded to fp32: 6e-8).
__global__ void gram_pack_lattice_kernel(const float *codebook, int K, int c, int n, GHeader *hdr)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double v[GRAM_MAX_C], ss = 0.0;
    int rem = k;
    for (int j = c - 1; j >= 0; --j) {
        v[j] = -1.0 + 2.0 * (double)(rem % n) / (double)(n - 1);
        rem /= n;
        ss += v[j] * v[j];
    }
    const double inv = ss > 0.0 ? 1.0 / sqrt(ss) : 1.0;
    bool unit = true, plain = true;
    for (int j = 0; j < c; ++j) {
        const double x = (double)codebook[(size_t)k * c + j];
        unit = unit && fabs(x - v[j] * inv) <= 1e-6;
        plain = plain && fabs(x - v[j]) <= 1e-6;
    }
    if (!unit) hdr->lattice_miss[0] = 1;
    if (!plain) hdr->lattice_miss[1] = 1;
}
__global__ void gram_pack_lattice_verdict_kernel(GHeader *hdr)
{
    if (hdr->lattice_miss[0] && hdr->lattice_miss[1]) hdr->lattice_n = 0;
}

// codebook image: fragment (lcb, tile, ks) = A operand: lane 16g+r holds codebook row 16 tile + r, slots
// 32ks+8g+j; slot s -> component s/6, term s%6 -> piece APART[term] of codebook[row][component].
__global__ void gram_pack_cb_kernel(Geom g, const float *codebook, uint16_t *img, float *cb_f32)
{
    const int lcb = g.shared_cb ? 1 : g.L;
    const size_t total = (size_t)lcb * g.KTP * g.NKC * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t f = i >> 9;
        const int ks = (int)(f % g.NKC); f /= g.NKC;
        const int tile = (int)(f % g.KTP);
        const int lv = (int)(f / g.KTP);
        const int row = 16 * tile + (lane & 15), s = 32 * ks + 8 * (lane >> 4) + j;
        const int comp = s / 6, term = s % 6;
        // rows past K repeat row 0: a repeat can tie with, never beat, the row it repeats, and ties go to the lower row
        // block -- so a scan over whole tiles needs no masking (gram_fused_kernel; gram_levels_kernel masks anyway)
        const int rsrc = row < g.K ? row : 0;
        float v = 0.0f;
        if (comp < g.c) v = codebook[((size_t)lv * g.K + rsrc) * g.c + comp];
        img[i] = piece_of(v, APART[term]);
    }
    const size_t nf = (size_t)lcb * g.K * g.c, nall = nf + 64 * g.c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nall; i += (size_t)gridDim.x * blockDim.x)
        cb_f32[i] = i < nf ? codebook[i] : 0.0f;
}

// W_out image: fragment (et, ks) = A operand: lane 16g+r holds element 16et+r, slots 32ks+8g+j; the slot of
// (level l, component jj, term) is (l / GL) * GS + ((l % GL) * CP + jj) * 6 + term.
__global__ void gram_pack_wout_kernel(Geom g, const float *w_out, uint16_t *img)
{
    const size_t total = (size_t)g.NET * g.NKSC * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t f = i >> 9;
        const int ks = (int)(f % g.NKSC);
        const int et = (int)(f / g.NKSC);
        const int e = 16 * et + (lane & 15), s = 32 * ks + 8 * (lane >> 4) + j;
        const int lg = s / g.GS, rs = s % g.GS;
        const int term = rs % 6, cj = rs / 6, li = cj / g.CP, jj = cj % g.CP, l = lg * g.GL + li;
        float v = 0.0f;
        if (rs < 6 * g.CP * g.GL && l < g.L && jj < g.c) v = w_out[((size_t)l * g.d + e) * g.c + jj];
        img[i] = piece_of(v, APART[term]);
    }
}

__global__ void gram_pack_bias_kernel(Geom g, const float *b_out, float *bias_prefix)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.d) return;
    double s = 0.0;
    for (int l = 0; l < g.L; ++l) {
        s += (double)b_out[(size_t)l * g.d + e];
        bias_prefix[(size_t)l * g.d + e] = (float)s;
    }
}

// RQAE.normalize_codebooks as the level loop applies it (rqae/model.py:126-131, called once per level from quantize, :196): every row of
// EVERY level is divided by its 2-norm once per level of the call -- `levels` times in all, in place -- and level l is quantised
// against its rows as they stand after l + 1 of them (renormalising a unit row is not a bit-wise no-op until it reaches a fixed
// point, usually after one or two steps).  Rows are independent, so one thread takes one row through all `levels` steps: torch-CPU's
// order of the 2-norm (oracle/rqae_oracle.c: aten_norm2 -- groups of 8 as fma lanes, one group of 4 as rounded squares, an fma tail),
// IEEE square root and divisions: the reference's bits.  used[l] (l < levels) receives level l's snapshot; cb the final state.
__global__ void normalize_codebooks_kernel(float *cb, int L, int K, int c, int levels, float *used)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (long long)L * K) return;
    const int lam = (int)(row / K);
    float v[16], snap[16];
    for (int j = 0; j < 16; ++j) { v[j] = j < c ? cb[row * c + j] : 0.0f; snap[j] = v[j]; }
    bool snapped = false;
    for (int t = 1; t <= levels; ++t) {
        float acc = 0.0f;
        int dd = 0;
        if (c >= 8) {
            float buf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int d0 = 0; d0 + 8 <= c; d0 += 8) {
                for (int j = 0; j < 8; ++j) buf[j] = __builtin_fmaf(v[d0 + j], v[d0 + j], buf[j]);
                dd = d0 + 8;
            }
            acc = buf[0];
            for (int j = 1; j < 8; ++j) acc = acc + buf[j];
        }
        if (c - dd >= 4) {
            for (int j = 0; j < 4; ++j) { const float sq = v[dd + j] * v[dd + j]; acc = acc + sq; }
            dd += 4;
        }
        for (; dd < c; ++dd) acc = __builtin_fmaf(v[dd], v[dd], acc);
        const float nrm = __builtin_sqrtf(acc);
        bool same = true;
        for (int j = 0; j < 16; ++j)
            if (j < c) {
                const float nv = v[j] / nrm;
                same = same && (__builtin_bit_cast(unsigned, nv) == __builtin_bit_cast(unsigned, v[j]));
                v[j] = nv;
            }
        if (t == lam + 1) {
            for (int j = 0; j < 16; ++j) snap[j] = v[j];
            snapped = true;
        }
        if (same) break;      // a fixed point: every further step leaves the row as it is
    }
    if (!snapped)
        for (int j = 0; j < 16; ++j) snap[j] = v[j];      // the fixed point was reached before step lam + 1 (or lam >= levels: unused)
    for (int j = 0; j < c; ++j) cb[row * c + j] = v[j];
    if (used)      // (levels past the call's depth: their final rows -- unused by the call, but never uninitialised memory)
        for (int j = 0; j < c; ++j) used[row * c + j] = lam < levels ? snap[j] : v[j];
}

// ------------------------------------------------------------------------------------------------ kernel A: project
struct ProjParams {
    const void *x;
    const uint16_t *win_img;
    float *P;              // [NOG][tok_pad][16]
    float *dump;           // 1 KiB of scratch (pipelined form: stores of rows / groups that do not exist)
    long long n_tokens, tok_pad;
    int d, NOG, KSA;
};

constexpr int A_KC = 4;     // k-steps per staged weight chunk
constexpr int A_TT = 2;     // 16-token tiles per wave
constexpr int A_NW = 4;     // waves per workgroup
constexpr int A_BUF = A_KC * 9 * 1024;

// x fragment of one k-step as bf16 pieces (B operand: lane 16g+n holds token n, elements 32ks+8g+0..7)
template <int XT> struct XFrag;
template <> struct XFrag<RQAE_X_BF16> {
    static constexpr int NP = 1;
    u32x4 raw;
    __device__ __forceinline__ void load(const void *x, long long row, int d, int e0) { raw = *reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(x) + row * d + e0); }
    __device__ __forceinline__ void parts(u32x4 (&o)[3]) const { o[0] = raw; }
};
template <> struct XFrag<RQAE_X_F16> {
    static constexpr int NP = 2;
    u32x4 raw;
    __device__ __forceinline__ void load(const void *x, long long row, int d, int e0) { raw = *reinterpret_cast<const u32x4 *>(static_cast<const uint16_t *>(x) + row * d + e0); }
    __device__ __forceinline__ void parts(u32x4 (&o)[3]) const
    {
        // fp16 (11 significant bits) = hi (8 bits) + lo (<= 3 bits), both exact bf16
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            const unsigned rw = raw[w];
            const h2 hv = __builtin_bit_cast(h2, rw);
            const float a = (float)hv[0], b = (float)hv[1];
            const unsigned ah = __builtin_bit_cast(unsigned, a) & 0xffff0000u, bh = __builtin_bit_cast(unsigned, b) & 0xffff0000u;
            const float al = a - __builtin_bit_cast(float, ah), bl = b - __builtin_bit_cast(float, bh);
            o[0][w] = pack_hi16(ah, bh);
            o[1][w] = pack_hi16(__builtin_bit_cast(unsigned, al), __builtin_bit_cast(unsigned, bl));
        }
    }
};
template <> struct XFrag<RQAE_X_F32> {
    static constexpr int NP = 3;
    u32x4 raw0, raw1;
    __device__ __forceinline__ void load(const void *x, long long row, int d, int e0)
    {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(static_cast<const float *>(x) + row * d + e0);
        raw0 = p[0]; raw1 = p[1];
    }
    __device__ __forceinline__ void parts(u32x4 (&o)[3]) const
    {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = f32_of(w < 2 ? raw0[2 * w] : raw1[2 * w - 4]);
            const float b = f32_of(w < 2 ? raw0[2 * w + 1] : raw1[2 * w - 3]);
            unsigned ah, am, al, bh, bm, bl;
            split3(a, ah, am, al);
            split3(b, bh, bm, bl);
            o[0][w] = pack_hi16(ah, bh);
            o[1][w] = pack_hi16(am, bm);
            o[2][w] = pack_hi16(al, bl);
        }
    }
};

// One workgroup = 4 waves x 32 tokens, up to 3 output groups (9 weight fragments per k-step: 3 groups x 3
// pieces).  The weight image streams L2 -> LDS by LDS-DMA in chunks of A_KC k-steps, double buffered; x goes
// straight to registers (every x byte is used by exactly one wave), one chunk ahead.
template <int XT>
__global__ __launch_bounds__(A_NW * 64) void gram_project_kernel(ProjParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;   // wave: provably uniform -> scalar branches
    const int ug0 = blockIdx.y * 3;
    const int ng = p.NOG - ug0 < 3 ? p.NOG - ug0 : 3;
    const long long tb = ((long long)blockIdx.x * A_NW + wave) * (16 * A_TT);
    long long row[A_TT];
#pragma unroll
    for (int T = 0; T < A_TT; ++T) {
        const long long r = tb + 16 * T + n;
        row[T] = r < p.n_tokens ? r : p.n_tokens - 1;
    }
    f32x4 acc[3][3][A_TT];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int T = 0; T < A_TT; ++T) acc[u][pp][T] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.KSA / A_KC;
    auto stage = [&](int cc, int buf) {
        const int npc = A_KC * ng * 3;
        for (int q = wave; q < npc; q += A_NW) {
            const int k = q / (ng * 3), rem = q - k * (ng * 3), u = rem / 3, pp = rem - 3 * u;
            const uint16_t *src = p.win_img + ((((size_t)(cc * A_KC + k) * p.NOG + ug0 + u) * 3 + pp) * 512) + lane * 8;
            glds16(src, smem + buf * A_BUF + ((k * 3 + u) * 3 + pp) * 1024);
        }
    };
    XFrag<XT> xc[A_KC][A_TT], xn[A_KC][A_TT];
    stage(0, 0);
#pragma unroll
    for (int k = 0; k < A_KC; ++k)
#pragma unroll
        for (int T = 0; T < A_TT; ++T) xc[k][T].load(p.x, row[T], p.d, 32 * k + 8 * g);
    __syncthreads();

    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) {
            stage(cc + 1, buf ^ 1);
#pragma unroll
            for (int k = 0; k < A_KC; ++k)
#pragma unroll
                for (int T = 0; T < A_TT; ++T) xn[k][T].load(p.x, row[T], p.d, 32 * ((cc + 1) * A_KC + k) + 8 * g);
        }
        const char *wb = smem + buf * A_BUF + lane * 16;
#pragma unroll
        for (int k = 0; k < A_KC; ++k) {
            u32x4 xp[A_TT][3];
#pragma unroll
            for (int T = 0; T < A_TT; ++T) xc[k][T].parts(xp[T]);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u < ng) {
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) {
                        const u32x4 a = *reinterpret_cast<const u32x4 *>(wb + ((k * 3 + u) * 3 + pp) * 1024);
#pragma unroll
                        for (int T = 0; T < A_TT; ++T)
#pragma unroll
                            for (int xq = 0; xq < XFrag<XT>::NP; ++xq)
                                if (pp + xq <= 2) acc[u][pp][T] = mfma16(a, xp[T][xq], acc[u][pp][T]);   // the six largest of the nine piece products
                    }
                }
            }
        }
        __syncthreads();   // next chunk landed (the barrier's fence drains the DMA), this buffer is free
#pragma unroll
        for (int k = 0; k < A_KC; ++k)
#pragma unroll
            for (int T = 0; T < A_TT; ++T) xc[k][T] = xn[k][T];
    }
    // D layout: lane (g, n) holds outputs 4g..4g+3 of token n; pieces summed small to large
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        if (u < ng) {
#pragma unroll
            for (int T = 0; T < A_TT; ++T) {
                const long long tok = tb + 16 * T + n;
                if (tok < p.n_tokens) {
                    const f32x4 v = (acc[u][2][T] + acc[u][1][T]) + acc[u][0][T];
                    *reinterpret_cast<f32x4 *>(p.P + (((size_t)(ug0 + u) * p.tok_pad + tok) * 16 + 4 * g)) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ kernel A, pipelined form
// Same arithmetic as gram_project_kernel (same fragments, same accumulators, same final sums: identical bits), different
// data movement.  The first form mixes an LDS-DMA weight stream with plain register loads of x and ends every chunk in
// __syncthreads(): hipcc then waits vmcnt(0) at each barrier (an LDS-DMA is a pending LDS write) and the x loads issued
// at the top of the chunk are drained with it -- every chunk pays a full HBM round trip (measured: 0.30 ms for a 1.2 GB
// read that the same access pattern streams in 0.195 ms, profiles/r02_load_patterns.txt).  Here every vector-memory
// operation of the main loop is an asm statement hipcc does not count, and the waits are counted by hand:
//   * weights: 3-stage LDS ring filled by global_load_lds_dwordx4, chunk cc+2 issued while chunk cc is multiplied;
//   * x: straight to registers (every byte is used by one wave), a 3-deep register ring, chunk cc+2 issued with it;
//   * per chunk ONE s_waitcnt vmcnt(#x loads of a chunk) -- the newest group stays in flight -- and ONE raw s_barrier.
// One workgroup = 8 waves x 32 tokens; KC k-steps per chunk (4 for 2-byte x, 2 for fp32 x: the ring holds 3 KC x-fragments).
constexpr int A2_NW = 8;
constexpr int A2_NS = 3;

template <int XT> struct XRaw;
template <> struct XRaw<RQAE_X_BF16> { static constexpr int NLD = 1, KC = 4, ESZ = 2; };
template <> struct XRaw<RQAE_X_F16> { static constexpr int NLD = 1, KC = 4, ESZ = 2; };
template <> struct XRaw<RQAE_X_F32> { static constexpr int NLD = 2, KC = 2, ESZ = 4; };

template <int XT>
__device__ __forceinline__ void xparts(const u32x4 (&raw)[XRaw<XT>::NLD], u32x4 (&o)[3])
{
    XFrag<XT> f;
    if constexpr (XT == RQAE_X_F32) { f.raw0 = raw[0]; f.raw1 = raw[1]; } else { f.raw = raw[0]; }
    f.parts(o);
}

// ABL: compile-time ablation bits for tools/ builds (-DRQAE_GRAM_ABLATE; 0 in the product): 1 = no MFMAs, 2 = no weight
// DMA, 4 = no barrier, 8 = no fragment reads from LDS
template <int XT, int ABL = 0>
__global__ __launch_bounds__(A2_NW * 64) void gram_project2_kernel(ProjParams p)
{
    constexpr int KC = XRaw<XT>::KC, NLD = XRaw<XT>::NLD, ESZ = XRaw<XT>::ESZ;
    constexpr int STAGE = KC * 9 * 1024;
    constexpr int XLOADS = KC * A_TT * NLD;          // x loads per chunk and wave
    constexpr int PSTORES = 3 * A_TT;                // P stores per tile and wave (issued whether or not the rows exist: see below)
    static_assert(XLOADS == 8, "the wait statements below name 8 x registers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int ug0 = blockIdx.y * 3;
    const size_t rowb = (size_t)p.d * ESZ;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int nchunks = p.KSA / KC;
    constexpr int TILE = A2_NW * 16 * A_TT;          // tokens per workgroup tile
    const long long ntiles = (p.n_tokens + TILE - 1) / TILE;
    // persistent: this workgroup owns tiles blockIdx.x, blockIdx.x + gridDim.x, ...; their chunks form ONE stream
    // s = 0 .. S-1 (tile s / nchunks, chunk s % nchunks), so the prefetch runs across tile boundaries
    const long long mytiles = ntiles > blockIdx.x ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long long S = mytiles * nchunks;
    if (S == 0) return;

    f32x4 acc[3][3][A_TT];
    u32x4 xr[A2_NS][KC][A_TT][NLD];

    // lane's row offset from its tile's first row; rows past the end re-read the last row (their results are not stored)
    auto row_off = [&](long long tile, int T) -> unsigned {
        const long long tile0 = tile * TILE;
        long long r = tile0 + (long long)wave * (16 * A_TT) + 16 * T + n;
        r = r < p.n_tokens ? r : p.n_tokens - 1;
        return (unsigned)((r - tile0) * (long long)rowb) + (unsigned)(g * 8 * ESZ);
    };
    // group s = this wave's weight pieces of the chunk (LDS-DMA) followed by its x fragments (registers), in that order.
    // Streams past the end (s >= S: at most 4 groups per workgroup) repeat the last chunk: the operation counts stay static.
    // (xr is passed in: clang rejects asm operands that name captures of a generic lambda)
    long long iti = 0;     // issue stream position: tile ordinal (of this workgroup's tiles) and chunk
    int icc = 0;
    auto issue = [&](auto slot_c, u32x4 (&xr)[A2_NS][KC][A_TT][NLD]) {
        constexpr int SLOT = decltype(slot_c)::value;
        const int cc = icc;
        const long long tile = blockIdx.x + iti * gridDim.x;
        if (!(iti == mytiles - 1 && icc == nchunks - 1)) {   // advance; at the end of the stream stay on the last chunk
            if (++icc == nchunks) { icc = 0; ++iti; }
        }
        for (int q = wave; q < ((ABL & 2) ? 0 : KC * 9); q += A2_NW) {
            const int k = q / 9, f = q - k * 9, u = f / 3;
            const int uu = ug0 + u < p.NOG ? ug0 + u : p.NOG - 1;     // groups past the last one: any valid fragment (never stored)
            const void *src = uniform_ptr(p.win_img + ((((size_t)(cc * KC + k) * p.NOG + uu) * 3 + (f - 3 * u)) * 512));
            dma_piece(src, lds0 + SLOT * STAGE + (k * 9 + f) * 1024, lane16);
        }
        const void *xb = uniform_ptr(static_cast<const char *>(p.x) + (size_t)tile * TILE * rowb + (size_t)cc * KC * 32 * ESZ);
        unsigned voff[A_TT];
#pragma unroll
        for (int T = 0; T < A_TT; ++T) voff[T] = row_off(tile, T);
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
    // ONE statement holds the wait and names the eight destinations, so no use or copy of them can be placed above the wait
    auto landed = [&](auto slot_c, u32x4 (&xr)[A2_NS][KC][A_TT][NLD]) {
        constexpr int SLOT = decltype(slot_c)::value;
        u32x4 *r = &xr[SLOT][0][0][0];
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "n"(XLOADS + PSTORES)
                     : "memory");
    };

    issue(std::integral_constant<int, 0>{}, xr);
    issue(std::integral_constant<int, 1>{}, xr);
    bool after_tile_end = false;       // the previous chunk ended a tile: its P stores are younger than group s+1
    long long cti = 0;                 // compute stream position
    int ccc = 0;
    for (long long s0 = 0; s0 < S; s0 += A2_NS) {
        static_for<0, A2_NS>([&](auto ph_c) {
            constexpr int PH = decltype(ph_c)::value;
            const long long s = s0 + PH;
            // Group s has landed when only the x loads of group s+1 are younger (vmcnt(8)) -- or, right after a tile end,
            // those and the tile's P stores (vmcnt(8 + 6)).  The operand-free statement is the stricter, conditional one.
            if (!after_tile_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XLOADS) : "memory");
            landed(ph_c, xr);
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();       // every wave's pieces of stage PH landed; every wave has left chunk s-1
            const long long ti = cti;
            const int cc = ccc;
            if (s < S) {
                if (++ccc == nchunks) { ccc = 0; ++cti; }
                if (cc == 0) {
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                            for (int T = 0; T < A_TT; ++T) acc[u][pp][T] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const char *wb = smem + PH * STAGE + lane * 16;
                // The chunk's KC*9 weight fragments, each feeding MF MFMAs.  Written as "all reads, then all MFMAs" and
                // put in order by sched_group_barrier: PF reads ahead, then one read per fragment's MFMAs -- left to
                // itself hipcc reads two fragments, waits for them, multiplies, and so exposes the LDS latency 18 times
                // per chunk (measured: 0.31 ms per 262144 tokens, no better than the unpipelined form).
                constexpr int NF = KC * 9, MF = A_TT * XFrag<XT>::NP, PF = 6;
                u32x4 xp[KC][A_TT][3];
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int T = 0; T < A_TT; ++T) xparts<XT>(xr[PH][k][T], xp[k][T]);
                u32x4 fr[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) fr[i] = (ABL & 8) ? xp[i % KC][0][0] : *reinterpret_cast<const u32x4 *>(wb + i * 1024);
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int k = i / 9, u = (i % 9) / 3, pp = i % 3;
#pragma unroll
                    for (int T = 0; T < A_TT; ++T)
#pragma unroll
                        for (int xq = 0; xq < XFrag<XT>::NP; ++xq) {
                            if (ABL & 1) { if (i < KC * A_TT) acc[u][pp][T] += __builtin_bit_cast(f32x4, xp[k][T][xq] ^ fr[i]); }
                            else if (pp + xq <= 2) acc[u][pp][T] = mfma16(fr[i], xp[k][T][xq], acc[u][pp][T]);   // W piece x x piece: the six largest of nine (mid*lo, lo*mid, lo*lo are below 2^-24 relative), as in the c-space contractions
                        }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int i = 0; i < NF - PF; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF * MF, 0);
            }
            // stage (s+2) % 3 was read last in chunk s-1, which every wave left before the barrier above
            issue(std::integral_constant<int, (PH + 2) % A2_NS>{}, xr);
            after_tile_end = (s < S) && (cc == nchunks - 1);
            if (after_tile_end) {
                // D layout: lane (g, n) holds outputs 4g..4g+3 of token n; pieces summed small to large.  Exactly PSTORES
                // store instructions per wave (rows past the end and groups past the last go to a dump row of the
                // workspace), so the wait above can count them.
                const long long tile = blockIdx.x + ti * gridDim.x;
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int T = 0; T < A_TT; ++T) {
                        const long long tok = tile * TILE + (long long)wave * (16 * A_TT) + 16 * T + n;
                        const bool live = tok < p.n_tokens && ug0 + u < p.NOG;
                        const f32x4 v = (acc[u][2][T] + acc[u][1][T]) + acc[u][0][T];
                        float *dst = live ? p.P + (((size_t)(ug0 + u) * p.tok_pad + tok) * 16 + 4 * g) : p.dump + lane * 4;
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated groups past the end of the stream
}

// ------------------------------------------------------------------------------------------------ kernel B: levels
struct LevParams {
    float *P;                 // [NOG][tok_pad][16]; columns of later level groups are updated in place
    const float *kconst;      // [NOG*16]
    const float *gneg;        // [L][L][GB]
    const uint16_t *cb_img;   // [lcb][KTP][NKC][512]
    const float *cb_f32;      // [lcb][K][c]
    unsigned *cvimg;          // [nks][tok_pad][16 dwords]: B operand rows of the reconstruction
    long long *idx;           // [n_tokens][idx_stride]
    long long n_tokens, tok_pad, ntiles;
    int levels, L, K, KTP, c, shared_cb;
    int idx_stride;           // row length of idx (= levels, but for a block of a deeper call: gram_deep3.hpp)
    const int *lat_flag;      // the block's lattice word (GHeader::lattice_n: n >= 2 when the codebook is an FSQ lattice), or NULL
    const int *sign_flag = nullptr;     // GHeader::sign_exact when the sign chain (gram_sign3.hpp) was launched ahead of this kernel, or NULL: it ran if the word is 1 and the lattice has two points
};

// One wave = 64 tokens, one token per lane for everything in c-space; for the cosine contraction the wave's
// tokens are 4 MFMA column tiles.  S[row][token] = sum_slots cb_piece[row][slot] * q_piece[token][slot] over the
// 6c slots of one (NKC) k-step(s): six bf16 x bf16 products per component reproduce the fp32 x fp32 product to
// 2^-24.  Scores are unnormalised (argmax of cos == argmax of the dot product for a positive norm); a zero,
// NaN or infinite projection takes index 0 as torch.argmax over NaNs does (rqae/model.py:182,188).
template <int CP, int NW, bool MULTI>
__global__ __launch_bounds__(NW * 64) void gram_levels_kernel(LevParams p)
{
    if (p.lat_flag && *reinterpret_cast<const volatile int *>(p.lat_flag) >= 2) return;      // the lattice form ran (gram_wide.hpp)
    constexpr int GL = CP <= 8 ? 8 : 4;
    constexpr int GP = ((GL * CP + 15) / 16) * 16;
    constexpr int GS = ((6 * CP * GL + 31) / 32) * 32;
    constexpr int NKC = (6 * CP + 31) / 32;
    constexpr int GB = (CP * CP + 3) & ~3;
    constexpr int XCHB = NKC * 4096;
    static_assert(NKC == 1, "one k-step per cosine contraction in the instantiated shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;   // wave: provably uniform -> scalar branches
    const int imgb = p.KTP * NKC * 1024;
    const int c = p.c;
    const int cbfb = ((p.K + 64) * c * 4 + 15) & ~15;   // fp32 codebook (+ 64 pad rows for the quad reads of the last block)
    char *cbl = smem;
    float *cbf_lds = reinterpret_cast<float *>(smem + imgb);
    char *xch = smem + imgb + cbfb + wave * XCHB;
    float *sgram = reinterpret_cast<float *>(smem + imgb + cbfb + NW * XCHB);     // the current level group's Gram blocks [li][li2][GB]

    // resident codebook image (shared codebooks: one image for every level)
    for (int q = wave; q < p.KTP * NKC; q += NW) glds16(p.cb_img + (size_t)q * 512 + lane * 8, cbl + q * 1024);
    // ... and its fp32 rows (rescoring and the straight-through value gather them: an LDS gather instead of ~20 dependent
    // L2 round trips per level)
    for (int i = tid; i < (p.K + 64) * c; i += NW * 64) cbf_lds[i] = p.cb_f32[i];
    __syncthreads();

    const int nlg = (p.levels + GL - 1) / GL;
    if (!MULTI) {
        // one level group: its Gram blocks into LDS (broadcast reads instead of dependent L2 round trips in every level)
        for (int i = tid; i < GL * GL * GB; i += NW * 64) {
            const int blk = i / GB, li = blk / GL, li2 = blk - li * GL;
            sgram[i] = (li < p.L && li2 < p.L) ? p.gneg[((size_t)li * p.L + li2) * GB + (i - blk * GB)] : 0.0f;
        }
        __syncthreads();
    }
    const int wq = (lane >> 2) & 3;            // row quad of this lane's token inside its 16-token tile
    for (long long tile = (long long)blockIdx.x * NW + wave; tile < p.ntiles; tile += (long long)gridDim.x * NW) {
        const long long t = tile * 64 + lane;  // < tok_pad
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
            unsigned img[GS / 2];                // this group's reconstruction slots, two bf16 per dword
#pragma unroll
            for (int i = 0; i < GS / 2; ++i) img[i] = 0u;

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
                    float xn[CP];
                    unit_dir<CP>(q, c, xn);

                    // ---- B operand rows of this wave's tokens: 32 slots = 64 bytes per token, through LDS
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
                        *reinterpret_cast<u32x4 *>(xch + lane * 64 + pos * 16) = u32x4{dw[4 * pc], dw[4 * pc + 1], dw[4 * pc + 2], dw[4 * pc + 3]};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
                    const char *cbl_l = cbl;   // shared codebook: same image every level
                    const int ngrp = p.KTP / 4;
                    u32x4 an[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(cbl_l + i * NKC * 1024 + lane * 16);
                    for (int grp = 0; grp < ngrp; ++grp) {
                        float mx[4];
                        u32x4 a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i] = an[i];
                        if (grp + 1 < ngrp) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4 *>(cbl_l + (((grp + 1) * 4 + i) * NKC) * 1024 + lane * 16);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            // four independent MFMAs back to back (one per token tile), then their maxima: the first
                            // result is ready by the time the fourth has issued
                            f32x4 s[4];
#pragma unroll
                            for (int T = 0; T < 4; ++T) s[T] = mfma16(a[i], B[T], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                            for (int T = 0; T < 4; ++T) {
                                // (rows past K repeat row 0 in the image -- gram_pack_cb_kernel -- so the last block needs no masking: a repeat
                                //  never beats the row it repeats, and ties go to the lower row block)
                                // IEEE-754-2019 maximum (v_maximum3_f32): a NaN score poisons the maximum, which then never
                                // beats `best`; the level takes index 0 through okq
                                const float m012 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s[T][0], s[T][1]), s[T][2]);
                                if (i == 0) mx[T] = __builtin_elementwise_maximum(m012, s[T][3]);
                                else mx[T] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m012, s[T][3]), mx[T]);
                            }
                        }
#pragma unroll
                        for (int T = 0; T < 4; ++T) {
                            const bool up = mx[T] > best[T];
                            best[T] = up ? mx[T] : best[T];
                            bgrp[T] = up ? grp : bgrp[T];
                        }
                    }
                    // ---- the four lanes (g) of a token hand their (maximum, group) to the token's own lane
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        *reinterpret_cast<u32x2 *>(xch + ((16 * T + n) * 4 + g) * 8) = u32x2{u32_of(best[T]), (unsigned)bgrp[T]};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const u32x4 h0 = *reinterpret_cast<const u32x4 *>(xch + lane * 32);
                    const u32x4 h1 = *reinterpret_cast<const u32x4 *>(xch + lane * 32 + 16);
                    float bv = f32_of(h0[0]);
                    int bgr = (int)h0[1], bg = 0;
                    bool amb = false;
                    {
                        const float v1 = f32_of(h0[2]), v2 = f32_of(h1[0]), v3 = f32_of(h1[2]);
                        const int g1 = (int)h0[3], g2 = (int)h1[1], g3 = (int)h1[3];
                        const float vv[3] = {v1, v2, v3};
                        const int gg[3] = {g1, g2, g3};
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
                    const float *cbf = cbf_lds;   // shared codebook, fp32 rows in LDS
                    int brow = 64 * bgr + 4 * bg;
                    float bsc = -INFINITY;
                    auto rescore4 = [&](int r0) {
                        float f[4 * CP];
                        if (CP == 4 || c == CP) {
                            // rows are c floats apart: a row quad is c aligned float4s
                            const f32x4 *qp = reinterpret_cast<const f32x4 *>(cbf + (size_t)r0 * c);
#pragma unroll
                            for (int w = 0; w < CP; ++w) {
                                const f32x4 v = (c == CP) ? qp[w] : f32x4{0.f, 0.f, 0.f, 0.f};
                                f[4 * w] = v[0]; f[4 * w + 1] = v[1]; f[4 * w + 2] = v[2]; f[4 * w + 3] = v[3];
                            }
                        }
                        if (c != CP) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int j = 0; j < CP; ++j) f[r * CP + j] = j < c ? cbf[(size_t)(r0 + r) * c + j] : 0.0f;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s = 0.0f;
#pragma unroll
                            for (int j = 0; j < CP; ++j) s = __builtin_fmaf(xn[j], f[r * CP + j], s);
                            const bool up = (s > bsc) && (r0 + r < p.K);
                            bsc = up ? s : bsc;
                            brow = up ? r0 + r : brow;
                        }
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i) rescore4(16 * (4 * bgr + i) + 4 * bg);
                    // equal maxima in one 64-row block on several lanes (duplicate rows): rescan the block in row order; rare, so
                    // the wave branches around it (a per-lane `if` alone is if-converted into every level of every token)
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
                    // ---- straight-through value (rqae/model.py:218-220) and its bf16 pieces for the reconstruction
                    float cvp[CP];
#pragma unroll
                    for (int j = 0; j < CP; ++j) {
                        const float cv = j < c ? cbf[(size_t)brow * c + j] : 0.0f;
                        const float df = cv - q[j];
                        cvp[j] = q[j] + df;
                        if (MULTI) cvh[li][j] = cvp[j];
                        unsigned h, m, lo;
                        split3(cvp[j], h, m, lo);
                        const int s0 = (li * CP + j) * 3;
                        img[s0] = pack_hi16(h, m);
                        img[s0 + 1] = pack_hi16(h, lo);
                        img[s0 + 2] = img[s0];
                    }
                    // ---- later levels of this group see the level just applied
#pragma unroll
                    for (int li2 = li + 1; li2 < GL; ++li2) {
                        const int l2 = lg * GL + li2;
                        if (l2 < p.levels) {
                            const float *gb = MULTI ? p.gneg + ((size_t)l * p.L + l2) * GB : sgram + (li * GL + li2) * GB;
#pragma unroll
                            for (int j = 0; j < CP; ++j)
#pragma unroll
                                for (int i = 0; i < CP; ++i) Q[li2][j] = __builtin_fmaf(gb[j * CP + i], cvp[i], Q[li2][j]);
                        }
                    }
                    // ---- reconstruction operand rows: every k-step this level completed (all of them after the last level)
                    {
                        const int ks_lo = (3 * CP * li) / 16;
                        const int ks_hi = (l == p.levels - 1 || li == GL - 1) ? (3 * CP * (li + 1) + 15) / 16 : (3 * CP * (li + 1)) / 16;
#pragma unroll
                        for (int ks = ks_lo; ks < GS / 32; ++ks)
                            if (ks < ks_hi) {
                                unsigned *dst = p.cvimg + (((size_t)(lg * (GS / 32) + ks) * p.tok_pad + t) * 16);
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    *reinterpret_cast<u32x4 *>(dst + 4 * v) = u32x4{img[16 * ks + 4 * v], img[16 * ks + 4 * v + 1], img[16 * ks + 4 * v + 2], img[16 * ks + 4 * v + 3]};
                            }
                    }
                } else {
                    codes[li] = 0;
                    if (MULTI) {
#pragma unroll
                        for (int j = 0; j < CP; ++j) cvh[li][j] = 0.0f;
                    }
                }
            });
            // ---- codes and reconstruction operand rows of this group
            if (t < p.n_tokens) {
#pragma unroll
                for (int li = 0; li < GL; ++li)
                    if (lg * GL + li < p.levels) p.idx[t * p.levels + lg * GL + li] = (long long)codes[li];
            }
            // ---- projections of the later level groups see this group's levels
            if (MULTI)
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
}

// ------------------------------------------------------------------------------------------------ kernel C: reconstruct
struct RecParams {
    const uint16_t *wout_img;    // [NET][NKSC][512]
    const float *bias;           // bias_prefix[levels-1][d]
    const unsigned *cvimg;       // [nks][tok_pad][16]
    float *out;                  // [n_tokens][d]
    long long n_tokens, tok_pad, npairs;
    int d, NKSC, nks, NET, nslices, LQ;
};

// element tiles per slice: as many (an even number) as fit 144 KiB of LDS beside NKS k-steps each
__host__ __device__ constexpr int rec_ets(int nks_t) { return ((144 / nks_t) / 2) * 2; }

// Weights stationary: a workgroup keeps the W_out fragments of ETS element tiles (one slice of d) in LDS and
// streams token groups (TT x 16 tokens per wave) through them: B = a token tile's cv pieces (registers, loaded
// once per group, the next group's one iteration ahead), A = LDS fragment, D = 16 elements x 16 tokens, accumulator
// initialised with the bias sum.  No barrier after the prologue.  The tile loop is unrolled at compile time and no
// load in the steady state is conditional: with a loop (or a branch) between a load and its use hipcc drains
// vmcnt(0), and since loads and stores retire in order that waits for every store of the previous group (measured:
// 0.69 ms per 262 144 tokens against 0.42 ms for the bare stores).
template <int NKS, int TT, int NW>
__global__ __launch_bounds__(NW * 64) void gram_reconstruct_kernel(RecParams p)
{
    constexpr int ETS = rec_ets(NKS);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, n = lane & 15;   // wave: provably uniform -> scalar branches
    // blocks b and b + 8 share an XCD (and its L2): an XCD owns every 8th token group and runs all slices on them
    const int b = blockIdx.x, xcd = b & 7, w = b >> 3;
    const int slice = w % p.nslices, lq = w / p.nslices;
    const int et0 = slice * ETS;
    const int net = p.NET - et0 < ETS ? p.NET - et0 : ETS;
    for (int q = wave; q < ETS * NKS; q += NW) {
        const int et = q / NKS, ks = q - et * NKS;
        if (ks < p.nks && et < net)
            glds16(p.wout_img + (((size_t)(et0 + et) * p.NKSC + ks) * 512) + lane * 8, smem + (et * NKS + ks) * 1024);
        else   // k-steps past the depth in use (their cv pieces are re-reads of the last real k-step) and tiles past d: zero
            *reinterpret_cast<u32x4 *>(smem + (et * NKS + ks) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // the slice's bias sums go to LDS too: a global load inside the tile loop would wait for every store issued before it
    float *sbias = reinterpret_cast<float *>(smem + (size_t)ETS * NKS * 1024);
    for (int i = tid; i < ETS * 16; i += NW * 64) sbias[i] = i < net * 16 ? p.bias[et0 * 16 + i] : 0.0f;
    __syncthreads();
    const long long stride = (long long)p.LQ * NW;
    const bool lo8 = n < 8;
    // The cv pieces of a token group (B operand, TT x NKS x 16 bytes per lane) are loaded by inline asm one group
    // ahead and waited for by hand.  hipcc cannot do this: loads and stores retire in order, so a compiler-tracked
    // load issued after a group's stores is waited for with a vmcnt that also waits for those stores, and at the loop
    // header it merges the two histories to the conservative count (ISA: vmcnt(16) = every store of the previous
    // group done) -- measured 0.65 ms per 262 144 tokens against 0.42 ms for the bare stores.  Here the loads are
    // invisible to hipcc's bookkeeping, the wait after a group's stores is vmcnt(#stores) (the stores of the steady
    // state are unconditional, so their number is static), and two register sets alternate so that no copy touches a
    // destination before its wait.
    const unsigned lane_off = (unsigned)(n * 64 + g * 16);   // bytes inside a 16-token x 64-byte fragment row block
    auto loadB = [&](u32x4 (&Bd)[TT][NKS], long long pair) {
        const long long tb = pair * (16 * TT);
#pragma unroll
        for (int T = 0; T < TT; ++T)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int kc = ks < p.nks ? ks : p.nks - 1;
                // wave-uniform base in SGPRs + one per-lane 32-bit offset: no address VGPRs per load
                const unsigned *base = p.cvimg + ((size_t)kc * p.tok_pad + tb + 16 * T) * 16;
                // s_nop 4: the base may come out of a v_readlane (SGPR spill); VALU-written SGPR -> VMEM address needs 5
                // wait states and hipcc pads nothing inside an asm statement (seen: memory access fault without it)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(Bd[T][ks]) : "v"(lane_off), "s"(base) : "memory");
            }
    };
    auto waitB = [&](u32x4 (&Bd)[TT][NKS], auto cnt_c) {
        constexpr int CNT = decltype(cnt_c)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
        // name every destination so that no use is scheduled above the wait
#pragma unroll
        for (int T = 0; T < TT; ++T)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(Bd[T][ks])::"memory");
    };
    constexpr int NSTORES = (ETS / 2) * TT * 2;             // per group in the steady state
    constexpr int WAITN = NSTORES < 63 ? NSTORES : 63;      // vmcnt is 6 bits; anything older than the newest 63 is done

    // one token group: two element tiles at a time; lanes n < 8 and n >= 8 of a 16-lane row trade one accumulator (DPP
    // row_ror:8), so that one store instruction writes 128 contiguous bytes (a whole line) for each of 8 tokens
    auto group = [&](u32x4 (&B)[TT][NKS], u32x4 (&Bnext)[TT][NKS], long long pair) {
        const long long tb = pair * (16 * TT);
        float *ob[TT][2];   // row bases once per group: tile pair `ip` is then a compile-time offset of 128 ip bytes
#pragma unroll
        for (int T = 0; T < TT; ++T) {
            ob[T][0] = p.out + (size_t)(tb + 16 * T + (n & 7)) * p.d + et0 * 16 + 4 * g + (lo8 ? 0 : 16);
            ob[T][1] = ob[T][0] + (size_t)8 * p.d;
        }
        auto tiles = [&](auto guard_c) {
            constexpr bool GUARD = decltype(guard_c)::value;
            static_for<0, ETS / 2>([&](auto ip) {
                constexpr int et = 2 * decltype(ip)::value;
                const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(sbias + et * 16 + 4 * g);
                const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(sbias + et * 16 + 16 + 4 * g);
                f32x4 acc0[TT], acc1[TT];
#pragma unroll
                for (int T = 0; T < TT; ++T) { acc0[T] = bias0; acc1[T] = bias1; }
                const char *wb = smem + (et * NKS) * 1024 + lane * 16;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const u32x4 a0 = *reinterpret_cast<const u32x4 *>(wb + ks * 1024);
                    const u32x4 a1 = *reinterpret_cast<const u32x4 *>(wb + (NKS + ks) * 1024);
#pragma unroll
                    for (int T = 0; T < TT; ++T) {
                        acc0[T] = mfma16(a0, B[T][ks], acc0[T]);
                        acc1[T] = mfma16(a1, B[T][ks], acc1[T]);
                    }
                    // keep hipcc from hoisting all 2 NKS fragment reads of the tile pair (64 VGPRs at NKS = 8)
                    if (ks % 2 == 1) asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int T = 0; T < TT; ++T) {
                    f32x4 s0, s1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v0 = acc0[T][r], v1 = acc1[T][r];
                        const float snd = lo8 ? v1 : v0;
                        const float rcv = f32_of((unsigned)__builtin_amdgcn_update_dpp(0, (int)u32_of(snd), 0x128, 0xf, 0xf, false));
                        s0[r] = lo8 ? v0 : rcv;    // n < 8: (token n, tile et)      n >= 8: (token n - 8, tile et + 1)
                        s1[r] = lo8 ? rcv : v1;    // n < 8: (token n + 8, tile et)  n >= 8: (token n, tile et + 1)
                    }
                    if (GUARD) {
                        const int es = (et0 + et) * 16 + 4 * g + (lo8 ? 0 : 16);
                        const long long tok0 = tb + 16 * T + (n & 7);
                        if (tok0 < p.n_tokens && es < p.d) *reinterpret_cast<f32x4 *>(ob[T][0] + et * 16) = s0;
                        if (tok0 + 8 < p.n_tokens && es < p.d) *reinterpret_cast<f32x4 *>(ob[T][1] + et * 16) = s1;
                    } else {
                        *reinterpret_cast<f32x4 *>(ob[T][0] + et * 16) = s0;
                        *reinterpret_cast<f32x4 *>(ob[T][1] + et * 16) = s1;
                    }
                }
            });
        };
        if (tb + 16 * TT <= p.n_tokens && net == ETS) {
            tiles(std::false_type{});
            waitB(Bnext, std::integral_constant<int, WAITN>{});
        } else {   // ragged end: conditional stores, so their number is unknown
            tiles(std::true_type{});
            waitB(Bnext, std::integral_constant<int, 0>{});
        }
    };
    u32x4 BX[TT][NKS], BY[TT][NKS];
    long long pi = (long long)lq * NW + wave;
    if (xcd + 8 * pi >= p.npairs) return;
    if constexpr (TT * NKS >= 16 || NW == 16) {
        // two register sets of cv pieces no longer fit without spills (and hipcc may spill an asm-loaded register before its wait):
        // one set, loaded and waited for per group
        for (; xcd + 8 * pi < p.npairs; pi += stride) {
            loadB(BX, xcd + 8 * pi);
            waitB(BX, std::integral_constant<int, 0>{});
            group(BX, BX, xcd + 8 * pi);
        }
        return;
    }
    loadB(BX, xcd + 8 * pi);
    waitB(BX, std::integral_constant<int, 0>{});
    for (;;) {
        {   // group in X, next group (past the end: this one again) into Y
            const long long pair = xcd + 8 * pi, pn = xcd + 8 * (pi + stride);
            loadB(BY, pn < p.npairs ? pn : pair);
            group(BX, BY, pair);
        }
        pi += stride;
        if (xcd + 8 * pi >= p.npairs) break;
        {
            const long long pair = xcd + 8 * pi, pn = xcd + 8 * (pi + stride);
            loadB(BX, pn < p.npairs ? pn : pair);
            group(BY, BX, pair);
        }
        pi += stride;
        if (xcd + 8 * pi >= p.npairs) break;
    }
}

#include "gram_fused.hpp"
#include "gram_stream.hpp"
#include "gram_wide.hpp"

constexpr int B_NW = 8;

// Kernel selection of one call.  The defaults are compiled in; a caller overrides them per call through
// rqae_hip_gram_options (include/rqae_hip.h) -- there is no process-wide setting (re-entrant, no global mutable
// state).  Tools builds (-DRQAE_GRAM_TOOLS: tools/ only) can also override the defaults from RQAE_GRAM_* environment
// variables, read once.
struct Tuning {
    int c_tt;            // reconstruct kernel: 16-token tiles per wave (2: 8 waves per workgroup, 1: 16 waves)
    long long chunk;     // tokens per pass
    int proj;            // projection kernel: 4 = gram_proj3_kernel (fp16 two-piece W_in, 32x32x16, every output tile in one launch: the default up to
                         // 256 outputs), 3 = phase A of the fused kernel (2-byte x), 2 = pipelined register form, 1 = first form
    int recon;           // reconstruction kernel: 3 = gram_recon3_kernel (fp16 two-piece operands, 32x32x16, 1-KiB store rows: the default wherever
                         // the level chain hands over fp32 planes), 2 = phase C of the fused kernel (one level group), 1 = weights-stationary form
    long long pipe_chunk; // tokens per pass of the pipelined schedule (0: never pipeline)
    int cap_ab;          // pipelined schedule: CUs for projection + level chain; the reconstruction of the pass before takes the rest
    int fused;           // 1: one persistent kernel where the shape allows it, 0: always the three-kernel form
    int stagger;         // fused kernel: 1 = half first tile on odd workgroups
    int lattice;         // 1: FSQ lattice codebooks take the lattice form of the level chain (the scan kernels stay behind it as the fallback)
    int lev2;            // level chain of one level group: 2 = lean form (fp32 values to the reconstruction) with 16 waves per workgroup
                         // (default), 1 = lean form with 8 waves, 0 = first form
};
constexpr size_t LEV_STAMP_BYTES = 1u << 20;   // tools builds (-DRQAE_LEV_TIMELINE): 16 stamps per wave of the level chain, behind the fused kernel's
constexpr long long GRAM_CHUNK = 262144;   // tokens per pass: sizes the workspace (rqae_hip_gram_workspace_bytes), so not a per-call option

Tuning default_tuning()
{
    Tuning v;
    v.chunk = GRAM_CHUNK;
    v.c_tt = 2;
    v.proj = 4;
    v.recon = 3;
    v.fused = 0;
    v.pipe_chunk = 0;       // off: measured slower (profiles/r02_pipeline_trace.txt): the kernels are bound per CU, not by HBM
    v.cap_ab = 160;
    v.lev2 = 2;
    v.lattice = 1;          // FSQ lattice codebooks: the argmax without the scan (gram_wide.hpp, LAT), the scan kernels behind it as the fallback
    v.stagger = 0;
#ifdef RQAE_GRAM_TOOLS
    static const Tuning env = [v]() mutable {
        if (const char *e = getenv("RQAE_GRAM_PROJ")) { const int a = atoi(e); v.proj = a >= 1 && a <= 4 ? a : 4; }
        if (const char *e = getenv("RQAE_GRAM_RECON")) { const int a = atoi(e); v.recon = a >= 1 && a <= 3 ? a : 3; }
        if (const char *e = getenv("RQAE_GRAM_FUSED")) v.fused = atoi(e) != 0;
        if (const char *e = getenv("RQAE_GRAM_PIPE")) { const long long c = atoll(e); v.pipe_chunk = c <= 0 ? 0 : ((c + 255) & ~255LL); }
        if (const char *e = getenv("RQAE_GRAM_CAP_AB")) { const int c = atoi(e); if (c >= 8) v.cap_ab = c; }
        if (const char *e = getenv("RQAE_GRAM_LEV2")) { const int a = atoi(e); v.lev2 = a < 0 || a > 2 ? 1 : a; }
        if (const char *e = getenv("RQAE_GRAM_LATTICE")) v.lattice = atoi(e) != 0;
        if (const char *e = getenv("RQAE_GRAM_STAGGER")) v.stagger = atoi(e) != 0;
        if (const char *e = getenv("RQAE_GRAM_CTT")) v.c_tt = atoi(e) == 1 ? 1 : 2;
        return v;
    }();
    return env;
#else
    return v;
#endif
}

// defaults overridden by the caller's options (a negative / zero-sized field keeps the default)
int make_tuning(const rqae_hip_gram_options *o, Tuning &t)
{
    t = default_tuning();
    if (!o) return 0;
    if (o->size != (int32_t)sizeof(rqae_hip_gram_options)) return gfail("gram options: size %d, this library expects %d", (int)o->size, (int)sizeof(rqae_hip_gram_options));
    if (o->fused >= 0) t.fused = o->fused != 0;
    if (o->pipe_chunk >= 0) t.pipe_chunk = o->pipe_chunk == 0 ? 0 : ((o->pipe_chunk + 255) & ~255LL);
    if (o->proj > 4 || o->recon > 3 || o->lean_levels > 5) return gfail("gram options: proj %d / recon %d / lean_levels %d out of range", o->proj, o->recon, o->lean_levels);
    if (o->proj >= 1) t.proj = o->proj;
    if (o->recon >= 1) t.recon = o->recon;
    // 0..3 name a scan kernel; 4 = lattice form with the scan kernel behind it; 5 = lattice form alone (the caller has asked
    // rqae_hip_gram_lattice about the block)
    if (o->lean_levels >= 0) { t.lev2 = o->lean_levels >= 4 ? 2 : o->lean_levels; t.lattice = o->lean_levels >= 4 ? o->lean_levels - 3 : 0; }
    return 0;
}

int device_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    static int cus[64];
    if (dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        cus[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    return cus[dev];
}

struct GView {
    Geom g;
    const uint16_t *win_img, *cb_img, *wout_img, *wout3_img, *win3_img, *g3_img, *sign_img;
    const float *kconst, *gneg, *cb_f32, *bias_prefix;
    const unsigned *wout_maxbits, *win_maxbits, *g3_maxbits, *sign_maxbits;
    const int *sign_exact;
};

// the header lives on the device; the caller passes (d, c, K, L) again and the geometry is recomputed
GView gview(const void *packed, int d, int c, int K, int L, int shared_cb)
{
    GView v;
    v.g = make_geom(d, c, K, L, shared_cb);
    const GLayout lo = make_glayout(v.g);
    const char *b = static_cast<const char *>(packed);
    v.win_img = reinterpret_cast<const uint16_t *>(b + lo.win_img);
    v.kconst = reinterpret_cast<const float *>(b + lo.kconst);
    v.gneg = reinterpret_cast<const float *>(b + lo.gneg);
    v.cb_img = reinterpret_cast<const uint16_t *>(b + lo.cb_img);
    v.cb_f32 = reinterpret_cast<const float *>(b + lo.cb_f32);
    v.wout_img = reinterpret_cast<const uint16_t *>(b + lo.wout_img);
    v.bias_prefix = reinterpret_cast<const float *>(b + lo.bias_prefix);
    v.wout3_img = reinterpret_cast<const uint16_t *>(b + lo.wout3_img);
    v.wout_maxbits = reinterpret_cast<const unsigned *>(b + offsetof(GHeader, wout_maxbits));
    v.win3_img = reinterpret_cast<const uint16_t *>(b + lo.win3_img);
    v.win_maxbits = reinterpret_cast<const unsigned *>(b + offsetof(GHeader, win_maxbits));
    v.g3_img = reinterpret_cast<const uint16_t *>(b + lo.g3_img);
    v.g3_maxbits = reinterpret_cast<const unsigned *>(b + offsetof(GHeader, g3_maxbits));
    v.sign_img = reinterpret_cast<const uint16_t *>(b + lo.sign_img);
    v.sign_maxbits = reinterpret_cast<const unsigned *>(b + offsetof(GHeader, sign_maxbits));
    v.sign_exact = reinterpret_cast<const int *>(b + offsetof(GHeader, sign_exact));
    return v;
}

// tokens per pass (GRAM_CHUNK = 262144; measured: 65536 1.39 ms, 131072 1.15 ms, 262144 1.11 ms per 262144
// tokens -- fewer launches beat cache residency of the intermediates)

inline long long pad64(long long n) { return (n + 63) & ~63LL; }

// bytes per token of the buffer between the level chain and the reconstruction: bf16-piece operand rows (64 per k-step) or the
// fp32 straight-through values of one level group (gram_levels2_kernel), whichever is larger
size_t cv_row_bytes(const Geom &g, int levels)
{
    const size_t rows = (size_t)nks_recon(g, levels) * 64, f32 = (size_t)8 * g.CP * 4;
    const size_t planes = (size_t)levels * (g.CP <= 5 ? g.CP : 16) * 4;      // fp32 planes of every level (gram_recon3_kernel's input)
    const size_t a = rows > f32 ? rows : f32;
    return a > planes ? a : planes;
}

constexpr long long GRAM_CHUNK_DEEP = 131072;  // tokens per pass of a deep call (32.5 KB of workspace per token at L = 1024: 4.3 GB a set)
inline size_t deep_set_bytes(const Geom &g, long long tp)
{
    return al256((size_t)rqae_p3::padded_groups(g.NOG) * tp * 16 * sizeof(float)) + al256((size_t)rqae_d3::BLOCK_LEVELS * tp * 4 * sizeof(float)) +
           al256(rqae_d3::pieces_bytes(tp, g.L));
}

size_t ws_bytes(const Geom &g, long long n_tokens, int levels)
{
    if (levels > GRAM_DEEP_L) {
        // deep calls (gram_deep3.hpp): the projections of every level, fp32 planes of one block of levels, the operand fragments of
        // every straight-through value, dump rows
        const long long ch = n_tokens < GRAM_CHUNK_DEEP ? n_tokens : GRAM_CHUNK_DEEP;
        const long long tp = pad64(ch);
        // (two sets when the call runs in several passes: consecutive passes go down two streams, one's level chain beside the other's products)
        return (n_tokens > GRAM_CHUNK_DEEP ? 2 : 1) * deep_set_bytes(g, tp) + 32768;
    }
    const long long ch = n_tokens < GRAM_CHUNK ? n_tokens : GRAM_CHUNK;
    const long long tp = pad64(ch);
    // projections, TWO operand-row buffers (the pipelined schedule reconstructs pass i while pass i+1 fills the other one), dump
    // rows, timeline stamps (tools build)
    return al256((size_t)rqae_p3::padded_groups(g.NOG) * tp * 16 * sizeof(float)) + 2 * al256(cv_row_bytes(g, levels) * tp) + 32768 + (2048 + 8 * 36 * 4) * 8 + LEV_STAMP_BYTES;
}

bool levels_resident_ok(const Geom &g) { return g.shared_cb && (size_t)g.KTP * g.NKC * 1024 + ((size_t)g.K + 64) * g.c * 4 + 16 + B_NW * g.NKC * 4096 + (size_t)g.GL * g.GL * g.GB * 4 <= 158 * 1024; }

// a codebook too large for LDS streams through it (gram_stream.hpp): one shared codebook, one k-step per cosine contraction
bool levels_stream_ok(const Geom &g) { return g.shared_cb && g.NKC == 1 && g.CP <= 5; }

// wide codebook vectors (6 <= c <= 13: three k-steps per cosine contraction, four levels per group): gram_wide.hpp
bool levels_wide_ok(const Geom &g) { return g.CP == 13 && g.NKC == 3 && g.d % 128 == 0 && g.NLG <= 4 && (g.shared_cb || g.K % 16 == 0); }

// one codebook PER LEVEL (any quantization_method but the FSQ lattices: rqae/model.py:46-54): nothing stays resident across levels,
// so every level streams its own image -- gram_wide.hpp's kernel for c <= 5 too (whole row tiles: its rescoring reads past K otherwise)
bool levels_free_ok(const Geom &g) { return !g.shared_cb && g.K % 16 == 0 && ((g.CP <= 5 && g.NKC == 1) || levels_wide_ok(g)); }

// the one-kernel form: one level group, whole chunks of 2 k-steps, everything resident fits the CU's 160 KiB of LDS
bool fused_shape_ok(const Geom &g, int levels)
{
    return g.shared_cb && g.CP <= 5 && levels <= 8 && g.d % 64 == 0 && g.NKC == 1 && fused_lds_bytes(g.d, g.c, g.K) <= 160 * 1024;
}

// Second stream for the pipelined schedule (encode_gram_impl): the reconstruction of pass i runs there, beside the projection
// and the level chain of pass i+1 on the caller's stream.  One per device, created on first use; the events order the
// two streams (fork from, and join back into, the caller's stream: the call stays asynchronous and capturable).
struct Side {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, evB[4] = {nullptr, nullptr, nullptr, nullptr}, evC[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = false, tried = false;
    std::mutex mu;
};
Side *side_for_device()
{
    static Side sides[64];
    static std::mutex init_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Side &sd = sides[dev];
    std::lock_guard<std::mutex> lock(init_mu);
    if (!sd.tried) {
        sd.tried = true;
        bool ok = hipStreamCreateWithFlags(&sd.st, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; ++i)
            ok = hipEventCreateWithFlags(&sd.evB[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&sd.evC[i], hipEventDisableTiming) == hipSuccess;
        sd.ok = ok;
    }
    return sd.ok ? &sd : nullptr;
}

int recon_nks_template(int nks)
{
    const int cand[] = {2, 3, 4, 6, 8, 12, 16, 24};
    for (int v : cand)
        if (nks <= v) return v;
    return 0;
}

}  // namespace

#ifndef RQAE_GRAM_NO_ABI   /* tools/quick_isa.py compiles the kernels it names without the launch code below (which instantiates every kernel) */
// ==================================================================================================== C ABI
extern "C" {

int rqae_hip_gram_supported(int d, int c, int K, int L, int codebook_shared)
{
    if (!gram_shape_ok(d, c, K, L)) return 0;
    const Geom g = make_geom(d, c, K, L, codebook_shared ? 1 : 0);
    return ((g.CP <= 5 && levels_resident_ok(g)) || levels_stream_ok(g) || (g.shared_cb && levels_wide_ok(g)) || levels_free_ok(g)) ? 1 : 0;
}

// Can one call with `levels` levels run on this arm?  The codes always can (on a supported shape); the reconstruction needs one of
// the instantiated reconstruction kernels: up to 24 k-steps of 32 slots (6 c slots per level: 32 levels at c = 4, 24 at c = 5), or
// the wide kernel's 16 levels.
int rqae_hip_encode_gram_supported(int d, int c, int K, int L, int codebook_shared, int levels, int want_recon)
{
    if (!rqae_hip_gram_supported(d, c, K, L, codebook_shared) || levels < 1 || levels > L) return 0;
    const Geom g = make_geom(d, c, K, L, codebook_shared ? 1 : 0);
    // more than 64 levels: the blocked path of gram_deep3.hpp (whole blocks of 32 levels of an FSQ lattice codebook, c = 4; the call
    // must vouch for the lattice: lean_levels = 5)
    if (levels > GRAM_DEEP_L) return deep_block_ok(g) && levels % rqae_d3::BLOCK_LEVELS == 0 && rqae_p3::supported(d, g.NOG, RQAE_X_BF16, false) ? 1 : 0;
    if (!want_recon) return 1;
    if (rqae_r3::supported(d, c, levels) && (g.CP == 13 || levels <= 8)) return 1;      // gram_recon3_kernel behind the planes of the wide / lean level chains
    if (g.CP == 13) return levels <= 16 ? 1 : 0;
    return recon_nks_template(nks_recon(g, levels)) != 0 ? 1 : 0;
}

size_t rqae_hip_gram_packed_bytes(int d, int c, int K, int L, int codebook_shared)
{
    if (!rqae_hip_gram_supported(d, c, K, L, codebook_shared)) {
        gfail("gram arm: unsupported shape d=%d c=%d K=%d L=%d shared=%d", d, c, K, L, codebook_shared);
        return 0;
    }
    return make_glayout(make_geom(d, c, K, L, codebook_shared ? 1 : 0)).total;
}

int rqae_hip_gram_pack(void *packed, int d, int c, int K, int L, const float *w_in, const float *b_in,
                       const float *codebook, const float *w_out, const float *b_out, int codebook_shared, const float *in_scale,
                       void *stream)
{
    if (!rqae_hip_gram_supported(d, c, K, L, codebook_shared)) return gfail("gram_pack: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !w_in || !b_in || !codebook || !w_out || !b_out) return gfail("gram_pack: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geom g = make_geom(d, c, K, L, codebook_shared ? 1 : 0);
    const GLayout lo = make_glayout(g);
    char *b = static_cast<char *>(packed);
    GHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = GRAM_MAGIC; h.d = d; h.c = c; h.K = K; h.L = L; h.shared_cb = g.shared_cb;
    // a lattice candidate: K = n^c for an n the threshold nibbles of lattice_best hold (c <= 13 < 16 in any case)
    int nlat = 0;
    if (g.shared_cb)
        for (int n = 2; n <= 16 && !nlat; ++n) {
            long long pw = 1;
            for (int j = 0; j < c && pw <= K; ++j) pw *= n;
            if (pw == K) nlat = n;
        }
    h.lattice_n = nlat;
    G_TRY(hipMemcpyAsync(b, &h, sizeof(h), hipMemcpyHostToDevice, s));
    if (nlat) {
        hipLaunchKernelGGL(gram_pack_lattice_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, codebook, K, c, nlat, reinterpret_cast<GHeader *>(b));
        G_TRY(hipGetLastError());
        hipLaunchKernelGGL(gram_pack_lattice_verdict_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<GHeader *>(b));
        G_TRY(hipGetLastError());
    }
    G_TRY(hipMemsetAsync(b + lo.kconst, 0, sizeof(float) * (size_t)g.NOG * 16, s));
    hipLaunchKernelGGL(gram_pack_win_kernel, dim3(2048), dim3(256), 0, s, g, w_in, in_scale, reinterpret_cast<uint16_t *>(b + lo.win_img));
    G_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_gram_kernel, dim3((unsigned)L, (unsigned)L + 1), dim3(256), 0, s, g, w_in, b_in, w_out, b_out,
                       reinterpret_cast<float *>(b + lo.gneg), reinterpret_cast<float *>(b + lo.kconst));
    G_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_cb_kernel, dim3(1024), dim3(256), 0, s, g, codebook, reinterpret_cast<uint16_t *>(b + lo.cb_img),
                       reinterpret_cast<float *>(b + lo.cb_f32));
    G_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_wout_kernel, dim3(2048), dim3(256), 0, s, g, w_out, reinterpret_cast<uint16_t *>(b + lo.wout_img));
    G_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_bias_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, s, g, b_out,
                       reinterpret_cast<float *>(b + lo.bias_prefix));
    G_TRY(hipGetLastError());
    // the fp16 two-piece image of W_out (gram_recon3.hpp): its power-of-two scale follows from max |W_out|, found on the device
    if (int rc = rqae_r3::pack(b + lo.wout3_img, reinterpret_cast<unsigned *>(b + offsetof(GHeader, wout_maxbits)), d, c, L, w_out, s)) return rc;
    // ... of the Gram blocks for the cross-block corrections of deep calls (gram_deep3.hpp)
    if (deep_block_ok(g))
        if (int rc = rqae_d3::pack_g3(b + lo.g3_img, reinterpret_cast<unsigned *>(b + offsetof(GHeader, g3_maxbits)), reinterpret_cast<const float *>(b + lo.gneg), L, g.GB, s)) return rc;
    // ... for the level chain of a two-point lattice (gram_sign3.hpp)
    if (rqae_s3::supported(c, L, g.shared_cb))
        if (int rc = rqae_s3::pack(b + lo.sign_img, reinterpret_cast<unsigned *>(b + offsetof(GHeader, sign_maxbits)), reinterpret_cast<int *>(b + offsetof(GHeader, sign_exact)),
                                   reinterpret_cast<const float *>(b + lo.gneg), codebook, c, L, K, g.GB, s)) return rc;
    // ... and of W_in (gram_proj3.hpp), in_scale folded in like the bf16 image's
    if (int rc = rqae_p3::pack(b + lo.win3_img, reinterpret_cast<unsigned *>(b + offsetof(GHeader, win_maxbits)), d, c, L, g.CP, g.GL, g.GP, g.NOG, w_in, in_scale, s)) return rc;
    return 0;
}

// RQAE.normalize_codebooks through a `levels`-level forward call (rqae/model.py:126-131,196): codebook [L][K][c] is renormalised
// `levels` times in place, used_out [L][K][c] (or NULL) receives for every level l < levels its rows after l + 1 renormalisations
// (what quantize sees at that level); torch-CPU's arithmetic order: the reference's bits.
int rqae_hip_normalize_codebooks(float *codebook, int L, int K, int c, int levels, float *used_out, void *stream)
{
    if (!codebook || L < 1 || K < 1 || c < 1 || c > 16 || levels < 0 || levels > L) return gfail("normalize_codebooks: bad arguments (L=%d K=%d c=%d levels=%d)", L, K, c, levels);
    if (levels == 0) return 0;
    const long long rows = (long long)L * K;
    hipLaunchKernelGGL(normalize_codebooks_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), codebook, L, K, c, levels, used_out);
    G_TRY(hipGetLastError());
    return 0;
}

// The codebook part of a block packed with codebook_shared = 0 built again from codebook [L][K][c] (the per-level snapshots of one
// call): the bf16-piece images and the fp32 rows; everything else in the block (the linear maps' images, Gram blocks) stays.
int rqae_hip_gram_repack_codebooks(void *packed, int d, int c, int K, int L, const float *codebook, void *stream)
{
    if (!packed || !codebook) return gfail("gram_repack_codebooks: null pointer");
    if (!rqae_hip_gram_supported(d, c, K, L, 0)) return gfail("gram_repack_codebooks: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    const Geom g = make_geom(d, c, K, L, 0);
    const GLayout lo = make_glayout(g);
    char *b = static_cast<char *>(packed);
    hipLaunchKernelGGL(gram_pack_cb_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), g, codebook, reinterpret_cast<uint16_t *>(b + lo.cb_img),
                       reinterpret_cast<float *>(b + lo.cb_f32));
    G_TRY(hipGetLastError());
    return 0;
}

// What rqae_hip_gram_pack found out about the block's codebook: n >= 2 when it is the FSQ lattice {linspace(-1, 1, n)}^c, else 0
// (-1: not a block).  Waits for `stream` (the pack that wrote the header) -- the one call of this arm that does.
int rqae_hip_gram_lattice(const void *packed, void *stream)
{
    if (!packed) { gfail("gram_lattice: null pointer"); return -1; }
    GHeader h;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&h, packed, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        gfail("gram_lattice: reading the header failed");
        return -1;
    }
    if (h.magic != GRAM_MAGIC) { gfail("gram_lattice: not a block of rqae_hip_gram_pack"); return -1; }
    return h.lattice_n >= 2 ? h.lattice_n : 0;
}

size_t rqae_hip_gram_workspace_bytes(int d, int c, int K, int L, int64_t n_tokens, int levels)
{
    if (!gram_shape_ok(d, c, K, L) || n_tokens < 0 || levels < 1 || levels > L) {
        gfail("gram_workspace_bytes: bad arguments");
        return 0;
    }
    return ws_bytes(make_geom(d, c, K, L, 1), n_tokens > 0 ? n_tokens : 1, levels);
}

}  // extern "C"

namespace {
// ev: NULL, or 4 events recorded before the projection, between the kernels and after the reconstruction of the LAST
// chunk (rqae_hip_encode_gram_phases)
int encode_gram_impl(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *x, int x_dtype,
                     int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *workspace, const rqae_hip_gram_options *opts,
                     void *stream, hipEvent_t *ev, float *inv_rms = nullptr, float eps = 0.0f, const float *inv1pw = nullptr, int patch_S = 0,
                     int patch_skip_bos = 0)
{
    Tuning tun;
    if (int rc = make_tuning(opts, tun)) return rc;
    auto tuning = [&]() -> const Tuning & { return tun; };
    if (!rqae_hip_gram_supported(d, c, K, L, codebook_shared)) return gfail("encode_gram: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !x || !indices || !workspace) return gfail("encode_gram: null pointer");
    if (levels < 1 || levels > L) return gfail("encode_gram: levels=%d outside 1..%d", levels, L);
    if (x_dtype < RQAE_X_F32 || x_dtype > RQAE_X_F16) return gfail("encode_gram: bad x_dtype %d", x_dtype);
    if (n_tokens < 0) return gfail("encode_gram: negative n_tokens");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return gfail("encode_gram: x, q_out and workspace must be 16-byte aligned");
    if (n_tokens == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const GView v = gview(packed, d, c, K, L, codebook_shared ? 1 : 0);
    const Geom &g = v.g;
    const int nks = nks_recon(g, levels);
    const int nks_t = recon_nks_template(nks);
    const bool wide = g.CP == 13;
    const size_t xs = x_dtype == RQAE_X_F32 ? 4 : 2;
    const int ncu = device_cus();

    const long long chunk = n_tokens < tuning().chunk ? n_tokens : tuning().chunk;
    const long long tp = pad64(chunk);
    float *P = static_cast<float *>(workspace);
    unsigned *cvimg = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + al256((size_t)rqae_p3::padded_groups(g.NOG) * tp * 16 * sizeof(float)));
    unsigned *cvimg2 = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(cvimg) + al256(cv_row_bytes(g, levels) * tp));
    float *dump = reinterpret_cast<float *>(reinterpret_cast<char *>(cvimg2) + al256(cv_row_bytes(g, levels) * tp));

    const bool p3_go = tuning().proj == 4 && rqae_p3::supported(d, g.NOG, x_dtype, inv_rms != nullptr);
    if (inv_rms && !p3_go && !(tuning().proj >= 3 && g.CP <= 5 && x_dtype != RQAE_X_F32 && d % 64 == 0))
        return gfail("encode_gram_gemma: the folded norm is served for bf16 / fp16 hidden states with d %% 64 == 0 (d=%d, x_dtype=%d)", d, x_dtype);
    if (levels > GRAM_DEEP_L) {
        // ---- deep calls (gram_deep3.hpp): blocks of 32 levels; across blocks the corrections are matrix products, as is the reconstruction
        constexpr int BL = rqae_d3::BLOCK_LEVELS;
        if (!deep_block_ok(g) || levels % BL) return gfail("encode_gram: %d levels need c = 4, L %% 64 == 0 and whole blocks of %d levels (L=%d c=%d)", levels, BL, L, c);
        if (inv1pw) return gfail("encode_gram_gemma_patch: one level group only (levels <= 8)");
        if (tuning().lattice != 2) return gfail("encode_gram: calls of more than %d levels run the lattice form alone: pass lean_levels = 5 for a block rqae_hip_gram_lattice named a lattice", GRAM_DEEP_L);
        if (!p3_go) return gfail("encode_gram: the deep path needs the fp16 projection (proj = 4) for d=%d x_dtype=%d", d, x_dtype);
        const int *latw = reinterpret_cast<const int *>(static_cast<const char *>(packed) + offsetof(GHeader, lattice_n));
        const long long dchunk = n_tokens < GRAM_CHUNK_DEEP ? n_tokens : GRAM_CHUNK_DEEP;
        const long long dtp = pad64(dchunk);
        const int nblocks = levels / BL;
        const int KSall = L * 4 / 16;
        // Several passes: consecutive passes alternate between the caller's stream and a second one (two workspace sets).  A pass is a
        // strict sequence of ~100 launches in which the level chain (one wave per SIMD at this pass size: latency-bound, 26 KB of LDS)
        // alternates with matrix products (64-110 KB of LDS, two waves per SIMD): side by side, one pass's chain runs in the shadow
        // of the other's products on the same CUs.
        Side *sd = (!ev && n_tokens > dchunk) ? side_for_device() : nullptr;
        std::unique_lock<std::mutex> side_lock;
        if (sd) {
            side_lock = std::unique_lock<std::mutex>(sd->mu);
            G_TRY(hipEventRecord(sd->fork, s));
            G_TRY(hipStreamWaitEvent(sd->st, sd->fork, 0));
        }
        const hipStream_t s_main = s;
        int pass_i = 0;
        for (long long t0 = 0; t0 < n_tokens; t0 += dchunk, ++pass_i) {
            const hipStream_t s = (sd && (pass_i & 1)) ? sd->st : s_main;       // (shadows the caller's stream inside the pass)
            char *wset = static_cast<char *>(workspace) + ((sd && (pass_i & 1)) ? deep_set_bytes(g, dtp) : 0);
            float *dP = reinterpret_cast<float *>(wset);
            float *planes = reinterpret_cast<float *>(wset + al256((size_t)rqae_p3::padded_groups(g.NOG) * dtp * 16 * sizeof(float)));
            char *pieces = reinterpret_cast<char *>(planes) + al256((size_t)BL * dtp * 4 * sizeof(float));
            const long long nt = n_tokens - t0 < dchunk ? n_tokens - t0 : dchunk;
            const bool mark = ev && t0 + dchunk >= n_tokens;
            if (mark) G_TRY(hipEventRecord(ev[0], s));
            rqae_p3::Args ap;
            ap.img = v.win3_img; ap.maxbits = v.win_maxbits; ap.x = static_cast<const char *>(x) + (size_t)t0 * d * xs; ap.P = dP;
            ap.inv_rms = inv_rms ? inv_rms + t0 : nullptr; ap.eps = eps; ap.n_tokens = nt; ap.tok_pad = dtp; ap.d = d; ap.nog = g.NOG; ap.x_dtype = x_dtype;
            if (int rc = rqae_p3::launch(ap, ncu, s)) return rc;
            if (mark) G_TRY(hipEventRecord(ev[1], s));
            for (int b = 0; b < nblocks; ++b) {
                // Corrections, two-level (the straight-through values' fragments are 4 bytes per value and token: left-looking over ALL
                // earlier levels reads 262 KB of them per token at L = 1024 -- 34 GB per 131072 tokens, 8 ms of HBM time -- so):
                //   * across SUPER-blocks of 8 blocks (256 levels) right-looking: when a super-block is done, ONE product adds its
                //     1024 values' contribution to every later column (4 KB of fragments per token, re-read per pass from the
                //     Infinity Cache; the columns are read and written once per super-block);
                //   * inside a super-block left-looking as before: the block's 128 columns see the super-block's earlier blocks.
                constexpr int SBK = 8;                                   // blocks per super-block
                const int sb0 = (b / SBK) * SBK;                        // first block of this block's super-block
                if (b > sb0) {
                    rqae_d3::Gemm gm;
                    gm.apieces = pieces; gm.wimg = v.g3_img; gm.maxbits = v.g3_maxbits; gm.n_tokens = nt; gm.tok_pad = dtp; gm.KSA = KSall; gm.KSW = KSall;
                    gm.ks0 = sb0 * (BL * 4 / 16); gm.nks = (b - sb0) * (BL * 4 / 16); gm.nt0 = b * (BL * 4 / 32); gm.ntiles = BL * 4 / 32;
                    gm.P = dP; gm.bias = nullptr; gm.out = nullptr; gm.d = d;
                    if (int rc = rqae_d3::corrections(gm, ncu, s)) return rc;
                }
                // the lattice form of the level chain on the block's 32 levels (four level groups: corrections between them through P)
                LevParams pl;
                pl.P = dP + (size_t)(b * (BL * 4 / 16)) * dtp * 16; pl.kconst = v.kconst + b * BL * 4;
                pl.gneg = v.gneg + ((size_t)(b * BL) * L + b * BL) * g.GB; pl.cb_img = v.cb_img; pl.cb_f32 = v.cb_f32;
                pl.cvimg = reinterpret_cast<unsigned *>(planes); pl.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels + b * BL;
                pl.idx_stride = levels; pl.n_tokens = nt; pl.tok_pad = dtp; pl.ntiles = (nt + 63) / 64;
                pl.levels = BL; pl.L = L; pl.K = K; pl.KTP = g.KTP; pl.c = c; pl.shared_cb = 1; pl.lat_flag = latw;
                long long nb = (pl.ntiles + 3) / 4;
                if (nb > 8LL * ncu) nb = 8LL * ncu;
                const size_t lds = (((size_t)K * c * 4 + 15) & ~(size_t)15) + (size_t)(BL / g.GL) * g.GL * g.GL * g.GB * 4;
                // (the chain of a five-point lattice -- the released shape -- chooses by approximate scores: lattice_fast, gram_wide.hpp)
                auto chain = [&](auto kern) -> int {
                    if (lds > 64 * 1024) G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, s, pl);
                    G_TRY(hipGetLastError());
                    return 0;
                };
                if (int rc = lattice_points(K, c) == 5 ? chain(gram_levels_wide_kernel<4, 4, true, false, true, true, 5>)
                                                       : chain(gram_levels_wide_kernel<4, 4, true, false, true, true>)) return rc;
                if (q_out || b + 1 < nblocks)
                    if (int rc = rqae_d3::convert(planes, dtp, BL, b * BL, pieces, L, s)) return rc;
                if ((b + 1) % SBK == 0 && b + 1 < nblocks) {
                    // the super-block that just ended, onto every later column of the call
                    rqae_d3::Gemm gm;
                    gm.apieces = pieces; gm.wimg = v.g3_img; gm.maxbits = v.g3_maxbits; gm.n_tokens = nt; gm.tok_pad = dtp; gm.KSA = KSall; gm.KSW = KSall;
                    gm.ks0 = sb0 * (BL * 4 / 16); gm.nks = SBK * (BL * 4 / 16); gm.nt0 = (b + 1) * (BL * 4 / 32); gm.ntiles = (nblocks - (b + 1)) * (BL * 4 / 32);
                    gm.P = dP; gm.bias = nullptr; gm.out = nullptr; gm.d = d;
                    if (int rc = rqae_d3::corrections(gm, ncu, s)) return rc;
                }
            }
            if (mark) G_TRY(hipEventRecord(ev[2], s));
            if (q_out) {
                rqae_d3::Gemm gm;
                gm.apieces = pieces; gm.wimg = v.wout3_img; gm.maxbits = v.wout_maxbits; gm.n_tokens = nt; gm.tok_pad = dtp; gm.KSA = KSall; gm.KSW = g.KS3;
                gm.ks0 = 0; gm.nks = levels * 4 / 16; gm.nt0 = 0; gm.ntiles = d / 32; gm.P = nullptr; gm.bias = v.bias_prefix + (size_t)(levels - 1) * d;
                gm.out = q_out + (size_t)t0 * d; gm.d = d;
                if (int rc = rqae_d3::reconstruct(gm, ncu, s)) return rc;
            }
            if (mark) G_TRY(hipEventRecord(ev[3], s));
        }
        if (sd) {      // join: the caller's stream waits for the second one
            G_TRY(hipEventRecord(sd->evC[0], sd->st));
            G_TRY(hipStreamWaitEvent(s_main, sd->evC[0], 0));
        }
        return 0;
    }
    if (!inv_rms && tuning().fused && x_dtype != RQAE_X_F32 && fused_shape_ok(g, levels)) {
        FusedParams pf;
        pf.x = x; pf.win_img = v.win_img; pf.kconst = v.kconst; pf.gneg = v.gneg; pf.cb_img = v.cb_img; pf.cb_f32 = v.cb_f32;
        pf.wout_img = v.wout_img; pf.bias = v.bias_prefix + (size_t)(levels - 1) * d; pf.out = q_out;
        pf.idx = reinterpret_cast<long long *>(indices); pf.dump = dump; pf.n_tokens = n_tokens;
        pf.stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dump) + 32768);
        pf.d = d; pf.K = K; pf.KTP = g.KTP; pf.c = c; pf.levels = levels; pf.L = L; pf.KSA = g.KSA; pf.NOG = g.NOG; pf.NET = g.NET;
        pf.NKSC = g.NKSC; pf.stagger = tuning().stagger;
        const size_t lds = fused_lds_bytes(d, c, K);
        long long nb = ncu & ~1;                          // an even number of workgroups (ftile_off alternates whole and half first tiles)
        if (nb < 2) nb = 2;
        while (nb > 2 && ftile_count(n_tokens, nb, pf.stagger) < nb) nb -= 2;   // few tokens: no more workgroups than tiles
        if (ftile_count(n_tokens, nb, pf.stagger) < nb) nb = ftile_count(n_tokens, nb, pf.stagger);
        if (ev) for (int i = 0; i < 3; ++i) G_TRY(hipEventRecord(ev[i], s));
        auto go = [&](auto kern) -> int {
            G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(F_NW * 64), lds, s, pf);
            G_TRY(hipGetLastError());
            return 0;
        };
        const bool ceq = c == g.CP;
        int rc;
#ifdef RQAE_FUSED_QUICK   /* tools builds: one instantiation, for fast compile-and-look cycles */
#define FUSED_GO(CPV, CEQV) go(gram_fused_kernel<RQAE_X_BF16, 5, true>)
#else
#define FUSED_GO(CPV, CEQV) (x_dtype == RQAE_X_BF16 ? go(gram_fused_kernel<RQAE_X_BF16, CPV, CEQV>) : go(gram_fused_kernel<RQAE_X_F16, CPV, CEQV>))
#endif
        if (g.CP == 4) rc = ceq ? FUSED_GO(4, true) : FUSED_GO(4, false);
        else rc = ceq ? FUSED_GO(5, true) : FUSED_GO(5, false);
#undef FUSED_GO
        if (rc) return rc;
        if (ev) G_TRY(hipEventRecord(ev[3], s));
        return 0;
    }
#ifdef RQAE_FUSED_QUICK
    return gfail("tools build: only the fused kernel is compiled in");
#else
    // one pass = projection + level chain (phaseAB) and reconstruction (phaseC) of the tokens [t0, t0 + nt); `ncu` caps the
    // persistent grids, `cvimg` is the operand-row buffer the two halves share
    const bool recon2_ok = tuning().recon >= 2 && g.CP <= 5 && levels <= 8 && d % 16 == 0 &&
        pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW <= 160 * 1024;
    // inv1pw != NULL: the hook's way back folded into the reconstruction's stores (gram_fused_kernel MODE 6): x is patched in place
    const bool patch = inv1pw != nullptr;
    // gram_recon3_kernel (fp16 two-piece operands): any shape whose values fit 16 k-steps, fed by fp32 planes
    const bool r3_want = q_out && !patch && tuning().recon == 3 && rqae_r3::supported(d, c, levels);
    // FSQ lattice codebooks: the lattice form first (it returns at once unless the block's header names a lattice), the scan kernel of
    // the shape behind it (it returns at once if the header does)
    const int *lat = tuning().lattice && g.shared_cb && (g.CP > 5 || (size_t)K * c * 4 <= 150 * 1024) ? reinterpret_cast<const int *>(static_cast<const char *>(packed) + offsetof(GHeader, lattice_n)) : nullptr;
    // one level group: the lean level chain hands fp32 straight-through values to the reconstruction (which splits them).  The scan
    // kernel behind the lattice form needs its tables in LDS (K <= ~1300); a caller who vouches for the lattice (lean_levels = 5) runs
    // the lattice form alone, which needs only the fp32 rows there (K = 4096 at depth <= 8 used to fail the hook)
    const bool lev2 = tuning().lev2 && g.CP <= 5 && levels <= 8 && g.shared_cb && g.NKC == 1 &&
                      (lev2_lds_bytes(c, K) <= 160 * 1024 || (lat && tuning().lattice == 2)) && (!q_out || recon2_ok || r3_want);
    // narrow codebooks outside the lean path (several level groups, or a codebook too large for the scan's tables): the scan kernels
    // write bf16 operand rows, so fp32 planes for every level exist only when the caller vouches for a lattice (lean_levels = 5: the
    // lattice form runs alone)
    const bool lat_planes = r3_want && !wide && !lev2 && lat && tuning().lattice == 2;
    const bool r3_go = r3_want && (wide || lev2 || lat_planes);
    if (q_out && !r3_go && !nks_t && !wide) return gfail("encode_gram: reconstruction with %d k-steps is not instantiated (codes only: pass q_out = NULL)", nks);
    if (patch && !(inv_rms && recon2_ok && lev2 && patch_S >= 1))
        return gfail("encode_gram_gemma_patch: needs the folded norm, one level group (levels <= 8) and the token-stationary reconstruction");
    auto phaseAB = [&](long long t0, long long nt, hipStream_t s, int ncu, unsigned *cvimg, bool mark, bool strict) -> int {
        if (mark) G_TRY(hipEventRecord(ev[0], s));
        // ---- A
        ProjParams pa;
        pa.x = static_cast<const char *>(x) + (size_t)t0 * d * xs;
        pa.win_img = v.win_img; pa.P = P; pa.dump = dump; pa.n_tokens = nt; pa.tok_pad = tp; pa.d = d; pa.NOG = g.NOG; pa.KSA = g.KSA;
        // the LDS-DMA projection covers 3 (c = 5) or 2 (c <= 4) output groups per workgroup; deeper shapes launch ny workgroups per token
        // tile, placed on one XCD so that the tile's x is read from HBM once (RQAE_GRAM_PROJ=2: the register-pipelined kernel instead)
        if (p3_go) {
            rqae_p3::Args ap;
            ap.img = v.win3_img; ap.maxbits = v.win_maxbits; ap.x = pa.x; ap.P = P; ap.inv_rms = inv_rms ? inv_rms + t0 : nullptr; ap.eps = eps;
            ap.n_tokens = nt; ap.tok_pad = tp; ap.d = d; ap.nog = g.NOG; ap.x_dtype = x_dtype;
            if (int rc = rqae_p3::launch(ap, ncu, s)) return rc;
        } else if (tuning().proj >= 3 && g.CP <= 5 && x_dtype != RQAE_X_F32 && d % 64 == 0) {
            FusedParams pf;
            memset(&pf, 0, sizeof(pf));
            pf.x = pa.x; pf.win_img = v.win_img; pf.P = P; pf.tok_pad = tp; pf.n_tokens = nt; pf.d = d; pf.K = K; pf.KTP = g.KTP; pf.c = c;
            pf.levels = levels; pf.L = L; pf.KSA = g.KSA; pf.NOG = g.NOG; pf.NET = g.NET; pf.NKSC = g.NKSC; pf.stagger = 0; pf.dump = dump;
            pf.stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dump) + 32768);
            const int ngf = g.CP == 4 ? FusedGeom<4>::NG : FusedGeom<5>::NG;
            const size_t lds = g.CP == 4 ? FusedGeom<4>::LDS_A : FusedGeom<5>::LDS_A;
            const int ny = (g.NOG + ngf - 1) / ngf;
            const long long ntl = (nt + F_TILE - 1) / F_TILE;
            long long nb;
            if (ny > 1) {
                long long gx = (ncu / ny) & ~7LL;                 // tiles in flight: a multiple of 8 (one id residue per XCD)
                if (gx < 8) gx = 8;
                const long long need = (ntl + 7) & ~7LL;
                if (gx > need) gx = need;
                nb = gx * ny;
                pf.ny = ny;
            } else {
                nb = ntl < ncu ? ntl : ncu;
                pf.ny = 1;
            }
            pf.ug0 = 0;
            pf.inv_rms = inv_rms ? inv_rms + t0 : nullptr;
            pf.eps = eps;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(F_NW * 64), lds, s, pf);
                G_TRY(hipGetLastError());
                return 0;
            };
            int rc;
            if (inv_rms) {     // Gemma's RMSNorm folded into the projection (MODE 5)
                if (g.CP == 4) rc = x_dtype == RQAE_X_BF16 ? go(gram_fused_kernel<RQAE_X_BF16, 4, true, 5>) : go(gram_fused_kernel<RQAE_X_F16, 4, true, 5>);
                else rc = x_dtype == RQAE_X_BF16 ? go(gram_fused_kernel<RQAE_X_BF16, 5, true, 5>) : go(gram_fused_kernel<RQAE_X_F16, 5, true, 5>);
            } else if (g.CP == 4) rc = x_dtype == RQAE_X_BF16 ? go(gram_fused_kernel<RQAE_X_BF16, 4, true, 1>) : go(gram_fused_kernel<RQAE_X_F16, 4, true, 1>);
            else rc = x_dtype == RQAE_X_BF16 ? go(gram_fused_kernel<RQAE_X_BF16, 5, true, 1>) : go(gram_fused_kernel<RQAE_X_F16, 5, true, 1>);
            if (rc) return rc;
        } else if ((tuning().proj >= 2 || wide) && g.KSA % 4 == 0) {
            const unsigned ny = (unsigned)((g.NOG + 2) / 3);
            long long gx = (nt + A2_NW * 16 * A_TT - 1) / (A2_NW * 16 * A_TT);     // persistent: one workgroup per CU walks the tiles
            const long long cap = ncu / ny > 0 ? ncu / ny : 1;
            if (gx > cap) gx = cap;
            // several output-group blocks per token tile (deep shapes): workgroup (bx, by) has the linear id bx + by gx, and ids that
            // are equal mod 8 share an XCD -- with gx a multiple of 8 the ny workgroups of a tile walk it on ONE XCD at the same
            // time, so its x comes from HBM once and from that L2 ny - 1 times (C4, bf16 x: 2.2 GB fetched for 0.47 GB of x before)
            if (ny > 1 && gx >= 8) gx &= ~7LL;
            const dim3 grid((unsigned)gx, ny);
            auto go = [&](auto kern, int kc) -> int {
                const size_t lds = (size_t)A2_NS * kc * 9 * 1024;
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(A2_NW * 64), lds, s, pa);
                G_TRY(hipGetLastError());
                return 0;
            };
            int rc;
#ifdef RQAE_GRAM_ABLATE
            static const int abl = getenv("RQAE_GRAM_ABL") ? atoi(getenv("RQAE_GRAM_ABL")) : 0;
            if (x_dtype == RQAE_X_BF16 && abl) {
                switch (abl) {
#define ABLC(N) case N: rc = go(gram_project2_kernel<RQAE_X_BF16, N>, XRaw<RQAE_X_BF16>::KC); break;
                ABLC(1) ABLC(2) ABLC(3) ABLC(4) ABLC(5) ABLC(7) ABLC(8) ABLC(9) ABLC(11) ABLC(15)
#undef ABLC
                default: return gfail("RQAE_GRAM_ABL=%d not instantiated", abl);
                }
            } else
#endif
            if (x_dtype == RQAE_X_BF16) rc = go(gram_project2_kernel<RQAE_X_BF16>, XRaw<RQAE_X_BF16>::KC);
            else if (x_dtype == RQAE_X_F16) rc = go(gram_project2_kernel<RQAE_X_F16>, XRaw<RQAE_X_F16>::KC);
            else rc = go(gram_project2_kernel<RQAE_X_F32>, XRaw<RQAE_X_F32>::KC);
            if (rc) return rc;
        } else {
            const dim3 grid((unsigned)((nt + A_NW * 16 * A_TT - 1) / (A_NW * 16 * A_TT)), (unsigned)((g.NOG + 2) / 3));
            const size_t lds = 2 * A_BUF;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(A_NW * 64), lds, s, pa);
                G_TRY(hipGetLastError());
                return 0;
            };
            int rc;
            if (x_dtype == RQAE_X_BF16) rc = go(gram_project_kernel<RQAE_X_BF16>);
            else if (x_dtype == RQAE_X_F16) rc = go(gram_project_kernel<RQAE_X_F16>);
            else rc = go(gram_project_kernel<RQAE_X_F32>);
            if (rc) return rc;
        }
        if (mark) G_TRY(hipEventRecord(ev[1], s));
        // ---- B
        // The lattice form and the scan kernel behind it write the same outputs in the same formats: fp32 planes for the lean / wide
        // reconstructions, operand rows otherwise.
        if (lat) {
            LevParams pl;
            pl.P = P; pl.kconst = v.kconst; pl.gneg = v.gneg; pl.cb_img = v.cb_img; pl.cb_f32 = v.cb_f32; pl.cvimg = cvimg;
            pl.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels;
            pl.n_tokens = nt; pl.tok_pad = tp; pl.ntiles = (nt + 63) / 64;
            pl.levels = levels; pl.L = L; pl.K = K; pl.KTP = g.KTP; pl.c = c; pl.shared_cb = 1; pl.lat_flag = lat; pl.idx_stride = levels;
            // narrow codebooks: 4 waves per workgroup, the fp32 rows and the Gram blocks inside each level group in LDS; wide ones: 8
            // waves, persistent, the whole upper triangle of Gram blocks in LDS (82 KB at depth 16: one workgroup per CU)
            const int nw = wide ? 8 : 4;
            long long nb = (pl.ntiles + nw - 1) / nw;
            if (nb > (wide ? 1LL : 8LL) * ncu) nb = (wide ? 1LL : 8LL) * ncu;
            const size_t lds = g.CP <= 5 ? (((size_t)K * c * 4 + 15) & ~(size_t)15) + (size_t)((levels + g.GL - 1) / g.GL) * g.GL * g.GL * g.GB * 4
                                         : (size_t)L * (L - 1) / 2 * g.GB * 4;
            auto go = [&](auto kern) -> int {
                if (lds > 64 * 1024) G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(nw * 64), lds, s, pl);
                G_TRY(hipGetLastError());
                return 0;
            };
            // a two-point lattice at C4's shape: the chain as small matrix products (gram_sign3.hpp), ahead of the lattice form, which
            // returns when that kernel ran (the block's sign_exact word and lattice_n == 2, both read on the device)
            if (wide && tuning().recon == 3 && levels == L && rqae_s3::supported(c, L, g.shared_cb)) {
                rqae_s3::Args sa;
                sa.P = P; sa.kconst = v.kconst; sa.img = v.sign_img; sa.maxbits = v.sign_maxbits; sa.lat_flag = lat; sa.exact_flag = v.sign_exact; sa.cb = v.cb_f32;
                sa.planes = reinterpret_cast<float *>(cvimg); sa.idx = pl.idx; sa.n_tokens = nt; sa.tok_pad = tp; sa.c = c; sa.L = L; sa.K = K; sa.idx_stride = levels;
                if (int rc = rqae_s3::launch(sa, ncu, s)) return rc;
                pl.sign_flag = v.sign_exact;
            }
            const bool multi = levels > g.GL;
            int rc;
            const bool ceq = c == g.CP;       // no padded coordinate: the common case, with c a compile-time fact
#define LATK(CPV, MV, IV) (ceq ? go(gram_levels_wide_kernel<CPV, 4, MV, IV, true, true>) : go(gram_levels_wide_kernel<CPV, 4, MV, IV, true, false>))
            if (wide) rc = ceq ? go(gram_levels_wide_kernel<13, 8, true, false, true, true>) : go(gram_levels_wide_kernel<13, 8, true, false, true, false>);
            else if (lev2) rc = g.CP == 4 ? LATK(4, false, false) : LATK(5, false, false);
            else if (lat_planes && multi && g.CP == 4 && ceq && lattice_points(K, c) == 8)
                rc = go(gram_levels_wide_kernel<4, 4, true, false, true, true, 8>);      // C5's lattice: approximate scores choose (lattice_fast)
            else if (lat_planes && multi && g.CP == 4 && ceq && lattice_points(K, c) == 5)
                rc = go(gram_levels_wide_kernel<4, 4, true, false, true, true, 5>);      // the released shape's
            else if (lat_planes) rc = multi ? (g.CP == 4 ? LATK(4, true, false) : LATK(5, true, false)) : (g.CP == 4 ? LATK(4, false, false) : LATK(5, false, false));
            else if (multi) rc = g.CP == 4 ? LATK(4, true, true) : LATK(5, true, true);
            else rc = g.CP == 4 ? LATK(4, false, true) : LATK(5, false, true);
#undef LATK
            if (rc) return rc;
            if (tuning().lattice == 2) return 0;          // the caller vouches for the block: no scan kernel behind the lattice form
        }
        if (wide) {
            LevParams pb;
            pb.P = P; pb.kconst = v.kconst; pb.gneg = v.gneg; pb.cb_img = v.cb_img; pb.cb_f32 = v.cb_f32; pb.cvimg = cvimg;
            pb.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels;
            pb.n_tokens = nt; pb.tok_pad = tp; pb.ntiles = (nt + 63) / 64;
            pb.levels = levels; pb.L = L; pb.K = K; pb.KTP = g.KTP; pb.c = c; pb.shared_cb = g.shared_cb; pb.lat_flag = lat; pb.idx_stride = levels;
            // 4 waves per workgroup (one per SIMD) until the batch fills the chip twice over: 8 then halve the image's L2 -> LDS traffic
#ifdef W_FORCE_NW8   /* tools build: two waves per SIMD on half the CUs */
            const int nw = 8;
#else
            const int nw = (pb.ntiles + 7) / 8 >= 2LL * ncu ? 8 : 4;
#endif
            const size_t lds = wide_levels_lds_bytes(g.NKC, nw);
            long long nb = (pb.ntiles + nw - 1) / nw;
            if (nb > ncu) nb = ncu;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3((nw + W_NWL) * 64), lds, s, pb);     // + the loader waves
                G_TRY(hipGetLastError());
                return 0;
            };
            return nw == 8 ? go(gram_levels_wide_kernel<13, 8, true>) : go(gram_levels_wide_kernel<13, 4, true>);
        }
        if (lev2 && tuning().lev2 == 3 && c == g.CP) {
            // the streamed form of gram_wide.hpp for one level group: 8 computing waves (two per SIMD) + 2 loader waves, the codebook
            // image through a 24 KB ring instead of resident, fp32 rows and Gram blocks from L2: 56 KB of LDS instead of 156
            LevParams pb;
            pb.P = P; pb.kconst = v.kconst; pb.gneg = v.gneg; pb.cb_img = v.cb_img; pb.cb_f32 = v.cb_f32; pb.cvimg = cvimg;
            pb.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels;
            pb.n_tokens = nt; pb.tok_pad = tp; pb.ntiles = (nt + 63) / 64;
            pb.levels = levels; pb.L = L; pb.K = K; pb.KTP = g.KTP; pb.c = c; pb.shared_cb = g.shared_cb; pb.lat_flag = lat; pb.idx_stride = levels;
            const int nw = 8;
            const size_t lds = wide_levels_lds_bytes(g.NKC, nw);
            long long nb = (pb.ntiles + nw - 1) / nw;
            if (nb > ncu) nb = ncu;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3((nw + W_NWL) * 64), lds, s, pb);
                G_TRY(hipGetLastError());
                return 0;
            };
            return g.CP == 4 ? go(gram_levels_wide_kernel<4, 8, false>) : go(gram_levels_wide_kernel<5, 8, false>);
        }
        if (lev2) {
            Lev2Params pl;
            pl.P = P; pl.kconst = v.kconst; pl.gneg = v.gneg; pl.cb_img = v.cb_img; pl.cb_f32 = v.cb_f32;
            pl.cvf = reinterpret_cast<float *>(cvimg); pl.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels;
            pl.n_tokens = nt; pl.tok_pad = tp; pl.ntiles = (nt + 63) / 64;
            pl.levels = levels; pl.L = L; pl.K = K; pl.KTP = g.KTP; pl.c = c; pl.lat_flag = lat;
            pl.stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dump) + 32768 + (2048 + 8 * 36 * 4) * 8);
            // 16 waves per workgroup (four per SIMD: one wave's serial part behind the others' scans) where the tables leave room
            const int nw = (tuning().lev2 == 2 && c == g.CP && lev2_lds_bytes(c, K, 16) <= 160 * 1024) ? 16 : F_NW;
            const size_t lds = lev2_lds_bytes(c, K, nw);
            long long nb = (pl.ntiles + nw - 1) / nw;
            const long long bcap = (strict || nw == 16) ? ncu : 2LL * ncu;
            if (nb > bcap) nb = bcap;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(nw * 64), lds, s, pl);
                G_TRY(hipGetLastError());
                return 0;
            };
            const bool ceq = c == g.CP;
            if (nw == 16) return g.CP == 4 ? go(gram_levels2_kernel<4, true, 16>) : go(gram_levels2_kernel<5, true, 16>);
            return g.CP == 4 ? (ceq ? go(gram_levels2_kernel<4, true>) : go(gram_levels2_kernel<4, false>))
                             : (ceq ? go(gram_levels2_kernel<5, true>) : go(gram_levels2_kernel<5, false>));
        }
        LevParams pb;
        pb.P = P; pb.kconst = v.kconst; pb.gneg = v.gneg; pb.cb_img = v.cb_img; pb.cb_f32 = v.cb_f32; pb.cvimg = cvimg;
        pb.idx = reinterpret_cast<long long *>(indices) + (size_t)t0 * levels;
        pb.n_tokens = nt; pb.tok_pad = tp; pb.ntiles = (nt + 63) / 64;
        pb.levels = levels; pb.L = L; pb.K = K; pb.KTP = g.KTP; pb.c = c; pb.shared_cb = g.shared_cb; pb.lat_flag = lat; pb.idx_stride = levels;
        if (g.CP <= 5 && g.NKC == 1 && (!g.shared_cb || (!levels_resident_ok(g) && g.CP == 4 && c == 4 && tuning().lev2 == 3))) {
            // per-level codebooks (levels_free_ok): every level streams its own image through the ring.  Also a
            // variant (lean_levels = 3) for a narrow codebook too large for LDS (C5: K 4096, depth 32): gram_wide.hpp's streamed form
            // -- 8 computing waves + 2 loader waves, scan maxima one tile late -- writing the operand rows gram_reconstruct_kernel
            // reads.  Measured at C5, 262 144 tokens: 2.42 ms against gram_levels_stream_kernel's 2.31 (60 cycles per matrix
            // instruction and wave in the scan, two waves per SIMD), so the default stays the older kernel
            const int nw = 8;
            const size_t lds = wide_levels_lds_bytes(g.NKC, nw);
            long long nb = (pb.ntiles + nw - 1) / nw;
            if (nb > ncu) nb = ncu;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3((nw + W_NWL) * 64), lds, s, pb);
                G_TRY(hipGetLastError());
                return 0;
            };
            if (int rc = g.CP == 4 ? go(gram_levels_wide_kernel<4, 8, true>) : go(gram_levels_wide_kernel<5, 8, true>)) return rc;
        } else if (!levels_resident_ok(g)) {
            // persistent: a workgroup's waves walk the ring together.  8 waves per workgroup halve the L2 -> LDS refill per token;
            // 4 when 8 would leave CUs without a workgroup
            const int nw = (pb.ntiles + SL_NW - 1) / SL_NW >= ncu ? SL_NW : 4;
            const size_t lds = stream_lds_bytes(nw);
            long long nb = (pb.ntiles + nw - 1) / nw;
            if (nb > ncu) nb = ncu;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(nw * 64), lds, s, pb);
                G_TRY(hipGetLastError());
                return 0;
            };
            const bool multi = levels > g.GL;
            int rc;
#define SLK(CPV, MV) (nw == SL_NW ? go(gram_levels_stream_kernel<CPV, MV, SL_NW>) : go(gram_levels_stream_kernel<CPV, MV, 4>))
            if (g.CP == 4) rc = multi ? SLK(4, true) : SLK(4, false);
            else rc = multi ? SLK(5, true) : SLK(5, false);
#undef SLK
            if (rc) return rc;
        } else {
            const size_t lds = (size_t)g.KTP * g.NKC * 1024 + ((((size_t)K + 64) * c * 4 + 15) & ~(size_t)15) + (size_t)B_NW * g.NKC * 4096 +
                               (size_t)g.GL * g.GL * g.GB * 4;
            long long nb = (pb.ntiles + B_NW - 1) / B_NW;
            const long long bcap = strict ? ncu : 2LL * ncu;     // one workgroup fits a CU; the plain schedule queues two rounds
            if (nb > bcap) nb = bcap;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(B_NW * 64), lds, s, pb);
                G_TRY(hipGetLastError());
                return 0;
            };
            const bool multi = levels > g.GL;
            const int rc = g.CP == 4 ? (multi ? go(gram_levels_kernel<4, B_NW, true>) : go(gram_levels_kernel<4, B_NW, false>))
                                     : (multi ? go(gram_levels_kernel<5, B_NW, true>) : go(gram_levels_kernel<5, B_NW, false>));
            if (rc) return rc;
        }
        return 0;
    };
    auto phaseC = [&](long long t0, long long nt, hipStream_t s, int ncu, unsigned *cvimg, bool mark) -> int {
        if (mark) G_TRY(hipEventRecord(ev[2], s));
        // ---- C
        if (patch) {
            FusedParams pf;
            memset(&pf, 0, sizeof(pf));
            pf.wout_img = v.wout_img; pf.bias = v.bias_prefix + (size_t)(levels - 1) * d; pf.out = nullptr;
            pf.cvimg = cvimg; pf.cvf = reinterpret_cast<const float *>(cvimg); pf.tok_pad = tp; pf.n_tokens = nt; pf.d = d; pf.K = K; pf.KTP = g.KTP; pf.c = c; pf.levels = levels; pf.L = L;
            pf.KSA = g.KSA; pf.NOG = g.NOG; pf.NET = g.NET; pf.NKSC = g.NKSC; pf.stagger = 0; pf.dump = dump; pf.nks_live = nks;
            pf.stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dump) + 32768);
            pf.hs_out = const_cast<char *>(static_cast<const char *>(x)) + (size_t)t0 * d * xs; pf.hs_dtype = x_dtype; pf.inv1pw = inv1pw;
            pf.inv_rms = inv_rms + t0; pf.tok0 = t0; pf.S = patch_S; pf.skip_bos = patch_skip_bos;
            long long nb = (nt + F_TILE - 1) / F_TILE;
            if (nb > ncu) nb = ncu;
            const size_t lds = 2 * pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW;
            if (lds > 160 * 1024) return gfail("encode_gram_gemma_patch: d=%d does not leave room for 1 / (1 + w) in LDS", d);
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(F_NW * 64), lds, s, pf);
                G_TRY(hipGetLastError());
                return 0;
            };
            if (int rc = g.CP == 4 ? go(gram_fused_kernel<RQAE_X_BF16, 4, true, 6>) : go(gram_fused_kernel<RQAE_X_BF16, 5, true, 6>)) return rc;
        } else if (r3_go) {
            rqae_r3::Args ar;
            ar.img = v.wout3_img; ar.maxbits = v.wout_maxbits; ar.bias = v.bias_prefix + (size_t)(levels - 1) * d;
            ar.planes = reinterpret_cast<const float *>(cvimg); ar.out = q_out + (size_t)t0 * d; ar.n_tokens = nt; ar.tok_pad = tp;
            ar.d = d; ar.c = c; ar.L = L; ar.levels = levels; ar.cvs = wide ? 16 : g.CP;
            if (int rc = rqae_r3::launch(ar, ncu, s)) return rc;
        } else if (q_out && wide) {
            RecWideParams pw;
            pw.wout_img = v.wout_img; pw.bias = v.bias_prefix + (size_t)(levels - 1) * d; pw.cvf = reinterpret_cast<const float *>(cvimg);
            pw.out = q_out + (size_t)t0 * d; pw.n_tokens = nt; pw.tok_pad = tp; pw.d = d; pw.NKSC = g.NKSC; pw.NET = g.NET; pw.levels = levels;
            long long nb = (nt + 127) / 128;
            if (nb > ncu) nb = ncu;
            auto go = [&](auto kern, size_t lds) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(512), lds, s, pw);
                G_TRY(hipGetLastError());
                return 0;
            };
            int rc;
            switch (g.NLG) {
            case 1: rc = go(gram_recon_wide_kernel<13, 1>, wide_recon_lds_bytes<13, 1>()); break;
            case 2: rc = go(gram_recon_wide_kernel<13, 2>, wide_recon_lds_bytes<13, 2>()); break;
            case 3: rc = go(gram_recon_wide_kernel<13, 3>, wide_recon_lds_bytes<13, 3>()); break;
            case 4: rc = go(gram_recon_wide_kernel<13, 4>, wide_recon_lds_bytes<13, 4>()); break;
            default: return gfail("encode_gram: wide reconstruction serves up to 16 levels");
            }
            if (rc) return rc;
        } else if (q_out && recon2_ok) {
            FusedParams pf;
            memset(&pf, 0, sizeof(pf));
            pf.wout_img = v.wout_img; pf.bias = v.bias_prefix + (size_t)(levels - 1) * d; pf.out = q_out + (size_t)t0 * d;
            pf.cvimg = cvimg; pf.cvf = reinterpret_cast<const float *>(cvimg); pf.tok_pad = tp; pf.n_tokens = nt; pf.d = d; pf.K = K; pf.KTP = g.KTP; pf.c = c; pf.levels = levels; pf.L = L;
            pf.KSA = g.KSA; pf.NOG = g.NOG; pf.NET = g.NET; pf.NKSC = g.NKSC; pf.stagger = 0; pf.dump = dump; pf.nks_live = nks;
            pf.stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dump) + 32768);
            long long nb = (nt + F_TILE - 1) / F_TILE;
            if (nb > ncu) nb = ncu;
            const size_t lds = pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(F_NW * 64), lds, s, pf);
                G_TRY(hipGetLastError());
                return 0;
            };
            const int rc = lev2 ? (g.CP == 4 ? go(gram_fused_kernel<RQAE_X_BF16, 4, true, 3>) : go(gram_fused_kernel<RQAE_X_BF16, 5, true, 3>))
                                : (g.CP == 4 ? go(gram_fused_kernel<RQAE_X_BF16, 4, true, 2>) : go(gram_fused_kernel<RQAE_X_BF16, 5, true, 2>));
            if (rc) return rc;
        } else if (q_out) {
            RecParams pc;
            pc.wout_img = v.wout_img; pc.bias = v.bias_prefix + (size_t)(levels - 1) * d; pc.cvimg = cvimg;
            pc.out = q_out + (size_t)t0 * d;
            const int ctt = nks_t <= 8 ? tuning().c_tt : 1;   // token tiles per wave: the cv pieces of a wave's tokens stay in registers
            const int cnw = ctt == 1 && nks_t <= 8 ? 16 : 8;  // waves per workgroup
            pc.n_tokens = nt; pc.tok_pad = tp; pc.npairs = (nt + 16 * ctt - 1) / (16 * ctt);
            pc.d = d; pc.NKSC = g.NKSC; pc.nks = nks; pc.NET = g.NET;
            const int ets = rec_ets(nks_t);
            pc.nslices = (g.NET + ets - 1) / ets;
            int lq = ncu / (8 * pc.nslices);
            if (lq < 1) lq = 1;
            // no more workgroups than there are token groups to hand out
            while (lq > 1 && 8LL * (lq - 1) * cnw >= pc.npairs) --lq;
            pc.LQ = lq;
            const size_t lds = (size_t)ets * nks_t * 1024 + (size_t)ets * 16 * sizeof(float);
            const unsigned nb = 8u * pc.nslices * lq;
            auto go = [&](auto kern) -> int {
                G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(nb), dim3(cnw * 64), lds, s, pc);
                G_TRY(hipGetLastError());
                return 0;
            };
            int rc = 1;
#define REC(NKS_) (ctt == 2 ? go(gram_reconstruct_kernel<NKS_, 2, 8>) : go(gram_reconstruct_kernel<NKS_, 1, 16>))
            switch (nks_t) {
            case 2: rc = REC(2); break;
            case 3: rc = REC(3); break;
            case 4: rc = REC(4); break;
            case 6: rc = REC(6); break;
            case 8: rc = REC(8); break;
            case 12: rc = go(gram_reconstruct_kernel<12, 1, 8>); break;
            case 16: rc = go(gram_reconstruct_kernel<16, 1, 8>); break;
            case 24: rc = go(gram_reconstruct_kernel<24, 1, 8>); break;
            default: return gfail("encode_gram: no reconstruct kernel for %d k-steps", nks_t);
            }
#undef REC
            if (rc) return rc;
        }
        if (mark) G_TRY(hipEventRecord(ev[3], s));
        return 0;
    };
    // ---- pipelined schedule: passes of pipe_chunk tokens; the reconstruction of pass i (HBM writes) runs on the second stream
    // beside the projection (HBM reads) and the level chain (matrix cores, no memory traffic) of pass i+1.  Every workgroup of
    // these kernels fills a CU's LDS, so capped grids split the CUs between the two streams.
    if (!ev && q_out && !patch && tuning().pipe_chunk > 0 && n_tokens > tuning().pipe_chunk && tuning().pipe_chunk <= tp) {
        Side *sd = side_for_device();
        if (sd) {
            const long long pch = tuning().pipe_chunk;
            int cap_ab = tuning().cap_ab < ncu - 8 ? tuning().cap_ab : ncu - 8;
            if (cap_ab < 8) cap_ab = 8;
            const int cap_c = ncu - cap_ab;
            std::lock_guard<std::mutex> lock(sd->mu);
            G_TRY(hipEventRecord(sd->fork, s));
            G_TRY(hipStreamWaitEvent(sd->st, sd->fork, 0));
            int i = 0;
            for (long long t0 = 0; t0 < n_tokens; t0 += pch, ++i) {
                const long long nt = n_tokens - t0 < pch ? n_tokens - t0 : pch;
                unsigned *cvb = (i & 1) ? cvimg2 : cvimg;
                if (i >= 2) G_TRY(hipStreamWaitEvent(s, sd->evC[(i - 2) & 3], 0));     // the buffer's last reader
                if (int rc = phaseAB(t0, nt, s, cap_ab, cvb, false, true)) return rc;
                G_TRY(hipEventRecord(sd->evB[i & 3], s));
                G_TRY(hipStreamWaitEvent(sd->st, sd->evB[i & 3], 0));
                if (int rc = phaseC(t0, nt, sd->st, cap_c, cvb, false)) return rc;
                G_TRY(hipEventRecord(sd->evC[i & 3], sd->st));
            }
            G_TRY(hipStreamWaitEvent(s, sd->evC[(i - 1) & 3], 0));                      // join
            return 0;
        }
    }
    for (long long t0 = 0; t0 < n_tokens; t0 += chunk) {
        const long long nt = n_tokens - t0 < chunk ? n_tokens - t0 : chunk;
        const bool mark = ev && t0 + chunk >= n_tokens;
        if (int rc = phaseAB(t0, nt, s, ncu, cvimg, mark, false)) return rc;
        if (int rc = phaseC(t0, nt, s, ncu, cvimg, mark)) return rc;
    }
    return 0;
#endif
}
}  // namespace

extern "C" {

int rqae_hip_encode_gram(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *x, int x_dtype,
                         int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *workspace,
                         const rqae_hip_gram_options *options, void *stream)
{
    return encode_gram_impl(packed, d, c, K, L, codebook_shared, x, x_dtype, n_tokens, levels, q_out, indices, workspace, options,
                            stream, nullptr);
}

// RQAE.hook's `norm` + forward for a Gemma-2 model (rqae/model.py:279-281, rqae/llm.py:65-66) on the matrix-core arm: x is the
// hooked layer's output itself; the packed block must have been built with in_scale = 1 + norm.weight.
int rqae_hip_encode_gram_gemma_supported(int d, int c, int K, int L, int codebook_shared, int x_dtype)
{
    return rqae_hip_gram_supported(d, c, K, L, codebook_shared) && cp_of(c) <= 5 && d % 64 == 0 && (x_dtype == RQAE_X_BF16 || x_dtype == RQAE_X_F16) ? 1 : 0;
}

int rqae_hip_encode_gram_gemma(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *hs, int hs_dtype,
                               float eps, int64_t n_tokens, int levels, float *q_out, int64_t *indices, float *inv_rms,
                               void *workspace, const rqae_hip_gram_options *options, void *stream)
{
    if (!inv_rms) return gfail("encode_gram_gemma: inv_rms is NULL");
    if (!rqae_hip_encode_gram_gemma_supported(d, c, K, L, codebook_shared, hs_dtype))
        return gfail("encode_gram_gemma: unsupported shape / dtype d=%d c=%d K=%d L=%d hs_dtype=%d", d, c, K, L, hs_dtype);
    return encode_gram_impl(packed, d, c, K, L, codebook_shared, hs, hs_dtype, n_tokens, levels, q_out, indices, workspace, options,
                            stream, nullptr, inv_rms, eps);
}

// The whole of RQAE.hook's arithmetic for a Gemma-2 layer output, in place (rqae/model.py:276-289 with store = None, replace = True):
// norm folded into the projection, level chain, and the reconstruction kernel writes new = (q / (1 + w)) / rs over hs in its own
// dtype, BOS rows (token % S == 0) kept when skip_bos.  `quantized` (fp32) is never written.  inv_1pw = 1 / (1 + norm.weight), [d]
// padded to a whole KiB; inv_rms: [n_tokens] floats of scratch.
// Can rqae_hip_encode_gram_gemma_patch serve this call?  One level group (levels <= 8), 2-byte hidden states, the token-stationary
// reconstruction's LDS, and fp32 planes from the level chain: the scan kernel's tables must fit LDS (K <= ~1300) unless the caller
// vouches for an FSQ lattice (lattice_vouched: rqae_hip_gram_lattice said so and the call passes lean_levels = 5), whose form needs
// only the fp32 rows there.  0 -> run rqae_hip_encode_gram_gemma + rqae_hip_gemma_denorm_patch instead.
int rqae_hip_encode_gram_gemma_patch_supported(int d, int c, int K, int L, int codebook_shared, int levels, int hs_dtype, int lattice_vouched)
{
    if (!rqae_hip_encode_gram_gemma_supported(d, c, K, L, codebook_shared, hs_dtype) || levels < 1 || levels > 8 || levels > L) return 0;
    const Geom g = make_geom(d, c, K, L, codebook_shared ? 1 : 0);
    if (g.CP > 5 || g.NKC != 1 || !g.shared_cb || d % 16) return 0;
    if (2 * pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW > 160 * 1024) return 0;
    if (lev2_lds_bytes(c, K) <= 160 * 1024) return 1;
    return lattice_vouched && (size_t)K * c * 4 <= 150 * 1024 ? 1 : 0;
}

int rqae_hip_encode_gram_gemma_patch(const void *packed, int d, int c, int K, int L, int codebook_shared, void *hs_inout, int hs_dtype,
                                     float eps, const float *inv_1pw, int S, int skip_bos, int64_t n_tokens, int levels,
                                     int64_t *indices, float *inv_rms, void *workspace, const rqae_hip_gram_options *options,
                                     void *stream)
{
    if (!inv_rms || !inv_1pw || !hs_inout) return gfail("encode_gram_gemma_patch: null pointer");
    if (S < 1) return gfail("encode_gram_gemma_patch: S=%d", S);
    if (!rqae_hip_encode_gram_gemma_supported(d, c, K, L, codebook_shared, hs_dtype) || levels > 8)
        return gfail("encode_gram_gemma_patch: unsupported shape / dtype d=%d c=%d K=%d L=%d levels=%d hs_dtype=%d", d, c, K, L, levels, hs_dtype);
    return encode_gram_impl(packed, d, c, K, L, codebook_shared, hs_inout, hs_dtype, n_tokens, levels, nullptr, indices, workspace, options,
                            stream, nullptr, inv_rms, eps, inv_1pw, S, skip_bos);
}

// RQAE.decode on the matrix-core arm: out = sum_l (W_out[l] codebook[idx_l] + b_out[l]) as the reconstruction kernel computes it
// (bf16-piece products, fp32 accumulation), the codebook rows gathered by the codes inside the kernel.
int rqae_hip_decode_gram_supported(int d, int c, int K, int L, int codebook_shared, int levels)
{
    if (!rqae_hip_gram_supported(d, c, K, L, codebook_shared) || levels < 1 || levels > L || levels > 8 || d % 16) return 0;
    const Geom g = make_geom(d, c, K, L, codebook_shared ? 1 : 0);
    if (g.CP > 5) return 0;
    return pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW <= 160 * 1024 ? 1 : 0;
}

int rqae_hip_decode_gram(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *indices, int idx_dtype,
                         int64_t n_tokens, int levels, float *out, int32_t *status, void *stream)
{
#ifdef RQAE_FUSED_QUICK
    return gfail("tools build: only the fused kernel is compiled in");
#else
    if (!rqae_hip_decode_gram_supported(d, c, K, L, codebook_shared, levels))
        return gfail("decode_gram: unsupported shape d=%d c=%d K=%d L=%d levels=%d (one level group: levels <= 8)", d, c, K, L, levels);
    if (!packed || !indices || !out) return gfail("decode_gram: null pointer");
    if (idx_dtype != RQAE_IDX_I64 && idx_dtype != RQAE_IDX_I32 && idx_dtype != RQAE_IDX_U16) return gfail("decode_gram: bad idx_dtype %d", idx_dtype);
    if (n_tokens < 0) return gfail("decode_gram: negative n_tokens");
    if (reinterpret_cast<uintptr_t>(out) & 15) return gfail("decode_gram: out must be 16-byte aligned");
    if (n_tokens == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const GView v = gview(packed, d, c, K, L, codebook_shared ? 1 : 0);
    const Geom &g = v.g;
    FusedParams pf;
    memset(&pf, 0, sizeof(pf));
    pf.wout_img = v.wout_img; pf.bias = v.bias_prefix + (size_t)(levels - 1) * d; pf.out = out; pf.cb_f32 = v.cb_f32; pf.codes = indices; pf.status = status;
    pf.n_tokens = n_tokens; pf.tok_pad = pad64(n_tokens); pf.d = d; pf.K = K; pf.KTP = g.KTP; pf.c = c; pf.levels = levels; pf.L = L;
    pf.KSA = g.KSA; pf.NOG = g.NOG; pf.NET = g.NET; pf.NKSC = g.NKSC; pf.stagger = 0; pf.nks_live = nks_recon(g, levels);
    long long nb = (n_tokens + F_TILE - 1) / F_TILE;
    const int ncu = device_cus();
    if (nb > ncu) nb = ncu;
    const size_t lds = pad1k((size_t)d * 4) + 3 * (size_t)(g.CP == 4 ? FusedGeom<4>::STAGE_C : FusedGeom<5>::STAGE_C) + F_NW * F_TRW;
    auto go = [&](auto kern) -> int {
        G_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(F_NW * 64), lds, s, pf);
        G_TRY(hipGetLastError());
        return 0;
    };
#define DEC(CPV) (idx_dtype == RQAE_IDX_I64 ? go(gram_fused_kernel<RQAE_IDX_I64, CPV, true, 4>) \
                  : idx_dtype == RQAE_IDX_I32 ? go(gram_fused_kernel<RQAE_IDX_I32, CPV, true, 4>) : go(gram_fused_kernel<RQAE_IDX_U16, CPV, true, 4>))
    const int rc = g.CP == 4 ? DEC(4) : DEC(5);
#undef DEC
    return rc;
#endif
}

int rqae_hip_encode_gram_phases(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *x, int x_dtype,
                                int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *workspace,
                                const rqae_hip_gram_options *options, void *stream, float *phase_ms)
{
    if (!phase_ms) return gfail("encode_gram_phases: phase_ms is NULL");
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) G_TRY(hipEventCreate(&ev[i]));
    int rc = encode_gram_impl(packed, d, c, K, L, codebook_shared, x, x_dtype, n_tokens, levels, q_out, indices, workspace, options, stream, ev);
    if (!rc && n_tokens > 0) {
        hipError_t e = hipEventSynchronize(ev[3]);
        if (e != hipSuccess) rc = gfail("encode_gram_phases: %s", hipGetErrorString(e));
        for (int i = 0; i < 3 && !rc; ++i)
            if (hipEventElapsedTime(&phase_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = gfail("encode_gram_phases: event timing failed");
    } else if (!rc) {
        phase_ms[0] = phase_ms[1] = phase_ms[2] = 0.0f;
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

}  // extern "C"

#ifdef W_TIMELINE
extern "C" int rqae_hip_debug_wide_stamps(unsigned long long *host_out, size_t n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * n);
}
#endif
#endif  /* RQAE_GRAM_NO_ABI */