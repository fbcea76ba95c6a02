// rqae_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the RQAE hot path.
//
// What runs here (reference file:line relative to /root/reference):
//   rqae_hip_encode         RQAE.forward level loop            rqae/model.py:199-230
//   rqae_hip_quantize       RQAE.quantize at temperature 0     rqae/model.py:187-197
//   rqae_hip_decode[_values] RQAE.decode                       rqae/model.py:232-252
//   rqae_hip_gemma_norm / _denorm_patch   the steps either side of forward in RQAE.hook   rqae/model.py:276-289
//   rqae_hip_intensity[_prepare]          RQAEFeature.intensity                             rqae/feature.py:102-129
//
// Arithmetic contract (restated on the CPU by
// oracle/rqae_oracle.c, which these kernels match bit for bit):
//   * everything is fp32, one rounding per written operation; compiled with
//     -ffp-contract=off, fused multiply-adds appear only as explicit fmaf()
//   * lin_in  : d is zero-padded to dp = 256*P and cut into 4 slices of 64*P;
//               element elem_of(P, s, i, lane) belongs to (slice s, lane, step i); each
//               (slice, lane) runs a fmaf chain over i from +0; the 64 lanes are
//               summed by the pairing lane^32, lane^16, lane^8, 7-mirror, lane^1,
//               lane^2; slices as ((s0+s1)+(s2+s3)); bias last
//   * ||q||   : ATen's 2-norm order (8-wide fma chunks, one 4-chunk of rounded
//               squares, fma tail), IEEE sqrt; q/||q|| IEEE division
//   * cos-sim : k-ordered fmaf chain from +0; argmax = first maximum
//   * STE     : q + (cv - q);  lin_out: k-ordered fmaf chain from +0, then + bias
//
// Work decomposition of the encode kernels: one workgroup owns TG tokens for all
// levels.  Wave (slice, half) keeps the residual and the running reconstruction of
// its TPW tokens x 64*P elements in VGPRs for the whole level loop (the N x d
// residual never goes back to HBM); weights stream from L2 once per level per
// workgroup (LDS-DMA into LDS in rqae_encode_lds_kernel, the main path); the only
// exchange per level is 4c partial sums in and c floats out per token through LDS.
// Files: encode_lds.hpp (rqae_encode_lds_kernel, the main path; its d = 2304 instantiations are the
// translation unit encode_main.hip), this file (pack, rqae_encode_kernel for generic shapes, dispatch,
// C ABI), decode.hpp (quantize + decode kernels), hook.hpp, intensity.hpp, rqae_common.hpp (shared
// helpers).  DESIGN.md section 5 describes each kernel with its roofline.
#include "encode_lds.hpp"

namespace {

// Fill the pass records (see make_layout).  One thread per float.
__global__ void pack_records_kernel(Header h, int nq, int qout, float *rec_p, const float *w_in, const float *w_out,
                                    const float *b_out)
{
    const int P = h.dp / 256;
    const size_t total = (size_t)(h.L + 1) * NSLICE * nq * 256;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += stride) {
        const int e4 = (int)(u & 3), lane = (int)((u >> 2) & 63);
        size_t rest = u >> 8;                      // (pl*4 + slice)*nq + quad
        const int quad = (int)(rest % nq); rest /= nq;
        const int slice = (int)(rest & 3);
        const size_t pl = rest >> 2;
        float val = 0.0f;
        if (quad < qout) {                         // lin_out stream of level pl-1
            const int n = quad * 4 + e4, i = n / (h.c + 1), k = n % (h.c + 1);
            if (i < P && pl >= 1) {
                const int el = elem_of(P, slice, i, lane);
                if (el < h.d) val = k < h.c ? w_out[((pl - 1) * h.d + el) * h.c + k] : b_out[(pl - 1) * h.d + el];
            }
        } else {                                   // lin_in stream of level pl
            const int n = (quad - qout) * 4 + e4, i = n / h.c, j = n % h.c;
            if (i < P && pl < (size_t)h.L) {
                const int el = elem_of(P, slice, i, lane);
                if (el < h.d) val = w_in[(pl * h.c + j) * h.d + el];
            }
        }
        rec_p[u] = val;
    }
}

// Level-0 codebook in the LDS layout of stage_rows (see make_layout).  One thread per float of the image.
__global__ void pack_cb_image_kernel(int c, int K, int n_img, float *img, const float *codebook)
{
    const int tail = c > 4 ? c - 4 : 0;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_img; u += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (u < K * 4) {
            const int r = u >> 2, k = u & 3;
            if (k < c) v = codebook[(size_t)r * c + k];
        } else if (u < K * 4 + K * tail) {
            const int r = (u - K * 4) / tail, k = 4 + (u - K * 4) % tail;
            v = codebook[(size_t)r * c + k];
        }
        img[u] = v;
    }
}

__global__ void pack_kernel(Header *hdr, Header h, float *w_in_p, float *w_out_p, float *b_out_p,
                            const float *w_in, const float *w_out, const float *b_out)
{
    const size_t n_w = (size_t)h.L * h.c * h.dp;
    const size_t n_b = (size_t)h.L * h.dp;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_w; i += stride) {
        const int e = (int)(i % h.dp);
        const size_t lj = i / h.dp;  // l*c + j
        const int j = (int)(lj % h.c);
        const size_t l = lj / h.c;
        w_in_p[i] = e < h.d ? w_in[lj * h.d + e] : 0.0f;
        w_out_p[i] = e < h.d ? w_out[(l * h.d + e) * h.c + j] : 0.0f;  // transposed: [l][k][e]
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_b; i += stride) {
        const int e = (int)(i % h.dp);
        const size_t l = i / h.dp;
        b_out_p[i] = e < h.d ? b_out[l * h.d + e] : 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = h;
}

// P  : elements per lane per slice (dp = 256*P)      C : codebook_dim
// TPW: tokens held by one wave                        NH: token halves (waves = 4*NH)
// GEN: P and C are upper bounds, the live sizes come from EncParams (slow generic path)
template <int P, int C, int TPW, int NH, bool RECON, bool GEN>
__global__ void __launch_bounds__(256 * NH, NH)
rqae_encode_kernel(const EncParams p)
{
    constexpr int NW = NSLICE * NH;
    constexpr int TG = TPW * NH;
    constexpr int TQ = (TG + NW - 1) / NW;
    constexpr int NV = TPW * C;
    constexpr int NF = (NV + 63) / 64;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *part = smem;                       // [NH][NSLICE][NV]
    float *cvs = part + NH * NSLICE * NV;     // [TG][C]
    float *cbl = cvs + ((TG * C + 3) & ~3);   // [kch][cp]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = wave & 3;
    const int half = wave >> 2;
    const int cp = GEN ? p.c : C;
    const int Pp = GEN ? p.dp / 256 : P;
    const int dp = GEN ? p.dp : 256 * P;
    auto el = [&](int i) { return elem_of(Pp, slice, i, lane); };   // this lane's element at step i
    const long long tok0 = (long long)blockIdx.x * TG;  // first token of the workgroup
    const long long tokw = tok0 + half * TPW;           // first token of this wave

    const bool shared_cb = p.hdr->shared_cb != 0;
    const int nchunk = (p.K + p.kch - 1) / p.kch;
    const bool stage_once = shared_cb && nchunk == 1;

    float r[P][TPW];
    float out[RECON ? P : 1][RECON ? TPW : 1];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int e = el(i);
            const long long tok = tokw + t;
            const bool ok = (!GEN || i < Pp) && e < p.d && tok < p.n_tokens;
            r[i][t] = ok ? load_x(p.x, p.x_dtype, (size_t)tok * p.d + e) : 0.0f;
            if constexpr (RECON) out[i][t] = 0.0f;
        }

    for (int l = 0; l < p.levels; ++l) {
        // ---- lin_in partials: acc[t*C + j] = sum_i r[i][t] * W_in[l][j][e_i]   (model.py:211)
        float acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0f;
        {
            const float *wl = p.w_in + (size_t)l * cp * dp;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (GEN && i >= Pp) continue;
                float w[C];
#pragma unroll
                for (int j = 0; j < C; ++j) w[j] = (!GEN || j < cp) ? wl[(size_t)j * dp + el(i)] : 0.0f;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (!GEN || j < cp) acc[t * C + j] = fmaf(r[i][t], w[j], acc[t * C + j]);
            }
        }
        // ---- 64-lane tree per value, results scattered over lanes, then to LDS
        {
            float fin[NF];
            ReduceScatter<NV, 0, NF>::run(acc, fin, lane);
            const int v0 = rs_value_of_lane(lane);
            float *dst = part + (half * NSLICE + slice) * NV;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (64 * f + v0 < NV) dst[64 * f + v0] = fin[f];
        }
        if (l == 0 ? true : !stage_once) {
            const int rows = p.kch < p.K ? p.kch : p.K;
            stage_floats(cbl, p.cb + (size_t)(shared_cb ? 0 : l) * p.K * cp, rows * cp);
        }
        __syncthreads();

        // ---- per token: q, ||q||, cosine argmax, STE value            (model.py:187-193, 218-220)
        {
            float q[TQ][C], xn[TQ][C], best[TQ];
            int bidx[TQ];
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const int tl = wave * TQ + tq;  // token inside the workgroup
                const int th = tl / TPW, tt = tl % TPW;
                const float *ps = part + (size_t)th * NSLICE * NV + tt * C;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if ((GEN && j >= cp) || tl >= TG) { q[tq][j] = 0.0f; continue; }
                    const float s01 = ps[0 * NV + j] + ps[1 * NV + j];
                    const float s23 = ps[2 * NV + j] + ps[3 * NV + j];
                    q[tq][j] = (s01 + s23) + p.b_in[l * cp + j];
                }
                const float nrm = aten_norm2<C>(q[tq], cp);
#pragma unroll
                for (int j = 0; j < C; ++j) xn[tq][j] = q[tq][j] / nrm;
                best[tq] = -INFINITY;
                bidx[tq] = 0;
            }
            for (int ch = 0; ch < nchunk; ++ch) {
                const int row_base = ch * p.kch;
                const int rows = (p.K - row_base) < p.kch ? (p.K - row_base) : p.kch;
                if (ch > 0) {
                    __syncthreads();
                    stage_floats(cbl, p.cb + ((size_t)(shared_cb ? 0 : l) * p.K + row_base) * cp, rows * cp);
                    __syncthreads();
                }
                if (wave * TQ < TG) scan_rows<C, TQ>(xn, cbl, rows, row_base, cp, lane, best, bidx);
            }
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const int tl = wave * TQ + tq;
                if (tl >= TG) continue;
                wave_argmax(best[tq], bidx[tq]);
                const float *row = p.cb + ((size_t)(shared_cb ? 0 : l) * p.K + bidx[tq]) * cp;
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < C; ++j)
                        if (!GEN || j < cp) {
                            const float diff = row[j] - q[tq][j];
                            cvs[tl * C + j] = q[tq][j] + diff;
                        }
                    const long long tok = tok0 + tl;
                    if (tok < p.n_tokens) p.idx[tok * p.levels + l] = bidx[tq];
                }
            }
        }
        __syncthreads();

        // ---- lin_out, residual and reconstruction update               (model.py:221-224)
        {
            float cv[TPW][C];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int k = 0; k < C; ++k) cv[t][k] = cvs[(half * TPW + t) * C + k];
            const float *wl = p.w_out + (size_t)l * cp * dp;
            const float *bl = p.b_out + (size_t)l * dp;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if (GEN && i >= Pp) continue;
                float w[C];
#pragma unroll
                for (int k = 0; k < C; ++k) w[k] = (!GEN || k < cp) ? wl[(size_t)k * dp + el(i)] : 0.0f;
                const float b = bl[el(i)];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    float y = 0.0f;
#pragma unroll
                    for (int k = 0; k < C; ++k)
                        if (!GEN || k < cp) y = fmaf(cv[t][k], w[k], y);
                    y = y + b;
                    r[i][t] = r[i][t] - y;
                    if constexpr (RECON) out[i][t] = out[i][t] + y;  // level 0: 0 + y (model.py:201)
                }
            }
        }
        // the next level's writes to `part` happen after this level's reads (barrier above);
        // its reads of `cvs` happen after the barrier that follows the next writes.
    }

    if constexpr (RECON) {
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int e = el(i);
                const long long tok = tokw + t;
                if ((!GEN || i < Pp) && e < p.d && tok < p.n_tokens) p.q_out[(size_t)tok * p.d + e] = out[i][t];
            }
    }
}

template <int P, int C, int TPW, int NH, bool GEN>
int launch_encode(const EncParams &p, hipStream_t stream)
{
    constexpr int TG = TPW * NH;
    constexpr int NV = TPW * C;
    EncParams q = p;
    const size_t fixed = sizeof(float) * (NH * NSLICE * NV + ((TG * C + 3) & ~3));
    int kch = (int)((LDS_CODEBOOK_BUDGET) / (sizeof(float) * p.c));
    kch = (kch / 64) * 64;
    if (kch >= p.K) kch = ((p.K + 63) / 64) * 64;
    q.kch = kch;
    const int rows = kch < p.K ? kch : p.K;
    const size_t lds = fixed + sizeof(float) * (size_t)rows * p.c;
    const long long nblk = (p.n_tokens + TG - 1) / TG;
    if (nblk > 0x7fffffffLL) return fail("encode: too many tokens for one launch");
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256 * NH), lds, stream, q);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (p.q_out) return go(rqae_encode_kernel<P, C, TPW, NH, true, GEN>);
    return go(rqae_encode_kernel<P, C, TPW, NH, false, GEN>);
}

enum Variant { V_AUTO = 0, V_A = 1, V_B = 2, V_GEN = 3, V_M = 4 };

Variant env_variant()
{
    const char *s = getenv("RQAE_HIP_VARIANT");
    if (!s || !*s) return V_AUTO;
    if (!strcmp(s, "A")) return V_A;
    if (!strcmp(s, "B")) return V_B;
    if (!strcmp(s, "GEN")) return V_GEN;
    if (!strcmp(s, "M")) return V_M;
    return V_AUTO;
}

// Which instantiation serves (P, c).  Returns a name; launches when p != nullptr.
const char *dispatch_encode(int P, int c, const EncParams *p, hipStream_t stream, int *rc)
{
    const Variant v = env_variant();
#ifdef RQAE_EXPERIMENT_ONLY_95   // compile-time experiments: one instantiation, fast rebuilds
    {
        if (p) *rc = launch_encode_lds<9, 5>(*p, stream, v == V_M);
        (void)v; (void)P; (void)c;
        return "experiment";
    }
#else
#define RUN(name, ...)                                  \
    do {                                                \
        if (p) *rc = launch_encode<__VA_ARGS__>(*p, stream); \
        return name;                                    \
    } while (0)
#define RUNF(name, ...)                                       \
    do {                                                      \
        if (p) *rc = launch_encode_lds<__VA_ARGS__>(*p, stream, v == V_M); \
        return name;                                          \
    } while (0)
    // M = the arm that puts lin_out and the cosine contraction on the matrix pipe (v_mfma_f32_4x4x1_16b).
    // Bit-identical output; measured 10-17 % slower than the packed-VALU arm on MI355X.
    if (v == V_AUTO || v == V_M) {
        const bool m = v == V_M;
        if (P == 9 && c == 5) RUNF(m ? "encode_lds<P=9,C=5> MFMA 4x4x1 arm" : "encode_lds<P=9,C=5> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 9, 5);
        if (P == 9 && c == 4) RUNF(m ? "encode_lds<P=9,C=4> MFMA 4x4x1 arm" : "encode_lds<P=9,C=4> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 9, 4);
        if (P == 1 && c == 4) RUNF(m ? "encode_lds<P=1,C=4> MFMA 4x4x1 arm" : "encode_lds<P=1,C=4> 8 waves x 8 tokens, LDS-DMA weights, two-sweep pass in packed fp32", 1, 4);
    }
#undef RUNF
    if (v != V_GEN) {
        if (P == 9 && c == 5) {
            if (v == V_B) RUN("encode<P=9,C=5,TPW=8,NH=2> 8 waves, 16 tokens/WG", 9, 5, 8, 2, false);
            RUN("encode<P=9,C=5,TPW=16,NH=1> 4 waves, 16 tokens/WG", 9, 5, 16, 1, false);
        }
        if (P == 9 && c == 4) RUN("encode<P=9,C=4,TPW=16,NH=1> 4 waves, 16 tokens/WG", 9, 4, 16, 1, false);
        if (P == 1 && c == 4) RUN("encode<P=1,C=4,TPW=16,NH=1> 4 waves, 16 tokens/WG", 1, 4, 16, 1, false);
        // (4 tokens per wave: with 8, residual + reconstruction alone are 224 registers and hipcc parks values in
        //  AGPRs at one wave per SIMD -- measured 3.39 vs 2.20 ms for 16 384 tokens, depth 4)
        if (P == 14 && c == 13) RUN("encode<P=14,C=13,TPW=4,NH=2> 8 waves, 8 tokens/WG", 14, 13, 4, 2, false);
    }
    if (P <= 8) RUN("encode<generic P<=8,C<=16,TPW=2>", 8, 16, 2, 1, true);
    RUN("encode<generic P<=32,C<=16,TPW=2>", 32, 16, 2, 1, true);
#undef RUN
#endif
}

}  // namespace

#include "decode.hpp"
#include "hook.hpp"
#include "intensity.hpp"
#include "tables.hpp"

namespace {

struct PackedView {
    const Header *hdr;
    const float *w_in, *w_out, *b_out, *b_in, *cb, *rec_p, *cb_img;
    int dp, cb_img_pieces;
};

PackedView view_of(const void *packed, int d, int c, int K, int L)
{
    const Layout lo = make_layout(d, c, K, L);
    const char *b = static_cast<const char *>(packed);
    PackedView v;
    v.hdr = reinterpret_cast<const Header *>(b);
    v.w_in = reinterpret_cast<const float *>(b + lo.w_in);
    v.w_out = reinterpret_cast<const float *>(b + lo.w_out);
    v.b_out = reinterpret_cast<const float *>(b + lo.b_out);
    v.b_in = reinterpret_cast<const float *>(b + lo.b_in);
    v.cb = reinterpret_cast<const float *>(b + lo.cb);
    v.cb_img = reinterpret_cast<const float *>(b + lo.cb_img);
    v.cb_img_pieces = lo.cb_img_pieces;
    v.rec_p = reinterpret_cast<const float *>(b + lo.rec_p);
    v.dp = lo.dp;
    return v;
}

int decode_common(const void *packed, int d, int c, int K, int L, const void *src, bool from_idx,
                  int idx_dtype, int64_t n_tokens, int nq, const uint8_t *level_mask, float *out,
                  int *status, void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("decode: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !src || !out) return fail("decode: null pointer");
    if (n_tokens < 0 || nq < 1) return fail("decode: bad sizes n_tokens=%lld nq=%d", (long long)n_tokens, nq);
    if (L > MAX_MASK_LEVELS) return fail("decode: L=%d exceeds %d", L, MAX_MASK_LEVELS);
    if (from_idx && (idx_dtype < RQAE_IDX_I64 || idx_dtype > RQAE_IDX_U16)) return fail("decode: bad idx_dtype %d", idx_dtype);
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    DecParams p;
    memset(&p, 0, sizeof(p));
    int used = 0, top = 0;
    for (int l = 0; l < L; ++l) {
        const bool on = level_mask ? level_mask[l] != 0 : l < nq;
        if (!on) continue;
        if (l >= nq) return fail("decode: level %d selected but indices hold only %d levels", l, nq);
        p.mask[l >> 5] |= 1u << (l & 31);
        ++used;
        top = l + 1;
    }
    if (!used) return fail("decode: no level selected");
    p.w_out = v.w_out; p.b_out = v.b_out; p.cb0 = v.cb; p.rec_p = v.rec_p;
    p.src = src; p.out = out; p.n_tokens = n_tokens;
    p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.nlev = top; p.nq = nq; p.idx_dtype = idx_dtype;
    p.aligned16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    p.status = status;
    return dispatch_decode(v.dp / 256, c, p, from_idx, static_cast<hipStream_t>(stream));
}

}  // namespace

// =================================================================== C ABI
#ifdef RQAE_TIMELINE
extern "C" int rqae_hip_debug_timeline(long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), sizeof(long long) * 4 * 8 * 64);
}
#endif

extern "C" {

int rqae_hip_abi_version(void) { return RQAE_HIP_ABI_VERSION; }

const char *rqae_hip_last_error(void) { return g_err; }

size_t rqae_hip_packed_bytes(int d, int c, int K, int L)
{
    if (!shape_ok(d, c, K, L)) {
        fail("unsupported shape d=%d c=%d K=%d L=%d (need 1<=d<=%d, 1<=c<=%d)", d, c, K, L, MAX_D, MAX_C);
        return 0;
    }
    return make_layout(d, c, K, L).total;
}

int rqae_hip_pack(void *packed, int d, int c, int K, int L, const float *w_in, const float *b_in,
                  const float *codebook, const float *w_out, const float *b_out, int codebook_shared,
                  void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("pack: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !w_in || !b_in || !codebook || !w_out || !b_out) return fail("pack: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Layout lo = make_layout(d, c, K, L);
    char *b = static_cast<char *>(packed);
    Header h = {MAGIC, d, c, K, L, lo.dp, codebook_shared ? 1 : 0, 0};
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<Header *>(b), h,
                       reinterpret_cast<float *>(b + lo.w_in), reinterpret_cast<float *>(b + lo.w_out),
                       reinterpret_cast<float *>(b + lo.b_out), w_in, w_out, b_out);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pack_records_kernel, dim3(2048), dim3(256), 0, s, h, lo.nq, rec_out_quads(lo.dp / 256, c),
                       reinterpret_cast<float *>(b + lo.rec_p), w_in, w_out, b_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b + lo.b_in, b_in, sizeof(float) * (size_t)L * c, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + lo.cb, codebook, sizeof(float) * (size_t)L * K * c, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(pack_cb_image_kernel, dim3(256), dim3(256), 0, s, c, K, lo.cb_img_pieces * 256,
                       reinterpret_cast<float *>(b + lo.cb_img), codebook);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_encode(const void *packed, int d, int c, int K, int L, const void *x, int x_dtype,
                    int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("encode: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!packed || !x || !indices) return fail("encode: null pointer");
    if (levels < 1 || levels > L) return fail("encode: levels=%d outside 1..%d", levels, L);
    if (x_dtype < RQAE_X_F32 || x_dtype > RQAE_X_F16) return fail("encode: bad x_dtype %d", x_dtype);
    if (n_tokens < 0) return fail("encode: negative n_tokens");
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    EncParams p;
    memset(&p, 0, sizeof(p));
    p.w_in = v.w_in; p.w_out = v.w_out; p.b_out = v.b_out; p.b_in = v.b_in; p.cb = v.cb; p.hdr = v.hdr;
    p.rec_p = v.rec_p; p.cb_img = v.cb_img; p.cb_img_pieces = v.cb_img_pieces;
    p.x = x; p.q_out = q_out; p.idx = reinterpret_cast<long long *>(indices);
    p.n_tokens = n_tokens; p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.levels = levels; p.x_dtype = x_dtype;
    p.aligned16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q_out)) & 15) == 0;
    int rc = 0;
    dispatch_encode(v.dp / 256, c, &p, static_cast<hipStream_t>(stream), &rc);
    return rc;
}

// shapes whose encode kernel can apply the Gemma norm on load (the LDS kernel's instantiations)
static bool gemma_fusable(int d, int c)
{
    const int P = (d + 255) / 256;
    return env_variant() == V_AUTO && d % 256 == 0 && ((P == 9 && (c == 5 || c == 4)) || (P == 1 && c == 4));
}

int rqae_hip_encode_gemma_supported(int d, int c, int K, int L)
{
    return shape_ok(d, c, K, L) && gemma_fusable(d, c) ? 1 : 0;
}

int rqae_hip_encode_gemma(const void *packed, int d, int c, int K, int L, const void *hs, int hs_dtype,
                          const float *norm_weight, float eps, int64_t n_tokens, int levels, float *q_out,
                          int64_t *indices, float *inv_rms, void *stream)
{
    if (!shape_ok(d, c, K, L)) return fail("encode_gemma: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if (!gemma_fusable(d, c)) return fail("encode_gemma: no fused kernel for d=%d c=%d (see rqae_hip_encode_gemma_supported)", d, c);
    if (!packed || !hs || !indices || !norm_weight || !inv_rms || !q_out) return fail("encode_gemma: null pointer");
    if (levels < 1 || levels > L) return fail("encode_gemma: levels=%d outside 1..%d", levels, L);
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("encode_gemma: bad hs_dtype %d", hs_dtype);
    if (n_tokens < 0) return fail("encode_gemma: negative n_tokens");
    if (n_tokens == 0) return 0;
    const PackedView v = view_of(packed, d, c, K, L);
    EncParams p;
    memset(&p, 0, sizeof(p));
    p.w_in = v.w_in; p.w_out = v.w_out; p.b_out = v.b_out; p.b_in = v.b_in; p.cb = v.cb; p.hdr = v.hdr;
    p.rec_p = v.rec_p; p.cb_img = v.cb_img; p.cb_img_pieces = v.cb_img_pieces;
    p.x = hs; p.q_out = q_out; p.idx = reinterpret_cast<long long *>(indices);
    p.n_tokens = n_tokens; p.d = d; p.dp = v.dp; p.c = c; p.K = K; p.levels = levels; p.x_dtype = hs_dtype;
    p.norm_w = norm_weight; p.inv_rms = inv_rms; p.norm_eps = eps;
    p.aligned16 = ((reinterpret_cast<uintptr_t>(hs) | reinterpret_cast<uintptr_t>(q_out) |
                    reinterpret_cast<uintptr_t>(norm_weight)) & 15) == 0;
    int rc = 0;
    dispatch_encode(v.dp / 256, c, &p, static_cast<hipStream_t>(stream), &rc);
    return rc;
}

int rqae_hip_kernel_info(int d, int c, int K, int L, char *buf, size_t buflen)
{
    if (!shape_ok(d, c, K, L)) return fail("kernel_info: unsupported shape");
    if (!buf || !buflen) return fail("kernel_info: null buffer");
    int rc = 0;
    const char *name = dispatch_encode(((d + 255) / 256), c, nullptr, nullptr, &rc);
    snprintf(buf, buflen, "%s", name);
    return 0;
}

int rqae_hip_quantize(const float *pin, int64_t n, int c, int K, const float *codebook_l, float *cv,
                      int64_t *idx, void *stream)
{
    if (c < 1 || c > MAX_C || K < 1) return fail("quantize: unsupported c=%d K=%d", c, K);
    if (!pin || !codebook_l || !cv || !idx) return fail("quantize: null pointer");
    if (n < 0) return fail("quantize: negative n");
    if (n == 0) return 0;
    QuantParams p;
    p.p = pin; p.cb = codebook_l; p.cv = cv; p.idx = reinterpret_cast<long long *>(idx);
    p.n = n; p.c = c; p.K = K;
    int kch = (int)(LDS_CODEBOOK_BUDGET / (sizeof(float) * c));
    kch = (kch / 64) * 64;
    if (kch >= K) kch = ((K + 63) / 64) * 64;
    p.kch = kch;
    const int rows = kch < K ? kch : K;
    const size_t lds = sizeof(float) * (size_t)rows * c;
    // token groups per workgroup: enough workgroups to fill the chip several times over, at most 16 groups each
    int iters = 1;
    if (kch >= K) {
        const long long groups = (n + 15) / 16;
        while (iters < 16 && groups / (iters * 2) >= 256 * 8) iters *= 2;
    }
    p.iters = iters;
    const long long nblk = (n + 16LL * iters - 1) / (16LL * iters);
    if (nblk > 0x7fffffffLL) return fail("quantize: too many rows for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (c <= 4) return go(rqae_quantize_kernel<4>);
    if (c == 5) return go(rqae_quantize_kernel<5>);
    if (c <= 8) return go(rqae_quantize_kernel<8>);
    return go(rqae_quantize_kernel<16>);
}

int rqae_hip_decode(const void *packed, int d, int c, int K, int L, const void *indices, int idx_dtype,
                    int64_t n_tokens, int idx_levels, const uint8_t *level_mask, float *out, int32_t *status, void *stream)
{
    return decode_common(packed, d, c, K, L, indices, true, idx_dtype, n_tokens, idx_levels, level_mask, out, status, stream);
}

int rqae_hip_decode_values(const void *packed, int d, int c, int K, int L, const float *cv, int64_t n_tokens,
                           int nq, const uint8_t *level_mask, float *out, void *stream)
{
    return decode_common(packed, d, c, K, L, cv, false, 0, n_tokens, nq, level_mask, out, nullptr, stream);
}

int rqae_hip_gemma_norm(const void *hs, int hs_dtype, const float *norm_weight, float eps, int64_t n_tokens, int d,
                        float *normed, float *inv_rms, void *stream)
{
    if (!hs || !norm_weight || !normed || !inv_rms) return fail("gemma_norm: null pointer");
    if (n_tokens < 0 || d < 1) return fail("gemma_norm: bad sizes");
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("gemma_norm: bad hs_dtype %d", hs_dtype);
    if (n_tokens == 0) return 0;
    const long long nblk = (n_tokens + 3) / 4;
    if (nblk > 0x7fffffffLL) return fail("gemma_norm: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t hsz = hs_dtype == RQAE_X_F32 ? 4 : 2;
    const int P = d / 256;
    const bool vec = d % 256 == 0 && (P == 1 || P == 4 || P == 9 || P == 14) &&
                     (reinterpret_cast<uintptr_t>(hs) % (4 * hsz)) == 0 && (reinterpret_cast<uintptr_t>(normed) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(norm_weight) & 15) == 0;
    if (vec) {
#define NORM_VEC(HT, PP) hipLaunchKernelGGL((gemma_norm_vec_kernel<HT, PP>), dim3((unsigned)nblk), dim3(256), 0, s, \
                                            (const HT *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms)
#define NORM_VEC_D(HT) do { if (P == 1) NORM_VEC(HT, 1); else if (P == 4) NORM_VEC(HT, 4); else if (P == 9) NORM_VEC(HT, 9); else NORM_VEC(HT, 14); } while (0)
        if (hs_dtype == RQAE_X_F16) NORM_VEC_D(_Float16);
        else if (hs_dtype == RQAE_X_BF16) NORM_VEC_D(uint16_t);
        else NORM_VEC_D(float);
#undef NORM_VEC_D
#undef NORM_VEC
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (hs_dtype == RQAE_X_F16)
        hipLaunchKernelGGL(gemma_norm_kernel<_Float16>, dim3((unsigned)nblk), dim3(256), 0, s, (const _Float16 *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    else if (hs_dtype == RQAE_X_BF16)
        hipLaunchKernelGGL(gemma_norm_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), 0, s, (const uint16_t *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    else
        hipLaunchKernelGGL(gemma_norm_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, (const float *)hs, norm_weight, eps, (long long)n_tokens, d, normed, inv_rms);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_gemma_denorm_patch(const float *q_out, const float *norm_weight, const float *inv_rms, int64_t n_tokens,
                                int d, int seq_len, int skip_bos, void *hs_inout, int hs_dtype, float *new_out,
                                void *stream)
{
    if (!q_out || !norm_weight || !inv_rms || !hs_inout) return fail("gemma_denorm_patch: null pointer");
    if (n_tokens < 0 || d < 1 || seq_len < 1) return fail("gemma_denorm_patch: bad sizes");
    if (hs_dtype < RQAE_X_F32 || hs_dtype > RQAE_X_F16) return fail("gemma_denorm_patch: bad hs_dtype %d", hs_dtype);
    if (n_tokens == 0) return 0;
    const long long nblk = (n_tokens + 3) / 4;
    if (nblk > 0x7fffffffLL) return fail("gemma_denorm_patch: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t hsz = hs_dtype == RQAE_X_F32 ? 4 : 2;
    const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(hs_inout) % (4 * hsz)) == 0 &&
                     (reinterpret_cast<uintptr_t>(q_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(norm_weight) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(new_out) & 15) == 0;
    if (vec) {
#define PATCH_VEC(HT) hipLaunchKernelGGL(gemma_denorm_patch_vec_kernel<HT>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, \
                                         norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (HT *)hs_inout, new_out)
        if (hs_dtype == RQAE_X_F16) PATCH_VEC(_Float16);
        else if (hs_dtype == RQAE_X_BF16) PATCH_VEC(uint16_t);
        else PATCH_VEC(float);
#undef PATCH_VEC
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (hs_dtype == RQAE_X_F16)
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<_Float16>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (_Float16 *)hs_inout, new_out);
    else if (hs_dtype == RQAE_X_BF16)
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (uint16_t *)hs_inout, new_out);
    else
        hipLaunchKernelGGL(gemma_denorm_patch_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, q_out, norm_weight, inv_rms, (long long)n_tokens, d, seq_len, skip_bos, (float *)hs_inout, new_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_intensity_prepare(const void *codebook_sims, const int *center, const void *layer_weights, int M, int K,
                               void *table, void *wcum, void *stream)
{
    if (!codebook_sims || !center || !layer_weights || !table || !wcum) return fail("intensity_prepare: null pointer");
    if (M < 1 || K < 1) return fail("intensity_prepare: bad sizes M=%d K=%d", M, K);
    hipLaunchKernelGGL(intensity_prepare_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const _Float16 *)codebook_sims, center, (const _Float16 *)layer_weights, M, K,
                       (_Float16 *)table, (_Float16 *)wcum);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_intensity(const void *table, const void *layer_weights, const void *wcum, int M, int K,
                       const void *indices, int idx_dtype, int64_t n_tokens, int nq, const int *layers_dev,
                       int n_layers, void *out, int32_t *status, void *stream)
{
    if (!table || !layer_weights || !wcum || !indices || !layers_dev || !out) return fail("intensity: null pointer");
    if (M < 1 || K < 1 || nq < M || n_layers < 1) return fail("intensity: bad sizes M=%d K=%d nq=%d n_layers=%d", M, K, nq, n_layers);
    if (idx_dtype < RQAE_IDX_I64 || idx_dtype > RQAE_IDX_U16) return fail("intensity: bad idx_dtype %d", idx_dtype);
    if (n_tokens < 0) return fail("intensity: negative n_tokens");
    if (n_tokens == 0) return 0;
    IntensityParams p;
    p.table = (const _Float16 *)table; p.w = (const _Float16 *)layer_weights; p.wcum = (const _Float16 *)wcum;
    p.idx = indices; p.layers = layers_dev; p.out = (_Float16 *)out;
    p.n_tokens = n_tokens; p.nq = nq; p.M = M; p.K = K; p.n_layers = n_layers; p.status = status;
    const size_t es = idx_dtype == RQAE_IDX_I64 ? 8 : idx_dtype == RQAE_IDX_I32 ? 4 : 2;
    const int TL = 16;
    int LC = TL;
    while (LC > 1 && (size_t)LC * K * 2 > 48 * 1024) LC >>= 1;
    if ((size_t)LC * K * 2 > 48 * 1024) return fail("intensity: K=%d too large for the LDS table", K);
    if (((size_t)LC * K * 2) % 16 != 0) return fail("intensity: K=%d needs LC*K*2 %% 16 == 0 (LC=%d)", K, LC);
    p.LC = LC;
    p.vec_ok = (reinterpret_cast<uintptr_t>(indices) & 15) == 0 && ((size_t)nq * es) % 16 == 0 &&
               (reinterpret_cast<uintptr_t>(table) & 15) == 0;
    if ((reinterpret_cast<uintptr_t>(table) & 15) != 0) return fail("intensity: table must be 16-byte aligned");
    const int RS = (int)(TL / (16 / es)) + 1;
    const size_t lds = (((size_t)LC * K * 2 + 15) & ~(size_t)15) + 64 * 4 + 16 + (size_t)4 * 64 * RS * 16;
    const long long nblk = (n_tokens + 255) / 256;
    if (nblk > 0x7fffffffLL) return fail("intensity: too many tokens for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, p);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if (idx_dtype == RQAE_IDX_I64) return go(intensity_kernel<long long>);
    if (idx_dtype == RQAE_IDX_I32) return go(intensity_kernel<int>);
    return go(intensity_kernel<uint16_t>);
}

int rqae_hip_codebook_sims(const float *codebook0, int K, int c, void *out, void *stream)
{
    if (!codebook0 || !out) return fail("codebook_sims: null pointer");
    if (K < 1 || c < 1 || c > MAX_C) return fail("codebook_sims: bad sizes K=%d c=%d (need 1<=c<=%d)", K, c, MAX_C);
    const dim3 grid((unsigned)((K + 255) / 256), (unsigned)((K + 31) / 32));
    if (grid.y > 65535u) return fail("codebook_sims: K=%d too large for one launch", K);
    hipLaunchKernelGGL(codebook_sims_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), codebook0, K, c,
                       (_Float16 *)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rqae_hip_subfeatures(const float *codebook, const float *w_out, const float *b_out, int L, int K, int c, int d,
                         float *out, void *stream)
{
    if (!codebook || !w_out || !b_out || !out) return fail("subfeatures: null pointer");
    if (!shape_ok(d, c, K, L) || L > 65535) return fail("subfeatures: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int RPB = 64;
    const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int per_wave = vec ? 256 : 64;
    const dim3 grid((unsigned)((d + per_wave - 1) / per_wave), (unsigned)((K + RPB - 1) / RPB), (unsigned)L);
    if (grid.y > 65535u) return fail("subfeatures: K=%d too large for one launch", K);
#define SUBF(CB, V) hipLaunchKernelGGL((subfeatures_kernel<CB, V>), grid, dim3(256), 0, s, codebook, w_out, b_out, K, c, d, RPB, out)
    if (vec) {
        if (c <= 5) SUBF(5, 4);
        else if (c <= 8) SUBF(8, 4);
        else SUBF(MAX_C, 4);
    } else {
        if (c <= 8) SUBF(8, 1);
        else SUBF(MAX_C, 1);
    }
#undef SUBF
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t rqae_hip_tables_scratch_bytes(int L, int K, int c)
{
    if (L < 1 || K < 1 || c < 1 || c > MAX_C) {
        fail("tables_scratch_bytes: bad sizes L=%d K=%d c=%d", L, K, c);
        return 0;
    }
    return (size_t)L * tables_level_doubles(K, c) * sizeof(double);
}

int rqae_hip_subfeature_sims(const float *codebook, const float *w_out, const float *b_out, int L, int K, int c, int d,
                             void *scratch, void *sims_out, float *layer_norms_out, int scale_by_layer_norms, void *stream)
{
    const int scaled = scale_by_layer_norms ? 1 : 0;
    if (!codebook || !w_out || !b_out || !scratch) return fail("subfeature_sims: null pointer");
    if (!sims_out && !layer_norms_out) return fail("subfeature_sims: nothing to compute");
    if (!shape_ok(d, c, K, L) || L > 65535) return fail("subfeature_sims: unsupported shape d=%d c=%d K=%d L=%d", d, c, K, L);
    if ((reinterpret_cast<uintptr_t>(scratch) & 7) != 0) return fail("subfeature_sims: scratch must be 8-byte aligned");
    if (sims_out && (reinterpret_cast<uintptr_t>(sims_out) & 15) != 0) return fail("subfeature_sims: sims_out must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(tables_prep_kernel, dim3((unsigned)L, (unsigned)((K + 1023) / 1024)), dim3(256), 0, s, codebook, w_out, b_out, K, c, d,
                       (double *)scratch, layer_norms_out);
    HIP_TRY(hipGetLastError());
    if (!sims_out) return 0;
    _Float16 *sims = (_Float16 *)sims_out;
    const double *scr = (const double *)scratch;
    if (K < 64) {
        hipLaunchKernelGGL(subfeature_sims_small_kernel, dim3((unsigned)((K * K + 255) / 256), (unsigned)L), dim3(256), 0, s,
                           codebook, scr, K, c, scaled, sims);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int MB = c + 1 <= 5 ? 5 : c + 1 <= 6 ? 6 : c + 1 <= 9 ? 9 : c + 1 <= 14 ? 14 : MAX_C + 1;
    const int NC = MB <= 9 ? 8 : 4;                 // elements per piece: NC * MB doubles of registers per lane
    int g = K % NC, h = NC;                         // PR = NC / gcd(K, NC): rows between equal alignments
    while (g) { const int t = h % g; h = g; g = t; }
    const int PR = NC / h;
    const int maxnp = (K + 2 * NC - 2) / NC;
    int TR = MB <= 6 ? 256 : 128;                   // rows per block: fewer when that leaves the chip short of blocks
    while (TR > 32 && (long long)L * ((K + TR - 1) / TR) * ((maxnp + 255) / 256) < 2048) TR >>= 1;
    const int ncol = K < NC * 257 ? K : NC * 257;   // columns a block stages in LDS
    const size_t plane = (size_t)ncol + ncol / NC + 1;   // padded column count (tables.hpp)
    const size_t lds = ((size_t)(TR + 1) * MB + plane) * sizeof(double) + plane * c * sizeof(float);
    const dim3 grid((unsigned)((K + TR - 1) / TR), (unsigned)L, (unsigned)((maxnp + 255) / 256));
    if (grid.z > 65535u) return fail("subfeature_sims: K=%d too large for one launch", K);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, codebook, scr, K, c, TR, PR, ncol, scaled, sims);
        return 0;
    };
    switch (MB) {
    case 5: if (go(subfeature_sims_kernel<5, 8>)) return 1; break;
    case 6: if (go(subfeature_sims_kernel<6, 8>)) return 1; break;
    case 9: if (go(subfeature_sims_kernel<9, 8>)) return 1; break;
    case 14: if (go(subfeature_sims_kernel<14, 4>)) return 1; break;
    default: if (go(subfeature_sims_kernel<MAX_C + 1, 4>)) return 1; break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
