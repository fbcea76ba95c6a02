// hook.hpp -- the steps either side of forward() inside RQAE.hook for a Gemma-2 LLM: norm and denorm+patch kernels; included by rqae_hip.hip.
#pragma once
#include "rqae_common.hpp"

namespace {

// =============================================================== hook pre/post
// The step either side of forward() in RQAE.hook (rqae/model.py:276-289) for a Gemma-2 LLM
// (rqae/llm.py:65-73):  normed = (hs * rsqrt(mean(hs^2) + eps)) * (1 + w)   and afterwards
// new = (q / (1 + w)) / rsqrt(mean(hs^2) + eps), new[:, 0] = hs[:, 0], written back in the LLM's dtype.
// One wave per token.  Order of the mean: lane owns elements 4*(lane + 64*j) + {0..3}, j ascending,
// fma chain of squares from +0 per lane, lanes summed by the tree of section 3, divided by d;
// rs = 1 / sqrt(ms + eps) with IEEE sqrt and divide (oracle/oracle.py:gemma_norm_canonical restates it).
// Sum of the four slice totals of every lane's value (in: per-lane partial of slice s in ss[s]) in the order of
// section 3: the 64-lane tree per slice, then (s0+s1)+(s2+s3).  Every lane returns the row total.
__device__ __forceinline__ float canonical_row_total(const float (&ss)[NSLICE], int lane)
{
    float fin[1];
    ReduceScatter<NSLICE, 0, 1>::run(ss, fin, lane);      // lane holds the tree sum of slice (bit5 + 2*bit4)
    const float s0 = lane_bcast(fin[0], 0), s1 = lane_bcast(fin[0], 32);
    const float s2 = lane_bcast(fin[0], 16), s3 = lane_bcast(fin[0], 48);
    return (s0 + s1) + (s2 + s3);
}

// One wave per token, any d (runtime P = ceil(d / 256)), scalar loads.
template <typename HT>
__global__ void __launch_bounds__(256)
gemma_norm_kernel(const HT *hs, const float *w, float eps, long long n_tokens, int d, float *normed, float *rs_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const int P = (d + 255) / 256;
    const HT *row = hs + (size_t)tok * d;
    float ss[NSLICE];
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        float a = 0.0f;
        for (int i = 0; i < P; ++i) {
            const int e = elem_of(P, s, i, lane);
            const float x = e < d ? widen<HT>(row[e]) : 0.0f;
            a = fmaf(x, x, a);
        }
        ss[s] = a;
    }
    const float total = canonical_row_total(ss, lane);
    const float ms = total / (float)d;
    const float rs = 1.0f / __builtin_sqrtf(ms + eps);
    if (lane == 0) rs_out[tok] = rs;
    float *orow = normed + (size_t)tok * d;
    for (int e = lane; e < d; e += 64) {
        const float t = widen<HT>(row[e]) * rs;
        orow[e] = t * (1.0f + w[e]);
    }
}

// Vector form for d == 256 * P with 16-byte aligned rows: hs is read ONCE with the tile loads of the encode
// kernel (runs of 4 / 2 / 1 consecutive elements per lane), kept in registers across the reduction, and normed
// is written with the matching vector stores.  Same arithmetic and order as gemma_norm_kernel.
template <typename HT, int P>
__global__ void __launch_bounds__(256)
gemma_norm_vec_kernel(const HT *hs, const float *w, float eps, long long n_tokens, int d, float *normed, float *rs_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    float x[NSLICE][P][1];
    float ss[NSLICE];
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        load_tile<P, 1, HT, true>(x[s], hs, tok, n_tokens, d, s, lane);
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < P; ++i) a = fmaf(x[s][i][0], x[s][i][0], a);
        ss[s] = a;
    }
    const float total = canonical_row_total(ss, lane);
    const float ms = total / (float)d;
    const float rs = 1.0f / __builtin_sqrtf(ms + eps);
    if (lane == 0) rs_out[tok] = rs;
    float *orow = normed + (size_t)tok * d;
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
        float w1[P][1];
        load_tile<P, 1, float, true>(w1, w, 0, 1, d, s, lane);
#pragma unroll
        for (int i = 0; i < P; ++i)
            store_run_ending_at<P>(orow, s, lane, i, [&](int k) { return (x[s][k][0] * rs) * (1.0f + w1[k][0]); });
    }
}

template <typename HT>
__device__ __forceinline__ HT narrow(float v);
template <>
__device__ __forceinline__ _Float16 narrow<_Float16>(float v) { return (_Float16)v; }
template <>
__device__ __forceinline__ float narrow<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint16_t narrow<uint16_t>(float v)   // round-to-nearest-even bf16 (as torch .to(bfloat16))
{
    return __builtin_bit_cast(uint16_t, (__bf16)v);
}

template <typename HT>
__global__ void __launch_bounds__(256)
gemma_denorm_patch_kernel(const float *q, const float *w, const float *rs, long long n_tokens, int d, int seq_len,
                          int skip_bos, HT *hs_inout, float *new_out)
{
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const bool bos = skip_bos && (tok % seq_len) == 0;
    const float r = rs[tok];
    const float *qrow = q + (size_t)tok * d;
    HT *hrow = hs_inout + (size_t)tok * d;
    float *nrow = new_out ? new_out + (size_t)tok * d : nullptr;
    for (int e = lane; e < d; e += 64) {
        float v;
        if (bos) v = widen<HT>(hrow[e]);                       // q_out[:, 0] = hs[:, 0]  (model.py:285-286)
        else {
            const float t = qrow[e] / (1.0f + w[e]);           // rqae/llm.py:69
            v = t / r;                                          // rqae/llm.py:70-72
        }
        if (nrow) nrow[e] = v;
        if (!bos) hrow[e] = narrow<HT>(v);                      // output[0].data.copy_(q_out)  (model.py:289)
    }
}

// Vector form of the patch kernel (d % 4 == 0, 16-byte aligned rows): float4 in, 4 elements of the LLM's
// dtype out per lane and step.  Same arithmetic as gemma_denorm_patch_kernel.
template <typename HT>
__global__ void __launch_bounds__(256)
gemma_denorm_patch_vec_kernel(const float *q, const float *w, const float *rs, long long n_tokens, int d, int seq_len,
                              int skip_bos, HT *hs_inout, float *new_out)
{
    typedef typename VecOf<HT, 4>::type H4;
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= n_tokens) return;
    const bool bos = skip_bos && (tok % seq_len) == 0;      // wave-uniform
    const float r = rs[tok];
    const float *qrow = q + (size_t)tok * d;
    HT *hrow = hs_inout + (size_t)tok * d;
    float *nrow = new_out ? new_out + (size_t)tok * d : nullptr;
    if (bos) {                                               // q_out[:, 0] = hs[:, 0]  (model.py:285-286): hs stays
        if (nrow)
            for (int e0 = 4 * lane; e0 < d; e0 += 256) {
                const H4 h = *reinterpret_cast<const H4 *>(hrow + e0);
                *reinterpret_cast<f4 *>(nrow + e0) = (f4){widen<HT>(h.x), widen<HT>(h.y), widen<HT>(h.z), widen<HT>(h.w)};
            }
        return;
    }
    for (int e0 = 4 * lane; e0 < d; e0 += 256) {
        const f4 qv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(qrow + e0));
        const f4 wv = *reinterpret_cast<const f4 *>(w + e0);
        f4 v;
        v.x = (qv.x / (1.0f + wv.x)) / r; v.y = (qv.y / (1.0f + wv.y)) / r;   // rqae/llm.py:69-72
        v.z = (qv.z / (1.0f + wv.z)) / r; v.w = (qv.w / (1.0f + wv.w)) / r;
        if (nrow) *reinterpret_cast<f4 *>(nrow + e0) = v;
        H4 o;
        o.x = narrow<HT>(v.x); o.y = narrow<HT>(v.y); o.z = narrow<HT>(v.z); o.w = narrow<HT>(v.w);
        *reinterpret_cast<H4 *>(hrow + e0) = o;             // output[0].data.copy_(q_out)  (model.py:289)
    }
}

}  // namespace
