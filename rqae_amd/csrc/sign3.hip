// sign3.hip -- translation unit of librqae_hip.so: the level chain of a two-point FSQ lattice on the matrix cores (gram_sign3.hpp;
// BASELINE configs[3]).  gram.hip drives it through sign3.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "sign3.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];

namespace {

int s3fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int s3fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define S3_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return s3fail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#include "gram_small.hpp"
#include "gram_pieces.hpp"
#include "gram_sign3.hpp"

__global__ void sign_set_word_kernel(int *w, int v, unsigned *bits) { *w = v; *bits = 0u; }

}  // namespace

namespace rqae_s3 {

bool supported(int c, int L, int shared_cb) { return c == 13 && L == S3G<13>::L && shared_cb != 0; }

size_t image_bytes(int c, int L) { return supported(c, L, 1) ? (size_t)S3G<13>::NITEM * 2048 : 0; }

int pack(void *img, unsigned *maxbits_dev, int *exact_dev, const float *gneg, const float *codebook, int c, int L, int K, int GB, hipStream_t s)
{
    if (!supported(c, L, 1)) return s3fail("sign chain: c=%d L=%d", c, L);
    hipLaunchKernelGGL(sign_set_word_kernel, dim3(1), dim3(1), 0, s, exact_dev, K == (1 << c) ? 1 : 0, maxbits_dev);
    S3_TRY(hipGetLastError());
    if (K == (1 << c)) {
        hipLaunchKernelGGL(sign_pack_exact_kernel<13>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, codebook, K, exact_dev);
        S3_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(gram_pack_absmax_kernel, dim3(256), dim3(256), 0, s, gneg, (size_t)L * L * GB, maxbits_dev);
    S3_TRY(hipGetLastError());
    hipLaunchKernelGGL(sign_pack_image_kernel<13>, dim3(512), dim3(256), 0, s, gneg, GB, maxbits_dev, static_cast<uint16_t *>(img));
    S3_TRY(hipGetLastError());
    return 0;
}

int launch(const Args &a, int ncu, hipStream_t s)
{
    if (!supported(a.c, a.L, 1)) return s3fail("sign chain: c=%d L=%d", a.c, a.L);
    if (a.tok_pad < a.n_tokens) return s3fail("sign chain: tok_pad %lld < %lld tokens", a.tok_pad, a.n_tokens);
    Sign3Params p;
    p.P = a.P; p.kconst = a.kconst; p.img = static_cast<const uint16_t *>(a.img); p.maxbits = a.maxbits; p.lat_flag = a.lat_flag; p.exact_flag = a.exact_flag;
    p.cb = a.cb; p.planes = a.planes; p.idx = a.idx; p.n_tokens = a.n_tokens; p.tok_pad = a.tok_pad; p.K = a.K; p.idx_stride = a.idx_stride;
    const long long ntile = (a.n_tokens + S3_TILE - 1) / S3_TILE;
    const long long nb = ntile < ncu ? ntile : ncu;
    const size_t lds = (size_t)S3G<13>::NITEM * 2048 + (size_t)S3G<13>::NCOL * sizeof(float);
    S3_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sign_chain_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sign_chain_kernel<13>, dim3((unsigned)nb), dim3(S3_NW * 64), lds, s, p);
    S3_TRY(hipGetLastError());
    return 0;
}

}  // namespace rqae_s3
