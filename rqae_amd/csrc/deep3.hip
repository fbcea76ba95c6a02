// deep3.hip -- translation unit of librqae_hip.so: the matrix-core arm beyond 64 levels (gram_deep3.hpp; the released model's 1024
// levels, rqae/model.py:18-24).  gram.hip drives it through deep3.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "deep3.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];

namespace {

int d3fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int d3fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define D3_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return d3fail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#include "gram_small.hpp"
#include "gram_pieces.hpp"
#include "gram_deep3.hpp"

}  // namespace

namespace rqae_d3 {

size_t g3_bytes(int L) { return (size_t)(L * 4 / 32) * (L * 4 / 16) * 2048; }

int pack_g3(void *img, unsigned *maxbits_dev, const float *gneg, int L, int GB, hipStream_t s)
{
    hipLaunchKernelGGL(gram_pack_absmax_kernel, dim3(1024), dim3(256), 0, s, gneg, (size_t)L * L * GB, maxbits_dev);
    D3_TRY(hipGetLastError());
    hipLaunchKernelGGL(deep_pack_g3_kernel, dim3(4096), dim3(256), 0, s, gneg, L, GB, maxbits_dev, static_cast<uint16_t *>(img));
    D3_TRY(hipGetLastError());
    return 0;
}

size_t pieces_bytes(long long tok_pad, int L) { return (size_t)(tok_pad / 32) * (L * 4 / 16) * 2048; }

int convert(const float *planes, long long tok_pad, int nlev, int l0, void *pieces, int L, hipStream_t s)
{
    if (nlev % 4 || l0 % 4 || tok_pad % 32) return d3fail("deep convert: levels %d + %d, tok_pad %lld", l0, nlev, tok_pad);
    const long long work = (tok_pad / 32) * (nlev / 4);
    long long nb = (work + 3) / 4;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(deep_pieces_kernel, dim3((unsigned)nb), dim3(256), 0, s, planes, tok_pad, nlev, static_cast<uint16_t *>(pieces), L * 4 / 16, l0 / 4);
    D3_TRY(hipGetLastError());
    return 0;
}

static int launch(int mode, const Gemm &g, int ncu, hipStream_t s)
{
    if (g.ntiles % 4 || g.nks < 1) return d3fail("deep gemm: %d output tiles, %d k-steps", g.ntiles, g.nks);
    const int ntb = g.ntiles % 8 == 0 ? 8 : 4;      // eight tiles per pass wherever the columns allow (gram_deep3.hpp: operand traffic per matrix instruction)
    Deep3Params p;
    p.apieces = static_cast<const uint16_t *>(g.apieces); p.wimg = static_cast<const uint16_t *>(g.wimg); p.maxbits = g.maxbits;
    p.P = g.P; p.bias = g.bias; p.out = g.out; p.n_tokens = g.n_tokens; p.tok_pad = g.tok_pad; p.KSA = g.KSA; p.KSW = g.KSW; p.ks0 = g.ks0; p.nks = g.nks;
    p.nt0 = g.nt0; p.npass = g.ntiles / ntb; p.d = g.d;
    const long long items = ((g.n_tokens + D3_TILE - 1) / D3_TILE) * p.npass;
    long long nb = items < ncu ? items : ncu;
    const size_t lds = d3_lds_bytes(mode, g.d, ntb);
    if (lds > 160 * 1024) return d3fail("deep gemm: d=%d does not fit LDS", g.d);
    auto go = [&](auto kern) -> int {
        D3_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(D3_NW * 64), lds, s, p);
        D3_TRY(hipGetLastError());
        return 0;
    };
    if (ntb == 8) return mode == 0 ? go(deep_gemm_kernel<0, 8>) : go(deep_gemm_kernel<1, 8>);
    return mode == 0 ? go(deep_gemm_kernel<0, 4>) : go(deep_gemm_kernel<1, 4>);
}

int corrections(const Gemm &g, int ncu, hipStream_t s) { return launch(0, g, ncu, s); }
int reconstruct(const Gemm &g, int ncu, hipStream_t s) { return launch(1, g, ncu, s); }

}  // namespace rqae_d3
