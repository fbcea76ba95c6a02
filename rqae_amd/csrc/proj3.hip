// proj3.hip -- translation unit of librqae_hip.so: the matrix-core arm's all-levels projection on fp16 two-piece operands
// (gram_proj3.hpp; rqae/model.py:211 for every level at once).  gram.hip calls it through proj3.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../include/rqae_hip.h"
#include "proj3.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];

namespace {

int p3fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int p3fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define P3_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return p3fail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#include "gram_small.hpp"
#include "gram_pieces.hpp"
#include "gram_proj3.hpp"

}  // namespace

namespace rqae_p3 {

size_t image_bytes(int d, int nog) { return (size_t)(d / 16) * out_tiles(nog) * 2048; }

int pack(void *img, unsigned *maxbits_dev, int d, int c, int L, int CP, int GL, int GP, int nog, const float *w_in, const float *in_scale,
         hipStream_t s)
{
    hipLaunchKernelGGL(gram_pack_absmax_in_kernel, dim3(512), dim3(256), 0, s, w_in, in_scale, (size_t)L * c * d, d, maxbits_dev);
    P3_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_win3_kernel, dim3(2048), dim3(256), 0, s, d, c, L, CP, GL, GP, out_tiles(nog), w_in, in_scale, maxbits_dev,
                       static_cast<uint16_t *>(img));
    P3_TRY(hipGetLastError());
    return 0;
}

bool supported(int d, int nog, int x_dtype, bool norm)
{
    const int ot = out_tiles(nog);
    if (d % 128 != 0 || ot < 1 || (ot > 8 && ot % 8 != 0)) return false;      // beyond 8 tiles: whole column blocks of 8
    if (norm && x_dtype == RQAE_X_F32) return false;
    return true;
}

int launch(const Args &a, int ncu, hipStream_t s)
{
    if (!supported(a.d, a.nog, a.x_dtype, a.inv_rms != nullptr)) return p3fail("proj3: no kernel for d=%d groups=%d x_dtype=%d", a.d, a.nog, a.x_dtype);
    Proj3Params pp;
    pp.x = a.x; pp.wimg = static_cast<const uint16_t *>(a.img); pp.maxbits = a.maxbits; pp.P = a.P; pp.inv_rms = a.inv_rms; pp.eps = a.eps;
    pp.n_tokens = a.n_tokens; pp.tok_pad = a.tok_pad; pp.d = a.d; pp.OTT = out_tiles(a.nog); pp.NOG = a.nog;
    const long long ntl = (a.n_tokens + P3_TILE - 1) / P3_TILE;
    const int ott = pp.OTT;
    const int otk = ott > 8 ? 8 : kernel_tiles(a.nog);      // instantiated tile counts
    long long nb;
    pp.ny = 1;
    if (ott > 8) {
        // column blocks of 8 output tiles: the ny workgroups of a token tile get ids equal mod 8 (one XCD, one L2)
        pp.ny = ott / 8;
        long long gx = (ncu / pp.ny) & ~7LL;                 // token tiles in flight: a multiple of 8 (one id residue per XCD)
        if (gx < 8) gx = 8;
        const long long need = (ntl + 7) & ~7LL;
        if (gx > need) gx = need;
        nb = gx * pp.ny;
    } else {
        nb = ntl < ncu ? ntl : ncu;
    }
    const size_t lds = p3_lds_bytes(a.x_dtype, otk);
    auto go = [&](auto kern) -> int {
        P3_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(P3_NW * 64), lds, s, pp);
        P3_TRY(hipGetLastError());
        return 0;
    };
    const bool norm = a.inv_rms != nullptr;
#define P3_GO(OTV)                                                                                                                 \
    (a.x_dtype == RQAE_X_BF16 ? (norm ? go(gram_proj3_kernel<RQAE_X_BF16, OTV, true>) : go(gram_proj3_kernel<RQAE_X_BF16, OTV, false>))    \
     : a.x_dtype == RQAE_X_F16 ? (norm ? go(gram_proj3_kernel<RQAE_X_F16, OTV, true>) : go(gram_proj3_kernel<RQAE_X_F16, OTV, false>))     \
                               : go(gram_proj3_kernel<RQAE_X_F32, OTV, false>))
    switch (otk) {
    case 1: return P3_GO(1);
    case 2: return P3_GO(2);
    case 4: return P3_GO(4);
    case 8: return P3_GO(8);
    }
#undef P3_GO
    return p3fail("proj3: no kernel for %d output tiles", ott);
}

}  // namespace rqae_p3
