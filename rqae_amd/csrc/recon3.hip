// recon3.hip -- translation unit of librqae_hip.so: the matrix-core arm's reconstruction on fp16 two-piece operands
// (gram_recon3.hpp; rqae/model.py:221,224 summed as the decode sum).  Its own unit so that the kernel compiles in seconds; gram.hip
// calls it through recon3.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "recon3.h"

extern __attribute__((visibility("hidden"))) __thread char rqae_g_err[512];

namespace {

int r3fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int r3fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rqae_g_err, sizeof(rqae_g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define R3_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return r3fail("%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#include "gram_small.hpp"
#include "gram_pieces.hpp"
#include "gram_recon3.hpp"

}  // namespace

namespace rqae_r3 {

size_t image_bytes(int d, int c, int L) { return (size_t)(d / 32) * ((L * c + 15) / 16) * 2048; }

int pack(void *img, unsigned *maxbits_dev, int d, int c, int L, const float *w_out, hipStream_t s)
{
    hipLaunchKernelGGL(gram_pack_absmax_kernel, dim3(512), dim3(256), 0, s, w_out, (size_t)L * d * c, maxbits_dev);
    R3_TRY(hipGetLastError());
    hipLaunchKernelGGL(gram_pack_wout3_kernel, dim3(2048), dim3(256), 0, s, d, c, L, w_out, maxbits_dev, static_cast<uint16_t *>(img), (L * c + 15) / 16);
    R3_TRY(hipGetLastError());
    return 0;
}

bool supported(int d, int c, int levels)
{
    const int kt = r3_ks_template(r3_ksteps(c, levels));
    return kt != 0 && d % 128 == 0 && r3_lds_bytes(d, kt) <= 160 * 1024;
}

int launch(const Args &a, int ncu, hipStream_t s)
{
    const int kt = r3_ks_template(r3_ksteps(a.c, a.levels));
    if (!supported(a.d, a.c, a.levels)) return r3fail("recon3: no kernel for d=%d c=%d levels=%d", a.d, a.c, a.levels);
    Rec3Params pr;
    pr.wimg = static_cast<const uint16_t *>(a.img); pr.maxbits = a.maxbits; pr.bias = a.bias; pr.cvf = a.planes; pr.out = a.out;
    pr.n_tokens = a.n_tokens; pr.tok_pad = a.tok_pad; pr.d = a.d; pr.c = a.c; pr.levels = a.levels; pr.KS3 = (a.L * a.c + 15) / 16; pr.cvs = a.cvs;
    // planes of exactly c floats (the lean level chains, c == CP) at a depth the registers hold one tile ahead: the prefetching forms
    bool pf4 = a.cvs == 4 && a.c == 4, pf5 = a.cvs == 5 && a.c == 5;
    // two token sets per wave (512-token tiles): measured at 262 144 tokens (tools/r04_run_o.sh) -- C5's 8 k-steps 0.670 -> 0.621 ms, the
    // store-bound depths no better (C2 0.479 / 0.489, P@16 0.485 / 0.502): taken at 8 k-steps when every CU still gets two tiles
    int ts = kt == 8 && a.n_tokens >= 2LL * 2 * R3_TILE * ncu ? 2 : 1;
#ifdef R3_TOOLS   /* tools builds: R3_FORCE=nopref (values loaded at the tile's start), ts1 / ts2 */
    if (const char *e = getenv("R3_FORCE")) {
        if (strstr(e, "nopref")) pf4 = pf5 = false;
        if (strstr(e, "ts1")) ts = 1;
        if (strstr(e, "ts2") && kt <= 8) ts = 2;
    }
#endif
    const long long tile = (long long)R3_TILE * ts;
    long long nb = (a.n_tokens + tile - 1) / tile;
    if (nb > ncu) nb = ncu;
    const size_t lds = r3_lds_bytes(a.d, kt, ts);
    auto go = [&](auto kern) -> int {
        R3_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(R3_NW * 64), lds, s, pr);
        R3_TRY(hipGetLastError());
        return 0;
    };
    if (ts == 2)
        switch (kt) {
#ifdef R3_TOOLS
        case 2: return pf4 ? go(gram_recon3_kernel<2, 4, 2>) : go(gram_recon3_kernel<2, 0, 2>);
        case 3: return pf5 ? go(gram_recon3_kernel<3, 5, 2>) : pf4 ? go(gram_recon3_kernel<3, 4, 2>) : go(gram_recon3_kernel<3, 0, 2>);
        case 4: return pf4 ? go(gram_recon3_kernel<4, 4, 2>) : go(gram_recon3_kernel<4, 0, 2>);
#endif
        case 8: return go(gram_recon3_kernel<8, 0, 2>);
        }
    switch (kt) {
    case 2: return pf4 ? go(gram_recon3_kernel<2, 4>) : go(gram_recon3_kernel<2, 0>);
    case 3: return pf5 ? go(gram_recon3_kernel<3, 5>) : pf4 ? go(gram_recon3_kernel<3, 4>) : go(gram_recon3_kernel<3, 0>);
    case 4: return pf4 ? go(gram_recon3_kernel<4, 4>) : go(gram_recon3_kernel<4, 0>);
    case 8: return go(gram_recon3_kernel<8, 0>);       // (64 prefetched values beside 64 piece registers spill)
    case 13: return go(gram_recon3_kernel<13, 0>);
    case 16: return go(gram_recon3_kernel<16, 0>);
    }
    return r3fail("recon3: no kernel for %d k-steps", kt);
}

}  // namespace rqae_r3
