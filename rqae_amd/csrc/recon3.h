// recon3.h -- what gram.hip needs of recon3.hip (the fp16 two-piece reconstruction of the matrix-core arm, gram_recon3.hpp): plain
// C++ entry points inside librqae_hip.so, not part of the C ABI (include/rqae_hip.h).
#ifndef RQAE_RECON3_H
#define RQAE_RECON3_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rqae_r3 {

// bytes of the W_out image: [d / 32][ceil(L c / 16)][2 pieces][1 KiB]
__attribute__((visibility("hidden"))) size_t image_bytes(int d, int c, int L);
// max |W_out| into *maxbits_dev (a zeroed device word), then the image; both on `stream`
__attribute__((visibility("hidden"))) int pack(void *img, unsigned *maxbits_dev, int d, int c, int L, const float *w_out, hipStream_t stream);
// is there a kernel for `levels` levels of c values over d elements (instantiated depth, d % 128 == 0, LDS)?
__attribute__((visibility("hidden"))) bool supported(int d, int c, int levels);

struct Args {
    const void *img;             // pack()'s image
    const unsigned *maxbits;     // pack()'s scale word
    const float *bias;           // [d]: the bias sums of the levels in use
    const float *planes;         // fp32 straight-through values [level][tok_pad][cvs]
    float *out;                  // [n_tokens][d]
    long long n_tokens, tok_pad;
    int d, c, L, levels, cvs;
};
__attribute__((visibility("hidden"))) int launch(const Args &a, int ncu, hipStream_t stream);

}  // namespace rqae_r3
#endif
