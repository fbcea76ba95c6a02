// proj3.h -- what gram.hip needs of proj3.hip (the all-levels projection on fp16 two-piece operands, gram_proj3.hpp): plain C++ entry
// points inside librqae_hip.so, not part of the C ABI (include/rqae_hip.h).
#ifndef RQAE_PROJ3_H
#define RQAE_PROJ3_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rqae_p3 {

// output tiles of 32 for nog groups of 16 outputs
inline int out_tiles(int nog) { return (nog + 1) / 2; }
// output tiles the kernel instantiation for nog groups computes (1, 2, 4 or 8), and the groups of 16 outputs the projection buffer
// must hold so that every one of its stores lands ([groups][tok_pad][16] floats)
inline int kernel_tiles(int nog) { const int t = out_tiles(nog); return t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : 8; }
inline int padded_groups(int nog) { return out_tiles(nog) <= 8 ? 2 * kernel_tiles(nog) : nog; }
// bytes of the W_in image: [d / 16][out_tiles][2 pieces][1 KiB]
__attribute__((visibility("hidden"))) size_t image_bytes(int d, int nog);
// max |W_in (x in_scale)| into *maxbits_dev (a zeroed device word), then the image; CP / GL / GP: the arm's output geometry
__attribute__((visibility("hidden"))) int pack(void *img, unsigned *maxbits_dev, int d, int c, int L, int CP, int GL, int GP, int nog,
                                               const float *w_in, const float *in_scale, hipStream_t stream);
// one workgroup covers up to 8 output tiles (256 outputs); deeper shapes run whole column blocks of 8 tiles side by side
__attribute__((visibility("hidden"))) bool supported(int d, int nog, int x_dtype, bool norm);

struct Args {
    const void *img;             // pack()'s image
    const unsigned *maxbits;     // pack()'s scale word
    const void *x;               // [n_tokens][d], x_dtype (RQAE_X_*)
    float *P;                    // [nog][tok_pad][16]
    float *inv_rms;              // NULL, or [n_tokens]: Gemma's RMSNorm folded in (1 + w is in the image: pack's in_scale)
    float eps;
    long long n_tokens, tok_pad;
    int d, nog, x_dtype;
};
__attribute__((visibility("hidden"))) int launch(const Args &a, int ncu, hipStream_t stream);

}  // namespace rqae_p3
#endif
