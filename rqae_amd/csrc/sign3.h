// sign3.h -- what gram.hip needs of sign3.hip (the level chain of a two-point FSQ lattice on the matrix cores, gram_sign3.hpp):
// plain C++ entry points inside librqae_hip.so, not part of the C ABI (include/rqae_hip.h).
#ifndef RQAE_SIGN3_H
#define RQAE_SIGN3_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rqae_s3 {

// the shapes the kernel is built for: BASELINE configs[3] (c = 13, depth 16: four level groups of four levels, 64 projection
// columns each), every level run, one shared codebook
__attribute__((visibility("hidden"))) bool supported(int c, int L, int shared_cb);
// fp16 two-piece image of the Gram blocks -(W_in[l2] W_out[l]) a level's straight-through values are multiplied by
__attribute__((visibility("hidden"))) size_t image_bytes(int c, int L);
// image + its scale word + `exact` (1 when every codebook row is, bit for bit, its sign pattern applied to the rows 0 / K - 1)
__attribute__((visibility("hidden"))) int pack(void *img, unsigned *maxbits_dev, int *exact_dev, const float *gneg, const float *codebook, int c, int L, int K,
                                               int GB, hipStream_t stream);

struct Args {
    const float *P;              // [groups of 16 columns][tok_pad][16]: the projections of every level (gram_proj3.hpp)
    const float *kconst;         // [columns]
    const void *img;             // pack()'s image
    const unsigned *maxbits;     // its scale word
    const int *lat_flag;         // the block's lattice word: the kernel runs when it is 2 ...
    const int *exact_flag;       // ... and the codebook is exact (pack)
    const float *cb;             // [K][c] fp32
    float *planes;               // out: straight-through values [levels][tok_pad][16] fp32 (what gram_recon3.hpp reads)
    long long *idx;              // out: [n_tokens][idx_stride]
    long long n_tokens, tok_pad;
    int c, L, K, idx_stride;
};
__attribute__((visibility("hidden"))) int launch(const Args &a, int ncu, hipStream_t stream);

}  // namespace rqae_s3
#endif
