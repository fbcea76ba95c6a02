// deep3.h -- what gram.hip needs of deep3.hip (the matrix-core arm beyond 64 levels, gram_deep3.hpp): plain C++ entry points inside
// librqae_hip.so, not part of the C ABI (include/rqae_hip.h).
#ifndef RQAE_DEEP3_H
#define RQAE_DEEP3_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rqae_d3 {

constexpr int BLOCK_LEVELS = 32;     // levels per block of the deep level chain (four level groups of 8; c = 4: 128 values = 8 k-steps)

// the G = -(W_in W_out) image for the cross-block corrections: [L c / 32][L c / 16][2][1 KiB]  (c = 4)
__attribute__((visibility("hidden"))) size_t g3_bytes(int L);
__attribute__((visibility("hidden"))) int pack_g3(void *img, unsigned *maxbits_dev, const float *gneg, int L, int GB, hipStream_t stream);
// the straight-through values' operand fragments: [tok_pad / 32][L c / 16][2][1 KiB]
__attribute__((visibility("hidden"))) size_t pieces_bytes(long long tok_pad, int L);
// planes [nlev][tok_pad][4] of levels l0 .. l0 + nlev - 1 -> the fragments of their k-steps (nlev % 4 == 0, l0 % 4 == 0)
__attribute__((visibility("hidden"))) int convert(const float *planes, long long tok_pad, int nlev, int l0, void *pieces, int L, hipStream_t stream);

struct Gemm {
    const void *apieces;         // convert()'s fragments
    const void *wimg;            // pack_g3()'s image, or rqae_r3's W_out image
    const unsigned *maxbits;     // that image's scale word
    long long n_tokens, tok_pad;
    int KSA, KSW;                // k-steps per tile in the two images
    int ks0, nks;                // k-steps [ks0, ks0 + nks) are multiplied
    int nt0, ntiles;             // output tiles of 32 columns: nt0 .. nt0 + ntiles - 1 (ntiles % 4 == 0)
    float *P;                    // corrections: added into [groups][tok_pad][16]
    const float *bias;           // reconstruction: [d] and out [n_tokens][d]
    float *out;
    int d;
};
__attribute__((visibility("hidden"))) int corrections(const Gemm &g, int ncu, hipStream_t stream);
__attribute__((visibility("hidden"))) int reconstruct(const Gemm &g, int ncu, hipStream_t stream);

}  // namespace rqae_d3
#endif
