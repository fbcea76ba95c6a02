// gram_pieces.hpp -- fp16 two-piece operands for v_mfma_f32_32x32x16_f16 (shared by recon3.hip and proj3.hip; included inside each
// unit's anonymous namespace behind gram_small.hpp).  An fp32 value v, scaled by an exact power of two into fp16's range, is cut
// into hi = rn16(v) and lo = rn16(v - hi): the remainder is below 2^-24 |v|, and hi*hi + hi*lo + lo*hi of two such values
// reproduces their fp32 product to ~2^-23 -- three matrix products where three bf16 pieces per value needed six.
constexpr int R3_CV_SHIFT = 10;          // straight-through values are scaled by 2^10 before they are cut into fp16 pieces (unit rows: |cv| <= 1 + one ulp of the projection)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// fp32 -> two fp16 pieces, round to nearest: v = hi + lo + r, |r| <= 2^-24 |v| (for |v| inside fp16's normal range; below it the
// absolute error is at most half a subnormal step, 2^-25).  A value beyond fp16's range (65504) converts to inf and its remainder
// to -inf: every product it enters is NaN -- the token's projection is NaN, it takes index 0 and a NaN reconstruction, as a NaN /
// inf input row does in the reference (rqae/model.py:188); out-of-range input fails loudly instead of saturating.
__device__ __forceinline__ void split2(float v, unsigned short &hi, unsigned short &lo)
{
    const _Float16 h = (_Float16)v;
    const float r = v - (float)h;
    const _Float16 l = (_Float16)r;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}


// the exact power of two that brings `maxbits` (a float's bits) into [2^14, 2^15): its exponent T (a weight w goes in as w 2^T)
__host__ __device__ inline int r3_shift_of(unsigned maxbits)
{
    const int e = (int)((maxbits >> 23) & 255u);      // max = 1.m x 2^(e - 127); zero / subnormal: no scaling worth doing
    if (e == 0 || e == 255) return 0;
    return 14 - (e - 127);
}
__device__ __forceinline__ float f32_bits(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float r3_pow2(int t) { return __builtin_bit_cast(float, (unsigned)((t + 127) << 23)); }     // -126 <= t <= 127


// largest |w| of an array as float bits (non-negative floats order like unsigned integers)
__attribute__((unused)) __global__ void gram_pack_absmax_kernel(const float *w, size_t n, unsigned *out_bits)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = __builtin_fabsf(w[i]);
        m = a > m ? a : m;        // NaN never wins: the image of a NaN weight is NaN whatever the scale
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));
}

