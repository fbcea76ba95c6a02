/*
 * rqae_hip.h -- C ABI of the MI355X (gfx950) RQAE encode/decode library, librqae_hip.so.
 *
 * This is the drop-in boundary for the residual-quantisation hot path of
 * harish-kamath/rqae.  The reference has no FFI: the path lives in one Python
 * class, rqae/model.py:RQAE.  Each entry point below replaces the arithmetic of
 * one method of that class; the Python mirror (rqae_amd/model.py) and any other
 * host language bind these symbols (ctypes / cffi / cgo / JNI all see plain
 * pointers and sizes; no torch types cross this line).
 *
 * Conventions
 *   - every pointer except `level_mask` is DEVICE memory on the current HIP device
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is enqueued on it, nothing synchronises the device, no allocation happens
 *   - inputs are borrowed and never written; outputs are caller-allocated
 *   - return 0 on success; non-zero on error, text via rqae_hip_last_error()
 *   - arithmetic is fp32 in the order fixed by DESIGN.md section 3; integer
 *     outputs are bit-exact against oracle/rqae_oracle.c
 *
 * Shapes:  d = RQAE dim, c = codebook_dim, K = rows per codebook, L = num_quantizers.
 */
#ifndef RQAE_HIP_H
#define RQAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQAE_HIP_ABI_VERSION 4

/* element type of the activation matrix x */
enum { RQAE_X_F32 = 0, RQAE_X_BF16 = 1, RQAE_X_F16 = 2 };
/* element type of a code-index matrix handed to decode */
enum { RQAE_IDX_I64 = 0, RQAE_IDX_I32 = 1, RQAE_IDX_U16 = 2 };

/*
 * Status word.  Entry points that gather rows by caller-supplied code indices take `int32_t *status`: a DEVICE int32 (or
 * NULL) into which the kernels OR the bits below when they meet bad input -- instead of faulting, asserting or reading out of
 * bounds.  The kernels never clear it; the caller zeroes it, reads it back after the stream work (or lazily) and turns a set
 * bit into its own error.  The reference raises a recoverable IndexError for such input (rqae/model.py:234,
 * rqae/feature.py:120, demo/server/server.py:55): rqae_amd/_native.py raises IndexError from this word and the process
 * lives on.  A flagged call's output rows for the bad codes are those of row 0 (decode) or unspecified finite values
 * (intensity / search); every other row is unaffected.
 */
#define RQAE_STATUS_CODE_RANGE 1    /* a code index outside its codebook / table / dataset */

int rqae_hip_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char *rqae_hip_last_error(void);

/*
 * Parameter block.  The reference keeps L pairs of nn.Linear plus one codebook
 * Parameter (rqae/model.py:29-54); the kernels want them stacked, zero-padded to
 * a multiple of 256 along d and lin_out transposed.  rqae_hip_pack builds that
 * block on the device from the stacked fp32 tensors:
 *   w_in [L][c][d]  = layers.{l}.0.weight     b_in  [L][c] = layers.{l}.0.bias
 *   w_out[L][d][c]  = layers.{l}.1.weight     b_out [L][d] = layers.{l}.1.bias
 *   codebook [L][K][c]  (must be finite)
 * codebook_shared != 0 asserts that all L codebooks are identical (FSQ methods,
 * rqae/model.py:63-72), which lets encode stage the codebook in LDS once.
 * rqae_hip_packed_bytes returns 0 for an unsupported shape (d > 8192, c > 16).
 */
size_t rqae_hip_packed_bytes(int d, int c, int K, int L);
int rqae_hip_pack(void *packed, int d, int c, int K, int L, const float *w_in, const float *b_in,
                  const float *codebook, const float *w_out, const float *b_out,
                  int codebook_shared, void *stream);

/*
 * RQAE.forward  (rqae/model.py:199-230): the level loop
 *   q = lin_in_l(r); idx = argmax cos(q, codebook_l); cv = q + (codebook_l[idx] - q);
 *   y = lin_out_l(cv); r -= y; out += y
 * for l = 0 .. levels-1 (levels = min(L, max_layers)), one launch.
 *   x        [n_tokens][d]       x_dtype; bf16/f16 are widened exactly to fp32
 *   q_out    [n_tokens][d] fp32  or NULL for codes only
 *   indices  [n_tokens][levels]  int64, contiguous (the reference returns the same
 *                                values as a permuted view, rqae/model.py:226)
 */
int rqae_hip_encode(const void *packed, int d, int c, int K, int L, const void *x, int x_dtype,
                    int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *stream);

/*
 * RQAE.quantize / quantize_gumbel at temperature 0 (rqae/model.py:187-197):
 *   p [n][c] fp32, codebook_l [K][c] fp32 -> cv [n][c] = codebook_l[idx], idx [n] int64
 */
int rqae_hip_quantize(const float *p, int64_t n, int c, int K, const float *codebook_l, float *cv,
                      int64_t *idx, void *stream);

/*
 * RQAE.decode (rqae/model.py:250-252): codebook[0][indices] for every level
 * (rqae/model.py:234), then the ascending sum of lin_out_l over the levels whose
 * level_mask[l] != 0 (rqae/model.py:236-248).  level_mask is a HOST array of L
 * bytes or NULL (= all levels).  Levels >= idx_levels must be masked out.
 *   indices [n_tokens][idx_levels] idx_dtype      out [n_tokens][d] fp32
 * It is an error if no level is selected (the reference returns None; the host
 * mirror handles that case before calling).
 */
/* Codes: -K <= code < K as in torch indexing (rqae/model.py:234: a negative code counts from the last row); any other value
 * is read as row 0 and sets RQAE_STATUS_CODE_RANGE in *status (the reference raises IndexError). */
int rqae_hip_decode(const void *packed, int d, int c, int K, int L, const void *indices,
                    int idx_dtype, int64_t n_tokens, int idx_levels, const uint8_t *level_mask,
                    float *out, int32_t *status, void *stream);

/* RQAE.decode_from_codebook_values (rqae/model.py:236-248): cv [n_tokens][nq][c] fp32. */
int rqae_hip_decode_values(const void *packed, int d, int c, int K, int L, const float *cv,
                           int64_t n_tokens, int nq, const uint8_t *level_mask, float *out,
                           void *stream);

/*
 * The step either side of forward() inside RQAE.hook (rqae/model.py:276-289) when the LLM is Gemma-2
 * (rqae/llm.py:65-73).  hs is the hooked layer's output [n_tokens][d] in hs_dtype (RQAE_X_*).
 *   rqae_hip_gemma_norm:  normed = (float(hs) * rsqrt(mean(hs^2) + eps)) * (1 + norm_weight)   fp32 [n][d]
 *                         inv_rms[n] = rsqrt(mean(hs^2) + eps)   (kept for the way back)
 *   rqae_hip_gemma_denorm_patch:  new = (q_out / (1 + norm_weight)) / inv_rms; rows with token %
 *                         seq_len == 0 keep hs when skip_bos != 0 (rqae/model.py:285-286); new is written
 *                         back into hs_inout in its own dtype (rqae/model.py:289) and, if new_out != NULL,
 *                         also as fp32 [n][d] (what the reference hands to store("new", ...)).
 */
/*
 * rqae_hip_encode_gemma: rqae_hip_gemma_norm folded into rqae_hip_encode -- the hooked layer's output goes in, the
 * RMSNorm is applied while the tile sits in registers (same arithmetic and order as rqae_hip_gemma_norm, so the
 * codes and q_out are bit-equal to the two-call form), normed is never written.  inv_rms [n_tokens] comes out for
 * rqae_hip_gemma_denorm_patch.  Only for shapes served by the LDS kernel: rqae_hip_encode_gemma_supported
 * returns 1 for them, 0 otherwise (then call rqae_hip_gemma_norm + rqae_hip_encode).
 */
int rqae_hip_encode_gemma_supported(int d, int c, int K, int L);
int rqae_hip_encode_gemma(const void *packed, int d, int c, int K, int L, const void *hs, int hs_dtype,
                          const float *norm_weight, float eps, int64_t n_tokens, int levels, float *q_out,
                          int64_t *indices, float *inv_rms, void *stream);
int rqae_hip_gemma_norm(const void *hs, int hs_dtype, const float *norm_weight, float eps, int64_t n_tokens,
                        int d, float *normed, float *inv_rms, void *stream);
int rqae_hip_gemma_denorm_patch(const float *q_out, const float *norm_weight, const float *inv_rms,
                                int64_t n_tokens, int d, int seq_len, int skip_bos, void *hs_inout,
                                int hs_dtype, float *new_out, void *stream);

/*
 * RQAEFeature.intensity (rqae/feature.py:102-129): per token the running, weight-normalised similarity of its
 * codes to a feature's centre codes, at the requested levels (the consumer of the codes this library writes).
 *   rqae_hip_intensity_prepare, once per (feature, M = max(layers) + 1):
 *       table[m][k] = codebook_sims[center[m]][k]   (fp16 [M][K]; codebook_sims is RQAE.codebook_sims, fp16 [K][K])
 *       wcum[m]     = layer_weights[:M].cumsum()[m] in torch-CPU's order (fp32 accumulator, fp16 prefixes)
 *   rqae_hip_intensity:
 *       out[n][j] = fp16( fp16( sum_{m <= layers[j]} fp16(table[m][code[n][m]] * w[m]) ) / wcum[layers[j]] )
 *       indices [n_tokens][nq] idx_dtype (RQAE_IDX_*), only columns < M are read; a code outside [0, K) contributes an
 *       unspecified finite value (an LDS read, never a fault) and sets RQAE_STATUS_CODE_RANGE in *status (may be NULL);
 *       layers_dev [n_layers] int32 DEVICE array, each < M, any order, duplicates allowed;
 *       table must be 16-byte aligned and allocated in whole 16-byte pieces; out [n_tokens][n_layers] fp16.
 * Bit-equal to the reference (the level sum is a sequential fp32 chain per token; nothing is re-associated).
 */
int rqae_hip_intensity_prepare(const void *codebook_sims, const int *center, const void *layer_weights, int M, int K,
                               void *table, void *wcum, void *stream);
int rqae_hip_intensity(const void *table, const void *layer_weights, const void *wcum, int M, int K,
                       const void *indices, int idx_dtype, int64_t n_tokens, int nq, const int *layers_dev,
                       int n_layers, void *out, int32_t *status, void *stream);

/*
 * The model's cached tables (rqae/model.py:133-178): what RQAEFeature and the demo server read
 * from the module besides the codes.  All pointers are device pointers; the fp16 tables are row-major.
 *
 *   rqae_hip_codebook_sims      replaces RQAE.codebook_sims (rqae/model.py:133-143):
 *       out[K][K] fp16 = F.normalize(codebook0) @ F.normalize(codebook0).T with torch-CPU's orders written out
 *       (same bits as the reference on CPU).  codebook0 is codebook[0], [K][c] fp32.
 *   rqae_hip_subfeatures        replaces RQAE.subfeatures (rqae/model.py:145-156):
 *       out[L][K][d] fp32 = lin_out_l(codebook[l]) (same bits as the reference on CPU).
 *       codebook [L][K][c], w_out [L][d][c], b_out [L][d], all fp32.
 *   rqae_hip_subfeature_sims    replaces RQAE.subfeature_sims and RQAE.layer_norms (rqae/model.py:158-178):
 *       sims_out[L][K][K] fp16 = cosine of subfeature rows, layer_norms_out[L] fp32 = mean column norm of
 *       W_out[l]; either may be NULL.  Computed from the (c+1)x(c+1) Gram matrix of [W_out[l] | b_out[l]] in
 *       fp64 -- the (L,K,d) subfeatures are never materialised.  scratch: rqae_hip_tables_scratch_bytes(L,K,c)
 *       bytes, 8-byte aligned, owned by the caller for the duration of the call's stream work.
 *       scale_by_layer_norms != 0 writes what the demo server keeps after `sims *= model.layer_norms[:, None, None]`
 *       (demo/server/server.py:101-112): fp16(float(fp16 cosine) * layer_norms[l]) -- one pass instead of two.
 */
int rqae_hip_codebook_sims(const float *codebook0, int K, int c, void *out, void *stream);
int rqae_hip_subfeatures(const float *codebook, const float *w_out, const float *b_out, int L, int K, int c, int d,
                         float *out, void *stream);
size_t rqae_hip_tables_scratch_bytes(int L, int K, int c);
int rqae_hip_subfeature_sims(const float *codebook, const float *w_out, const float *b_out, int L, int K, int c, int d,
                             void *scratch, void *sims_out, float *layer_norms_out, int scale_by_layer_norms,
                             void *stream);

/*
 * The matrix-core arm of RQAE.forward ("gram arm", csrc/gram.hip).  Same inputs and outputs as
 * rqae_hip_encode (rqae/model.py:199-230), different arithmetic: lin_in of every level as ONE [tokens,d] x [d,L*c]
 * product, the level chain in c-space on precomputed c x c Gram blocks W_in[l] W_out[m], the
 * reconstruction as the decode sum (rqae/model.py:236-248) -- products on the matrix cores with every fp32 operand cut into
 * low-precision pieces: two fp16 pieces per value (hi = rn16(v 2^T), lo = rn16(v 2^T - hi), T an exact power of two found at pack
 * time; three products per operand pair -- hi hi, hi lo, lo hi; the lo lo product, below 2^-22 of the term, is not formed, and
 * a 2-byte activation is a single piece: two products) on v_mfma_f32_32x32x16_f16 in the default kernels of round 4 (proj = 4, recon = 3, the
 * sign chain, the products of calls deeper than 64 levels), three bf16 pieces (six products) on v_mfma_f32_16x16x32_bf16 in the
 * kernels of rounds 2-3.  Results carry fp32-level rounding but are not the bits of rqae_hip_encode: parity is SURVEY 8(c) Tier 2
 * (codes identical wherever the top-1/top-2 cosine margin is >= 1e-5; q_out within 1e-5 relative).  Exact duplicate codebook rows
 * still resolve to the lowest index.  An input value beyond fp16's range (|x| > 65504 in bf16 / fp16 / fp32 hidden states) makes its
 * token's projection non-finite: row 0 at every level and a NaN reconstruction, as a non-finite row does (rqae/model.py:182,188).
 *   rqae_hip_gram_supported      1 if the shape is served (d % 128 == 0, c <= 13, L <= 1024; see DESIGN.md sections 9, 12, 13 for
 *                                the codebook forms), else 0 -> use rqae_hip_encode
 *   rqae_hip_gram_packed_bytes   size of the arm's own parameter block (0 = unsupported)
 *   rqae_hip_gram_pack           builds it from the stacked fp32 tensors (same arguments as rqae_hip_pack)
 *   rqae_hip_gram_workspace_bytes scratch the caller lends to one rqae_hip_encode_gram call: the projections
 *                                [groups][tokens][16] fp32 and the reconstruction operand rows [k-steps][tokens][32]
 *                                bf16 of at most 131072 tokens at a time (longer batches run in passes)
 *   rqae_hip_encode_gram         x, q_out (or NULL), workspace 16-byte aligned; indices int64 [n_tokens][levels]
 *   rqae_hip_encode_gram_phases  the same call with HIP events between its three kernels (measurement only: it
 *                                synchronises the stream); phase_ms[3] = projection, level chain, reconstruction of the
 *                                last pass, in milliseconds
 */
int rqae_hip_gram_supported(int d, int c, int K, int L, int codebook_shared);
/* one call with `levels` levels.  Up to 64 levels the codes can always be had on a supported shape and the reconstruction
 * (want_recon != 0) needs one of the instantiated reconstruction kernels (32 levels at c = 4, 24 at c = 5, 16 at 6 <= c <= 13).  Beyond 64 levels
 * (the released model's 1024: DESIGN.md 13.4) the call runs whole blocks of 32 levels of a c = 4 shape with L % 64 == 0 and a shared
 * FSQ lattice codebook, and must pass lean_levels = 5. */
int rqae_hip_encode_gram_supported(int d, int c, int K, int L, int codebook_shared, int levels, int want_recon);
size_t rqae_hip_gram_packed_bytes(int d, int c, int K, int L, int codebook_shared);
int rqae_hip_gram_pack(void *packed, int d, int c, int K, int L, const float *w_in, const float *b_in,
                       const float *codebook, const float *w_out, const float *b_out, int codebook_shared,
                       const float *in_scale /* [d] or NULL: a diagonal map in front of lin_in, folded into the W_in image */,
                       void *stream);
/* RQAE.normalize_codebooks as a `levels`-level forward applies it (rqae/model.py:126-131, called once per level from quantize, :196;
 * any quantization_method but the FSQ lattices): codebook [L][K][c] fp32 is renormalised `levels` times IN PLACE (every level's rows,
 * as the reference does), and used_out [L][K][c] (or NULL) receives for each level l < levels its rows after l + 1 renormalisations
 * -- what that level is quantised against.  One launch; torch-CPU's order of the row norm and IEEE divisions: the reference's bits.
 * rqae_hip_gram_repack_codebooks rebuilds just the codebook part (images + fp32 rows) of a block packed with codebook_shared = 0 from
 * such per-level rows; the linear maps' images and the Gram blocks stay. */
int rqae_hip_normalize_codebooks(float *codebook, int L, int K, int c, int levels, float *used_out, void *stream);
int rqae_hip_gram_repack_codebooks(void *packed, int d, int c, int K, int L, const float *codebook, void *stream);
/* n >= 2 when rqae_hip_gram_pack found the block's (shared) codebook to be the FSQ lattice {linspace(-1, 1, n)}^c of
 * rqae/model.py:63-72 (rows of unit length, or plain), else 0; -1 if `packed` is no block.  Waits for `stream` -- the pack that
 * wrote the answer; the only call of this arm that blocks.  A caller that got n >= 2 may pass lean_levels = 5 below. */
int rqae_hip_gram_lattice(const void *packed, void *stream);
size_t rqae_hip_gram_workspace_bytes(int d, int c, int K, int L, int64_t n_tokens, int levels);
/* Which kernels one rqae_hip_encode_gram call launches: per call, no process-wide state (NULL = the defaults, in
 * parentheses; a negative field keeps its default).  Every combination computes the same function under the same parity
 * policy; they differ in speed.
 *   size        sizeof(rqae_hip_gram_options) (lets the struct grow)
 *   fused       1: one persistent kernel (projection + level chain + reconstruction per 256-token tile)        (0)
 *   proj        4: fp16 two-piece projection, every output tile in one launch; 3: LDS-DMA projection kernel
 *               of round 3, 2: register-pipelined, 1: first form                                                  (4)
 *   recon       3: fp16 two-piece reconstruction (with it C4's level chain runs as matrix products, 13.5);
 *               2: token-stationary reconstruction of round 3 with 512-byte store rows, 1: weights-stationary     (3)
 *   lean_levels the level chain.  4 = for a codebook that rqae_hip_gram_pack found to be an FSQ lattice ({linspace(-1, 1, n)}^c,
 *               rows of unit length or plain: rqae/model.py:63-72) the lattice form -- the rows that can win follow from the
 *               signs and the order of |projection| (6 of 1024 at n = 4, c = 5), each scored like the scan's rescoring scores
 *               a row; no scan -- and for any other codebook the scan kernel 2 names (launched behind the lattice form; on a
 *               lattice it returns at once: 5 us).  5 = the lattice form alone, for a caller whom rqae_hip_gram_lattice has told
 *               that the block holds a lattice (on any other block such a call computes nothing).  0..3 name a scan kernel and
 *               switch the lattice form off: 2 = (one level group, levels <= 8) fp32 straight-through values handed to the
 *               reconstruction kernel, 16 waves per workgroup, tables resident in LDS; 1 = the same with 8 waves;
 *               0 = the first form; 3 = the streamed form (codebook image through a 24 KB ring by loader waves)     (4)
 *   pipe_chunk  > 0: passes of that many tokens, reconstruction of pass i on a second stream beside pass i+1    (0) */
typedef struct rqae_hip_gram_options {
    int32_t size;
    int32_t fused, proj, recon, lean_levels;
    int64_t pipe_chunk;
} rqae_hip_gram_options;
int rqae_hip_encode_gram(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *x, int x_dtype,
                         int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *workspace,
                         const rqae_hip_gram_options *options, void *stream);
/* The step in front of forward inside RQAE.hook when the LLM is Gemma-2 (rqae/model.py:279-281; Gemma2.norm, rqae/llm.py:65-66),
 * on the matrix-core arm: hs [n_tokens][d] bf16 / fp16 is the hooked layer's output itself.  The norm is linear in front of
 * lin_in: normed = (hs rs)(1 + w), rs = 1 / sqrt(mean(hs^2) + eps) per token, so the caller packs with in_scale = 1 + w
 * (rqae_hip_gram_pack) and the projection kernel sums hs^2 while the tile passes through its registers and scales the
 * projections by rs; `normed` never exists in memory.  q_out / indices as rqae_hip_encode_gram (in normed space: the reference's
 * `quantized`); inv_rms [n_tokens] fp32 receives rs for the way back (rqae_hip_gemma_denorm_patch).  Tier-2 parity. */
int rqae_hip_encode_gram_gemma_supported(int d, int c, int K, int L, int codebook_shared, int hs_dtype);
int rqae_hip_encode_gram_gemma(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *hs, int hs_dtype,
                               float eps, int64_t n_tokens, int levels, float *q_out, int64_t *indices, float *inv_rms,
                               void *workspace, const rqae_hip_gram_options *options, void *stream);
/* RQAE.hook's arithmetic end to end for a Gemma-2 layer output, IN PLACE (rqae/model.py:276-289 with the default no-op store and
 * replace = True): the norm folded into the projection as above, the level chain, and a reconstruction kernel whose stores are the
 * way back -- new = (quantized / (1 + w)) / rs (rqae/llm.py:68-73) rounded to hs' own dtype over hs, rows of the first position of
 * every sequence (token % S == 0) left as they are when skip_bos != 0 (rqae/model.py:285-286).  The fp32 `quantized` tensor is never
 * written: 2 d + 2 d bytes per token move instead of 2 d + 4 d + (4 d + 2 d + 2 d).  inv_1pw [d] fp32 = 1 / (1 + norm.weight),
 * allocated in whole KiB; inv_rms [n_tokens] fp32 is scratch (it holds rs afterwards); levels <= 8.  Tier-2 parity for the codes;
 * the patched values within one rounding step of hs' dtype of the unfused path's. */
/* 1 when rqae_hip_encode_gram_gemma_patch serves the call: levels <= 8 (one level group), 2-byte hs, and the level chain's tables fit
 * LDS -- or, for a codebook rqae_hip_gram_lattice named an FSQ lattice and a call that passes lean_levels = 5 (lattice_vouched != 0),
 * just its fp32 rows do (K = 4096 at c = 4).  0: take rqae_hip_encode_gram_gemma + rqae_hip_gemma_denorm_patch. */
int rqae_hip_encode_gram_gemma_patch_supported(int d, int c, int K, int L, int codebook_shared, int levels, int hs_dtype, int lattice_vouched);
int rqae_hip_encode_gram_gemma_patch(const void *packed, int d, int c, int K, int L, int codebook_shared, void *hs_inout, int hs_dtype,
                                     float eps, const float *inv_1pw, int S, int skip_bos, int64_t n_tokens, int levels,
                                     int64_t *indices, float *inv_rms, void *workspace, const rqae_hip_gram_options *options,
                                     void *stream);
/* RQAE.decode (rqae/model.py:250-252) on the matrix-core arm, every level of indices [n_tokens][levels] selected: the decode sum
 * as rqae_hip_encode_gram's reconstruction kernel computes it (bf16-piece products, fp32 accumulation: within 1e-6 relative of
 * rqae_hip_decode, not its bits), codebook rows gathered by the codes in the kernel.  HBM-write-bound: 9 216 + 8 levels bytes per
 * token at d = 2304.  Served for levels <= 8 (one level group); codes int64 / int32 / uint16 checked as in rqae_hip_decode
 * (-K <= code < K, else row 0 and RQAE_STATUS_CODE_RANGE in *status). */
int rqae_hip_decode_gram_supported(int d, int c, int K, int L, int codebook_shared, int levels);
int rqae_hip_decode_gram(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *indices, int idx_dtype,
                         int64_t n_tokens, int levels, float *out, int32_t *status, void *stream);
int rqae_hip_encode_gram_phases(const void *packed, int d, int c, int K, int L, int codebook_shared, const void *x,
                                int x_dtype, int64_t n_tokens, int levels, float *q_out, int64_t *indices, void *workspace,
                                const rqae_hip_gram_options *options, void *stream, float *phase_ms);

/* ---- Similarity search over a resident dataset of codes ------------------------------
 * Replaces the arithmetic of demo/server/server.py: get_intensities (:41-68) and IntensityEngine.find_examples
 * (:160-325).  The caller (rqae_amd/search.py mirrors find_examples) walks the layer ranges; per range [l0, l1):
 *   rqae_hip_search_query_table  table_out[l - l0][k][q] fp16 = sims16[l][ query_codes[q * query_stride + l] ][k]
 *                                (server.py:186-200 with the row's two axes swapped: one dataset code then needs ONE
 *                                contiguous row).  sims16 [L][K][K] fp16 is the table the server keeps after
 *                                `sims *= layer_norms` (:101-112; rqae_hip_subfeature_sims writes it).  Rows padded to
 *                                128 query positions and K to a multiple of 16 (zeros);
 *                                rqae_hip_search_table_bytes(K, l1 - l0) bytes.  Query codes must be in [0, K).
 *   rqae_hip_search_accumulate   acc[t][q] (fp16, [n_tokens][128], n_tokens = sequences x positions) takes the range's
 *                                intensities: per (t, q) the fp32 sum over ascending levels of
 *                                table[l][codes[t][l]][q], rounded to fp16 per chunk of <= 64 levels, chunks added
 *                                in fp16 (server.py:214-252), and the range added to acc in fp16 (:257-260; first != 0
 *                                stores instead, as the reference's first range does).  codes [n_tokens][L] int32 or
 *                                uint16 (RQAE_IDX_*), device-resident; a code outside [0, K) contributes an unspecified finite
 *                                value (an LDS read, never a fault) and sets RQAE_STATUS_CODE_RANGE in *status.
 *   rqae_hip_search_rank         max_out[q][r] fp16 / order_out[q][r] int32: per query position the sequences by
 *                                descending max over positions of acc (server.py:263-266); equal maxima in ascending
 *                                sequence order (torch.argsort there promises none).  Sequence n = tokens
 *                                [n S, (n + 1) S).  workspace: rqae_hip_search_rank_workspace_bytes(n_seqs, Q).
 *   rqae_hip_search_gather       out[q][e][s] fp16 = acc[examples[q * E + e]][s][q]  (server.py:293-311); an example outside
 *                                [0, n_seqs) yields zeros and sets RQAE_STATUS_CODE_RANGE in *status
 * Supported: Q <= 128, K <= 1280 (rqae_hip_search_supported).  All pointers are device pointers; acc and table_out
 * 16-byte aligned. */
int rqae_hip_search_supported(int K, int Q);
size_t rqae_hip_search_table_bytes(int K, int n_levels);
int rqae_hip_search_query_table(const void *sims16, int L, int K, const int32_t *query_codes, int Q, int64_t query_stride,
                                int l0, int l1, void *table_out, void *stream);
int rqae_hip_search_accumulate(const void *codes, int code_dtype, int64_t n_tokens, int L, const void *table, int K, int Q,
                               int l0, int l1, void *acc, int first, int32_t *status, void *stream);
size_t rqae_hip_search_rank_workspace_bytes(int64_t n_seqs, int Q);
int rqae_hip_search_rank(const void *acc, int64_t n_seqs, int S, int Q, void *max_out, int32_t *order_out, void *workspace,
                         size_t workspace_bytes, void *stream);
int rqae_hip_search_gather(const void *acc, int64_t n_seqs, int S, int Q, const int32_t *examples, int E, void *out,
                           int32_t *status, void *stream);

/* Human-readable name of the encode kernel variant chosen for a shape. */
int rqae_hip_kernel_info(int d, int c, int K, int L, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RQAE_HIP_H */
