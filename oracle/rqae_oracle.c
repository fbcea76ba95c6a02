/*
 * rqae_oracle.c -- CPU restatement of the reference RQAE encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rqae_amd/ may import, link or call
 * this file; it is the checker the HIP path is compared against (tests/,
 * __graft_entry__.smoke(), and the cpu_baseline leg of bench.py).
 *
 * What it restates (all file:line relative to /root/reference):
 *   forward   rqae/model.py:199-230   level loop: lin_in -> quantize -> STE ->
 *                                     lin_out -> residual -= / quantized_out +=
 *   quantize  rqae/model.py:187-197   x/||x|| -> x @ codebook[l].T -> argmax ->
 *                                     codebook[l][idx]
 *   argmax    rqae/model.py:180-185   first maximum wins, NaN counts as maximum
 *   decode    rqae/model.py:232-252   codebook[0][idx] -> sum_l lin_out_l(.)
 *
 * Arithmetic is fp32 with every rounding step written out, so that the HIP
 * kernels can reproduce it bit for bit:
 *   - x @ codebook.T  : k-ordered fmaf chain from +0   (== ATen mm on CPU, probed)
 *   - lin_out         : k-ordered fmaf chain from +0, then + bias (== ATen addmm)
 *   - ||q||           : ATen's vectorised 2-norm order (8-lane fma chunks, one
 *                       4-chunk of rounded squares, fma tail), then sqrtf
 *   - q / ||q||       : IEEE division
 *   - STE             : q + (cv - q), two roundings          (model.py:218-220)
 *   - lin_in (length-d dot): ATen's blocked GEMM order is not specified, so the
 *     order is DEFINED here ("canonical order"): d is
 *     zero-padded to a multiple of 256 and cut into 4 contiguous slices of
 *     64*P elements; elem_of(P, s, i, lane) belongs to (slice s, lane, step i);
 *     each (slice, lane) runs an fmaf chain over i = 0..P-1 from +0; the 64
 *     lanes of a slice are summed by the tree in lane_tree(); the slices are
 *     summed ((s0+s1)+(s2+s3)); the bias is added last.
 *   The reference's own output differs from this only through that one
 *   reduction order; parity with it is pinned by the tests/golden fixtures.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off -mfma ...).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NSLICE 4
#define NLANE 64

static inline int pad_dim(int d) { return ((d + 255) / 256) * 256; }

/* Sum of the 64 per-lane partials of one slice.  The pairing mirrors cheap
 * cross-lane moves on a 64-wide wavefront: lane^32, lane^16, lane^8, mirror
 * inside each group of 8, lane^1, lane^2.  Every step adds a lane and its
 * partner (fp32 add is commutative), so both partners hold the same bits. */
static inline float lane_tree(const float *v)
{
    float a[32], b[16], c[8];
    for (int l = 0; l < 32; ++l) a[l] = v[l] + v[l + 32];
    for (int l = 0; l < 16; ++l) b[l] = a[l] + a[l + 16];
    for (int l = 0; l < 8; ++l) c[l] = b[l] + b[l + 8];
    float u0 = c[0] + c[7], u1 = c[1] + c[6], u2 = c[2] + c[5], u3 = c[3] + c[4];
    return (u0 + u1) + (u2 + u3);
}

/* torch.linalg.vector_norm(x, 2) over a contiguous last dim of length c, as the
 * ATen CPU kernel orders it (rqae/model.py:188 -> x.norm(dim=-1)). */
static inline float aten_norm2(const float *x, int c)
{
    int d = 0;
    float acc = 0.0f;
    int n8 = (c / 8) * 8;
    if (n8) {
        float buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int d0 = 0; d0 < n8; d0 += 8)
            for (int j = 0; j < 8; ++j) buf[j] = fmaf(x[d0 + j], x[d0 + j], buf[j]);
        acc = buf[0];
        for (int j = 1; j < 8; ++j) acc = acc + buf[j];
        d = n8;
    }
    if (c - d >= 4) {
        for (int j = 0; j < 4; ++j) {
            float sq = x[d + j] * x[d + j];
            acc = acc + sq;
        }
        d += 4;
    }
    for (; d < c; ++d) acc = fmaf(x[d], x[d], acc);
    return sqrtf(acc);
}

/* torch.argmax semantics: a beats b if a > b, or a is NaN and b is not. */
static inline int beats(float a, float b)
{
    return (a > b) || (isnan(a) && !isnan(b));
}

/* nearest codebook row by cosine similarity; sims is scratch of K floats.
 * cbT is the level's codebook transposed to [c][K]. */
static int64_t nearest_row(const float *xn, const float *cbT, int c, int K, float *sims,
                           float *margin_out)
{
    for (int n = 0; n < K; ++n) sims[n] = 0.0f;
    for (int k = 0; k < c; ++k) {
        const float xk = xn[k];
        const float *row = cbT + (size_t)k * K;
        for (int n = 0; n < K; ++n) sims[n] = fmaf(xk, row[n], sims[n]);
    }
    int64_t bidx = 0;
    float best = sims[0];
    for (int n = 1; n < K; ++n)
        if (beats(sims[n], best)) { best = sims[n]; bidx = n; }
    if (margin_out) {
        /* top-1 minus the best similarity among rows that point in a different direction: rows that are bit-for-bit
         * copies of the winner (the FSQ lattices hold such duplicates) are excluded, every other row counts --
         * two distinct rows whose fp32 similarities are equal give margin 0 (the tie that torch.argmax breaks by index).
         * This is the margin of SURVEY 8(c)'s Tier-2 policy. */
        float second = -INFINITY;
        for (int n = 0; n < K; ++n) {
            if (n == bidx) continue;
            int same = 1;
            for (int k = 0; k < c && same; ++k) same = cbT[(size_t)k * K + n] == cbT[(size_t)k * K + bidx];
            if (!same && sims[n] > second) second = sims[n];
        }
        *margin_out = isnan(best) ? NAN : best - second;
    }
    return bidx;
}

typedef struct {
    int d, dp, c, K, L, P;
    float *w_in;  /* [L][c][dp]  zero padded */
    float *w_outT;/* [L][c][dp]  transposed + zero padded */
    float *b_out; /* [L][dp] */
    float *cbT;   /* [L][c][K] */
    const float *b_in; /* [L][c] */
    const float *cb;   /* [L][K][c] */
} packed_t;

static void pack(packed_t *p, int d, int c, int K, int L, const float *w_in, const float *b_in,
                 const float *cb, const float *w_out, const float *b_out)
{
    p->d = d; p->c = c; p->K = K; p->L = L;
    p->dp = pad_dim(d); p->P = p->dp / 256;
    p->b_in = b_in; p->cb = cb;
    size_t dp = p->dp;
    p->w_in = (float *)calloc((size_t)L * c * dp, sizeof(float));
    p->w_outT = (float *)calloc((size_t)L * c * dp, sizeof(float));
    p->b_out = (float *)calloc((size_t)L * dp, sizeof(float));
    p->cbT = (float *)calloc((size_t)L * c * K, sizeof(float));
    for (int l = 0; l < L; ++l) {
        for (int j = 0; j < c; ++j)
            memcpy(p->w_in + ((size_t)l * c + j) * dp, w_in + ((size_t)l * c + j) * d,
                   sizeof(float) * d);
        for (int e = 0; e < d; ++e) {
            for (int k = 0; k < c; ++k)
                p->w_outT[((size_t)l * c + k) * dp + e] = w_out[((size_t)l * d + e) * c + k];
            p->b_out[(size_t)l * dp + e] = b_out[(size_t)l * d + e];
        }
        for (int n = 0; n < K; ++n)
            for (int k = 0; k < c; ++k)
                p->cbT[((size_t)l * c + k) * K + n] = cb[((size_t)l * K + n) * c + k];
    }
}

static void unpack(packed_t *p)
{
    free(p->w_in); free(p->w_outT); free(p->b_out); free(p->cbT);
}

/* Canonical lane<->element map.  Slice s owns elements [s*64P, (s+1)*64P).
 * Inside a slice the P element steps of a lane are grouped so that a lane's elements are contiguous
 * in memory in runs of 4, then 2, then 1 (wide loads/stores on the GPU):
 *   steps i <  4*floor(P/4)            : block of 256 elements per 4 steps, lane owns 4*lane + (i%4)
 *   next 2 steps if at least 2 remain  : block of 128 elements,            lane owns 2*lane + j
 *   last step if one remains           : block of 64 elements,             lane owns lane
 * The fma chain of a lane still runs over i = 0..P-1 in ascending order. */
static inline int elem_of(int P, int s, int i, int lane)
{
    const int base = s * 64 * P;
    const int q4 = (P / 4) * 4;
    if (i < q4) return base + 256 * (i / 4) + 4 * lane + (i % 4);
    const int rem = P - q4, j = i - q4;
    if (rem >= 2 && j < 2) return base + 64 * q4 + 2 * lane + j;
    return base + 64 * q4 + (rem >= 2 ? 128 : 0) + lane;
}

/* lin_in of one token at one level in the canonical order (see header). */
static void lin_in_canonical(const packed_t *p, int l, const float *r, float *q)
{
    const int P = p->P, dp = p->dp, c = p->c;
    for (int j = 0; j < c; ++j) {
        const float *w = p->w_in + ((size_t)l * c + j) * dp;
        float t[NSLICE];
        for (int s = 0; s < NSLICE; ++s) {
            float part[NLANE];
            for (int ln = 0; ln < NLANE; ++ln) part[ln] = 0.0f;
            for (int i = 0; i < P; ++i)
                for (int ln = 0; ln < NLANE; ++ln) {
                    const int e = elem_of(P, s, i, ln);
                    part[ln] = fmaf(r[e], w[e], part[ln]);
                }
            t[s] = lane_tree(part);
        }
        q[j] = ((t[0] + t[1]) + (t[2] + t[3])) + p->b_in[(size_t)l * c + j];
    }
}

/*
 * RQAE.forward (rqae/model.py:199-230) on x[n_tokens][d] fp32.
 *   q_out   [n_tokens][d] fp32 or NULL
 *   indices [n_tokens][levels] int64
 *   margins [n_tokens][levels] fp32 or NULL (top-1 minus the best row that is not a copy of the winner)
 * levels = min(L, max_layers) is passed by the caller.
 */
void rqae_oracle_forward(const float *x, int64_t n_tokens, int d, int c, int K, int L, int levels,
                         const float *w_in, const float *b_in, const float *cb,
                         const float *w_out, const float *b_out, float *q_out, int64_t *indices,
                         float *margins, int n_threads)
{
    packed_t p;
    pack(&p, d, c, K, L, w_in, b_in, cb, w_out, b_out);
    const int dp = p.dp;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        float *r = (float *)malloc(sizeof(float) * dp);
        float *out = (float *)malloc(sizeof(float) * dp);
        float *y = (float *)malloc(sizeof(float) * dp);
        float *sims = (float *)malloc(sizeof(float) * K);
        float *q = (float *)malloc(sizeof(float) * c * 3);
        float *xn = q + c, *cv = q + 2 * c;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < n_tokens; ++n) {
            memcpy(r, x + n * d, sizeof(float) * d);
            for (int e = d; e < dp; ++e) r[e] = 0.0f;
            for (int l = 0; l < levels; ++l) {
                lin_in_canonical(&p, l, r, q);                       /* model.py:211 */
                const float nrm = aten_norm2(q, c);                    /* model.py:188 */
                for (int k = 0; k < c; ++k) xn[k] = q[k] / nrm;
                const float *cbT = p.cbT + (size_t)l * c * K;
                int64_t idx = nearest_row(xn, cbT, c, K, sims,        /* model.py:190-191 */
                                          margins ? margins + n * levels + l : NULL);
                indices[n * levels + l] = idx;
                const float *row = p.cb + ((size_t)l * K + idx) * c;  /* model.py:192 */
                for (int k = 0; k < c; ++k) {                          /* model.py:218-220 */
                    float diff = row[k] - q[k];
                    cv[k] = q[k] + diff;
                }
                for (int e = 0; e < dp; ++e) y[e] = 0.0f;              /* model.py:221 */
                for (int k = 0; k < c; ++k) {
                    const float ck = cv[k];
                    const float *w = p.w_outT + ((size_t)l * c + k) * dp;
                    for (int e = 0; e < dp; ++e) y[e] = fmaf(ck, w[e], y[e]);
                }
                const float *bo = p.b_out + (size_t)l * dp;
                if (l == 0) {
                    for (int e = 0; e < dp; ++e) {
                        float ye = y[e] + bo[e];
                        r[e] = r[e] - ye;                               /* model.py:223 */
                        out[e] = 0.0f + ye;                             /* model.py:201,224 */
                    }
                } else {
                    for (int e = 0; e < dp; ++e) {
                        float ye = y[e] + bo[e];
                        r[e] = r[e] - ye;
                        out[e] = out[e] + ye;
                    }
                }
            }
            if (q_out) memcpy(q_out + n * d, out, sizeof(float) * d);
        }
        free(r); free(out); free(y); free(sims); free(q);
    }
    unpack(&p);
}

/*
 * Adjudicator for SURVEY 8(c)'s Tier-2 policy: the level loop of RQAE.forward (rqae/model.py:199-230) in DOUBLE
 * precision along a GIVEN code path.  For token n and level l the residual is advanced with codes[n][l] (whatever
 * produced them), and the level reports
 *   gap[n][l]    = max_k cos(q_l, codebook[l][k]) - cos(q_l, codebook[l][codes[n][l]])   (>= 0; 0 = the code is the
 *                  double-precision argmax or a copy of it)
 *   argmax[n][l] = lowest index attaining the double-precision maximum
 * so a code path is "an argmax path up to rounding" exactly when every gap is at rounding level.  A zero or
 * non-finite projection reports gap 0 for code 0 (torch.argmax over NaNs) and +inf otherwise.
 */
void rqae_oracle_adjudicate(const float *x, int64_t n_tokens, int d, int c, int K, int L, int levels,
                            const float *w_in, const float *b_in, const float *cb, const float *w_out,
                            const float *b_out, const int64_t *codes, double *gap, int64_t *argmax, int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)L;
#pragma omp parallel
    {
        double *r = (double *)malloc(sizeof(double) * d);
        double *q = (double *)malloc(sizeof(double) * 2 * c);
        double *cv = q + c;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < n_tokens; ++n) {
            for (int e = 0; e < d; ++e) r[e] = (double)x[n * d + e];
            for (int l = 0; l < levels; ++l) {
                double n2 = 0.0;
                for (int j = 0; j < c; ++j) {
                    const float *w = w_in + ((size_t)l * c + j) * d;
                    double s = 0.0;
                    for (int e = 0; e < d; ++e) s += (double)w[e] * r[e];
                    q[j] = s + (double)b_in[(size_t)l * c + j];
                    n2 += q[j] * q[j];
                }
                const double nrm = sqrt(n2);
                const int64_t code = codes[n * levels + l];
                const float *cbl = cb + (size_t)l * K * c;
                if (!(nrm > 0.0) || !isfinite(nrm)) {
                    gap[n * levels + l] = code == 0 ? 0.0 : INFINITY;
                    argmax[n * levels + l] = 0;
                } else {
                    double best = -INFINITY, mine = 0.0;
                    int64_t bi = 0;
                    for (int k = 0; k < K; ++k) {
                        double s = 0.0;
                        for (int j = 0; j < c; ++j) s += (double)cbl[(size_t)k * c + j] * q[j];
                        s /= nrm;
                        if (s > best) { best = s; bi = k; }
                        if (k == code) mine = s;
                    }
                    gap[n * levels + l] = best - mine;
                    argmax[n * levels + l] = bi;
                }
                const float *row = cbl + (size_t)code * c;
                for (int j = 0; j < c; ++j) cv[j] = q[j] + ((double)row[j] - q[j]);
                for (int e = 0; e < d; ++e) {
                    const float *w = w_out + ((size_t)l * d + e) * c;
                    double y = (double)b_out[(size_t)l * d + e];
                    for (int j = 0; j < c; ++j) y += (double)w[j] * cv[j];
                    r[e] -= y;
                }
            }
        }
        free(r); free(q);
    }
}

/*
 * RQAE.quantize (rqae/model.py:187-197) for one level: pin [n][c] fp32 ->
 * cv [n][c] (exact codebook rows), idx [n] int64.  cb_l is that level's [K][c].
 */
void rqae_oracle_quantize(const float *pin, int64_t n, int c, int K, const float *cb_l, float *cv,
                          int64_t *idx, float *margins, int n_threads)
{
    float *cbT = (float *)malloc(sizeof(float) * c * K);
    for (int r = 0; r < K; ++r)
        for (int k = 0; k < c; ++k) cbT[(size_t)k * K + r] = cb_l[(size_t)r * c + k];
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        float *sims = (float *)malloc(sizeof(float) * K);
        float *xn = (float *)malloc(sizeof(float) * c);
#pragma omp for schedule(static)
        for (int64_t t = 0; t < n; ++t) {
            const float *q = pin + t * c;
            const float nrm = aten_norm2(q, c);
            for (int k = 0; k < c; ++k) xn[k] = q[k] / nrm;
            int64_t b = nearest_row(xn, cbT, c, K, sims, margins ? margins + t : NULL);
            idx[t] = b;
            memcpy(cv + t * c, cb_l + (size_t)b * c, sizeof(float) * c);
        }
        free(sims); free(xn);
    }
    free(cbT);
}

/*
 * RQAE.decode (rqae/model.py:232-252): codebook[0][idx] for every level
 * (model.py:234), then sum over levels ascending of lin_out_l(cv_l), first
 * included level assigns, later ones += (model.py:244-247).  level_mask[l]==0
 * skips level l (the `layers` whitelist); NULL keeps all.  idx is [n][idx_levels].
 * Levels >= L are ignored like the reference's enumerate(self.layers).
 * Returns the number of levels summed (0 => reference returns None; out untouched).
 */
int rqae_oracle_decode(const int64_t *idx, int64_t n_tokens, int idx_levels, int d, int c, int K,
                       int L, const float *cb0, const float *w_out, const float *b_out,
                       const uint8_t *level_mask, float *out, int n_threads)
{
    const int dp = pad_dim(d);
    int used = 0;
    (void)K;
    const int lv = idx_levels < L ? idx_levels : L;
    for (int l = 0; l < lv; ++l)
        if (!level_mask || level_mask[l]) ++used;
    if (!used) return 0;
    float *wT = (float *)calloc((size_t)L * c * dp, sizeof(float));
    for (int l = 0; l < L; ++l)
        for (int e = 0; e < d; ++e)
            for (int k = 0; k < c; ++k)
                wT[((size_t)l * c + k) * dp + e] = w_out[((size_t)l * d + e) * c + k];
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * dp);
        float *y = (float *)malloc(sizeof(float) * dp);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < n_tokens; ++n) {
            int first = 1;
            for (int l = 0; l < lv; ++l) {
                if (level_mask && !level_mask[l]) continue;
                const float *row = cb0 + (size_t)idx[n * idx_levels + l] * c;
                for (int e = 0; e < d; ++e) y[e] = 0.0f;
                for (int k = 0; k < c; ++k) {
                    const float ck = row[k];
                    const float *w = wT + ((size_t)l * c + k) * dp;
                    for (int e = 0; e < d; ++e) y[e] = fmaf(ck, w[e], y[e]);
                }
                const float *bo = b_out + (size_t)l * d;
                if (first) {
                    for (int e = 0; e < d; ++e) acc[e] = y[e] + bo[e];
                    first = 0;
                } else {
                    for (int e = 0; e < d; ++e) {
                        float ye = y[e] + bo[e];
                        acc[e] = acc[e] + ye;
                    }
                }
            }
            memcpy(out + n * d, acc, sizeof(float) * d);
        }
        free(acc); free(y);
    }
    free(wT);
    return used;
}

/*
 * The cached tables of the model (rqae/model.py:133-178), the rows SURVEY 8(f) ranks f4.
 *
 * codebook_sims (model.py:141-142): F.normalize(cb0, dim=-1) @ F.normalize(cb0, dim=-1).T.
 * F.normalize divides by max(||row||, 1e-12) with the norm ordered as x.norm(dim=-1) (aten_norm2
 * above); the (K,c)@(c,K) product is a k-ordered fma chain from +0, like `x @ codebook.T` of
 * quantize.  Output fp32; the caller rounds to fp16 (model.py:142).
 */
void rqae_oracle_codebook_sims(const float *cb0, int K, int c, float *out)
{
    float *unit = (float *)malloc(sizeof(float) * (size_t)K * c);
    for (int n = 0; n < K; ++n) {
        float nr = aten_norm2(cb0 + (size_t)n * c, c);
        if (!(nr > 1e-12f)) nr = 1e-12f;
        for (int k = 0; k < c; ++k) unit[(size_t)n * c + k] = cb0[(size_t)n * c + k] / nr;
    }
#pragma omp parallel for schedule(static)
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            float acc = 0.0f;
            for (int k = 0; k < c; ++k) acc = fmaf(unit[(size_t)a * c + k], unit[(size_t)b * c + k], acc);
            out[(size_t)a * K + b] = acc;
        }
    free(unit);
}

/*
 * subfeatures (model.py:145-156): out[l][n][:] = lin_out_l(codebook[l][n]) -- the same chain as the
 * level step of decode (fma chain over k from +0, then + bias), on codebook[l] rather than codebook[0].
 */
void rqae_oracle_subfeatures(const float *cb, const float *w_out, const float *b_out, int L, int K,
                             int c, int d, float *out)
{
#pragma omp parallel for schedule(static) collapse(2)
    for (int l = 0; l < L; ++l)
        for (int n = 0; n < K; ++n) {
            const float *row = cb + ((size_t)l * K + n) * c;
            float *o = out + ((size_t)l * K + n) * d;
            for (int e = 0; e < d; ++e) {
                const float *w = w_out + ((size_t)l * d + e) * c;
                float y = 0.0f;
                for (int k = 0; k < c; ++k) y = fmaf(row[k], w[k], y);
                o[e] = y + b_out[(size_t)l * d + e];
            }
        }
}

/*
 * subfeature_sims (model.py:158-168) and layer_norms (model.py:170-178).  The reference's
 * (K,d)@(d,K) product runs in whatever order the BLAS picks, so there are no reference bits to follow:
 * this is the value in double precision from the fp32 subfeatures above (cosine of rows, each row divided
 * by max(norm, 1e-12)), which the fp16 result of the reference must equal to within its own fp32 rounding.
 * sub is [K][d] for one level; out fp64 [K][K].
 */
void rqae_oracle_subfeature_sims_level(const float *sub, int K, int d, double *out)
{
    double *inv = (double *)malloc(sizeof(double) * K);
    for (int n = 0; n < K; ++n) {
        double s = 0.0;
        for (int e = 0; e < d; ++e) s += (double)sub[(size_t)n * d + e] * sub[(size_t)n * d + e];
        s = sqrt(s);
        inv[n] = 1.0 / (s > 1e-12 ? s : 1e-12);
    }
#pragma omp parallel for schedule(static)
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            double s = 0.0;
            for (int e = 0; e < d; ++e) s += (double)sub[(size_t)a * d + e] * sub[(size_t)b * d + e];
            out[(size_t)a * K + b] = s * inv[a] * inv[b];
        }
    free(inv);
}

/* layer_norms[l] = mean over the c columns of ||W_out[l][:, k]||_2 (model.py:174), in double. */
void rqae_oracle_layer_norms(const float *w_out, int L, int c, int d, double *out)
{
    for (int l = 0; l < L; ++l) {
        double m = 0.0;
        for (int k = 0; k < c; ++k) {
            double s = 0.0;
            for (int e = 0; e < d; ++e) {
                double v = w_out[((size_t)l * d + e) * c + k];
                s += v * v;
            }
            m += sqrt(s);
        }
        out[l] = m / c;
    }
}

int rqae_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
