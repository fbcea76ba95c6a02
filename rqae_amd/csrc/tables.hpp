// tables.hpp -- the model's cached tables: codebook_sims, subfeatures, subfeature_sims and
// layer_norms of rqae/model.py:133-178; included by rqae_hip.hip.
#pragma once
#include "rqae_common.hpp"

namespace {

// =============================================================== codebook_sims (rqae/model.py:133-143)
// out[a][b] = fp16( sum_k unit[a][k] * unit[b][k] ), unit = F.normalize(codebook[0], dim=-1): the row divided by
// max(||row||, 1e-12) with the norm in x.norm(dim=-1)'s order (aten_norm2), the product a k-ordered fma chain from
// +0 -- the orders of torch-CPU, so the table has the reference's bits.  A block owns 32 rows x 256 columns; a
// lane owns one column, keeps its unit row in registers and reads the tile's unit rows from LDS (broadcast).
__global__ void __launch_bounds__(256)
codebook_sims_kernel(const float *__restrict__ cb, int K, int c, _Float16 *__restrict__ out)
{
    constexpr int C = MAX_C, TR = 32;
    __shared__ float urow[TR][C];
    auto unit_row = [&](int n, float (&u)[C]) {
#pragma unroll
        for (int k = 0; k < C; ++k) u[k] = k < c ? cb[(size_t)n * c + k] : 0.0f;
        float nr = aten_norm2<C>(u, c);
        if (!(nr > 1e-12f)) nr = 1e-12f;
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (k < c) u[k] = u[k] / nr;
    };
    const int row0 = blockIdx.y * TR, col = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < TR && row0 + threadIdx.x < K) {
        float u[C];
        unit_row(row0 + threadIdx.x, u);
#pragma unroll
        for (int k = 0; k < C; ++k) urow[threadIdx.x][k] = u[k];
    }
    __syncthreads();
    if (col >= K) return;
    float mine[C];
    unit_row(col, mine);
    const int nr = min(TR, K - row0);
    for (int r = 0; r < nr; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (k < c) acc = fmaf(urow[r][k], mine[k], acc);
        out[(size_t)(row0 + r) * K + col] = (_Float16)acc;
    }
}

// =============================================================== subfeatures (rqae/model.py:145-156)
// out[l][n][j] = (fma chain over k from +0 of codebook[l][n][k] * W_out[l][j][k]) + b_out[l][j] -- the level step
// of decode applied to every codebook row; same bits as the reference on CPU.  HBM-write-bound (L*K*d*4 bytes:
// 5.9 GB at the pretrained shape).  A lane owns V consecutive j (its V*c weights and V biases stay in registers),
// a wave walks codebook rows (uniform -> scalar loads) and stores V*64 consecutive floats per row.
template <int CB, int V>
__global__ void __launch_bounds__(256)
subfeatures_kernel(const float *__restrict__ cb, const float *__restrict__ w_out, const float *__restrict__ b_out,
                   int K, int c, int d, int rows_per_block, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = blockIdx.z;
    const int j0 = (blockIdx.x * 64 + lane) * V;
    if (j0 >= d) return;
    float w[V][CB], b[V];
    const float *wl = w_out + ((size_t)l * d + j0) * c;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        b[v] = b_out[(size_t)l * d + j0 + v];
#pragma unroll
        for (int k = 0; k < CB; ++k) w[v][k] = k < c ? wl[v * c + k] : 0.0f;
    }
    const int n0 = blockIdx.y * rows_per_block, n1 = min(K, n0 + rows_per_block);
    for (int n = n0 + wave; n < n1; n += 4) {
        const float *row = cb + ((size_t)l * K + n) * c;
        float x[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) x[k] = k < c ? row[k] : 0.0f;
        float y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < CB; ++k)
                if (k < c) acc = fmaf(x[k], w[v][k], acc);
            y[v] = acc + b[v];
        }
        float *o = out + ((size_t)l * K + n) * d + j0;
        if constexpr (V == 4) {
            f4 t = {y[0], y[1], y[2], y[3]};
            __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(o));
        } else {
            o[0] = y[0];
        }
    }
}

// =============================================================== subfeature_sims, layer_norms (rqae/model.py:158-178)
// The reference materialises subfeatures (L,K,d), normalises the rows and multiplies (K,d)@(d,K) per level:
// 2*K*K*d flops per level (1.8 TFLOP at the pretrained shape) in whatever order the BLAS picks.  A subfeature is
// affine in its codebook row, u_n = A v_n with A = [W_out[l] | b_out[l]] (d x (c+1)) and v_n = [codebook[l][n]; 1],
// so every inner product is a (c+1) x (c+1) quadratic form:
//     <u_a, u_b> = v_a' G v_b,   G = A'A,   sims[a][b] = v_a' G v_b / (max(|u_a|, 1e-12) max(|u_b|, 1e-12)).
// G costs d*(c+1)^2 flops per level, the table K*K*(c+1): the kernel is bound by the fp16 table it writes
// (L*K*K*2 bytes, 0.8 GB at the pretrained shape).  Everything is fp64 up to the final fp64 -> fp32 -> fp16
// conversion, so the result is the cosine of the exact affine images to ~1e-15 -- closer to the exact value than
// the reference's fp32 product of fp32-rounded subfeatures; tests/ bound the distance to the reference capture by
// the reference's own rounding (<= 1 fp16 ulp, > 99.9 % identical).  layer_norms[l] = mean_k sqrt(G[k][k]), k < c.
//
// scratch, per level (m = c+1): G [m][m] | inv [K] = 1/max(|u_n|, 1e-12) | gv [K][m] = (G v_n) * inv[n], doubles.
__host__ __device__ inline size_t tables_level_doubles(int K, int c)
{
    const size_t m = (size_t)c + 1;
    return m * m + (size_t)K + (size_t)K * m;
}

// tables_prep_kernel: one block per level.
__global__ void __launch_bounds__(256)
tables_prep_kernel(const float *__restrict__ cb, const float *__restrict__ w_out, const float *__restrict__ b_out,
                   int K, int c, int d, double *__restrict__ scratch, float *__restrict__ layer_norms)
{
    constexpr int M = MAX_C + 1, NPMAX = M * (M + 1) / 2;
    __shared__ float tile[256][M];
    __shared__ double part[256];
    __shared__ double sG[M][M];
    const int l = blockIdx.x, tid = threadIdx.x, m = c + 1, np = m * (m + 1) / 2;
    static_assert(NPMAX <= 256, "one thread per pair of columns");
    const int groups = 256 / np;                    // row groups per pair
    const int pair = tid % np, grp = tid / np;
    int pa = 0, pb = 0;
    {   // pair -> (a <= b), row-major over the upper triangle
        int q = pair;
        while (q >= m - pa) { q -= m - pa; ++pa; }
        pb = pa + q;
    }
    double acc = 0.0;
    for (int j0 = 0; j0 < d; j0 += 256) {
        const int nrow = min(256, d - j0);
        __syncthreads();
        for (int e = tid; e < nrow * c; e += 256) tile[e / c][e % c] = w_out[((size_t)l * d + j0) * c + e];
        if (tid < nrow) tile[tid][c] = b_out[(size_t)l * d + j0 + tid];
        __syncthreads();
        if (grp < groups)
            for (int r = grp; r < nrow; r += groups) acc = fma((double)tile[r][pa], (double)tile[r][pb], acc);
    }
    part[tid] = acc;
    __syncthreads();
    double *G = scratch + (size_t)l * tables_level_doubles(K, c);
    if (tid < np) {
        double s = 0.0;
        for (int g = 0; g < groups; ++g) s += part[g * np + tid];
        sG[pa][pb] = s;
        sG[pb][pa] = s;
        G[pa * m + pb] = s;
        G[pb * m + pa] = s;
    }
    __syncthreads();
    if (tid == 0 && layer_norms) {
        double s = 0.0;
        for (int k = 0; k < c; ++k) s += sqrt(sG[k][k]);
        layer_norms[l] = (float)(s / c);
    }
    double *inv = G + m * m, *gv = inv + K;
    for (int n = tid; n < K; n += 256) {
        const float *row = cb + ((size_t)l * K + n) * c;
        double v[M];
#pragma unroll
        for (int a = 0; a < M; ++a) v[a] = a < c ? (double)row[a] : (a == c ? 1.0 : 0.0);
        double q = 0.0;
        double *gvn = gv + (size_t)n * m;
#pragma unroll 1
        for (int a = 0; a < m; ++a) {           // G stays in LDS: unrolling both loops would hoist m*m doubles
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < M; ++b)
                if (b < m) t = fma(sG[a][b], v[b], t);
            gvn[a] = t;
            q = fma(a < c ? (double)row[a] : 1.0, t, q);
        }
        const double nr = sqrt(q > 0.0 ? q : 0.0);
        const double iv = 1.0 / (nr > 1e-12 ? nr : 1e-12);
        inv[n] = iv;
#pragma unroll 1
        for (int a = 0; a < m; ++a) gvn[a] *= iv;
    }
}

// subfeature_sims_kernel: a block owns TR rows x one segment of up to 256*NC columns of one level.  A lane owns NC
// consecutive columns (their v and inv in registers as doubles) and every nsub-th row; rows are done RC at a
// time: the lanes put their fp16 results into an LDS image of the RC row segments, laid out so that LDS and
// global addresses agree modulo 16 (rows are K*2 bytes apart in memory -- 1250 at K = 625 -- so no per-lane
// store could be aligned), then the block writes every segment with aligned 16-byte stores and finishes its ragged
// head and tail with 2-byte ones.  MB >= c+1 is the padded length of v (zeros beyond c+1).
template <int MB, int NC>
__global__ void __launch_bounds__(256)
subfeature_sims_kernel(const float *__restrict__ cb, const double *__restrict__ scratch, int K, int c, int TR,
                       int RC, _Float16 *__restrict__ out)
{
    constexpr int SEG = 256 * NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int l = blockIdx.y, row0 = blockIdx.x * TR, seg0 = blockIdx.z * SEG, tid = threadIdx.x, m = c + 1;
    const int nrow = min(TR, K - row0), W = min(SEG, K - seg0);        // this block's rows and columns
    const int SR = ((W * 2 + 15) & ~15) + 16;                           // LDS bytes per row segment
    double *gvs = reinterpret_cast<double *>(smem_raw);                 // [TR][MB]
    unsigned char *img = smem_raw + (size_t)TR * MB * sizeof(double);   // [RC][SR]
    const double *G = scratch + (size_t)l * tables_level_doubles(K, c), *inv = G + m * m, *gv = inv + K;
    const float *cbl = cb + (size_t)l * K * c;
    for (int e = tid; e < nrow * MB; e += 256) {
        const int r = e / MB, a = e % MB;
        gvs[e] = a < m ? gv[(size_t)(row0 + r) * m + a] : 0.0;
    }
    const int ncg = (W + NC - 1) / NC;                                  // <= 256
    const int nsub = 256 / ncg, cg = tid % ncg, sub = tid / ncg;
    const bool worker = sub < nsub;
    const int col0 = seg0 + cg * NC;
    double v[NC][MB], iv[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int col = min(col0 + j, K - 1);
        iv[j] = inv[col];
#pragma unroll
        for (int a = 0; a < MB; ++a) {
            const float x = cbl[min((size_t)col * c + a, (size_t)K * c - 1)];
            v[j][a] = a < c ? (double)x : (a == c ? 1.0 : 0.0);
        }
    }
    const int tpr = 256 / RC, crun = tid / tpr, clane = tid % tpr;     // copy-out: tpr lanes per row segment
    __syncthreads();
    for (int rc0 = 0; rc0 < nrow; rc0 += RC) {
        const int nrc = min(RC, nrow - rc0);
        if (worker)
            for (int r = sub; r < nrc; r += nsub) {
                const double *g = gvs + (rc0 + r) * MB;
                double acc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[j] = 0.0;
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const double ga = g[a];
#pragma unroll
                    for (int j = 0; j < NC; ++j) acc[j] = fma(ga, v[j][a], acc[j]);
                }
                const size_t e0 = ((size_t)l * K + row0 + rc0 + r) * K + seg0;      // element offset of the segment
                _Float16 *dst = reinterpret_cast<_Float16 *>(img + (size_t)r * SR) + (e0 & 7) + cg * NC;
#pragma unroll
                for (int j = 0; j < NC; ++j) dst[j] = (_Float16)(float)(acc[j] * iv[j]);
            }
        __syncthreads();
        if (crun < nrc) {
            const size_t e0 = ((size_t)l * K + row0 + rc0 + crun) * K + seg0;
            const int mis = (int)(e0 & 7);                               // elements before e0 in its 16-byte piece
            const int npiece = (mis + W + 7) / 8;
            _Float16 *gbase = out + (e0 - mis);                          // 16-byte aligned (out is)
            const unsigned char *lrow = img + (size_t)crun * SR;
            for (int pc = clane; pc < npiece; pc += tpr) {
                const int lo = pc * 8, hi = lo + 8;                      // elements of the piece, relative to gbase
                if (lo >= mis && hi <= mis + W) {
                    const f4 t = *reinterpret_cast<const f4 *>(lrow + (size_t)pc * 16);
                    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(gbase + lo));
                } else {
                    const _Float16 *lp = reinterpret_cast<const _Float16 *>(lrow) + lo;
                    for (int j = 0; j < 8; ++j)
                        if (lo + j >= mis && lo + j < mis + W) gbase[lo + j] = lp[j];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace
