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
// scratch, per level (m = c+1): G [m][m] | inv [K] = 1/max(|u_n|, 1e-12) | gv [K][m] = (G v_n) * inv[n] | norm, doubles.
__host__ __device__ inline size_t tables_level_doubles(int K, int c)
{
    const size_t m = (size_t)c + 1;
    return m * m + (size_t)K + (size_t)K * m + 1;   // ... | the level's layer norm (as the fp32 value, in a double)
}

// The table element from its fp64 value: fp16 as RQAE.subfeature_sims holds it, or -- scaled -- what the demo server
// keeps after `sims *= layer_norms` (demo/server/server.py:108): the fp16 value times the fp32 norm, rounded to fp16.
__device__ __forceinline__ _Float16 sims_finish(double v, float norm, bool scaled)
{
    const _Float16 h = (_Float16)(float)v;
    return scaled ? (_Float16)((float)h * norm) : h;
}

// tables_prep_kernel: blockIdx.x = level; blockIdx.y = a chunk of 1024 codebook rows (every chunk's block forms G
// itself: d*(c+1)^2/2 fmas, cheaper than a second launch or a grid-wide wait).
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
        if (blockIdx.y == 0) {
            G[pa * m + pb] = s;
            G[pb * m + pa] = s;
        }
    }
    __syncthreads();
    if (tid == 0 && blockIdx.y == 0) {
        double s = 0.0;
        for (int k = 0; k < c; ++k) s += sqrt(sG[k][k]);
        const float nrm = (float)(s / c);
        if (layer_norms) layer_norms[l] = nrm;
        G[m * m + K + (size_t)K * m] = (double)nrm;
    }
    double *inv = G + m * m, *gv = inv + K;
    for (int n = blockIdx.y * 1024 + tid; n < min(K, (int)(blockIdx.y + 1) * 1024); n += 256) {
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

// One element of the table from the scratch block (the slow path: ragged pieces and tiny codebooks).
__device__ __forceinline__ _Float16 sims_element(const float *__restrict__ cbl, const double *__restrict__ inv,
                                                 const double *__restrict__ gv, int c, int row, int col, float norm,
                                                 bool scaled)
{
    const int m = c + 1;
    const double iv = inv[col];
    double acc = 0.0;
    for (int a = 0; a < m; ++a) acc = fma(gv[(size_t)row * m + a], (a < c ? (double)cbl[(size_t)col * c + a] : 1.0) * iv, acc);
    return sims_finish(acc, norm, scaled);
}

// subfeature_sims_kernel: a block owns TR rows x up to 256 pieces of one level; a piece is NC consecutive fp16 of
// the output at an NC*2-byte aligned address (NC = 8: one 16-byte store).  Rows are K*2 bytes apart in memory
// (1250 at K = 625), so the columns a piece covers shift from row to row: row r of level l starts mis =
// (l*K*K + r*K) mod NC elements into a piece, and piece p of that row covers columns NC*p - mis .. + NC-1.  mis
// repeats every PR = NC / gcd(K, NC) rows, so a lane owns one piece index p, takes the rows class by class
// (rows i0, i0+PR, ... share mis), loads the NC columns' v * inv for the class into registers (doubles, from the
// block's LDS copy of the codebook segment) and then does one row per step: m fp64 fmas per element against the
// row's G v * inv (LDS), fp64 -> fp32 -> fp16, one aligned store.  No barrier after the prologue.
// The piece that straddles rows r-1 and r (when mis > 0) and the ragged end of the level's last row are written
// element by element at the end (blockIdx.z == 0).  MB >= c+1 is the padded length of v (zeros beyond c+1).
template <int MB, int NC>
__global__ void __launch_bounds__(256)
subfeature_sims_kernel(const float *__restrict__ cb, const double *__restrict__ scratch, int K, int c, int TR,
                       int PR, int ncol_lds, int scaled, _Float16 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int l = blockIdx.y, row0 = blockIdx.x * TR, P0 = blockIdx.z * 256, tid = threadIdx.x, m = c + 1;
    const int nrow = min(TR, K - row0);
    const size_t EL = (size_t)l * K * K;                                // element offset of the level
    const int maxnp = (K + 2 * NC - 2) / NC;                            // pieces a row can touch
    const int ncg = min(256, maxnp - P0);
    const int c_lo = max(0, NC * P0 - NC), c_hi = min(K, NC * (P0 + ncg));   // columns this block can need
    double *gvs = reinterpret_cast<double *>(smem_raw) + MB;           // [-1 .. TR-1][MB]: row0-1 first (ragged pieces)
    double *invs = gvs + (size_t)TR * MB;                               // [ncol_lds]
    // the codebook segment, transposed, column x at word x + x/NC of its plane: a lane reads columns NC*p+q for fixed
    // q, so consecutive lanes are NC+1 words apart and the reads are free of bank conflicts (likewise invs, doubles)
    const int plane = ncol_lds + ncol_lds / NC + 1;
    float *cbs = reinterpret_cast<float *>(invs + plane);               // [c][plane]
    auto pad = [](int x) { return x + x / NC; };
    const double *G = scratch + (size_t)l * tables_level_doubles(K, c), *inv = G + m * m, *gv = inv + K;
    const float norm = scaled ? (float)gv[(size_t)K * m] : 1.0f;
    const float *cbl = cb + (size_t)l * K * c;
    for (int e = tid - MB; e < nrow * MB; e += 256) {
        const int r = e < 0 ? -1 : e / MB, a = e < 0 ? e + MB : e % MB;
        gvs[e] = a < m && row0 + r >= 0 ? gv[(size_t)(row0 + r) * m + a] : 0.0;
    }
    for (int i = tid; i < c_hi - c_lo; i += 256) invs[pad(i)] = inv[c_lo + i];
    for (int i = tid; i < (c_hi - c_lo) * c; i += 256) cbs[(i % c) * plane + pad(i / c)] = cbl[(size_t)c_lo * c + i];
    __syncthreads();
    const int nsub = 256 / ncg, cg = tid % ncg, sub = tid / ncg;
    if (sub < nsub) {
        const int p = P0 + cg;
        for (int i0 = 0; i0 < PR && i0 < nrow; ++i0) {
            const int mis = (int)((EL + (size_t)(row0 + i0) * K) % NC);
            const int cfirst = NC * p - mis;
            if (cfirst < 0 || cfirst + NC > K) continue;               // ragged pieces: below
            double v[NC][MB];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int li = pad(cfirst + j - c_lo);
                const double iv = invs[li];
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const float x = cbs[min(a, c - 1) * plane + li];
                    v[j][a] = (a < c ? (double)x : (a == c ? 1.0 : 0.0)) * iv;
                }
            }
            _Float16 *dst = out + EL + (size_t)row0 * K + cfirst;       // + i * K: NC-aligned for rows of this class
            for (int i = i0 + PR * sub; i < nrow; i += PR * nsub) {
                const double *g = gvs + i * MB;
                double acc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[j] = 0.0;
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const double ga = g[a];
#pragma unroll
                    for (int j = 0; j < NC; ++j) acc[j] = fma(ga, v[j][a], acc[j]);
                }
                typedef _Float16 hvec __attribute__((ext_vector_type(NC)));
                hvec h;
#pragma unroll
                for (int j = 0; j < NC; ++j) h[j] = sims_finish(acc[j], norm, scaled != 0);
                *reinterpret_cast<hvec *>(dst + (size_t)i * K) = h;
            }
        }
    }
    if (blockIdx.z == 0) {
        auto element = [&](int i, int col) -> _Float16 {               // row row0 + i (i >= -1) x column col
            const bool in_lds = col >= c_lo && col < c_hi;
            const double iv = in_lds ? invs[pad(col - c_lo)] : inv[col];
            double acc = 0.0;
            for (int a = 0; a < m; ++a) {
                const double x = a < c ? (double)(in_lds ? cbs[a * plane + pad(col - c_lo)] : cbl[(size_t)col * c + a]) : 1.0;
                acc = fma(gvs[i * MB + a], x * iv, acc);
            }
            return sims_finish(acc, norm, scaled != 0);
        };
        for (int t = tid; t < nrow * NC; t += 256) {
            const int i = t / NC, j = t % NC, r = row0 + i;
            const size_t e_r = EL + (size_t)r * K;
            const int mis = (int)(e_r % NC);
            if (mis > 0 && (j >= mis || r > 0))                         // the piece shared with row r-1
                out[e_r - mis + j] = j < mis ? element(i - 1, K - mis + j) : element(i, j - mis);
            const int tailn = (int)((e_r + K) % NC);
            if (r == K - 1 && j < tailn)                                // the level's ragged end
                out[e_r + K - tailn + j] = element(i, K - tailn + j);
        }
    }
}

// K < 64: one lane per element.
__global__ void __launch_bounds__(256)
subfeature_sims_small_kernel(const float *__restrict__ cb, const double *__restrict__ scratch, int K, int c,
                             int scaled, _Float16 *__restrict__ out)
{
    const int l = blockIdx.y, m = c + 1;
    const double *G = scratch + (size_t)l * tables_level_doubles(K, c), *inv = G + m * m, *gv = inv + K;
    const float norm = scaled ? (float)gv[(size_t)K * m] : 1.0f;
    const float *cbl = cb + (size_t)l * K * c;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < K * K; e += gridDim.x * 256)
        out[(size_t)l * K * K + e] = sims_element(cbl, inv, gv, c, e / K, e % K, norm, scaled != 0);
}

}  // namespace
