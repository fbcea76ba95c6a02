/* A host program in plain C over include/rqae_hip.h: no Python, no torch in the process.
 * It is what a maintainer's cgo / JNI / ctypes stub would do: device memory from the HIP runtime,
 * plain pointers and sizes into librqae_hip.so.  tests/test_hip_parity.py::test_c_abi_from_a_c_program writes the
 * inputs, runs this program and compares its outputs with the oracle.
 *
 *   host_smoke <dir>     reads <dir>/meta.txt  (d c K L n levels), w_in.bin b_in.bin cb.bin w_out.bin b_out.bin x.bin
 *                        writes <dir>/idx.bin (int64), q.bin (fp32), dec.bin (fp32) -- the exact arm -- and idx_gram.bin, q_gram.bin,
 *                        lattice.txt -- the matrix-core arm with its per-call options struct and the lattice query
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rqae_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_RQAE(e) do { if ((e) != 0) { fprintf(stderr, "%s: %s\n", #e, rqae_hip_last_error()); return 3; } } while (0)

static void *slurp(const char *dir, const char *name, size_t bytes)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    void *buf = malloc(bytes);
    if (!f || !buf || fread(buf, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", path); exit(4); }
    fclose(f);
    return buf;
}

static int dump(const char *dir, const char *name, const void *dev, size_t bytes)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    void *host = malloc(bytes);
    if (hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(host, 1, bytes, f) != bytes) return 1;
    fclose(f);
    free(host);
    return 0;
}

static void *to_device(const void *host, size_t bytes)
{
    void *dev = NULL;
    if (hipMalloc(&dev, bytes) != hipSuccess || hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "device copy of %zu bytes failed\n", bytes);
        exit(5);
    }
    return dev;
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: host_smoke <dir>\n"); return 1; }
    const char *dir = argv[1];
    char path[4096];
    snprintf(path, sizeof path, "%s/meta.txt", dir);
    FILE *mf = fopen(path, "r");
    int d, c, K, L, n, levels;
    if (!mf || fscanf(mf, "%d %d %d %d %d %d", &d, &c, &K, &L, &n, &levels) != 6) { fprintf(stderr, "bad meta.txt\n"); return 1; }
    fclose(mf);
    if (rqae_hip_abi_version() != RQAE_HIP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    const size_t f = sizeof(float);
    float *w_in = to_device(slurp(dir, "w_in.bin", (size_t)L * c * d * f), (size_t)L * c * d * f);
    float *b_in = to_device(slurp(dir, "b_in.bin", (size_t)L * c * f), (size_t)L * c * f);
    float *cb = to_device(slurp(dir, "cb.bin", (size_t)L * K * c * f), (size_t)L * K * c * f);
    float *w_out = to_device(slurp(dir, "w_out.bin", (size_t)L * d * c * f), (size_t)L * d * c * f);
    float *b_out = to_device(slurp(dir, "b_out.bin", (size_t)L * d * f), (size_t)L * d * f);
    float *x = to_device(slurp(dir, "x.bin", (size_t)n * d * f), (size_t)n * d * f);

    const size_t pbytes = rqae_hip_packed_bytes(d, c, K, L);
    if (!pbytes) { fprintf(stderr, "packed_bytes: %s\n", rqae_hip_last_error()); return 3; }
    void *packed = NULL, *q = NULL, *idx = NULL, *dec = NULL;
    CHECK_HIP(hipMalloc(&packed, pbytes));
    CHECK_HIP(hipMalloc(&q, (size_t)n * d * f));
    CHECK_HIP(hipMalloc(&dec, (size_t)n * d * f));
    CHECK_HIP(hipMalloc(&idx, (size_t)n * levels * sizeof(int64_t)));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    CHECK_RQAE(rqae_hip_pack(packed, d, c, K, L, w_in, b_in, cb, w_out, b_out, 1, stream));
    CHECK_RQAE(rqae_hip_encode(packed, d, c, K, L, x, RQAE_X_F32, n, levels, (float *)q, (int64_t *)idx, stream));
    CHECK_RQAE(rqae_hip_decode(packed, d, c, K, L, idx, RQAE_IDX_I64, n, levels, NULL, (float *)dec, NULL /* status word */, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    /* error path: a shape the library refuses must come back as a code + message, not an abort */
    if (rqae_hip_encode(packed, 100000, c, K, L, x, RQAE_X_F32, n, levels, (float *)q, (int64_t *)idx, stream) == 0 ||
        strlen(rqae_hip_last_error()) == 0) { fprintf(stderr, "bad shape was not refused\n"); return 6; }

    if (dump(dir, "idx.bin", idx, (size_t)n * levels * sizeof(int64_t)) || dump(dir, "q.bin", q, (size_t)n * d * f) ||
        dump(dir, "dec.bin", dec, (size_t)n * d * f)) { fprintf(stderr, "cannot write outputs\n"); return 7; }

    /* the matrix-core arm: its own block, what the pack found out about the codebook, options per call */
    if (rqae_hip_gram_supported(d, c, K, L, 1)) {
        const size_t gbytes = rqae_hip_gram_packed_bytes(d, c, K, L, 1), wbytes = rqae_hip_gram_workspace_bytes(d, c, K, L, n, levels);
        if (!gbytes || !wbytes) { fprintf(stderr, "gram sizes: %s\n", rqae_hip_last_error()); return 3; }
        void *gpacked = NULL, *ws = NULL;
        CHECK_HIP(hipMalloc(&gpacked, gbytes));
        CHECK_HIP(hipMalloc(&ws, wbytes));
        CHECK_RQAE(rqae_hip_gram_pack(gpacked, d, c, K, L, w_in, b_in, cb, w_out, b_out, 1, NULL, stream));
        const int lattice_n = rqae_hip_gram_lattice(gpacked, stream);       /* waits for the pack */
        if (lattice_n < 0) { fprintf(stderr, "gram_lattice: %s\n", rqae_hip_last_error()); return 3; }
        rqae_hip_gram_options opts;
        memset(&opts, 0xff, sizeof opts);                                     /* every field negative: the defaults */
        opts.size = (int32_t)sizeof opts;
        opts.pipe_chunk = -1;
        if (lattice_n >= 2) opts.lean_levels = 5;                             /* the block is a lattice: no scan kernel behind the lattice form */
        CHECK_RQAE(rqae_hip_encode_gram(gpacked, d, c, K, L, 1, x, RQAE_X_F32, n, levels, (float *)q, (int64_t *)idx, ws, &opts, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        if (dump(dir, "idx_gram.bin", idx, (size_t)n * levels * sizeof(int64_t)) || dump(dir, "q_gram.bin", q, (size_t)n * d * f)) {
            fprintf(stderr, "cannot write outputs\n");
            return 7;
        }
        snprintf(path, sizeof path, "%s/lattice.txt", dir);
        FILE *lf = fopen(path, "w");
        if (!lf) return 7;
        fprintf(lf, "%d\n", lattice_n);
        fclose(lf);
        /* an options struct of another size is refused, not misread */
        opts.size = 8;
        if (rqae_hip_encode_gram(gpacked, d, c, K, L, 1, x, RQAE_X_F32, n, levels, (float *)q, (int64_t *)idx, ws, &opts, stream) == 0) {
            fprintf(stderr, "a short options struct was not refused\n");
            return 6;
        }
    }
    printf("host_smoke ok: d=%d c=%d K=%d L=%d n=%d levels=%d\n", d, c, K, L, n, levels);
    return 0;
}
