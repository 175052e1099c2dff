/* Oracle: the DEFINING arithmetic of the residual-vector-quantiser search.
 *
 * TEST INFRASTRUCTURE -- see oracle/__init__.py.  Not linked into the product.
 *
 * PARITY UNPINNED: the reference imports its quantiser from an external,
 * un-vendored, un-pinned module (`som_quantizer`, networks/vae.py:6; repository
 * "quantization-maps", README.md:10,34).  Its source is not in the reference
 * tree and it holds no fixtures for it, so this file restates the published
 * algorithm (README.md:48: "quantizing a signal, subtracting the quantized
 * signal from the original, and iteratively quantizing the residual") against
 * the call-site contract (vae.py:245-251, 315-318, 333) and FIXES the arithmetic
 * so that any implementation can be bit-exact against it:
 *
 *   dist(r, c) = sum_{d = 0 .. D-1, in that order} sq_d,
 *                sq_d = ((double)r[d] - (double)c[d])^2   -- one IEEE-754
 *                binary64 subtract, one multiply, one add per term, NO fused
 *                multiply-add, round-to-nearest-even.
 *   index      = the smallest k attaining the minimum dist.
 *   residual   r[d] <- r[d] - c_index[d]          (binary32)
 *   output     out[d] <- out[d] + c_index[d]      (binary32, stage order)
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/build.py).
 */
#include <stdint.h>
#include <stddef.h>

#pragma STDC FP_CONTRACT OFF

static double exact_dist(const float *r, const float *c, int dim) {
    double acc = 0.0;
    for (int d = 0; d < dim; ++d) {
        double diff = (double)r[d] - (double)c[d];
        double sq = diff * diff;
        acc = acc + sq;
    }
    return acc;
}

/* Full search: frames (n, dim), codebook (k, dim) -> idx (n).  O(n*k*dim). */
void rvq_exact_search(const float *frames, const float *codebook, int n, int k, int dim,
                      int64_t *idx) {
    for (int f = 0; f < n; ++f) {
        const float *r = frames + (size_t)f * dim;
        double best = exact_dist(r, codebook, dim);
        int64_t arg = 0;
        for (int j = 1; j < k; ++j) {
            double dj = exact_dist(r, codebook + (size_t)j * dim, dim);
            if (dj < best) { best = dj; arg = j; }
        }
        idx[f] = arg;
    }
}

/* Search restricted to a candidate list per frame.  cand is (n, max_cand)
 * int32, n_cand (n) says how many entries of each row are valid (>= 1).  The
 * caller guarantees the true arg-min is among them (oracle/rvq.py derives the
 * list from a float64 matmul with a margin far above its rounding error). */
void rvq_exact_among(const float *frames, const float *codebook, int n, int dim,
                     const int32_t *cand, const int32_t *n_cand, int max_cand,
                     int64_t *idx) {
    for (int f = 0; f < n; ++f) {
        const float *r = frames + (size_t)f * dim;
        const int32_t *row = cand + (size_t)f * max_cand;
        double best = 0.0;
        int64_t arg = -1;
        for (int j = 0; j < n_cand[f]; ++j) {
            int32_t kk = row[j];
            double dj = exact_dist(r, codebook + (size_t)kk * dim, dim);
            if (arg < 0 || dj < best || (dj == best && kk < arg)) { best = dj; arg = kk; }
        }
        idx[f] = arg;
    }
}

/* One stage's bookkeeping in binary32: r -= c[idx], out += c[idx]. */
void rvq_apply(float *frames, float *out, const float *codebook, const int64_t *idx,
               int n, int dim) {
    for (int f = 0; f < n; ++f) {
        const float *c = codebook + (size_t)idx[f] * dim;
        float *r = frames + (size_t)f * dim;
        float *o = out + (size_t)f * dim;
        for (int d = 0; d < dim; ++d) {
            r[d] = r[d] - c[d];
            o[d] = o[d] + c[d];
        }
    }
}
