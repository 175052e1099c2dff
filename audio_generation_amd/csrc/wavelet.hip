// Wavelet / multiresolution layers of networks/wavelets.py for gfx950.
// Both are per-channel (depthwise) and pure bandwidth: one read of the input,
// one write of the output, everything else stays in LDS / registers.
#include "mfma_tile.hpp"

namespace agx {

// ------------------------------------------------------------------- multires cascade
// CausalMultiresConv1d.forward (wavelets.py:79-96).  One block = one (b, c) row
// tile of TT outputs + the cascade's full receptive field as left halo; the
// depth levels ping-pong between two LDS rows, the mixed output accumulates in
// registers in the reference's order (deepest-index weight first).
constexpr int MR_TT = 1024;  // outputs per block (4 per thread)

__global__ __launch_bounds__(256) void multires_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ h0,
                                                       const float *__restrict__ h1,
                                                       const float *__restrict__ w, float *__restrict__ y,
                                                       int C, int L, int K, int depth, int halo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int W = MR_TT + halo;  // tile width incl. halo
    float *lo_a = sm, *lo_b = sm + W;
    float *hk0 = sm + 2 * W, *hk1 = hk0 + K;
    const int tid = threadIdx.x;
    // 1-D grid: block = (row, tile) with the tiles of a row adjacent (rows used to sit in gridDim.y: B * C <= 65535)
    const int n_tx = (L + MR_TT - 1) / MR_TT;
    const unsigned brow = blockIdx.x / n_tx, btile = blockIdx.x - brow * n_tx;
    const int c = brow % C;
    const size_t row = size_t(brow) * L;
    const int t0 = btile * MR_TT;
    const int g0 = t0 - halo;  // global index of tile position 0

    for (int i = tid; i < W; i += 256) {
        const int g = g0 + i;
        lo_a[i] = (g >= 0 && g < L) ? x[row + g] : 0.f;
    }
    if (tid < K) {
        hk0[tid] = h0[c * K + tid];
        hk1[tid] = h1[c * K + tid];
    }
    __syncthreads();

    float acc[MR_TT / 256], xin[MR_TT / 256];
#pragma unroll
    for (int u = 0; u < MR_TT / 256; ++u) {
        acc[u] = 0.f;
        xin[u] = lo_a[halo + tid + 256 * u];
    }
    const float *wc = w + size_t(c) * (depth + 2);
    float *cur = lo_a, *nxt = lo_b;
    int dil = 1;
    for (int lvl = depth; lvl >= 1; --lvl) {
        // high-pass output only where it is kept (the last TT positions)
        const float wl = wc[lvl];
#pragma unroll
        for (int u = 0; u < MR_TT / 256; ++u) {
            const int i = halo + tid + 256 * u;
            float hi = 0.f;
            for (int k = 0; k < K; ++k) {
                const int src = i - (K - 1 - k) * dil;
                hi = fmaf(hk1[k], src >= 0 ? cur[src] : 0.f, hi);
            }
            acc[u] += wl * hi;
        }
        // low-pass for the whole tile (feeds the next level)
        for (int i = tid; i < W; i += 256) {
            float lo = 0.f;
            for (int k = 0; k < K; ++k) {
                const int src = i - (K - 1 - k) * dil;
                lo = fmaf(hk0[k], src >= 0 ? cur[src] : 0.f, lo);
            }
            nxt[i] = lo;
        }
        __syncthreads();
        float *tmp = cur;
        cur = nxt;
        nxt = tmp;
        dil *= 2;
    }
    const float w0 = wc[0], wx = wc[depth + 1];
#pragma unroll
    for (int u = 0; u < MR_TT / 256; ++u) {
        const int t = t0 + tid + 256 * u;
        if (t < L) {
            float v = acc[u] + w0 * cur[halo + tid + 256 * u];
            v += xin[u] * wx;
            y[row + t] = gelu_erf(v);
        }
    }
}


// ---------------------------------------------------------------- multires cascade, backward
// Gradient of CausalMultiresConv1d.forward (wavelets.py:79-96) w.r.t. x, h0, h1 and w.  One block = one (b, c) row and
// MRB_TT OWNED positions; the tile spans [t0 - halo, t0 + MRB_TT + halo): the left halo re-forms every level of the
// cascade (kept in LDS: the backward needs each level's input), the right halo carries the output gradients that reach
// back to the owned positions through the anti-causal transposed filters.  Output gradients left of t0 are treated as
// zero: a gradient at position s only depends on output gradients at t >= s, so everything this block OWNS --
// dx[t0 .. t0+TT) and the t-terms of the parameter sums for t in that range -- is exact.  Parameter sums: one block
// reduction per parameter in a fixed order, per-block partials, then multires_bwd_reduce_kernel (b-major, tile order):
// deterministic, no atomics.
constexpr int MRB_TT = 512;

__device__ __forceinline__ float mrb_block_sum(float v, float *red, int slot) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[4 * slot + (threadIdx.x >> 6)] = v;
    return v;
}

__global__ __launch_bounds__(256) void multires_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                           const float *__restrict__ h0, const float *__restrict__ h1,
                                                           const float *__restrict__ w, float *__restrict__ dx,
                                                           float *__restrict__ part, int C, int L, int K, int depth,
                                                           int halo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int W = MRB_TT + 2 * halo;
    const int NP = 2 * K + depth + 2;          // [dh0 (K) | dh1 (K) | dw (depth + 2)]
    float *lev = sm;                            // [(depth + 1)][W]: level 0 = x, level j = low-pass after j levels
    float *dpre = lev + size_t(depth + 1) * W;  // [W]  output gradient -> gradient at the GELU input
    float *ga = dpre + W, *gb = ga + W;         // [W] each: pre-activation first, then gradient w.r.t. the low-pass levels
    float *hk0 = gb + W, *hk1 = hk0 + K;
    float *red = hk1 + K;                       // [NP][4]
    const int tid = threadIdx.x;
    const int n_tx = (L + MRB_TT - 1) / MRB_TT;     // 1-D grid: block = (row, tile), as the forward
    const unsigned brow = blockIdx.x / n_tx, btile = blockIdx.x - brow * n_tx;
    const int c = brow % C;
    const size_t row = size_t(brow) * L;
    const int t0 = btile * MRB_TT;
    const int g0 = t0 - halo;
    const float *wc = w + size_t(c) * (depth + 2);

    for (int i = tid; i < W; i += 256) {
        const int g = g0 + i;
        const bool in = g >= 0 && g < L;
        lev[i] = in ? x[row + g] : 0.f;
        dpre[i] = (in && i >= halo) ? dout[row + g] : 0.f;
        ga[i] = 0.f;
    }
    if (tid < K) {
        hk0[tid] = h0[c * K + tid];
        hk1[tid] = h1[c * K + tid];
    }
    __syncthreads();
    // forward: levels + pre-activation (ga) on [halo, W)
    {
        int dil = 1;
        for (int j = 0; j < depth; ++j, dil *= 2) {
            const float *cur = lev + size_t(j) * W;
            float *nxt = lev + size_t(j + 1) * W;
            const float wl = wc[depth - j];
            for (int i = tid; i < W; i += 256) {
                float lo = 0.f, hi = 0.f;
                for (int k = 0; k < K; ++k) {
                    const int src = i - (K - 1 - k) * dil;
                    const float v = src >= 0 ? cur[src] : 0.f;
                    lo = fmaf(hk0[k], v, lo);
                    hi = fmaf(hk1[k], v, hi);
                }
                nxt[i] = lo;
                ga[i] += wl * hi;
            }
            __syncthreads();
        }
    }
    const float w0 = wc[0], wx = wc[depth + 1];
    const float *top = lev + size_t(depth) * W;
    for (int i = tid; i < W; i += 256) {
        const float pre = ga[i] + w0 * top[i] + wx * lev[i];
        dpre[i] *= gelu_grad(pre);
    }
    __syncthreads();
    // parameter t-terms over the owned positions; dw[0], dw[depth + 1]
    const int own0 = halo + tid, own1 = halo + tid + 256;     // MRB_TT = 2 x 256 owned positions per thread
    {
        float a = dpre[own0] * top[own0] + dpre[own1] * top[own1];
        float b = dpre[own0] * lev[own0] + dpre[own1] * lev[own1];
        mrb_block_sum(a, red, 2 * K + 0);
        mrb_block_sum(b, red, 2 * K + depth + 1);
    }
    // gradient w.r.t. the deepest low-pass level
    for (int i = tid; i < W; i += 256) ga[i] = w0 * dpre[i];
    __syncthreads();
    float *gcur = ga, *gnext = gb;
    float acc0[2], acc1[2];
    for (int j = depth - 1; j >= 0; --j) {
        const int dil = 1 << j;
        const float *in = lev + size_t(j) * W;
        const float wl = wc[depth - j];
        // t-terms of this level: dw[depth - j] = sum dpre * hi_j;  dh1[k] = sum (wl dpre) * in[. - (K-1-k) dil];
        // dh0[k] = sum gcur * in[. - (K-1-k) dil]
        float hi0 = 0.f, hi1 = 0.f;
        for (int k = 0; k < K; ++k) {
            const int s0 = own0 - (K - 1 - k) * dil, s1 = own1 - (K - 1 - k) * dil;   // >= 0: halo covers the reach
            const float v0 = in[s0], v1 = in[s1];
            hi0 = fmaf(hk1[k], v0, hi0);
            hi1 = fmaf(hk1[k], v1, hi1);
            acc0[0] = gcur[own0] * v0; acc0[1] = gcur[own1] * v1;
            acc1[0] = wl * dpre[own0] * v0; acc1[1] = wl * dpre[own1] * v1;
            // each parameter's partial lands in its own red[] slot; levels accumulate in registers of wave-lane 0 below
            float s_h0 = acc0[0] + acc0[1], s_h1 = acc1[0] + acc1[1];
            for (int off = 32; off > 0; off >>= 1) { s_h0 += __shfl_xor(s_h0, off); s_h1 += __shfl_xor(s_h1, off); }
            if ((tid & 63) == 0) {
                const int wv = tid >> 6;
                red[4 * k + wv] = (j == depth - 1 ? 0.f : red[4 * k + wv]) + s_h0;            // levels added deepest first
                red[4 * (K + k) + wv] = (j == depth - 1 ? 0.f : red[4 * (K + k) + wv]) + s_h1;
            }
        }
        mrb_block_sum(dpre[own0] * hi0 + dpre[own1] * hi1, red, 2 * K + depth - j);
        // gradient w.r.t. this level's input: transposed (anti-causal) filters
        for (int s = tid; s < W; s += 256) {
            float g = 0.f;
            for (int k = 0; k < K; ++k) {
                const int t = s + (K - 1 - k) * dil;
                if (t < W) g = fmaf(hk0[k], gcur[t], fmaf(hk1[k] * wl, dpre[t], g));
            }
            gnext[s] = g;
        }
        __syncthreads();
        float *tmp = gcur; gcur = gnext; gnext = tmp;
    }
    if (depth == 0 && tid < 4 * 2 * K) red[tid] = 0.f;      // no level: the filters get no gradient
    // dx over the owned positions
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = halo + tid + 256 * u, t = t0 + tid + 256 * u;
        if (t < L) dx[row + t] = gcur[i] + wx * dpre[i];
    }
    __syncthreads();
    if (tid < NP) {
        const float *r = red + 4 * tid;
        part[size_t(blockIdx.x) * NP + tid] = (r[0] + r[1]) + (r[2] + r[3]);      // (row, tile) order, as before
    }
}

// d{h0,h1,w}[c][p] = sum over (b, tile) of part[((b C + c) tiles + tile) NP + p], in that order
__global__ __launch_bounds__(64) void multires_bwd_reduce_kernel(const float *__restrict__ part, int B, int C, int tiles,
                                                                 int K, int depth, float *__restrict__ dh0,
                                                                 float *__restrict__ dh1, float *__restrict__ dw) {
    const int NP = 2 * K + depth + 2;
    const int c = blockIdx.x, pidx = threadIdx.x;
    if (pidx >= NP) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < tiles; ++t) s += part[((size_t(b) * C + c) * tiles + t) * NP + pidx];
    if (pidx < K) dh0[c * K + pidx] = s;
    else if (pidx < 2 * K) dh1[c * K + pidx - K] = s;
    else dw[c * (depth + 2) + pidx - 2 * K] = s;
}

// out[i] = (gelu'(pre[i])) * sum_{j < group} g[i * group + j]
__global__ __launch_bounds__(256) void group_sum_kernel(const float *__restrict__ g, const float *__restrict__ pre,
                                                        float *__restrict__ out, int64_t n, int group) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int j = 0; j < group; ++j) s += g[i * group + j];
    out[i] = pre ? s * gelu_grad(pre[i]) : s;
}

// ---------------------------------------------------------------------- wavelet fold
// WaveletLayer.forward, the part between the two convs (wavelets.py:221-231).
// With n_points = scale * fold, window u of the flattened (l, p) signal covers
// points p >= (u % scale) * fold of frame l0 = u / scale and points
// p < (u % scale) * fold of frame l0 + 1, so
//     out[u] = h[l0] * S_hi[u % scale] + h[l0 + 1] * S_lo[u % scale]
// with S_* partial sums of the channel's wavelet cos(t) exp(-t^2 / sigma_c);
// the last scale - 1 outputs are the raw tail samples the reference appends.
__global__ __launch_bounds__(256) void wavelet_fold_kernel(const float *__restrict__ h,
                                                           const float *__restrict__ space,
                                                           const float *__restrict__ sigma, int sigma_len,
                                                           float *__restrict__ y, int C, int L, int P,
                                                           int scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *kern = sm;            // [P]
    float *s_lo = sm + P;        // [scale]
    float *s_hi = s_lo + scale;  // [scale]
    const int tid = threadIdx.x;
    const int n_bx = min((L * scale + 255) / 256, 64);     // 1-D grid: block = (row, stripe)
    const unsigned brow = blockIdx.x / n_bx, bstripe = blockIdx.x - brow * n_bx;
    const int c = brow % C;
    const float sg = sigma[sigma_len == 1 ? 0 : c];
    if (tid < P) {
        const float t = space[tid];
        kern[tid] = cosf(t) * expf(-(t * t) / sg);
    }
    __syncthreads();
    const int fold = P / scale;
    if (tid < scale) {
        float lo = 0.f, hi = 0.f;
        for (int p = 0; p < tid * fold; ++p) lo += kern[p];
        for (int p = tid * fold; p < P; ++p) hi += kern[p];
        s_lo[tid] = lo;
        s_hi[tid] = hi;
    }
    __syncthreads();
    const size_t in_row = size_t(brow) * L;
    const size_t out_row = in_row * scale;
    const int n_out = L * scale;
    const int n_win = (L - 1) * scale + 1;
    for (int u = bstripe * 256 + tid; u < n_out; u += n_bx * 256) {
        float v;
        if (u < n_win) {
            const int l0 = u / scale, ph = u - l0 * scale;
            v = h[in_row + l0] * s_hi[ph];
            if (ph > 0) v += h[in_row + l0 + 1] * s_lo[ph];
        } else {
            v = kern[P - (scale - 1) + (u - n_win)] * h[in_row + L - 1];
        }
        y[out_row + u] = v;
    }
}

// Backward of the fold: dh (B,C,L) and, per (b, c) row, the partial derivative w.r.t. the row's
// wavelet scale sigma (d kern[p] / d sigma = kern[p] t_p^2 / sigma^2).  One block per row.
__global__ __launch_bounds__(256) void wavelet_fold_bwd_kernel(const float *__restrict__ h,
                                                               const float *__restrict__ dout,
                                                               const float *__restrict__ space,
                                                               const float *__restrict__ sigma, int sigma_len,
                                                               float *__restrict__ dh,
                                                               float *__restrict__ dsig_part, int C, int L, int P,
                                                               int scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *kern = sm;               // [P]
    float *dkern = sm + P;          // [P]
    float *s_lo = dkern + P;        // [scale]
    float *s_hi = s_lo + scale;
    float *ds_lo = s_hi + scale;
    float *ds_hi = ds_lo + scale;
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int c = blockIdx.x % C;
    const float sg = sigma[sigma_len == 1 ? 0 : c];
    if (tid < P) {
        const float t = space[tid];
        const float k = cosf(t) * expf(-(t * t) / sg);
        kern[tid] = k;
        dkern[tid] = k * (t * t) / (sg * sg);
    }
    __syncthreads();
    const int fold = P / scale;
    if (tid < scale) {
        float lo = 0.f, hi = 0.f, dlo = 0.f, dhi = 0.f;
        for (int p = 0; p < tid * fold; ++p) { lo += kern[p]; dlo += dkern[p]; }
        for (int p = tid * fold; p < P; ++p) { hi += kern[p]; dhi += dkern[p]; }
        s_lo[tid] = lo; s_hi[tid] = hi; ds_lo[tid] = dlo; ds_hi[tid] = dhi;
    }
    __syncthreads();
    const size_t in_row = size_t(blockIdx.x) * L, out_row = in_row * scale;
    const int n_win = (L - 1) * scale + 1;
    float dsig = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float hl = h[in_row + l];
        float g = 0.f;
        // windows starting in frame l (u = l*scale + ph < n_win)
        for (int ph = 0; ph < scale; ++ph) {
            const int u = l * scale + ph;
            if (u < n_win) {
                const float d = dout[out_row + u];
                g = fmaf(d, s_hi[ph], g);
                dsig = fmaf(d * hl, ds_hi[ph], dsig);
            }
        }
        // windows starting in frame l-1 that spill into frame l (ph > 0)
        if (l >= 1)
            for (int ph = 1; ph < scale; ++ph) {
                const float d = dout[out_row + (l - 1) * scale + ph];
                g = fmaf(d, s_lo[ph], g);
                dsig = fmaf(d * hl, ds_lo[ph], dsig);
            }
        // the reference's raw-sample tail: out[n_win + e] = kern[P - (scale-1) + e] * h[L-1]
        if (l == L - 1)
            for (int e = 0; e < scale - 1; ++e) {
                const float d = dout[out_row + n_win + e];
                g = fmaf(d, kern[P - (scale - 1) + e], g);
                dsig = fmaf(d * hl, dkern[P - (scale - 1) + e], dsig);
            }
        dh[in_row + l] = g;
    }
    for (int off = 32; off > 0; off >>= 1) dsig += __shfl_xor(dsig, off);
    if ((tid & 63) == 0) red[tid >> 6] = dsig;
    __syncthreads();
    if (tid == 0) dsig_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dsigma[c] = sum_b part[b*C + c]   (or the sum over everything when sigma is shared)
__global__ __launch_bounds__(256) void wavelet_sigma_reduce_kernel(const float *__restrict__ part, int B, int C,
                                                                   int sigma_len, float *__restrict__ dsigma) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (sigma_len == 1) {
        if (c == 0) {
            float s = 0.f;
            for (int i = 0; i < B * C; ++i) s += part[i];
            dsigma[0] = s;
        }
        return;
    }
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[b * C + c];
    dsigma[c] = s;
}

}  // namespace agx

extern "C" {

int agx_wavelet_fold_backward(const float *h, const float *dout, const float *space, const float *sigma,
                              int32_t sigma_len, float *dh, float *dsigma, float *workspace, int32_t batch,
                              int32_t channels, int32_t length, int32_t n_points, int32_t scale, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || n_points <= 0 || scale <= 0 || n_points % scale != 0)
        return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold_backward: bad shape");
    if (n_points > 256 || scale > 256) return fail(AGX_ERR_UNSUPPORTED, "wavelet_fold_backward: n_points/scale > 256");
    if (sigma_len != 1 && sigma_len != channels) return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold_backward: sigma_len must be 1 or C");
    if (!h || !dout || !space || !sigma || !dh || !dsigma || !workspace)
        return fail(AGX_ERR_NULL_POINTER, "wavelet_fold_backward: NULL pointer");
    const int64_t rows = int64_t(batch) * channels;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (2 * n_points + 4 * scale) * sizeof(float);
    hipLaunchKernelGGL(wavelet_fold_bwd_kernel, dim3((unsigned)rows), dim3(256), lds, st, h, dout, space, sigma,
                       sigma_len, dh, workspace, channels, length, n_points, scale);
    hipLaunchKernelGGL(wavelet_sigma_reduce_kernel, dim3(ceil_div(channels, 256)), dim3(256), 0, st, workspace, batch,
                       channels, sigma_len, dsigma);
    return check_launch("wavelet_fold_backward");
}


int agx_multires_forward(const float *x, const float *h0, const float *h1, const float *w, float *y,
                         int32_t batch, int32_t channels, int32_t length, int32_t kernel, int32_t depth,
                         void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || kernel <= 0 || depth < 0 || depth > 20)
        return fail(AGX_ERR_BAD_SHAPE, "multires: bad shape");
    if (!x || !h0 || !h1 || !w || !y) return fail(AGX_ERR_NULL_POINTER, "multires: NULL pointer");
    const int64_t halo = int64_t(kernel - 1) * ((int64_t(1) << depth) - 1);
    const size_t lds = (2 * (MR_TT + halo) + 2 * kernel) * sizeof(float);
    if (lds > 150 * 1024) return fail(AGX_ERR_UNSUPPORTED, "multires: receptive field %lld too long for LDS", (long long)halo);
    const int64_t rows = int64_t(batch) * channels;
    if (rows * ceil_div(length, MR_TT) > INT32_MAX) return fail(AGX_ERR_BAD_SHAPE, "multires: too many tiles for one launch");
    auto kern = multires_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3((unsigned)(rows * ceil_div(length, MR_TT))), dim3(256), lds,
                       static_cast<hipStream_t>(stream), x, h0, h1, w, y, channels, length, kernel, depth, int(halo));
    return check_launch("multires");
}

size_t agx_multires_backward_workspace_bytes(int32_t batch, int32_t channels, int32_t length, int32_t kernel, int32_t depth) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || kernel <= 0 || depth < 0) return 0;
    return size_t(batch) * channels * ceil_div(length, MRB_TT) * (2 * kernel + depth + 2) * sizeof(float);
}

int agx_multires_backward(const float *x, const float *dout, const float *h0, const float *h1, const float *w, float *dx,
                          float *dh0, float *dh1, float *dw, void *workspace, size_t workspace_bytes, int32_t batch,
                          int32_t channels, int32_t length, int32_t kernel, int32_t depth, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || kernel <= 0 || depth < 0 || depth > 20)
        return fail(AGX_ERR_BAD_SHAPE, "multires_backward: bad shape");
    if (!x || !dout || !h0 || !h1 || !w || !dx || !dh0 || !dh1 || !dw || !workspace)
        return fail(AGX_ERR_NULL_POINTER, "multires_backward: NULL pointer");
    if (workspace_bytes < agx_multires_backward_workspace_bytes(batch, channels, length, kernel, depth))
        return fail(AGX_ERR_WORKSPACE, "multires_backward: workspace too small");
    const int np = 2 * kernel + depth + 2;
    if (np > 64) return fail(AGX_ERR_UNSUPPORTED, "multires_backward: 2 K + depth + 2 > 64");
    const int64_t halo = int64_t(kernel - 1) * ((int64_t(1) << depth) - 1);
    const size_t lds = (size_t(depth + 4) * (MRB_TT + 2 * halo) + 2 * kernel + 4 * np) * sizeof(float);
    if (lds > 150 * 1024) return fail(AGX_ERR_UNSUPPORTED, "multires_backward: receptive field %lld too long for LDS", (long long)halo);
    const int64_t rows = int64_t(batch) * channels;
    if (rows * ceil_div(length, MRB_TT) > INT32_MAX) return fail(AGX_ERR_BAD_SHAPE, "multires_backward: too many tiles for one launch");
    auto kern = multires_bwd_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = ceil_div(length, MRB_TT);
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(kern, dim3((unsigned)(rows * tiles)), dim3(256), lds, st, x, dout, h0, h1, w, dx, part, channels,
                       length, kernel, depth, int(halo));
    hipLaunchKernelGGL(multires_bwd_reduce_kernel, dim3(channels), dim3(64), 0, st, part, batch, channels, tiles, kernel,
                       depth, dh0, dh1, dw);
    return check_launch("multires_backward");
}

int agx_group_sum(const float *g, const float *gelu_pre, float *out, int64_t n_out, int32_t group, void *stream) {
    using namespace agx;
    if (n_out <= 0 || group <= 0) return fail(AGX_ERR_BAD_SHAPE, "group_sum: bad shape");
    if (!g || !out) return fail(AGX_ERR_NULL_POINTER, "group_sum: NULL pointer");
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)ceil_div64(n_out, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g, gelu_pre, out, n_out, group);
    return check_launch("group_sum");
}

int agx_wavelet_fold(const float *h, const float *space, const float *sigma, int32_t sigma_len, float *y,
                     int32_t batch, int32_t channels, int32_t length, int32_t n_points, int32_t scale,
                     void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || n_points <= 0 || scale <= 0 || n_points % scale != 0)
        return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold: bad shape (n_points must be divisible by scale)");
    if (n_points > 256 || scale > 256) return fail(AGX_ERR_UNSUPPORTED, "wavelet_fold: n_points/scale > 256");
    if (sigma_len != 1 && sigma_len != channels) return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold: sigma_len must be 1 or C");
    if (!h || !space || !sigma || !y) return fail(AGX_ERR_NULL_POINTER, "wavelet_fold: NULL pointer");
    const int64_t rows = int64_t(batch) * channels;
    const int gx = min(ceil_div(length * scale, 256), 64);     // (the kernel derives the same stripe count)
    if (rows * gx > INT32_MAX) return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold: too many blocks for one launch");
    const size_t lds = (n_points + 2 * scale) * sizeof(float);
    hipLaunchKernelGGL(wavelet_fold_kernel, dim3((unsigned)(rows * gx)), dim3(256), lds,
                       static_cast<hipStream_t>(stream), h, space, sigma, sigma_len, y, channels, length, n_points, scale);
    return check_launch("wavelet_fold");
}

}  // extern "C"
