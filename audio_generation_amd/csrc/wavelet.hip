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
    const int c = blockIdx.y % C;
    const size_t row = size_t(blockIdx.y) * L;
    const int t0 = blockIdx.x * MR_TT;
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
    const int c = blockIdx.y % C;
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
    const size_t in_row = size_t(blockIdx.y) * L;
    const size_t out_row = in_row * scale;
    const int n_out = L * scale;
    const int n_win = (L - 1) * scale + 1;
    for (int u = blockIdx.x * 256 + tid; u < n_out; u += gridDim.x * 256) {
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
    if (rows > 65535) return fail(AGX_ERR_BAD_SHAPE, "multires: B*C too large for one launch");
    auto kern = multires_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(ceil_div(length, MR_TT), (unsigned)rows), dim3(256), lds,
                       static_cast<hipStream_t>(stream), x, h0, h1, w, y, channels, length, kernel, depth, int(halo));
    return check_launch("multires");
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
    if (rows > 65535) return fail(AGX_ERR_BAD_SHAPE, "wavelet_fold: B*C too large for one launch");
    const int gx = min(ceil_div(length * scale, 256), 64);
    const size_t lds = (n_points + 2 * scale) * sizeof(float);
    hipLaunchKernelGGL(wavelet_fold_kernel, dim3(gx, (unsigned)rows), dim3(256), lds,
                       static_cast<hipStream_t>(stream), h, space, sigma, sigma_len, y, channels, length, n_points, scale);
    return check_launch("wavelet_fold");
}

}  // extern "C"
