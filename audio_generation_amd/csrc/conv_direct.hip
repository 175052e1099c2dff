// Direct fp32 (VALU) polyphase convolution -- any shape.
//
// This is the kernel behind the narrow layers of the codec (waveform -> 32
// channels, vae.py:257; 32 channels -> waveform, vae.py:281) which are pure
// bandwidth, and the catch-all for shapes the MFMA kernels do not tile.
// One thread owns one base position t and a chunk of CO_T output rows; the
// input tile (+ halo) is staged once per channel chunk in LDS with coalesced
// loads along time, weights are wave-uniform (scalar loads of the K-major
// packed image).
#include "common.hpp"

namespace agx {

template <int CO_T>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvPlan p, int ci_tile, int span,
                                                          const float *__restrict__ x,
                                                          const float *__restrict__ wp,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ res,
                                                          float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [ci_tile][span]
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 256;
    const int m0 = blockIdx.y * CO_T;
    const int b = blockIdx.z;
    const int t = t0 + tid;
    const int in0 = t0 * p.s - p.P;  // input index of xs[.][0]

    float acc[CO_T];
#pragma unroll
    for (int r = 0; r < CO_T; ++r) acc[r] = 0.f;

    const float *xb = x + size_t(b) * p.Cin * p.Lin;
    for (int c0 = 0; c0 < p.Cin; c0 += ci_tile) {
        const int nc = min(ci_tile, p.Cin - c0);
        __syncthreads();
        for (int e = tid; e < nc * span; e += 256) {
            const int c = e / span, i = e - c * span;
            const int pos = in0 + i;
            xs[e] = (pos >= 0 && pos < p.Lvalid) ? xb[size_t(c0 + c) * p.Lin + pos] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            const float *xr = xs + c * span + tid * p.s;
            for (int j = 0; j < p.J; ++j) {
                const float xv = xr[j * p.d];
                const float *wj = wp + packed_weight_index(c0 + c, j, 0, p.J, p.M);
#pragma unroll
                for (int r = 0; r < CO_T; ++r) {
                    const int m = min(m0 + r, p.M - 1);  // wave-uniform -> scalar load
                    acc[r] = fmaf(wj[size_t(m) * kWG], xv, acc[r]);
                }
            }
        }
    }

    if (t >= p.Lt) return;
#pragma unroll
    for (int r = 0; r < CO_T; ++r) {
        const int m = m0 + r;
        if (m >= p.M) break;
        const int co = m / p.q, ph = m - co * p.q;
        const int u = t * p.q + ph - p.oshift;
        if (u < 0 || u >= p.Lout) continue;
        float v = acc[r] + (bias ? bias[co] : 0.f);
        if (p.epilogue & AGX_EPI_LEAKY_PRE) v = v > 0.f ? v : v * p.slope;
        if (p.epilogue & AGX_EPI_GELU_PRE) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        const size_t o = (size_t(b) * p.Cout + co) * p.Lout + u;
        if (p.epilogue & AGX_EPI_RESIDUAL) v += res[o];
        if (p.epilogue & AGX_EPI_LEAKY_POST) v = v > 0.f ? v : v * p.slope;
        if (p.epilogue & AGX_EPI_MASK) v = p.mask[o] > 0.f ? v : v * p.slope;
        y[o] = v;
    }
}

const char *conv_direct_variant(const ConvPlan &p) {
    return p.M >= 32 ? "conv_direct<32>" : (p.M > 4 ? "conv_direct<16>" : (p.M > 1 ? "conv_direct<4>" : "conv_direct<1>"));
}

int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                       const float *res, float *y, hipStream_t st) {
    const int span = 255 * p.s + (p.J - 1) * p.d + 1;
    int ci_tile = (12 * 1024) / span;  // <= 48 KB of LDS
    if (ci_tile < 1) ci_tile = 1;
    if (ci_tile > p.Cin) ci_tile = p.Cin;
    const size_t lds = size_t(ci_tile) * span * sizeof(float);
    if (lds > 150 * 1024) return fail(AGX_ERR_UNSUPPORTED, "conv_direct: tile needs %zu B of LDS", lds);
    const int co_t = p.M >= 32 ? 32 : (p.M > 4 ? 16 : (p.M > 1 ? 4 : 1));
    dim3 grid(ceil_div(p.Lt, 256), ceil_div(p.M, co_t), p.B), block(256);
    if (grid.y > 65535 || grid.z > 65535) return fail(AGX_ERR_BAD_SHAPE, "conv_direct: grid too large");
    switch (co_t) {
        case 32:
            hipLaunchKernelGGL(conv_direct_kernel<32>, grid, block, lds, st, p, ci_tile, span, x, wp, bias, res, y);
            break;
        case 16:
            hipLaunchKernelGGL(conv_direct_kernel<16>, grid, block, lds, st, p, ci_tile, span, x, wp, bias, res, y);
            break;
        case 4:
            hipLaunchKernelGGL(conv_direct_kernel<4>, grid, block, lds, st, p, ci_tile, span, x, wp, bias, res, y);
            break;
        default:
            hipLaunchKernelGGL(conv_direct_kernel<1>, grid, block, lds, st, p, ci_tile, span, x, wp, bias, res, y);
            break;
    }
    return check_launch("conv_direct");
}

}  // namespace agx
