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
    // 2-D layers (Tout > 1 or kh > 1) put the (b, output row) index on grid.x: it can exceed 65535
    const bool two_d = p.Tout > 1 || p.kh > 1;
    const int t0 = (two_d ? blockIdx.z : blockIdx.x) * 256;
    const int m0 = blockIdx.y * CO_T;
    const int zrow = two_d ? blockIdx.x : blockIdx.z;
    const int b = zrow / p.Tout, trow = zrow - b * p.Tout;  // 1-D: Tout = 1
    const int t = t0 + tid;
    const int in0 = t0 * p.s - p.P;  // input index of xs[.][0]
    // grouped layers: this block's rows all sit in one group (the launcher picks CO_T | Cout / G)
    const int cpg = p.ncv / p.G;                       // (virtual) input channels per group
    const int grp = p.G > 1 ? m0 / (p.Cout / p.G) : 0;
    const int row0 = trow * p.sh - p.ph;

    float acc[CO_T];
#pragma unroll
    for (int r = 0; r < CO_T; ++r) acc[r] = 0.f;

    const float *xb = x + size_t(b) * p.cin_real * p.x_cstride + size_t(grp) * cpg * p.x_cstride;
    for (int c0 = 0; c0 < cpg; c0 += ci_tile) {
        const int nc = min(ci_tile, cpg - c0);
        __syncthreads();
        // all loads of a batch go out on clamped addresses before any is used (a conditional load is
        // waited for one by one); zeros are selected afterwards
        {
            const int total = nc * span;
            const float inv_span = 1.f / float(span);
            for (int e0 = tid; e0 < total; e0 += 256 * 8) {
                float v[8];
                bool okv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, total - 1);
                    const int c = int((float(e) + 0.5f) * inv_span), i = e - c * span;   // exact: e < 2^20
                    const int pos = in0 + i;
                    const int cv = c0 + c, ci = cv / p.kh, r = row0 + (cv - ci * p.kh);  // 1-D: kh = 1, r = 0
                    okv[u] = pos >= 0 && pos < p.Lvalid && r >= 0 && r < p.Tin;
                    v[u] = xb[size_t(ci) * p.x_cstride + size_t(min(max(r, 0), p.Tin - 1)) * p.Lin +
                              min(max(pos, 0), p.Lvalid - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < total) xs[e0 + u * 256] = okv[u] ? v[u] : 0.f;
            }
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            const float *xr = xs + c * span + tid * p.s;
            for (int j = 0; j < p.J; ++j) {
                const float xv = xr[j * p.d];
                const float *wj = wp + packed_weight_index(c0 + c, j, 0, p.J, p.M);  // image has cpg channels
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
        const size_t o = (size_t(b) * p.Cout + co) * p.y_cstride + size_t(trow) * p.Lout + u;
        if (p.epilogue & AGX_EPI_RESIDUAL) v += res[o];
        if (p.epilogue & AGX_EPI_LEAKY_POST) v = v > 0.f ? v : v * p.slope;
        if (p.epilogue & AGX_EPI_MASK) v = p.mask[o] > 0.f ? v : v * p.slope;
        y[o] = v;
    }
}

// ------------------------------------------------------------------ streaming fast path
// The two pure-bandwidth layers of every codec configuration are causal K=7, stride-1 convs
// between the waveform and first_block_channels (vae.py:257, :281).  No LDS, no barrier: one
// thread owns 4 consecutive outputs of CO_T rows, reads the 12-sample aligned window
// [t-8, t+4) of each input row as three float4 (neighbouring threads share two of them through
// L1) and writes float4.  Same summation order as the generic kernel (channel-major, tap-minor).
constexpr int kNarrowJ = 7, kNarrowP = 6;

static inline bool narrow_ok(const ConvPlan &p) {
    return p.G == 1 && p.kh == 1 && p.Tout == 1 && p.J == kNarrowJ && p.P == kNarrowP && p.s == 1 && p.d == 1 && p.q == 1 && p.oshift == 0 &&
           p.Lin % 4 == 0 && p.Lvalid % 4 == 0 && p.Lout % 4 == 0 && p.Lt == p.Lout && p.Lvalid >= 4 &&
           (p.M == 1 || p.M == 2 || p.M == 32);
}

template <int CO_T>
__global__ __launch_bounds__(256) void conv_narrow_kernel(ConvPlan p, const float *__restrict__ x,
                                                          const float *__restrict__ wp,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ res,
                                                          float *__restrict__ y) {
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.y;
    const int m0 = blockIdx.z * CO_T;  // M == 32 runs as two 16-row halves (register budget)
    const bool live = t < p.Lt;
    float acc[CO_T][4];
#pragma unroll
    for (int r = 0; r < CO_T; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[r][u] = 0.f;

    const float *xb = x + size_t(b) * p.Cin * p.Lin;
    // aligned window pieces [t-8,t-4) [t-4,t) [t,t+4): each wholly inside or outside [0, Lvalid)
    int off[3];
    bool ok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pos = t - 8 + 4 * k;
        ok[k] = pos >= 0 && pos + 4 <= p.Lvalid;
        off[k] = ok[k] ? pos : 0;
    }
    constexpr int CU = CO_T >= 16 ? 1 : 4;  // channels in flight per step
    for (int c0 = 0; c0 < p.Cin; c0 += CU) {
        float win[CU][12];
#pragma unroll
        for (int cc = 0; cc < CU; ++cc) {
            const int c = min(c0 + cc, p.Cin - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(xb + size_t(c) * p.Lin + off[k]);
                win[cc][4 * k + 0] = ok[k] ? v.x : 0.f;
                win[cc][4 * k + 1] = ok[k] ? v.y : 0.f;
                win[cc][4 * k + 2] = ok[k] ? v.z : 0.f;
                win[cc][4 * k + 3] = ok[k] ? v.w : 0.f;
            }
        }
#pragma unroll
        for (int cc = 0; cc < CU; ++cc) {
            if (c0 + cc >= p.Cin) break;
#pragma unroll
            for (int j = 0; j < kNarrowJ; ++j) {
                const float *wj = wp + packed_weight_index(c0 + cc, j, m0, kNarrowJ, p.M);
#pragma unroll
                for (int r = 0; r < CO_T; ++r) {
                    const float w = wj[size_t(r) * kWG];  // wave-uniform -> scalar load
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[r][u] = fmaf(w, win[cc][2 + u + j], acc[r][u]);
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < CO_T; ++r) {
        const size_t o = (size_t(b) * p.Cout + m0 + r) * p.Lout + t;
        const float bv = bias ? bias[m0 + r] : 0.f;
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), mv = rv;
        if (p.epilogue & AGX_EPI_RESIDUAL) rv = *reinterpret_cast<const float4 *>(res + o);
        if (p.epilogue & AGX_EPI_MASK) mv = *reinterpret_cast<const float4 *>(p.mask + o);
        const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
        float out[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = acc[r][u] + bv;
            if (p.epilogue & AGX_EPI_LEAKY_PRE) v = v > 0.f ? v : v * p.slope;
            if (p.epilogue & AGX_EPI_GELU_PRE) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
            if (p.epilogue & AGX_EPI_RESIDUAL) v += rr[u];
            if (p.epilogue & AGX_EPI_LEAKY_POST) v = v > 0.f ? v : v * p.slope;
            if (p.epilogue & AGX_EPI_MASK) v = mm[u] > 0.f ? v : v * p.slope;
            out[u] = v;
        }
        *reinterpret_cast<float4 *>(y + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// ------------------------------------------------------------------ few GEMM rows
// Dense 1-D layers with at most 16 rows (q * Cout): backward-data of the small-hop STFTs of the mel loss (8 / 16 phase
// channels from 514 spectrum rows: 36 + 11 ms of a training step on the tiled kernel above, whose blocks then own 16 rows
// x 256 outputs and walk the channels through LDS), of the waveform discriminator's first conv, 1-channel heads.  One
// thread owns ONE base position and all the rows: each x element is read once per tap (coalesced along t), the weights
// are wave-uniform scalar loads.  Summation order: channel-major, tap-minor (as the generic kernel).
template <int MM>
__global__ __launch_bounds__(256) void conv_fewrows_kernel(ConvPlan p, const float *__restrict__ x, const float *__restrict__ wp,
                                                           const float *__restrict__ bias, const float *__restrict__ res,
                                                           float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float ws[];   // weights of one 16-channel group: [j][c16][MM]
    __shared__ int tap_live[128];   // taps whose weights are all zero in this group are skipped (a short window inside a long
                                    // n_fft: 60 of the 64 taps of the mel loss's 32-sample window)
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const bool live = t < p.Lt;
    const float *xb = x + size_t(b) * p.cin_real * p.Lin;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const int ngrp = (p.cin_real + kWG - 1) / kWG, gsz = p.J * p.M * kWG;
    for (int g = 0; g < ngrp; ++g) {
        __syncthreads();
        for (int jj = threadIdx.x; jj < 128; jj += 256) tap_live[jj] = p.J > 128 ? 1 : 0;
        __syncthreads();
        for (int e = threadIdx.x; e < p.J * kWG * MM; e += 256) {   // image order of a group: [j][m][c16]
            const int m = e % MM, c16 = (e / MM) % kWG, jj = e / (MM * kWG);
            const float w = m < p.M ? wp[size_t(g) * gsz + (size_t(jj) * p.M + m) * kWG + c16] : 0.f;
            ws[e] = w;
            if (w != 0.f && jj < 128) tap_live[jj] = 1;   // benign race: every writer stores 1
        }
        __syncthreads();
        const int cn = min(kWG, p.cin_real - g * kWG);
        for (int c16 = 0; c16 < cn; ++c16) {
            const float *xr = xb + size_t(g * kWG + c16) * p.Lin;
            for (int jj = 0; jj < p.J; ++jj) {
                if (jj < 128 && !tap_live[jj]) continue;   // uniform
                const int i = t * p.s + jj * p.d - p.P;
                const float xv = (live && i >= 0 && i < p.Lvalid) ? xr[i] : 0.f;
                const float *wv = ws + (jj * kWG + c16) * MM;   // wave-uniform address: LDS broadcast
#pragma unroll
                for (int m = 0; m < MM; ++m) acc[m] = fmaf(wv[m], xv, acc[m]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m >= p.M) break;
        const int co = m / p.q, ph = m - co * p.q;
        const int u = t * p.q + ph - p.oshift;
        if (u < 0 || u >= p.Lout) continue;
        const size_t o = (size_t(b) * p.Cout + co) * p.Lout + u;
        float v = acc[m] + (bias ? bias[co] : 0.f);
        if (p.epilogue & AGX_EPI_LEAKY_PRE) v = v > 0.f ? v : v * p.slope;
        if (p.epilogue & AGX_EPI_GELU_PRE) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        if (p.epilogue & AGX_EPI_RESIDUAL) v += res[o];
        if (p.epilogue & AGX_EPI_LEAKY_POST) v = v > 0.f ? v : v * p.slope;
        if (p.epilogue & AGX_EPI_MASK) v = p.mask[o] > 0.f ? v : v * p.slope;
        y[o] = v;
    }
}

static inline bool fewrows_ok(const ConvPlan &p) {
    const int mm = p.M > 8 ? 16 : (p.M > 4 ? 8 : (p.M > 1 ? 4 : 1));
    return p.G == 1 && p.kh == 1 && p.Tout == 1 && p.M <= 16 && p.pm_R == 0 && p.prec == 0 && p.ncv == p.cin_real &&
           p.J * mm <= 1024;   // one group of weights (J x 16 x rows) in <= 64 KB of LDS
}

const char *conv_direct_variant(const ConvPlan &p) {
    if (p.G > 1) {
        const int rpg = p.Cout / p.G;
        return rpg % 32 == 0 ? "conv_direct<32>" : (rpg % 16 == 0 ? "conv_direct<16>" : (rpg % 4 == 0 ? "conv_direct<4>" : "conv_direct<1>"));
    }
    if (narrow_ok(p)) return p.M == 32 ? "conv_narrow<16>" : (p.M == 2 ? "conv_narrow<2>" : "conv_narrow<1>");
    if (fewrows_ok(p)) return p.M > 8 ? "conv_fewrows<16>" : (p.M > 4 ? "conv_fewrows<8>" : (p.M > 1 ? "conv_fewrows<4>" : "conv_fewrows<1>"));
    return p.M >= 32 ? "conv_direct<32>" : (p.M > 4 ? "conv_direct<16>" : (p.M > 1 ? "conv_direct<4>" : "conv_direct<1>"));
}

int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                       const float *res, float *y, hipStream_t st) {
    if (narrow_ok(p)) {
        dim3 grid(ceil_div(p.Lt, 1024), p.B, p.M == 32 ? 2 : 1), block(256);
        if (grid.y > 65535) return fail(AGX_ERR_BAD_SHAPE, "conv_narrow: grid too large");
        if (p.M == 32)
            hipLaunchKernelGGL(conv_narrow_kernel<16>, grid, block, 0, st, p, x, wp, bias, res, y);
        else if (p.M == 2)
            hipLaunchKernelGGL(conv_narrow_kernel<2>, grid, block, 0, st, p, x, wp, bias, res, y);
        else
            hipLaunchKernelGGL(conv_narrow_kernel<1>, grid, block, 0, st, p, x, wp, bias, res, y);
        return check_launch("conv_narrow");
    }
    if (fewrows_ok(p)) {
        dim3 grid(ceil_div(p.Lt, 256), p.B), block(256);
        if (grid.y > 65535) return fail(AGX_ERR_BAD_SHAPE, "conv_fewrows: grid too large");
        const int mm = p.M > 8 ? 16 : (p.M > 4 ? 8 : (p.M > 1 ? 4 : 1));
        const size_t lds = size_t(p.J) * kWG * mm * sizeof(float);
        if (mm == 16) hipLaunchKernelGGL(conv_fewrows_kernel<16>, grid, block, lds, st, p, x, wp, bias, res, y);
        else if (mm == 8) hipLaunchKernelGGL(conv_fewrows_kernel<8>, grid, block, lds, st, p, x, wp, bias, res, y);
        else if (mm == 4) hipLaunchKernelGGL(conv_fewrows_kernel<4>, grid, block, lds, st, p, x, wp, bias, res, y);
        else hipLaunchKernelGGL(conv_fewrows_kernel<1>, grid, block, lds, st, p, x, wp, bias, res, y);
        return check_launch("conv_fewrows");
    }
    const int span = 255 * p.s + (p.J - 1) * p.d + 1;
    const int cpg = p.ncv / p.G;
    int ci_tile = (12 * 1024) / span;  // <= 48 KB of LDS
    if (ci_tile < 1) ci_tile = 1;
    if (ci_tile > cpg) ci_tile = cpg;
    const size_t lds = size_t(ci_tile) * span * sizeof(float);
    if (lds > 150 * 1024) return fail(AGX_ERR_UNSUPPORTED, "conv_direct: tile needs %zu B of LDS", lds);
    int co_t = p.M >= 32 ? 32 : (p.M > 4 ? 16 : (p.M > 1 ? 4 : 1));
    if (p.G > 1) {
        if (p.q != 1) return fail(AGX_ERR_UNSUPPORTED, "conv_direct: grouped polyphase layers");
        const int rpg = p.Cout / p.G;  // a block's rows must share their group
        co_t = rpg % 32 == 0 ? 32 : (rpg % 16 == 0 ? 16 : (rpg % 4 == 0 ? 4 : 1));
    }
    dim3 grid(ceil_div(p.Lt, 256), ceil_div(p.M, co_t), p.B * p.Tout), block(256);
    if (p.Tout > 1 || p.kh > 1) grid = dim3(p.B * p.Tout, ceil_div(p.M, co_t), ceil_div(p.Lt, 256));
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
