// Backward of grouped Conv1d layers (AGX_CONV_PADDED with groups > 1): the strided grouped convs of the
// waveform discriminator (discriminator.py:33-38: 2-4 input channels per group, k = 41, stride 4).
// Too thin for MFMA tiles (K-dimension of 8-16 per group), so both are VALU kernels reading the torch
// weight layout directly; 1 / sigma of the spectral norm is applied on the fly.
#include "common.hpp"

namespace agx {

// dx[b, ci, i] = sum_{co in group(ci)} sum_{k : (i + P - k) % s == 0} W[co, ci_l, k] / sigma * dz[b, co, (i + P - k) / s]
__global__ __launch_bounds__(256) void grouped_bwd_data_kernel(const float *__restrict__ dz, const float *__restrict__ w,
                                                               const float *__restrict__ sigma,
                                                               const float *__restrict__ add,
                                                               const float *__restrict__ mask, float slope,
                                                               float *__restrict__ dx, int Cin, int Cout, int G, int K,
                                                               int s, int P, int Lin, int Lout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ci = blockIdx.y, b = blockIdx.z;
    if (i >= Lin) return;
    const int cpg = Cin / G, opg = Cout / G, grp = ci / cpg, cl = ci - grp * cpg;
    const float inv = sigma ? 1.f / sigma[0] : 1.f;
    float acc = 0.f;
    const int k0 = (i + P) % s;  // taps congruent to i + P
    for (int k = k0; k < K; k += s) {
        const int t = (i + P - k) / s;
        if (i + P - k < 0 || t >= Lout) continue;
        for (int o = 0; o < opg; ++o) {
            const int co = grp * opg + o;
            acc = fmaf(w[(size_t(co) * cpg + cl) * K + k], dz[(size_t(b) * Cout + co) * Lout + t], acc);
        }
    }
    acc *= inv;
    const size_t e = (size_t(b) * Cin + ci) * Lin + i;
    if (add) acc += add[e];
    if (mask) acc = mask[e] > 0.f ? acc : acc * slope;
    dx[e] = acc;
}

// part[slice][co][e]: e < cpg*K  <->  dW[co, ci_l, k] ; e == cpg*K  <->  dbias[co]   (sum over this slice's (b, t))
__global__ __launch_bounds__(256) void grouped_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                                 float *__restrict__ part, int B, int Cin, int Cout,
                                                                 int G, int K, int s, int P, int Lin, int Lout,
                                                                 int n_slices) {
    const int co = blockIdx.x, slice = blockIdx.y;
    const int cpg = Cin / G, opg = Cout / G, grp = co / opg;
    const int row = cpg * K + 1;
    const int64_t total = int64_t(B) * Lout;
    const int64_t per = (total + n_slices - 1) / n_slices;
    const int64_t lo = int64_t(slice) * per, hi = min(lo + per, total);
    for (int e = threadIdx.x; e < row; e += 256) {
        const bool is_bias = e == cpg * K;
        const int cl = is_bias ? 0 : e / K, k = is_bias ? 0 : e - cl * K;
        const int ci = grp * cpg + cl;
        float acc = 0.f;
        int b = int(lo / Lout), t = int(lo - int64_t(b) * Lout);
        for (int64_t n = hi - lo; n > 0; --n, ++t) {
            if (t == Lout) {
                t = 0;
                ++b;
            }
            const float d = dz[(size_t(b) * Cout + co) * Lout + t];   // wave-uniform address: one broadcast load
            if (is_bias) {
                acc += d;
            } else {
                const int pos = t * s + k - P;
                if (pos >= 0 && pos < Lin) acc = fmaf(d, x[(size_t(b) * Cin + ci) * Lin + pos], acc);
            }
        }
        part[(size_t(slice) * Cout + co) * row + e] = acc;
    }
}

// dw[co][e] / dbias[co] = sum over slices, fixed order
__global__ __launch_bounds__(256) void grouped_bwd_reduce_kernel(const float *__restrict__ part, int n_slices, int Cout,
                                                                 int row, float *__restrict__ dw,
                                                                 float *__restrict__ dbias) {
    const int co = blockIdx.x;
    for (int e = threadIdx.x; e < row; e += 256) {
        float acc = 0.f;
        for (int sl = 0; sl < n_slices; ++sl) acc += part[(size_t(sl) * Cout + co) * row + e];
        if (e < row - 1) dw[size_t(co) * (row - 1) + e] = acc;
        else if (dbias) dbias[co] = acc;
    }
}

static int grouped_slices(const ConvPlan &p) {
    const int64_t total = int64_t(p.B) * p.Lout;
    int64_t ns = (4096 + p.Cout - 1) / p.Cout;     // ~16 blocks per CU in total
    if (ns > total / 64) ns = total / 64;
    if (ns < 1) ns = 1;
    if (ns > 1024) ns = 1024;
    return int(ns);
}

}  // namespace agx

extern "C" {

int agx_conv_grouped_bwd_data(const agx_conv_desc *d, const float *dz, const float *w, const float *sigma,
                              const float *add, const float *mask, float slope, float *dx, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (d->kind != AGX_CONV_PADDED || p.d != 1) return fail(AGX_ERR_UNSUPPORTED, "grouped_bwd_data: AGX_CONV_PADDED, dilation 1 only");
    if (!dz || !w || !dx) return fail(AGX_ERR_NULL_POINTER, "grouped_bwd_data: NULL pointer");
    if (p.Cin > 65535 || p.B > 65535) return fail(AGX_ERR_BAD_SHAPE, "grouped_bwd_data: grid too large");
    hipLaunchKernelGGL(grouped_bwd_data_kernel, dim3(ceil_div(p.Lin, 256), p.Cin, p.B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dz, w, sigma, add, mask, slope, dx, p.Cin, p.Cout, p.G, p.J, p.s,
                       p.P, p.Lin, p.Lout);
    return check_launch("agx_conv_grouped_bwd_data");
}

size_t agx_conv_grouped_bwd_weight_workspace_bytes(const agx_conv_desc *d) {
    using namespace agx;
    ConvPlan p;
    if (lower_conv(d, &p) != AGX_OK) return 0;
    return size_t(grouped_slices(p)) * p.Cout * (size_t(p.Cin / p.G) * p.J + 1) * sizeof(float);
}

int agx_conv_grouped_bwd_weight(const agx_conv_desc *d, const float *x, const float *dz, float *dw, float *dbias,
                                void *workspace, size_t workspace_bytes, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (d->kind != AGX_CONV_PADDED || p.d != 1) return fail(AGX_ERR_UNSUPPORTED, "grouped_bwd_weight: AGX_CONV_PADDED, dilation 1 only");
    if (!x || !dz || !dw || !workspace) return fail(AGX_ERR_NULL_POINTER, "grouped_bwd_weight: NULL pointer");
    if (workspace_bytes < agx_conv_grouped_bwd_weight_workspace_bytes(d))
        return fail(AGX_ERR_WORKSPACE, "grouped_bwd_weight: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ns = grouped_slices(p), row = (p.Cin / p.G) * p.J + 1;
    if (p.Cout > 65535) return fail(AGX_ERR_BAD_SHAPE, "grouped_bwd_weight: grid too large");
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(grouped_bwd_weight_kernel, dim3(p.Cout, ns), dim3(256), 0, st, x, dz, part, p.B, p.Cin, p.Cout, p.G,
                       p.J, p.s, p.P, p.Lin, p.Lout, ns);
    hipLaunchKernelGGL(grouped_bwd_reduce_kernel, dim3(p.Cout), dim3(256), 0, st, part, ns, p.Cout, row, dw, dbias);
    return check_launch("agx_conv_grouped_bwd_weight");
}

}  // extern "C"
