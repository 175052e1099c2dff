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

// Tiled version: one block = (group, batch item, 256 input positions) for all cpg channels of the group; the
// group's weights (scaled by 1 / sigma) and the dz tile the positions can touch are staged in LDS.
constexpr int GD_TI = 256;
template <int CPG>
__global__ __launch_bounds__(256) void grouped_bwd_data_tiled_kernel(const float *__restrict__ dz,
                                                                     const float *__restrict__ w,
                                                                     const float *__restrict__ sigma,
                                                                     const float *__restrict__ add,
                                                                     const float *__restrict__ mask, float slope,
                                                                     float *__restrict__ dx, int Cin, int Cout, int G,
                                                                     int K, int s, int P, int Lin, int Lout, int TD) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int opg = Cout / G;
    float *ws = sm;                          // [opg][CPG][K]
    float *ds = sm + opg * CPG * K;          // [opg][TD]
    const int tid = threadIdx.x, grp = blockIdx.y, b = blockIdx.z;
    const int i0 = blockIdx.x * GD_TI, i = i0 + tid;
    const float inv = sigma ? 1.f / sigma[0] : 1.f;
    // first dz position any i of the tile can read: t = (i + P - k) / s with k <= K - 1
    const int num = i0 + P - (K - 1);
    const int tbase = num >= 0 ? num / s : -((-num + s - 1) / s);
    for (int e = tid; e < opg * CPG * K; e += 256) ws[e] = w[size_t(grp) * opg * CPG * K + e] * inv;
    {
        const float *db = dz + (size_t(b) * Cout + grp * opg) * Lout;
        const int total = opg * TD;
        const float inv_td = 1.f / float(TD);
        for (int e0 = tid; e0 < total; e0 += 256 * 4) {
            float v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * 256, total - 1);
                const int o = int((float(e) + 0.5f) * inv_td), tt = e - o * TD;
                const int t = tbase + tt;
                ok[u] = t >= 0 && t < Lout;
                v[u] = db[size_t(o) * Lout + min(max(t, 0), Lout - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 256 < total) ds[e0 + u * 256] = ok[u] ? v[u] : 0.f;
        }
    }
    __syncthreads();
    if (i >= Lin) return;
    float acc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) acc[c] = 0.f;
    for (int k = (i + P) % s; k < K; k += s) {
        const int tt = (i + P - k) / s - tbase;       // i + P - k >= 0 is not guaranteed: guard below
        if (i + P - k < 0) continue;
        for (int o = 0; o < opg; ++o) {
            const float d = ds[o * TD + tt];          // zero outside [0, Lout)
#pragma unroll
            for (int c = 0; c < CPG; ++c) acc[c] = fmaf(ws[(o * CPG + c) * K + k], d, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const size_t e = (size_t(b) * Cin + grp * CPG + c) * Lin + i;
        float v = acc[c];
        if (add) v += add[e];
        if (mask) v = mask[e] > 0.f ? v : v * slope;
        dx[e] = v;
    }
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

// Tiled version of the above for the shapes that matter (cpg * K <= 256 - opg threads, opg <= 16): one block =
// one group x a slice of (batch item, 256-position tile) pairs.  The x tile [cpg][256 s + K - 1] and the dz tile
// [opg][256] are staged in LDS; thread p < cpg*K owns weight tap (cl, k) for ALL opg output channels of the
// group (one x read feeds opg FMAs, the dz reads are wave-wide broadcasts); threads cpg*K .. cpg*K+opg-1 sum
// the bias gradients.  Same part[] layout as the simple kernel.
constexpr int GW_TT = 256;
template <int OPG>
__global__ __launch_bounds__(256) void grouped_bwd_weight_tiled_kernel(const float *__restrict__ x,
                                                                       const float *__restrict__ dz,
                                                                       float *__restrict__ part, int B, int Cin,
                                                                       int Cout, int G, int K, int s, int P, int Lin,
                                                                       int Lout, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cpg = Cin / G, span = (GW_TT - 1) * s + K;
    float *xs = sm;                    // [cpg][span]
    float *ds = sm + cpg * span;       // [OPG][GW_TT]
    const int grp = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x;
    const int npair = cpg * K;
    const bool is_pair = tid < npair, is_bias = tid >= npair && tid < npair + OPG;
    const int cl = is_pair ? tid / K : 0, k = is_pair ? tid - cl * K : 0;
    float acc[OPG];
#pragma unroll
    for (int o = 0; o < OPG; ++o) acc[o] = 0.f;
    float bacc = 0.f;
    const int tiles = (Lout + GW_TT - 1) / GW_TT, items = B * tiles;
    for (int item = slice; item < items; item += n_slices) {
        const int b = item / tiles, t0 = (item - b * tiles) * GW_TT;
        __syncthreads();
        // staging: batches of clamped loads, zeros selected afterwards
        {
            const float *xb = x + (size_t(b) * Cin + grp * cpg) * Lin;
            const int in0 = t0 * s - P, total = cpg * span;
            const float inv_span = 1.f / float(span);
            for (int e0 = tid; e0 < total; e0 += 256 * 8) {
                float v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, total - 1);
                    const int c = int((float(e) + 0.5f) * inv_span), i = e - c * span;
                    const int pos = in0 + i;
                    ok[u] = pos >= 0 && pos < Lin;
                    v[u] = xb[size_t(c) * Lin + min(max(pos, 0), Lin - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < total) xs[e0 + u * 256] = ok[u] ? v[u] : 0.f;
            }
            const float *db = dz + (size_t(b) * Cout + grp * OPG) * Lout;
            const int tt = tid;   // GW_TT == 256: one column per thread, OPG rows
            const bool tok = t0 + tt < Lout;
            float v[OPG];
#pragma unroll
            for (int o = 0; o < OPG; ++o) v[o] = db[size_t(o) * Lout + min(t0 + tt, Lout - 1)];
#pragma unroll
            for (int o = 0; o < OPG; ++o) ds[o * GW_TT + tt] = tok ? v[o] : 0.f;
        }
        __syncthreads();
        if (is_pair) {
            const float *xr = xs + cl * span + k;
#pragma unroll 4
            for (int t = 0; t < GW_TT; ++t) {
                const float xv = xr[t * s];
#pragma unroll
                for (int o = 0; o < OPG; ++o) acc[o] = fmaf(ds[o * GW_TT + t], xv, acc[o]);
            }
        } else if (is_bias) {
            const float *dr = ds + (tid - npair) * GW_TT;
            for (int t = 0; t < GW_TT; ++t) bacc += dr[t];
        }
    }
    const int row = npair + 1;
    if (is_pair) {
#pragma unroll
        for (int o = 0; o < OPG; ++o) part[(size_t(slice) * Cout + grp * OPG + o) * row + tid] = acc[o];
    } else if (is_bias) {
        part[(size_t(slice) * Cout + grp * OPG + (tid - npair)) * row + npair] = bacc;
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
    int64_t ns = (4096 + p.G - 1) / p.G;            // ~16 blocks per CU in total (tiled kernel: G x ns blocks)
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cpg = p.Cin / p.G, opg = p.Cout / p.G;
    const int TD = (GD_TI + p.J - 1) / p.s + 2;
    const size_t lds = (size_t(opg) * cpg * p.J + size_t(opg) * TD) * sizeof(float);
    if ((cpg == 4 || cpg == 2) && lds <= 64 * 1024 && p.G <= 65535) {
        dim3 grid(ceil_div(p.Lin, GD_TI), p.G, p.B);
        if (cpg == 4)
            hipLaunchKernelGGL(grouped_bwd_data_tiled_kernel<4>, grid, dim3(256), lds, st, dz, w, sigma, add, mask, slope, dx,
                               p.Cin, p.Cout, p.G, p.J, p.s, p.P, p.Lin, p.Lout, TD);
        else
            hipLaunchKernelGGL(grouped_bwd_data_tiled_kernel<2>, grid, dim3(256), lds, st, dz, w, sigma, add, mask, slope, dx,
                               p.Cin, p.Cout, p.G, p.J, p.s, p.P, p.Lin, p.Lout, TD);
        return check_launch("agx_conv_grouped_bwd_data");
    }
    hipLaunchKernelGGL(grouped_bwd_data_kernel, dim3(ceil_div(p.Lin, 256), p.Cin, p.B), dim3(256), 0, st, dz, w, sigma, add,
                       mask, slope, dx, p.Cin, p.Cout, p.G, p.J, p.s, p.P, p.Lin, p.Lout);
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
    const int cpg = p.Cin / p.G, opg = p.Cout / p.G;
    const size_t lds = (size_t(cpg) * ((GW_TT - 1) * p.s + p.J) + size_t(opg) * GW_TT) * sizeof(float);
    const bool tiled = cpg * p.J + opg <= 256 && (opg == 16 || opg == 8 || opg == 4) && lds <= 64 * 1024 && p.G <= 65535;
    if (tiled) {
        dim3 grid(p.G, ns);
        if (opg == 16)
            hipLaunchKernelGGL(grouped_bwd_weight_tiled_kernel<16>, grid, dim3(256), lds, st, x, dz, part, p.B, p.Cin, p.Cout,
                               p.G, p.J, p.s, p.P, p.Lin, p.Lout, ns);
        else if (opg == 8)
            hipLaunchKernelGGL(grouped_bwd_weight_tiled_kernel<8>, grid, dim3(256), lds, st, x, dz, part, p.B, p.Cin, p.Cout,
                               p.G, p.J, p.s, p.P, p.Lin, p.Lout, ns);
        else
            hipLaunchKernelGGL(grouped_bwd_weight_tiled_kernel<4>, grid, dim3(256), lds, st, x, dz, part, p.B, p.Cin, p.Cout,
                               p.G, p.J, p.s, p.P, p.Lin, p.Lout, ns);
    } else {
        hipLaunchKernelGGL(grouped_bwd_weight_kernel, dim3(p.Cout, ns), dim3(256), 0, st, x, dz, part, p.B, p.Cin, p.Cout,
                           p.G, p.J, p.s, p.P, p.Lin, p.Lout, ns);
    }
    hipLaunchKernelGGL(grouped_bwd_reduce_kernel, dim3(p.Cout), dim3(256), 0, st, part, ns, p.Cout, row, dw, dbias);
    return check_launch("agx_conv_grouped_bwd_weight");
}

}  // extern "C"
