// Weight-norm fold + repack of conv weights into the kernels' K-major image.
//
// Reference: networks/utils.py:34-42 (torch.nn.utils.weight_norm, dim=0:
// w = g * v / ||v||, norm over every dim but 0) applied to the convs of
// networks/vae.py:26-29, 54-56, 76-83.
#include "common.hpp"

namespace agx {

// scale[r] = g[r] / ||v[r, :]||  (one 256-thread block per row r)
__global__ __launch_bounds__(256) void wn_scale_kernel(const float *__restrict__ v,
                                                       const float *__restrict__ g,
                                                       float *__restrict__ scale, int inner) {
    const int r = blockIdx.x;
    const float *row = v + size_t(r) * inner;
    float acc = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) acc += row[i] * row[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (part[0] + part[1]) + (part[2] + part[3]);
        scale[r] = g[r] / sqrtf(tot);
    }
}

__global__ void fill_ones_kernel(float *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.f;
}

__device__ __forceinline__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// packed[((ci/16)*J + j) * M + co*q + ph][ci%16]  (common.hpp: packed_weight_index)
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ v,
                                                   const float *__restrict__ scale,
                                                   float *__restrict__ packed, int kind, int Cin,
                                                   int Cout, int K, int q, int J, int P, int up) {
    const int M = q * Cout;
    const int64_t total = packed_weight_floats(Cin, J, M);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    // e = ((g * J + j) * M + m) * 16 + c16
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int j = gj % J, ci = (gj / J) * kWG + c16;
    if (ci >= Cin) {  // zero padding of the last channel group
        packed[e] = 0.f;
        return;
    }
    const int co = m / q, ph = m % q;
    float out = 0.f;
    if (kind == AGX_CONV_CAUSAL || kind == AGX_CONV_SAME) {
        out = v[(size_t(co) * Cin + ci) * K + j] * scale[co];
    } else if (kind == AGX_CONV_UPSAMPLE) {
        // taps k of the high-rate 'same' conv that land on low-rate offset j - P
        const int pl = (K - 1) / 2;
        const float sc = scale[co];
        for (int k = 0; k < K; ++k)
            if (floordiv(ph + k - pl, up) == j - P) out += v[(size_t(co) * Cin + ci) * K + k] * sc;
    } else {  // AGX_CONV_TRANSPOSED: weight (Cin, Cout, K), norm over dim 0 = Cin
        const int k = ph + up * (J - 1 - j);
        if (k < K) out = v[(size_t(ci) * Cout + co) * K + k] * scale[ci];
    }
    packed[e] = out;
}

}  // namespace agx

extern "C" int agx_conv_pack(const agx_conv_desc *d, const float *v, const float *g, float *packed,
                             void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!v || !packed) return fail(AGX_ERR_NULL_POINTER, "agx_conv_pack: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool transposed = d->kind == AGX_CONV_TRANSPOSED;
    const int dim0 = transposed ? d->c_in : d->c_out;
    const int inner = (transposed ? d->c_out : d->c_in) * d->kernel;
    const int64_t n_w = packed_weight_floats(p.Cin, p.J, p.M);
    float *scale = packed + n_w;  // tail scratch reserved by agx_conv_packed_floats
    if (g)
        hipLaunchKernelGGL(wn_scale_kernel, dim3(dim0), dim3(256), 0, st, v, g, scale, inner);
    else
        hipLaunchKernelGGL(fill_ones_kernel, dim3(ceil_div(dim0, 256)), dim3(256), 0, st, scale, dim0);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div64(n_w, 256)), dim3(256), 0, st, v, scale,
                       packed, d->kind, p.Cin, p.Cout, d->kernel, p.q, p.J, p.P, d->stride);
    return check_launch("agx_conv_pack");
}
