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

// spectral norm: every row scaled by 1 / sigma (sigma lives on the device: no host sync)
__global__ void fill_inv_sigma_kernel(float *p, int n, const float *__restrict__ sigma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.f / sigma[0];
}

__device__ __forceinline__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Folded forward weight of the polyphase form: coefficient of x[ci, t*s + j*d - P] in y[co, q*t + ph].
__device__ __forceinline__ float fwd_weight(const float *__restrict__ v, const float *__restrict__ scale,
                                            int kind, int Cin, int Cout, int K, int J, int P, int up, int ci,
                                            int j, int co, int ph) {
    if (kind == AGX_CONV_CAUSAL || kind == AGX_CONV_SAME || kind == AGX_CONV_PADDED) return v[(size_t(co) * Cin + ci) * K + j] * scale[co];
    if (kind == AGX_CONV_UPSAMPLE) {
        // taps k of the high-rate 'same' conv that land on low-rate offset j - P
        const int pl = (K - 1) / 2;
        const float sc = scale[co];
        float out = 0.f;
        for (int k = 0; k < K; ++k)
            if (floordiv(ph + k - pl, up) == j - P) out += v[(size_t(co) * Cin + ci) * K + k] * sc;
        return out;
    }
    // AGX_CONV_TRANSPOSED: weight (Cin, Cout, K), norm over dim 0 = Cin
    const int k = ph + up * (J - 1 - j);
    return k < K ? v[(size_t(ci) * Cout + co) * K + k] * scale[ci] : 0.f;
}

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
    packed[e] = fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, j, m / q, m % q);
}

// tile image (common.hpp: tile_image_index): Wt[((ci/4)*J + j) * M + m][ci%4], rows m = co*q + ph as in the standard image
__global__ __launch_bounds__(256) void pack_tile_kernel(const float *__restrict__ v, const float *__restrict__ scale,
                                                        float *__restrict__ timg, int kind, int Cin, int Cout, int K,
                                                        int q, int J, int P, int up) {
    const int M = q * Cout;
    const int64_t total = tile_image_floats(Cin, J, M);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = int(e % 4);
    const int m = int((e / 4) % M);
    const int gj = int(e / (int64_t(4) * M));
    const int j = gj % J, ci = (gj / J) * 4 + c4;
    timg[e] = ci < Cin ? fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, j, m / q, m % q) : 0.f;
}

// bf16x3 image: [((g * J + j) * M + m) * 48 + plane * 16 + c16] bf16, w = h + m + l split with round-to-nearest
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float *__restrict__ v, const float *__restrict__ scale,
                                                          __bf16 *__restrict__ packed, int kind, int Cin, int Cout, int K,
                                                          int q, int J, int P, int up) {
    const int M = q * Cout;
    const int64_t total = int64_t((Cin + kWG - 1) / kWG) * J * M * kWG;   // one thread per weight
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int j = gj % J, ci = (gj / J) * kWG + c16;
    const float w = ci < Cin ? fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, j, m / q, m % q) : 0.f;
    const __bf16 h = (__bf16)w;
    const float r1 = w - (float)h;
    const __bf16 mm = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)mm);
    __bf16 *dst = packed + (size_t(gj) * M + m) * 48 + c16;
    dst[0] = h;
    dst[16] = mm;
    dst[32] = l;
}

// B3 tile image (common.hpp): [(((g * J + j) * 3 + plane) * 2 + lh) * M + m][8] bf16, rows m = co * q + ph as in the standard image
__global__ __launch_bounds__(256) void pack_b3_tile_kernel(const float *__restrict__ v, const float *__restrict__ scale,
                                                           __bf16 *__restrict__ timg, int kind, int Cin, int Cout, int K,
                                                           int q, int J, int P, int up) {
    const int M = q * Cout;
    const int64_t total = int64_t(Cin / kWG) * J * M * kWG;   // one thread per weight
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int j = gj % J, ci = (gj / J) * kWG + c16;
    const float w = fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, j, m / q, m % q);
    const __bf16 h = (__bf16)w;
    const float r1 = w - (float)h;
    const __bf16 mm = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)mm);
    int lh, slot;
    if (J == 1 && q == 1) {   // GEMM2 order: lane half = bit 2 of the channel, slot = (bit 3) * 4 + low two bits
        lh = (c16 >> 2) & 1;
        slot = ((c16 >> 3) << 2) | (c16 & 3);
    } else {
        lh = c16 >> 3;
        slot = c16 & 7;
    }
    __bf16 *dst = timg + ((size_t(gj) * 3 * 2 + lh) * M + m) * 8 + slot;
    const size_t plane = size_t(2) * M * 8;
    dst[0] = h;
    dst[plane] = mm;
    dst[2 * plane] = l;
}

// B3 tile image from a finished standard bf16x3 image (any plan: the Conv2d forward / backward-data images of conv2d.hip) -- a pure
// permutation: [((g J + j) M + m) 48 + plane 16 + c16]  ->  [(((g J + j) 3 + plane) 2 + c16 / 8) M + m][c16 % 8]
__global__ __launch_bounds__(256) void b3_tile_from_bf_kernel(const __bf16 *__restrict__ img, __bf16 *__restrict__ timg, int64_t gj_count,
                                                              int M) {
    const int64_t total = gj_count * M * 48;
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % 16), plane = int((e / 16) % 3);
    const int m = int((e / 48) % M);
    const int64_t gj = e / (int64_t(48) * M);
    timg[(((gj * 3 + plane) * 2 + (c16 >> 3)) * M + m) * 8 + (c16 & 7)] = img[e];
}
// B3 tile image of a STRIDED Conv2d layer's forward in space-to-depth form (conv_b3.hip: conv2d_b3_kernel): virtual channel
// cv = (rho_h sw + rho_w) Cin + ci, tap jv = ah KWv + aw  <->  W[co][ci][sh ah + rho_h][sw aw + rho_w] (zero past the kernel).
__global__ __launch_bounds__(256) void pack_s2d_b3_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                                          __bf16 *__restrict__ timg, int Cin, int Cout, int kh, int kw, int sh, int sw) {
    const int KHv = (kh + sh - 1) / sh, KWv = (kw + sw - 1) / sw, Jv = KHv * KWv, Cv = sh * sw * Cin, M = Cout;
    const int64_t total = int64_t(Cv / kWG) * Jv * M * kWG;   // one thread per virtual weight
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int jv = gj % Jv, cv = (gj / Jv) * kWG + c16;
    const int phase = cv / Cin, ci = cv - phase * Cin, rho_h = phase / sw, rho_w = phase - rho_h * sw;
    const int ah = jv / KWv, aw = jv - ah * KWv, dh = sh * ah + rho_h, dw = sw * aw + rho_w;
    const float wv = (dh < kh && dw < kw) ? w[((size_t(m) * Cin + ci) * kh + dh) * kw + dw] * scale[m] : 0.f;
    const __bf16 h = (__bf16)wv;
    const float r1 = wv - (float)h;
    const __bf16 mm = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)mm);
    __bf16 *dst = timg + ((size_t(gj) * 3 * 2 + (c16 >> 3)) * M + m) * 8 + (c16 & 7);
    const size_t plane = size_t(2) * M * 8;
    dst[0] = h;
    dst[plane] = mm;
    dst[2 * plane] = l;
}
void launch_pack_s2d_b3(const float *w, const float *scale, float *timg, int Cin, int Cout, int kh, int kw, int sh, int sw,
                        hipStream_t st) {
    const int64_t total = int64_t(sh * sw * Cin) * ceil_div(kh, sh) * ceil_div(kw, sw) * Cout;
    hipLaunchKernelGGL(pack_s2d_b3_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, scale,
                       reinterpret_cast<__bf16 *>(timg), Cin, Cout, kh, kw, sh, sw);
}
void launch_b3_tile_from_bf(const float *img, float *timg, int64_t gj_count, int M, hipStream_t st) {
    const int64_t total = gj_count * M * 48;
    hipLaunchKernelGGL(b3_tile_from_bf_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st,
                       reinterpret_cast<const __bf16 *>(img), reinterpret_cast<__bf16 *>(timg), gj_count, M);
}

// Packed image of the BACKWARD-DATA op (core.hip: lower_conv_bwd_data).  (Cin, Cout, K, q, J, P, s) are
// the FORWARD plan's; the image has fwd-Cout "input" channels and M_b = q_b * fwd-Cin rows.
__global__ __launch_bounds__(256) void pack_bwd_kernel(const float *__restrict__ v,
                                                       const float *__restrict__ scale,
                                                       float *__restrict__ packed, int kind, int Cin,
                                                       int Cout, int K, int q, int J, int P, int s, int up,
                                                       int Jb, int qb) {
    const int Mb = qb * Cin;
    const int64_t total = packed_weight_floats(Cout, Jb, Mb);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % Mb);
    const int gj = int(e / (int64_t(kWG) * Mb));
    const int jb = gj % Jb, co = (gj / Jb) * kWG + c16;  // bwd input channel = fwd output channel
    if (co >= Cout) {
        packed[e] = 0.f;
        return;
    }
    const int ci = m / qb, phb = m % qb;  // bwd output channel = fwd input channel
    float out = 0.f;
    if (q == 1 && s == 1) {
        out = fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, J - 1 - jb, co, 0);
    } else if (q == 1) {  // strided conv: tap p' + s*m, m = Jb-1-jb
        const int k = phb + s * (Jb - 1 - jb);
        if (k < J) out = fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, k, co, 0);
    } else {              // polyphase forward: k = jb, p = k % q, j = J-1 - k / q
        out = fwd_weight(v, scale, kind, Cin, Cout, K, J, P, up, ci, J - 1 - jb / q, co, jb % q);
    }
    packed[e] = out;
}

}  // namespace agx

static int pack_scales(const agx_conv_desc *d, const float *v, const float *g, float *scale, hipStream_t st) {
    using namespace agx;
    const bool transposed = d->kind == AGX_CONV_TRANSPOSED;
    const int dim0 = transposed ? d->c_in : d->c_out;
    const int inner = (transposed ? d->c_out : d->c_in) * d->kernel;
    if (g)
        hipLaunchKernelGGL(wn_scale_kernel, dim3(dim0), dim3(256), 0, st, v, g, scale, inner);
    else
        hipLaunchKernelGGL(fill_ones_kernel, dim3(ceil_div(dim0, 256)), dim3(256), 0, st, scale, dim0);
    return dim0;
}

extern "C" int64_t agx_conv_bwd_packed_floats(const agx_conv_desc *d) {
    agx::ConvPlan b;
    int rc = agx::lower_conv_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    const int dim0 = (d->kind == AGX_CONV_TRANSPOSED) ? d->c_in : d->c_out;
    return agx::packed_weight_floats(b.Cin, b.J, b.M) + dim0;
}

static int pack_backward(const agx_conv_desc *d, const float *v, const float *g, const float *sigma, float *packed,
                         hipStream_t st, const char *who) {
    using namespace agx;
    ConvPlan f, b;
    int rc = lower_conv(d, &f);
    if (rc != AGX_OK) return rc;
    rc = lower_conv_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    if (!v || !packed) return fail(AGX_ERR_NULL_POINTER, "%s: NULL pointer", who);
    const int64_t n_w = packed_weight_floats(b.Cin, b.J, b.M);
    float *scale = packed + n_w;
    if (sigma) {
        const int dim0 = d->kind == AGX_CONV_TRANSPOSED ? d->c_in : d->c_out;
        hipLaunchKernelGGL(fill_inv_sigma_kernel, dim3(ceil_div(dim0, 256)), dim3(256), 0, st, scale, dim0, sigma);
    } else {
        pack_scales(d, v, g, scale, st);
    }
    hipLaunchKernelGGL(pack_bwd_kernel, dim3((unsigned)ceil_div64(n_w, 256)), dim3(256), 0, st, v, scale, packed,
                       d->kind, f.Cin, f.Cout, d->kernel, f.q, f.J, f.P, f.s, d->stride, b.J, b.q);
    return check_launch(who);
}

extern "C" int agx_conv_pack_bwd(const agx_conv_desc *d, const float *v, const float *g, float *packed,
                                 void *stream) {
    return pack_backward(d, v, g, nullptr, packed, static_cast<hipStream_t>(stream), "agx_conv_pack_bwd");
}

extern "C" int agx_conv_pack_bwd_sigma(const agx_conv_desc *d, const float *w, const float *sigma, float *packed,
                                       void *stream) {
    if (!sigma) return agx::fail(AGX_ERR_NULL_POINTER, "agx_conv_pack_bwd_sigma: NULL sigma");
    return pack_backward(d, w, nullptr, sigma, packed, static_cast<hipStream_t>(stream), "agx_conv_pack_bwd_sigma");
}

static int pack_forward(const agx_conv_desc *d, const float *v, const float *g, const float *sigma,
                        float *packed, hipStream_t st, const char *who) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!v || !packed) return fail(AGX_ERR_NULL_POINTER, "%s: NULL pointer", who);
    const bool transposed = d->kind == AGX_CONV_TRANSPOSED;
    const int cpg = p.Cin / p.G;  // the weight tensor is (Cout, Cin / groups, K)
    const int dim0 = transposed ? d->c_in : d->c_out;
    const int inner = (transposed ? d->c_out : cpg) * d->kernel;
    const int64_t n_w = p.prec ? packed_weight_floats_bf(p.Cin, p.J, p.M) : packed_weight_floats(cpg, p.J, p.M);
    float *scale = packed + n_w;  // tail scratch reserved by agx_conv_packed_floats
    if (p.prec && p.G != 1) return fail(AGX_ERR_UNSUPPORTED, "%s: bf16x3 images are for dense layers", who);
    if (g)
        hipLaunchKernelGGL(wn_scale_kernel, dim3(dim0), dim3(256), 0, st, v, g, scale, inner);
    else if (sigma)
        hipLaunchKernelGGL(fill_inv_sigma_kernel, dim3(ceil_div(dim0, 256)), dim3(256), 0, st, scale, dim0, sigma);
    else
        hipLaunchKernelGGL(fill_ones_kernel, dim3(ceil_div(dim0, 256)), dim3(256), 0, st, scale, dim0);
    if (p.prec) {
        const int64_t nthreads = int64_t(ceil_div(p.Cin, kWG)) * p.J * p.M * kWG;
        hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)ceil_div64(nthreads, 256)), dim3(256), 0, st, v, scale,
                           reinterpret_cast<__bf16 *>(packed), d->kind, p.Cin, p.Cout, d->kernel, p.q, p.J, p.P, d->stride);
        if (p.tile_off >= 0)   // second copy in the DMA layout of resblock_b3.hip, behind the scale scratch
            hipLaunchKernelGGL(pack_b3_tile_kernel, dim3((unsigned)ceil_div64(nthreads, 256)), dim3(256), 0, st, v, scale,
                               reinterpret_cast<__bf16 *>(packed + p.tile_off), d->kind, p.Cin, p.Cout, d->kernel, p.q, p.J, p.P,
                               d->stride);
        return check_launch(who);
    }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div64(n_w, 256)), dim3(256), 0, st, v, scale,
                       packed, d->kind, cpg, p.Cout, d->kernel, p.q, p.J, p.P, d->stride);
    if (p.tile_off >= 0) {   // second copy in the LDS-DMA layout of resblock_p.hip / conv_p.hip, behind the scale scratch
        const int64_t n_t = tile_image_floats(p.Cin, p.J, p.M);
        hipLaunchKernelGGL(pack_tile_kernel, dim3((unsigned)ceil_div64(n_t, 256)), dim3(256), 0, st, v, scale,
                           packed + p.tile_off, d->kind, p.Cin, p.Cout, d->kernel, p.q, p.J, p.P, d->stride);
    }
    return check_launch(who);
}

extern "C" int agx_conv_pack(const agx_conv_desc *d, const float *v, const float *g, float *packed,
                             void *stream) {
    return pack_forward(d, v, g, nullptr, packed, static_cast<hipStream_t>(stream), "agx_conv_pack");
}

extern "C" int agx_conv_pack_sigma(const agx_conv_desc *d, const float *w, const float *sigma, float *packed,
                                   void *stream) {
    if (!sigma) return agx::fail(AGX_ERR_NULL_POINTER, "agx_conv_pack_sigma: NULL sigma");
    return pack_forward(d, w, nullptr, sigma, packed, static_cast<hipStream_t>(stream), "agx_conv_pack_sigma");
}

// 2-D layers: the (Cout, Cin, kh, kw) tensor read as (Cout, Cin * kh, kw) is already in virtual-channel
// order c' = ci * kh + dh (common.hpp), so the 1-D pack kernel applies as is.
namespace agx {
void launch_pack_tile2d(const float *w, const float *scale, const float *sigma, float *timg, int C, int M, int kh, int kw,
                        int bwd, hipStream_t st, int sh, int sw, int kh_w, int kw_w);   // conv2d.hip
int64_t conv2d_b3_tile_floats(const ConvPlan &p);   // conv_b3.hip
}
extern "C" int64_t agx_conv2d_packed_floats(const agx_conv2d_desc *d) {
    agx::ConvPlan p;
    int rc = agx::lower_conv2d(d, &p);
    if (rc != AGX_OK) return rc;
    const int64_t tile = p.tile_off < 0 ? 0 : p.prec ? agx::conv2d_b3_tile_floats(p)                      // B3 tile image (conv_b3.hip)
                                                     : agx::tile_image_floats(p.kh * p.Cin, p.J / p.kh, p.M);
    return (p.prec ? agx::packed_weight_floats_bf(p.ncv, p.J, p.M) : agx::packed_weight_floats(p.ncv, p.J, p.M)) + p.Cout + tile;
}

extern "C" int agx_conv2d_pack(const agx_conv2d_desc *d, const float *w, const float *sigma, float *packed,
                               void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv2d(d, &p);
    if (rc != AGX_OK) return rc;
    if (!w || !packed) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_pack: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_w = p.prec ? packed_weight_floats_bf(p.ncv, p.J, p.M) : packed_weight_floats(p.ncv, p.J, p.M);
    float *scale = packed + n_w;
    if (sigma)
        hipLaunchKernelGGL(fill_inv_sigma_kernel, dim3(ceil_div(p.Cout, 256)), dim3(256), 0, st, scale, p.Cout, sigma);
    else
        hipLaunchKernelGGL(fill_ones_kernel, dim3(ceil_div(p.Cout, 256)), dim3(256), 0, st, scale, p.Cout);
    if (p.prec) {
        const int64_t nthreads = int64_t(ceil_div(p.ncv, kWG)) * p.J * p.M * kWG;
        hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)ceil_div64(nthreads, 256)), dim3(256), 0, st, w, scale,
                           reinterpret_cast<__bf16 *>(packed), AGX_CONV_PADDED, p.ncv, p.Cout, p.J, 1, p.J, p.P, 1);
        if (p.tile_off >= 0 && (p.s > 1 || p.sh > 1))   // strided forward layer: the space-to-depth weights
            launch_pack_s2d_b3(w, scale, packed + p.tile_off, p.Cin, p.Cout, p.kh, p.J / p.kh, p.sh, p.s, st);
        else if (p.tile_off >= 0)   // second copy in the DMA layout of conv2d_b3_kernel, behind the scale scratch
            launch_b3_tile_from_bf(packed, packed + p.tile_off, int64_t(ceil_div(p.ncv, kWG)) * p.J, p.M, st);
        return check_launch("agx_conv2d_pack");
    }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div64(n_w, 256)), dim3(256), 0, st, w, scale, packed,
                       AGX_CONV_PADDED, p.ncv, p.Cout, p.J, 1, p.J, p.P, 1);
    if (p.tile_off >= 0)   // second copy for the ring kernel (conv_p.hip, D2 geometries)
        launch_pack_tile2d(w, scale, nullptr, packed + p.tile_off, p.Cin, p.M, p.kh, p.J / p.kh, 0, st, 1, 1, 0, 0);
    return check_launch("agx_conv2d_pack");
}
