// torch.nn.Conv2d on the 1-D polyphase conv kernels (discriminator.py:101-114, 150-167).
//
//   y[b, co, t, f] = bias[co] + sum_{ci, dh, dw} W[co, ci, dh, dw] x[b, ci, t*sh + dh - ph, f*sw + dw - pw]
//
// is, for one output row (b, t), a 1-D conv along f over the kh * Cin "virtual channels"
// c' = ci * kh + dh whose rows are rows t*sh + dh - ph of the input planes (zero rows outside the
// image).  The MFMA kernel's LDS-DMA staging takes its row pointers from a RowMap (mfma_tile.hpp), so
// the same tiles, pipeline and epilogue serve; blockIdx.x walks (b, t).
#include "common.hpp"

namespace agx {
int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                       const float *res, float *y, hipStream_t st);
int launch_conv_mfma(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                     const float *res, float *y, hipStream_t st);
bool conv_mfma_supported(const ConvPlan &p);
const char *conv_mfma_variant(const ConvPlan &p);
const char *conv_direct_variant(const ConvPlan &p);
bool conv_p2d_supported(const ConvPlan &p);
const char *conv_p2d_variant(const ConvPlan &p);
int launch_conv_p2d(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                    hipStream_t st);
bool conv2d_b3_supported(const ConvPlan &p);
const char *conv2d_b3_variant(const ConvPlan &p);
int launch_conv2d_b3(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                     hipStream_t st);
void launch_b3_tile_from_bf(const float *img, float *timg, int64_t gj_count, int M, hipStream_t st);   // pack.hip

int lower_conv2d(const agx_conv2d_desc *d, ConvPlan *p) {
    if (!d) return fail(AGX_ERR_NULL_POINTER, "conv2d descriptor is NULL");
    if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0)
        return fail(AGX_ERR_BAD_SHAPE, "conv2d: bad dimension");
    if (d->h_in + 2 * d->pad_h < d->kh || d->w_in + 2 * d->pad_w < d->kw)
        return fail(AGX_ERR_BAD_SHAPE, "conv2d: kernel larger than the padded input");
    if (d->epilogue & ~AGX_EPI_LEAKY_PRE) return fail(AGX_ERR_UNSUPPORTED, "conv2d: only the LEAKY_PRE epilogue");
    // Two lowerings (the packed image differs, so pack and forward both come through here):
    //  * patches   -- MFMA path for Cin % 16 == 0, Cout >= 32: real channels, taps j = dh * kw + dw, a tile is
    //                 several output rows x columns staged as one 2-D input patch per channel;
    //  * row-folded -- everything else (the 2-channel 7x7 first conv, the 1-channel final conv, the direct
    //                 kernel): virtual channels c' = ci * kh + dh, one output row per tile.
    const bool patch = d->impl != AGX_IMPL_DIRECT && d->c_in % kWG == 0 && d->c_out >= 8;   // rows >= M are clamped / masked
    p->B = d->batch;
    p->cin_real = d->c_in;
    p->ncv = patch ? d->c_in : d->c_in * d->kh;
    p->Cin = ceil_div(p->ncv, kWG) * kWG;
    p->Cout = d->c_out;
    p->Lin = p->Lvalid = d->w_in;
    p->q = 1;
    p->J = patch ? d->kh * d->kw : d->kw;
    p->s = d->stride_w;
    p->d = 1;
    p->P = d->pad_w;
    p->Lt = p->Lout = (d->w_in + 2 * d->pad_w - d->kw) / d->stride_w + 1;
    p->M = d->c_out;
    p->epilogue = d->epilogue;
    p->slope = d->slope;
    p->oshift = 0;
    p->mask = nullptr;
    p->G = 1;
    p->kh = d->kh;
    p->sh = d->stride_h;
    p->ph = d->pad_h;
    p->Tin = d->h_in;
    p->Tout = (d->h_in + 2 * d->pad_h - d->kh) / d->stride_h + 1;
    p->x_cstride = int64_t(d->h_in) * d->w_in;
    p->y_cstride = int64_t(p->Tout) * p->Lout;
    p->pm_R = patch ? 1 : 0;  // the launcher fixes the tile (conv_mfma.hip: tile_span)
    p->pm_WF = 0;
    p->qh = 1;
    p->oshift_h = 0;
    p->Tt = p->Tout;
    p->prec = (patch && d->impl == AGX_IMPL_MFMA_BF16X3) ? 1 : 0;   // bf16x3: patch layers only
    if (int64_t(p->B) * p->Tout > (int64_t(1) << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv2d: batch * output rows too large");
    // layers the persistent ring kernel covers (conv_p.hip, D2 geometries) carry a tile image behind the scale scratch
    p->tile_off = -1;
    if (patch && d->impl == AGX_IMPL_AUTO && conv_p2d_geometry(*p) != 0) p->tile_off = packed_weight_floats(p->ncv, p->J, p->M) + p->Cout;
    // bf16x3 layers with the ring form (conv_b3.hip: 3 x 3, stride 1) carry a B3 tile image behind the scale scratch
    if (p->prec == 1 && conv2d_b3_geometry(*p) != 0) p->tile_off = packed_weight_floats_bf(p->ncv, p->J, p->M) + p->Cout;
    return AGX_OK;
}

// Gradient w.r.t. the layer input.  With i' = i + ph = sh t' + a (and the same along the columns),
//   dx[ci, sh t' + a - ph, sw f' + c - pw] = sum_{co, mh, mw} W[co, ci, a + sh mh, c + sw mw] dy[co, t' - mh, f' - mw]:
// a stride-1 conv over dy with a ceil(kh/sh) x ceil(kw/sw) kernel whose M = Cin*sh*sw rows carry the output
// phases (a, c) -- the 2-D form of core.hip:lower_conv_bwd_data, on the patch tiles.  Layers the patch
// form does not cover (fewer than 32 rows: the 2-channel first conv) must be stride 1 and run row-folded
// with the flipped kernel.
int lower_conv2d_bwd_data(const agx_conv2d_desc *d, ConvPlan *b) {
    ConvPlan f;
    int rc = lower_conv2d(d, &f);
    if (rc != AGX_OK) return rc;
    const int Hout = f.Tout, Wout = f.Lout;
    const bool patch = d->impl != AGX_IMPL_DIRECT && d->c_out % kWG == 0 && d->c_in * d->stride_h * d->stride_w >= 32;
    *b = f;
    b->cin_real = d->c_out;
    b->Cout = d->c_in;
    b->Lin = b->Lvalid = Wout;
    b->Tin = Hout;
    b->Lout = d->w_in;
    b->Tout = d->h_in;
    b->x_cstride = int64_t(Hout) * Wout;
    b->y_cstride = int64_t(d->h_in) * d->w_in;
    b->epilogue = 0;
    b->mask = nullptr;
    b->prec = 0;
    b->d = 1;
    b->s = 1;
    b->sh = 1;
    if (patch) {
        const int Jh = ceil_div(d->kh, d->stride_h), Jw = ceil_div(d->kw, d->stride_w);
        b->ncv = d->c_out;
        b->Cin = d->c_out;
        b->q = d->stride_w;
        b->qh = d->stride_h;
        b->J = Jh * Jw;
        b->kh = Jh;
        b->P = Jw - 1;
        b->ph = Jh - 1;
        b->Lt = ceil_div(d->w_in + d->pad_w, d->stride_w);
        b->Tt = ceil_div(d->h_in + d->pad_h, d->stride_h);
        b->oshift = d->pad_w;
        b->oshift_h = d->pad_h;
        // stride-1 axes: the tight form (flipped kernel, padding k - 1 - p) has no shifted / wasted base positions
        if (d->stride_w == 1 && d->kw - 1 - d->pad_w >= 0) {
            b->P = d->kw - 1 - d->pad_w;
            b->Lt = d->w_in;
            b->oshift = 0;
        }
        if (d->stride_h == 1 && d->kh - 1 - d->pad_h >= 0) {
            b->ph = d->kh - 1 - d->pad_h;
            b->Tt = d->h_in;
            b->oshift_h = 0;
        }
        b->M = d->c_in * d->stride_h * d->stride_w;
        b->pm_R = 1;
        b->prec = d->impl == AGX_IMPL_MFMA_BF16X3 ? 1 : 0;
    } else if (d->c_out <= 4 || d->stride_h != 1 || d->stride_w != 1 || d->kh - 1 - d->pad_h < 0 || d->kw - 1 - d->pad_w < 0) {
        // few dy channels (the 1-channel final conv: kh * kw * Cout <= 32 MACs per dx element -- a bandwidth-bound
        // gather, 0.25-0.9 ms on the GEMM kernels against 0.03) and odd shapes (strided layers with few / unaligned
        // channels: the tiny test models): gather kernel below
        b->pm_R = -1;
        b->ncv = d->c_out;
        b->J = d->kh * d->kw;
        b->M = d->c_in;
        return AGX_OK;
    } else {
        b->ncv = d->c_out * d->kh;
        b->Cin = ceil_div(b->ncv, kWG) * kWG;
        b->q = 1;
        b->qh = 1;
        b->J = d->kw;
        b->kh = d->kh;
        b->P = d->kw - 1 - d->pad_w;
        b->ph = d->kh - 1 - d->pad_h;
        b->Lt = d->w_in;
        b->Tt = d->h_in;
        b->M = d->c_in;
        b->oshift = 0;
        b->oshift_h = 0;
        b->pm_R = 0;
    }
    b->pm_WF = 0;
    b->tile_off = -1;
    if (patch && d->impl == AGX_IMPL_AUTO && conv_p2d_geometry(*b) != 0) b->tile_off = packed_weight_floats(b->ncv, b->J, b->M);
    if (patch && b->prec == 1 && conv2d_b3_geometry(*b) != 0) b->tile_off = packed_weight_floats_bf(b->ncv, b->J, b->M);
    return AGX_OK;
}

// Tile image of a row-folded layer for conv_p.hip's D2 geometries: virtual channel v = a * C + c (kernel row a, input
// channel c of THIS op), Wt[((v / 4) * kw + j) * M + m][v % 4] (common.hpp: tile_image_index with Cin -> kh * C).
//   forward            C = Cin,  M = Cout:  w[m][c][a][j] * scale[m]
//   backward-data      C = Cout, (kh, kw) = the phase GEMM's (Jh, Jw) = ceil(k / s), M = Cin * sh * sw with
//                      m = (ci * sh + ra) * sw + rc:  w[c][ci][ra + sh (Jh-1-a)][rc + sw (Jw-1-j)] / sigma, 0 beyond the kernel
//                      (stride 1: the flipped kernel)
__global__ __launch_bounds__(256) void pack_tile2d_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                                          const float *__restrict__ sigma, float *__restrict__ timg, int C,
                                                          int M, int kh, int kw, int bwd, int sh, int sw, int kh_w, int kw_w) {
    const int64_t total = tile_image_floats(kh * C, kw, M);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = int(e % 4);
    const int m = int((e / 4) % M);
    const int gj = int(e / (int64_t(4) * M));
    const int j = gj % kw, v = (gj / kw) * 4 + c4;
    float out = 0.f;
    if (v < kh * C) {
        const int a = v / C, c = v - a * C;
        if (!bwd) {
            out = w[((size_t(m) * C + c) * kh + a) * kw + j] * (scale ? scale[m] : 1.f);
        } else {
            const int rc = m % sw, ra = (m / sw) % sh, ci = m / (sw * sh), Cin_w = M / (sh * sw);
            const int dh = ra + sh * (kh - 1 - a), dw = rc + sw * (kw - 1 - j);
            if (dh < kh_w && dw < kw_w)
                out = w[((size_t(c) * Cin_w + ci) * kh_w + dh) * kw_w + dw] * (sigma ? 1.f / sigma[0] : 1.f);
        }
    }
    timg[e] = out;
}

void launch_pack_tile2d(const float *w, const float *scale, const float *sigma, float *timg, int C, int M, int kh, int kw,
                        int bwd, hipStream_t st, int sh = 1, int sw = 1, int kh_w = 0, int kw_w = 0) {
    const int64_t nt = tile_image_floats(kh * C, kw, M);
    hipLaunchKernelGGL(pack_tile2d_kernel, dim3((unsigned)ceil_div64(nt, 256)), dim3(256), 0, st, w, scale, sigma, timg, C, M, kh,
                       kw, bwd, sh, sw, kh_w ? kh_w : kh, kw_w ? kw_w : kw);
}

// Catch-all backward-data: one thread per dx element, gather over (co, dh, dw).  wimg = the weight tensor
// (Cout, Cin, kh, kw) scaled by 1 / sigma.
__global__ __launch_bounds__(256) void conv2d_bwd_data_gather_kernel(const float *__restrict__ dy,
                                                                     const float *__restrict__ wimg,
                                                                     const float *__restrict__ add,
                                                                     const float *__restrict__ mask, float slope,
                                                                     float *__restrict__ dx, int B, int Cin, int Cout,
                                                                     int Hin, int Win, int Hout, int Wout, int kh, int kw,
                                                                     int sh, int sw, int ph, int pw) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = int64_t(B) * Cin * Hin * Win;
    if (e >= total) return;
    const int j = int(e % Win), i = int((e / Win) % Hin), ci = int((e / (int64_t(Win) * Hin)) % Cin);
    const int b = int(e / (int64_t(Win) * Hin * Cin));
    float acc = 0.f;
    for (int dh = 0; dh < kh; ++dh) {
        const int ti = i + ph - dh;
        if (ti < 0 || ti % sh) continue;
        const int t = ti / sh;
        if (t >= Hout) continue;
        for (int dw = 0; dw < kw; ++dw) {
            const int fj = j + pw - dw;
            if (fj < 0 || fj % sw) continue;
            const int f = fj / sw;
            if (f >= Wout) continue;
            for (int co = 0; co < Cout; ++co)
                acc = fmaf(wimg[((size_t(co) * Cin + ci) * kh + dh) * kw + dw],
                           dy[((size_t(b) * Cout + co) * Hout + t) * Wout + f], acc);
        }
    }
    if (add) acc += add[e];
    if (mask) acc = mask[e] > 0.f ? acc : acc * slope;
    dx[e] = acc;
}

__global__ __launch_bounds__(256) void scale_copy_kernel(const float *__restrict__ w, const float *__restrict__ sigma,
                                                         float *__restrict__ out, int64_t n) {
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e < n) out[e] = w[e] * (sigma ? 1.f / sigma[0] : 1.f);
}

// Backward-data of a layer with very few input channels (the 2-channel 7x7 first conv of the STFT discriminators):
// M = Cin would waste the MFMA tile (and the direct kernel manages 4 TFLOP/s), so the kernel's column axis is
// peeled off:   P[(c, dw)][i][j'] = sum_{co, dh} W[co, c, dh, dw] dy[co][i - dh + ph][j']     (a (kh x 1) Conv2d with
// kw * Cin output rows -- on the patch tiles)    then    dx[c][i][j] = sum_dw P[(c, dw)][i][j - dw + pw].
__global__ __launch_bounds__(256) void colsplit_weights_kernel(const float *__restrict__ w, const float *__restrict__ sigma,
                                                               float *__restrict__ wp, int Cin, int Cout, int kh, int kw) {
    // wp (Cin*kw, Cout, kh, 1):  wp[(c*kw + dw)][co][a] = w[co][c][kh-1-a][dw] / sigma
    const int64_t total = int64_t(Cin) * kw * Cout * kh;
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int a = int(e % kh), co = int((e / kh) % Cout), m = int(e / (int64_t(kh) * Cout));
    const int c = m / kw, dw = m - c * kw;
    wp[e] = w[((size_t(co) * Cin + c) * kh + (kh - 1 - a)) * kw + dw] * (sigma ? 1.f / sigma[0] : 1.f);
}

__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ pbuf, const float *__restrict__ add,
                                                     float *__restrict__ dx, int C, int kw, int H, int Wp, int Wx, int pw) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y, bc = blockIdx.z;   // bc = b * C + c
    if (j >= Wx) return;
    const int b = bc / C, c = bc - b * C;
    float acc = 0.f;
    for (int dw = 0; dw < kw; ++dw) {
        const int jp = j - dw + pw;
        if (jp >= 0 && jp < Wp) acc += pbuf[((size_t(b) * C * kw + c * kw + dw) * H + i) * Wp + jp];
    }
    const size_t e = (size_t(bc) * H + i) * Wx + j;
    dx[e] = acc + (add ? add[e] : 0.f);
}

// Packed image of the backward-data op (both lowerings above).
__global__ __launch_bounds__(256) void pack_bwd2d_kernel(const float *__restrict__ w, const float *__restrict__ sigma,
                                                         float *__restrict__ packed, int Cin, int Cout, int kh, int kw,
                                                         int sh, int sw, int patch, int nch, int J, int M, int bf) {
    const int64_t total = packed_weight_floats(nch, J, M);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int j = gj % J, c = (gj / J) * kWG + c16;
    float out = 0.f;
    if (c < nch) {
        int co, ci, dh, dw;
        if (patch) {   // channel = co; tap j = jh * Jw + jw; row m = (ci * sh + a) * sw + cph
            const int Jh = (kh + sh - 1) / sh, Jw = (kw + sw - 1) / sw;
            const int jh = j / Jw, jw = j - jh * Jw;
            const int cph = m % sw, a = (m / sw) % sh;
            ci = m / (sw * sh);
            co = c;
            dh = a + sh * (Jh - 1 - jh);
            dw = cph + sw * (Jw - 1 - jw);
        } else {       // channel = co * kh + dh'; tap j = dw'; row m = ci; flipped kernel
            co = c / kh;
            dh = kh - 1 - (c - co * kh);
            dw = kw - 1 - j;
            ci = m;
        }
        if (dh < kh && dw < kw) out = w[((size_t(co) * Cin + ci) * kh + dh) * kw + dw] * (sigma ? 1.f / sigma[0] : 1.f);
    }
    if (!bf) {
        packed[e] = out;
        return;
    }
    // bf16x3 image (mfma_tile.hpp): three planes per 16-channel group
    const __bf16 h = (__bf16)out;
    const float r1 = out - (float)h;
    const __bf16 mm = (__bf16)r1;
    __bf16 *dst = reinterpret_cast<__bf16 *>(packed) + (size_t(gj) * M + m) * 48 + c16;
    dst[0] = h;
    dst[16] = mm;
    dst[32] = (__bf16)(r1 - (float)mm);
}
// Forward of a layer with very few OUTPUT channels (the 1-channel final conv of the STFT discriminators, 512 -> 1 with a
// (1, k) kernel on a 35 x 16 ... 281 x 2 map): M = Cout rows would leave the MFMA tile empty (0.65-2.3 ms for 40 MMAC).
// One wave per output position: the lanes split the Cin * kh virtual channels of the row-folded image
// (packed[((c'/16) * J + j) * M + m][c' % 16], c' = ci * kh + dh), shuffle-reduce, lane 0 applies bias / LeakyReLU.
template <int MM>
__global__ __launch_bounds__(256) void conv2d_fewout_kernel(ConvPlan p, const float *__restrict__ x,
                                                            const float *__restrict__ wp, const float *__restrict__ bias,
                                                            float *__restrict__ y, int64_t npos) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (pos >= npos) return;   // whole waves leave together
    const int j = int(pos % p.Lout), i = int((pos / p.Lout) % p.Tout), b = int(pos / (int64_t(p.Lout) * p.Tout));
    const float *xb = x + size_t(b) * p.cin_real * p.x_cstride;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    for (int c = lane; c < p.ncv; c += 64) {
        const int ci = c / p.kh, dh = c - ci * p.kh;
        const int r = i * p.sh + dh - p.ph;
        if (r < 0 || r >= p.Tin) continue;
        const float *row = xb + size_t(ci) * p.x_cstride + size_t(r) * p.Lin;
        const float *wc = wp + size_t(c / kWG) * p.J * p.M * kWG + (c % kWG);
        for (int jj = 0; jj < p.J; ++jj) {
            const int col = j * p.s + jj - p.P;
            if (col < 0 || col >= p.Lin) continue;
            const float xv = row[col];
#pragma unroll
            for (int m = 0; m < MM; ++m)
                if (m < p.M) acc[m] = fmaf(wc[(size_t(jj) * p.M + m) * kWG], xv, acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc[m] += __shfl_xor(acc[m], o);
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (m < p.M) {
                float v = acc[m] + (bias ? bias[m] : 0.f);
                if ((p.epilogue & AGX_EPI_LEAKY_PRE) && v < 0.f) v *= p.slope;
                y[(size_t(b) * p.M + m) * p.y_cstride + size_t(i) * p.Lout + j] = v;
            }
    }
}

static bool conv2d_fewout(const agx_conv2d_desc *d, const ConvPlan &p) {
    // (a bf16x3 descriptor of a layer without a bf16x3 form -- prec 0 -- is an AUTO descriptor)
    return (d->impl == AGX_IMPL_AUTO || d->impl == AGX_IMPL_MFMA_BF16X3) && p.pm_R == 0 && p.prec == 0 && p.M <= 4 && p.q == 1 && p.d == 1;
}

}  // namespace agx

extern "C" {

int64_t agx_conv2d_bwd_packed_floats(const agx_conv2d_desc *d) {
    agx::ConvPlan b;
    int rc = agx::lower_conv2d_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    if (b.pm_R < 0) return int64_t(d->c_out) * d->c_in * d->kh * d->kw;
    const int64_t tile = b.tile_off < 0 ? 0 : b.prec ? agx::packed_weight_floats_bf(b.ncv, b.J, b.M)
                                                     : agx::tile_image_floats(b.kh * b.Cin, b.J / b.kh, b.M);
    return (b.prec ? agx::packed_weight_floats_bf(b.ncv, b.J, b.M) : agx::packed_weight_floats(b.ncv, b.J, b.M)) + tile;
}

int agx_conv2d_pack_bwd(const agx_conv2d_desc *d, const float *w, const float *sigma, float *packed, void *stream) {
    using namespace agx;
    ConvPlan b;
    int rc = lower_conv2d_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    if (!w || !packed) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_pack_bwd: NULL pointer");
    if (b.pm_R < 0) {
        const int64_t nraw = int64_t(d->c_out) * d->c_in * d->kh * d->kw;
        hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)ceil_div64(nraw, 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), w, sigma, packed, nraw);
        return check_launch("agx_conv2d_pack_bwd");
    }
    const int64_t n = packed_weight_floats(b.ncv, b.J, b.M);
    hipLaunchKernelGGL(pack_bwd2d_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, sigma, packed, d->c_in, d->c_out, d->kh, d->kw, d->stride_h,
                       d->stride_w, b.pm_R ? 1 : 0, b.ncv, b.J, b.M, b.prec);
    if (b.tile_off >= 0 && b.prec)
        launch_b3_tile_from_bf(packed, packed + b.tile_off, int64_t(ceil_div(b.ncv, kWG)) * b.J, b.M, static_cast<hipStream_t>(stream));
    else if (b.tile_off >= 0)
        launch_pack_tile2d(w, nullptr, sigma, packed + b.tile_off, b.Cin, b.M, b.kh, b.J / b.kh, 1, static_cast<hipStream_t>(stream),
                           d->stride_h, d->stride_w, d->kh, d->kw);
    return check_launch("agx_conv2d_pack_bwd");
}

int agx_conv2d_bwd_data(const agx_conv2d_desc *d, const float *dy, const float *packed_bwd, const float *add,
                        const float *mask, float slope, float *dx, void *stream) {
    using namespace agx;
    ConvPlan b;
    int rc = lower_conv2d_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    if (!dy || !packed_bwd || !dx) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_bwd_data: NULL pointer");
    b.epilogue = (add ? AGX_EPI_RESIDUAL : 0) | (mask ? AGX_EPI_MASK : 0);
    b.mask = mask;
    b.slope = slope;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (b.pm_R < 0) {
        const int64_t total = int64_t(d->batch) * d->c_in * d->h_in * d->w_in;
        hipLaunchKernelGGL(conv2d_bwd_data_gather_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, dy,
                           packed_bwd, add, mask, slope, dx, d->batch, d->c_in, d->c_out, d->h_in, d->w_in, b.Tin, b.Lin,
                           d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w);
        return check_launch("agx_conv2d_bwd_data");
    }
    if (b.prec == 1 && conv2d_b3_supported(b)) return launch_conv2d_b3(b, dy, packed_bwd, nullptr, add, dx, st);
    if (tuning().conv_impl == 1 && conv_p2d_supported(b)) return launch_conv_p2d(b, dy, packed_bwd, nullptr, add, dx, st);
    if (b.pm_R || (d->impl != AGX_IMPL_DIRECT && conv_mfma_supported(b)))
        return launch_conv_mfma(b, dy, packed_bwd, nullptr, add, dx, st);
    return launch_conv_direct(b, dy, packed_bwd, nullptr, add, dx, st);
}

int agx_conv2d_colsplit_weights(const agx_conv2d_desc *d, const float *w, const float *sigma, float *wp, void *stream) {
    using namespace agx;
    if (!d || !w || !wp) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_colsplit_weights: NULL pointer");
    const int64_t total = int64_t(d->c_in) * d->kw * d->c_out * d->kh;
    hipLaunchKernelGGL(colsplit_weights_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, sigma, wp, d->c_in, d->c_out, d->kh, d->kw);
    return check_launch("agx_conv2d_colsplit_weights");
}

int agx_conv2d_colsum(const agx_conv2d_desc *d, const float *pbuf, const float *add, float *dx, void *stream) {
    using namespace agx;
    ConvPlan f;
    int rc = lower_conv2d(d, &f);
    if (rc != AGX_OK) return rc;
    if (d->stride_h != 1 || d->stride_w != 1) return fail(AGX_ERR_UNSUPPORTED, "agx_conv2d_colsum: stride-1 layers only");
    if (!pbuf || !dx) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_colsum: NULL pointer");
    if (d->h_in > 65535 || int64_t(d->batch) * d->c_in > 65535) return fail(AGX_ERR_BAD_SHAPE, "agx_conv2d_colsum: grid too large");
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(d->w_in, 256), d->h_in, d->batch * d->c_in), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pbuf, add, dx, d->c_in, d->kw, d->h_in, f.Lout, d->w_in, d->pad_w);
    return check_launch("agx_conv2d_colsum");
}

int agx_conv2d_out_shape(const agx_conv2d_desc *d, int32_t *h_out, int32_t *w_out) {
    agx::ConvPlan p;
    int rc = agx::lower_conv2d(d, &p);
    if (rc != AGX_OK) return rc;
    if (h_out) *h_out = p.Tout;
    if (w_out) *w_out = p.Lout;
    return AGX_OK;
}

static int conv2d_impl(const agx_conv2d_desc *d, const agx::ConvPlan &p) {
    int impl = d->impl;
    if (impl == AGX_IMPL_MFMA_BF16X3) impl = p.prec ? AGX_IMPL_MFMA : AGX_IMPL_AUTO;   // layers without a bf16x3 form run fp32
    if (impl == AGX_IMPL_AUTO) impl = agx::conv_mfma_supported(p) ? AGX_IMPL_MFMA : AGX_IMPL_DIRECT;
    return impl;
}

int agx_conv2d_forward(const agx_conv2d_desc *d, const float *x, const float *packed, const float *bias, float *y,
                       void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv2d(d, &p);
    if (rc != AGX_OK) return rc;
    if (!x || !packed || !y) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_forward: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (conv2d_fewout(d, p)) {
        const int64_t npos = int64_t(p.B) * p.Tout * p.Lout;
        hipLaunchKernelGGL(conv2d_fewout_kernel<4>, dim3((unsigned)ceil_div64(npos, 4)), dim3(256), 0, st, p, x, packed, bias, y,
                           npos);
        return check_launch("agx_conv2d_forward");
    }
    if (p.prec == 1 && conv2d_b3_supported(p)) return launch_conv2d_b3(p, x, packed, bias, nullptr, y, st);
    if (tuning().conv_impl == 1 && conv_p2d_supported(p)) return launch_conv_p2d(p, x, packed, bias, nullptr, y, st);
    const int impl = conv2d_impl(d, p);
    if (p.pm_R && impl != AGX_IMPL_MFMA)
        return fail(AGX_ERR_UNSUPPORTED, "conv2d: no MFMA tile fits this layer (set impl = AGX_IMPL_DIRECT for pack and forward)");
    if (impl == AGX_IMPL_MFMA) return launch_conv_mfma(p, x, packed, bias, nullptr, y, st);
    if (impl == AGX_IMPL_DIRECT) return launch_conv_direct(p, x, packed, bias, nullptr, y, st);
    return fail(AGX_ERR_BAD_SHAPE, "conv2d: unknown impl %d", impl);
}

int agx_conv2d_kernel_name(const agx_conv2d_desc *d, char *buf, size_t buf_len) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv2d(d, &p);
    if (rc != AGX_OK) return rc;
    if (!buf || buf_len == 0) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_kernel_name: NULL buffer");
    snprintf(buf, buf_len, "%s", (p.prec == 1 && agx::conv2d_b3_supported(p)) ? agx::conv2d_b3_variant(p)
                                 : (agx::tuning().conv_impl == 1 && agx::conv_p2d_supported(p)) ? agx::conv_p2d_variant(p)
                                 : agx::conv2d_fewout(d, p) ? "conv2d_fewout<4>"
                                 : (conv2d_impl(d, p) == AGX_IMPL_MFMA ? conv_mfma_variant(p) : conv_direct_variant(p)));
    return AGX_OK;
}

int agx_conv2d_bwd_data_kernel_name(const agx_conv2d_desc *d, char *buf, size_t buf_len) {
    using namespace agx;
    ConvPlan b;
    int rc = lower_conv2d_bwd_data(d, &b);
    if (rc != AGX_OK) return rc;
    if (!buf || buf_len == 0) return fail(AGX_ERR_NULL_POINTER, "agx_conv2d_bwd_data_kernel_name: NULL buffer");
    snprintf(buf, buf_len, "%s", b.pm_R < 0 ? "conv2d_bwd_data_gather"
                                 : (b.prec == 1 && conv2d_b3_supported(b)) ? conv2d_b3_variant(b)
                                 : (tuning().conv_impl == 1 && conv_p2d_supported(b)) ? conv_p2d_variant(b)
                                 : (b.pm_R || (d->impl != AGX_IMPL_DIRECT && conv_mfma_supported(b))) ? conv_mfma_variant(b)
                                                                                                       : conv_direct_variant(b));
    return AGX_OK;
}

}  // extern "C"
