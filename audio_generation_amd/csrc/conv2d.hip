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
    const bool patch = d->impl != AGX_IMPL_DIRECT && d->c_in % kWG == 0 && d->c_out >= 32;
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
    if (int64_t(p->B) * p->Tout > (int64_t(1) << 30)) return fail(AGX_ERR_BAD_SHAPE, "conv2d: batch * output rows too large");
    return AGX_OK;
}
}  // namespace agx

extern "C" {

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
    snprintf(buf, buf_len, "%s", conv2d_impl(d, p) == AGX_IMPL_MFMA ? conv_mfma_variant(p) : conv_direct_variant(p));
    return AGX_OK;
}

}  // extern "C"
