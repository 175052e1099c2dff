// C-ABI entry points of the convolution family: shape checks + kernel dispatch.
#include "common.hpp"

namespace agx {
int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                       const float *res, float *y, hipStream_t st);
int launch_conv_mfma(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                     const float *res, float *y, hipStream_t st);
bool conv_mfma_supported(const ConvPlan &p);
const char *conv_mfma_variant(const ConvPlan &p);
const char *conv_direct_variant(const ConvPlan &p);
int launch_resblock_fused(const ConvPlan &p, const float *x, const float *w1, const float *b1,
                          const float *w2, const float *b2, float *y, int post_act, hipStream_t st);
bool resblock_fused_supported(const ConvPlan &p);
const char *resblock_variant(const ConvPlan &p);
int launch_resblock_p(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                      const float *b2, float *y, int post_act, hipStream_t st);
bool resblock_p_supported(const ConvPlan &p);
const char *resblock_p_variant(const ConvPlan &p);
int launch_resblock_b3(const ConvPlan &p, const float *x, const float *w1, const float *b1, const float *w2,
                       const float *b2, float *y, int post_act, hipStream_t st);
bool resblock_b3_supported(const ConvPlan &p);
const char *resblock_b3_variant(const ConvPlan &p);
int launch_conv_b3(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *y, hipStream_t st);
bool conv_b3_supported(const ConvPlan &p);
int launch_conv_b3_planes(const ConvPlan &p, const void *x_planes, const float *wp, const float *bias, float *y, void *y_planes,
                          hipStream_t st);
int launch_planes_split(const float *x, void *planes, int batch, int channels, int length, hipStream_t st);
const char *conv_b3_variant(const ConvPlan &p);
int launch_conv_p(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                  hipStream_t st);
bool conv_p_supported(const ConvPlan &p);
const char *conv_p_variant(const ConvPlan &p);

// dx *= gelu'(pre)  (exact erf GELU)
__global__ __launch_bounds__(256) void gelu_grad_mul_kernel(float *__restrict__ dx, const float *__restrict__ pre, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    dx[i] *= 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

static int run_conv(const ConvPlan &p, int impl, const float *x, const float *wp, const float *bias,
                    const float *res, float *y, hipStream_t st) {
    if ((impl == AGX_IMPL_AUTO || impl == AGX_IMPL_MFMA) && tuning().conv_impl == 1 && conv_p_supported(p))
        return launch_conv_p(p, x, wp, bias, res, y, st);
    if (impl == AGX_IMPL_MFMA_BF16X3 && tuning().conv_impl == 1 && conv_b3_supported(p)) return launch_conv_b3(p, x, wp, bias, y, st);
    if (impl == AGX_IMPL_AUTO) impl = conv_mfma_supported(p) ? AGX_IMPL_MFMA : AGX_IMPL_DIRECT;
    if (impl == AGX_IMPL_MFMA || impl == AGX_IMPL_MFMA_BF16X3) return launch_conv_mfma(p, x, wp, bias, res, y, st);
    if (impl == AGX_IMPL_DIRECT) return launch_conv_direct(p, x, wp, bias, res, y, st);
    return fail(AGX_ERR_BAD_SHAPE, "conv: unknown impl %d", impl);
}
}  // namespace agx

extern "C" {

int agx_conv_forward(const agx_conv_desc *d, const float *x, const float *packed, const float *bias,
                     const float *res, float *y, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!x || !packed || !y) return fail(AGX_ERR_NULL_POINTER, "agx_conv_forward: NULL pointer");
    if ((p.epilogue & AGX_EPI_RESIDUAL) && !res)
        return fail(AGX_ERR_NULL_POINTER, "agx_conv_forward: residual epilogue without res");
    return run_conv(p, d->impl, x, packed, bias, res, y, static_cast<hipStream_t>(stream));
}

size_t agx_planes_bytes(int32_t batch, int32_t channels, int32_t length) {
    if (batch <= 0 || channels <= 0 || length <= 0 || channels % 8 != 0) return 0;
    return size_t(batch) * (channels / 8) * 3 * size_t(length) * 16;
}

int agx_planes_split(const float *x, void *planes, int32_t batch, int32_t channels, int32_t length, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || length <= 0 || channels % 8 != 0)
        return fail(AGX_ERR_BAD_SHAPE, "agx_planes_split: bad shape B=%d C=%d L=%d (C must be a multiple of 8)", batch, channels, length);
    if (!x || !planes) return fail(AGX_ERR_NULL_POINTER, "agx_planes_split: NULL pointer");
    return launch_planes_split(x, planes, batch, channels, length, static_cast<hipStream_t>(stream));
}

int agx_conv_forward_planes(const agx_conv_desc *d, const void *x_planes, const float *packed, const float *bias, float *y,
                            void *y_planes, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!x_planes || !packed || (!y && !y_planes)) return fail(AGX_ERR_NULL_POINTER, "agx_conv_forward_planes: NULL pointer");
    if (d->impl != AGX_IMPL_MFMA_BF16X3 || tuning().conv_impl != 1 || !conv_b3_supported(p) || p.s != 1 || (p.q == 1 && p.J != 7))
        return fail(AGX_ERR_UNSUPPORTED, "agx_conv_forward_planes: the layer has no plane-fed bf16x3 ring form (ask agx_conv_planes_supported first)");
    return launch_conv_b3_planes(p, x_planes, packed, bias, y, y_planes, static_cast<hipStream_t>(stream));
}

int agx_conv_planes_supported(const agx_conv_desc *d) {
    using namespace agx;
    ConvPlan p;
    if (lower_conv(d, &p) != AGX_OK) return 0;
    if (d->impl != AGX_IMPL_MFMA_BF16X3 || tuning().conv_impl != 1 || !conv_b3_supported(p)) return 0;
    if (p.s != 1 || (p.q == 1 && p.J != 7)) return 0;   // the strided down-convs and the causal k = 3 layer take fp32 input only
    return (p.q == 1 && p.Cout % 8 == 0) ? 2 : 1;      // 2: the layer can also WRITE planes (one output phase)
}

int agx_conv_bwd_data(const agx_conv_desc *d, const float *dy, const float *packed_bwd, const float *add,
                      const float *mask, float slope, float *dx, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv_bwd_data(d, &p);
    if (rc != AGX_OK) return rc;
    if (!dy || !packed_bwd || !dx) return fail(AGX_ERR_NULL_POINTER, "agx_conv_bwd_data: NULL pointer");
    p.epilogue = (add ? AGX_EPI_RESIDUAL : 0) | (mask ? AGX_EPI_MASK : 0);
    p.mask = mask;
    p.slope = slope;
    return run_conv(p, d->impl, dy, packed_bwd, nullptr, add, dx, static_cast<hipStream_t>(stream));
}

int agx_conv_bwd_data_gelu(const agx_conv_desc *d, const float *dy, const float *packed_bwd, const float *add,
                           const float *pre, float *dx, void *stream) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv_bwd_data(d, &p);
    if (rc != AGX_OK) return rc;
    if (!dy || !packed_bwd || !dx || !pre) return fail(AGX_ERR_NULL_POINTER, "agx_conv_bwd_data_gelu: NULL pointer");
    p.epilogue = add ? AGX_EPI_RESIDUAL : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = run_conv(p, d->impl, dy, packed_bwd, nullptr, add, dx, st);
    if (rc != AGX_OK) return rc;
    // the GELU gradient as a second (elementwise) launch: erf + exp in the conv epilogue cost the MFMA kernels
    // their register budget (spills in every instantiation), and this op only exists on the 225-frame bottleneck
    const int64_t n = int64_t(p.B) * p.Cout * p.Lout;
    hipLaunchKernelGGL(gelu_grad_mul_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, dx, pre, n);
    return check_launch("agx_conv_bwd_data_gelu");
}

int agx_conv_kernel_name(const agx_conv_desc *d, char *buf, size_t buf_len) {
    using namespace agx;
    ConvPlan p;
    int rc = lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    if (!buf || buf_len == 0) return fail(AGX_ERR_NULL_POINTER, "agx_conv_kernel_name: NULL buffer");
    int impl = d->impl;
    if (impl == AGX_IMPL_MFMA_BF16X3 && tuning().conv_impl == 1 && conv_b3_supported(p)) {
        snprintf(buf, buf_len, "%s:bf16x3", conv_b3_variant(p));
        return AGX_OK;
    }
    if ((impl == AGX_IMPL_AUTO || impl == AGX_IMPL_MFMA) && tuning().conv_impl == 1 && conv_p_supported(p)) {
        snprintf(buf, buf_len, "%s", conv_p_variant(p));
        return AGX_OK;
    }
    if (impl == AGX_IMPL_AUTO) impl = conv_mfma_supported(p) ? AGX_IMPL_MFMA : AGX_IMPL_DIRECT;
    if (impl == AGX_IMPL_MFMA_BF16X3) snprintf(buf, buf_len, "%s:bf16x3", conv_mfma_variant(p));
    else snprintf(buf, buf_len, "%s", impl == AGX_IMPL_MFMA ? conv_mfma_variant(p) : conv_direct_variant(p));
    return AGX_OK;
}

size_t agx_resblock_workspace_bytes(const agx_conv_desc *d) {
    if (!d || d->batch <= 0 || d->c_out <= 0 || d->l_in <= 0) return 0;
    return size_t(d->batch) * d->c_out * d->l_in * sizeof(float);
}

int agx_resblock_kernel_name(const agx_conv_desc *d, char *buf, size_t buf_len) {
    using namespace agx;
    if (!d || !buf || buf_len == 0) return fail(AGX_ERR_NULL_POINTER, "agx_resblock_kernel_name: NULL pointer");
    agx_conv_desc d1 = *d;
    d1.epilogue = AGX_EPI_LEAKY_PRE;
    ConvPlan p;
    int rc = lower_conv(&d1, &p);
    if (rc != AGX_OK) return rc;
    if (d->impl != AGX_IMPL_DIRECT && tuning().rb_impl == 1 && resblock_p_supported(p)) {
        snprintf(buf, buf_len, "%s", resblock_p_variant(p));
    } else if (tuning().rb_impl == 1 && resblock_b3_supported(p)) {
        snprintf(buf, buf_len, "%s:bf16x3", resblock_b3_variant(p));
    } else if (d->impl != AGX_IMPL_DIRECT && resblock_fused_supported(p)) {
        snprintf(buf, buf_len, "%s%s", resblock_variant(p), p.prec ? ":bf16x3" : "");
    } else {
        int impl = d->impl;
        if (impl == AGX_IMPL_AUTO) impl = conv_mfma_supported(p) ? AGX_IMPL_MFMA : AGX_IMPL_DIRECT;
        snprintf(buf, buf_len, "2x:%s%s", impl == AGX_IMPL_DIRECT ? conv_direct_variant(p) : conv_mfma_variant(p),
                 p.prec ? ":bf16x3" : "");
    }
    return AGX_OK;
}

int agx_resblock_forward(const agx_conv_desc *d, const float *x, const float *packed1,
                         const float *bias1, const float *packed2, const float *bias2, float *y,
                         int32_t post_act, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace agx;
    if (!d) return fail(AGX_ERR_NULL_POINTER, "agx_resblock_forward: NULL descriptor");
    if (d->kind != AGX_CONV_CAUSAL || d->stride != 1 || d->c_in != d->c_out)
        return fail(AGX_ERR_BAD_SHAPE, "resblock: needs a stride-1 causal conv with c_in == c_out");
    if (!x || !packed1 || !packed2 || !y) return fail(AGX_ERR_NULL_POINTER, "agx_resblock_forward: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    agx_conv_desc d1 = *d;
    d1.epilogue = AGX_EPI_LEAKY_PRE;
    ConvPlan p1;
    int rc = lower_conv(&d1, &p1);
    if (rc != AGX_OK) return rc;
    if (d->impl != AGX_IMPL_DIRECT && tuning().rb_impl == 1 && resblock_p_supported(p1))
        return launch_resblock_p(p1, x, packed1, bias1, packed2, bias2, y, post_act, st);
    if (tuning().rb_impl == 1 && resblock_b3_supported(p1))
        return launch_resblock_b3(p1, x, packed1, bias1, packed2, bias2, y, post_act, st);
    if (d->impl != AGX_IMPL_DIRECT && resblock_fused_supported(p1))
        return launch_resblock_fused(p1, x, packed1, bias1, packed2, bias2, y, post_act, st);
    // two launches: h = leaky(conv1(x)+b1) -> workspace;  y = [leaky](x + conv2(h) + b2)
    if (!workspace || workspace_bytes < agx_resblock_workspace_bytes(d))
        return fail(AGX_ERR_WORKSPACE, "resblock: workspace too small (%zu < %zu)", workspace_bytes,
                    agx_resblock_workspace_bytes(d));
    float *h = static_cast<float *>(workspace);
    rc = run_conv(p1, d->impl, x, packed1, bias1, nullptr, h, st);
    if (rc != AGX_OK) return rc;
    agx_conv_desc d2 = *d;
    d2.kernel = 1;
    d2.dilation = 1;
    d2.epilogue = AGX_EPI_RESIDUAL | (post_act ? AGX_EPI_LEAKY_POST : 0);
    ConvPlan p2;
    rc = lower_conv(&d2, &p2);
    if (rc != AGX_OK) return rc;
    return run_conv(p2, d->impl, h, packed2, bias2, x, y, st);
}

}  // extern "C"
