// libagx core: error state, descriptor lowering, shape queries.
#include <cmath>
#include <cstring>

#include "common.hpp"

namespace agx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

Tuning &tuning() {
    static Tuning t;
    return t;
}

int lower_conv(const agx_conv_desc *d, ConvPlan *p) {
    if (!d) return fail(AGX_ERR_NULL_POINTER, "conv descriptor is NULL");
    if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->l_in <= 0 || d->kernel <= 0 ||
        d->stride <= 0 || d->dilation <= 0)
        return fail(AGX_ERR_BAD_SHAPE,
                    "conv: non-positive dimension (B=%d Cin=%d Cout=%d L=%d K=%d s=%d d=%d)", d->batch,
                    d->c_in, d->c_out, d->l_in, d->kernel, d->stride, d->dilation);
    const int K = d->kernel, s = d->stride, dil = d->dilation, L = d->l_in;
    p->B = d->batch;
    p->Cin = d->c_in;
    p->Cout = d->c_out;
    p->Lin = L;
    p->Lvalid = L;
    p->epilogue = d->epilogue;
    p->slope = d->slope;
    switch (d->kind) {
        case AGX_CONV_CAUSAL: {
            // vae.py:32   pad = dilation*(K-1) - stride + 1
            // vae.py:39-43 extra right pad from the UNdilated kernel size
            const int P = dil * (K - 1) - s + 1;
            const double nxt = double(L - K + P) / double(s) + 1.0;
            const int target = (int(std::ceil(nxt)) - 1) * s + K - P;
            const int E = target - L;
            const int padded = L + P + E;
            const int span = dil * (K - 1) + 1;
            if (padded < span) return fail(AGX_ERR_BAD_SHAPE, "conv: input too short (L=%d)", L);
            p->q = 1;
            p->J = K;
            p->s = s;
            p->d = dil;
            p->P = P;
            p->Lt = (padded - span) / s + 1;
            p->Lout = p->Lt;
            if (E < 0) p->Lvalid = L + E;  // F.pad with a negative pad crops
            break;
        }
        case AGX_CONV_SAME: {
            if (s != 1) return fail(AGX_ERR_BAD_SHAPE, "same-conv needs stride 1");
            p->q = 1;
            p->J = K;
            p->s = 1;
            p->d = dil;
            p->P = (dil * (K - 1)) / 2;  // torch: left = total // 2
            p->Lt = L;
            p->Lout = L;
            break;
        }
        case AGX_CONV_UPSAMPLE: {
            if (dil != 1) return fail(AGX_ERR_BAD_SHAPE, "upsample-conv needs dilation 1");
            // nearest x s, then 'same' conv: polyphase over the low-rate signal
            const int pl = (K - 1) / 2;
            auto fdiv = [](int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); };
            const int jmin = fdiv(-pl, s);
            const int jmax = fdiv(s - 1 + K - 1 - pl, s);
            p->q = s;
            p->J = jmax - jmin + 1;
            p->s = 1;
            p->d = 1;
            p->P = -jmin;
            p->Lt = L;
            p->Lout = L * s;
            break;
        }
        case AGX_CONV_TRANSPOSED: {
            if (dil != 1) return fail(AGX_ERR_BAD_SHAPE, "transposed conv needs dilation 1");
            if (K < s) return fail(AGX_ERR_UNSUPPORTED, "transposed conv with K < stride");
            p->q = s;
            p->J = (K + s - 1) / s;
            p->s = 1;
            p->d = 1;
            p->P = p->J - 1;
            p->Lt = L;
            p->Lout = L * s;  // (L-1)s + K, minus the K - s crop of vae.py:58,63-64
            break;
        }
        case AGX_CONV_PADDED: {
            // torch.nn.Conv1d: Lout = floor((L + 2 pad - d (K - 1) - 1) / s) + 1
            if (d->padding < 0) return fail(AGX_ERR_BAD_SHAPE, "conv: negative padding");
            const int span = dil * (K - 1) + 1;
            if (L + 2 * d->padding < span) return fail(AGX_ERR_BAD_SHAPE, "conv: input too short (L=%d)", L);
            p->q = 1;
            p->J = K;
            p->s = s;
            p->d = dil;
            p->P = d->padding;
            p->Lt = (L + 2 * d->padding - span) / s + 1;
            p->Lout = p->Lt;
            break;
        }
        default:
            return fail(AGX_ERR_BAD_SHAPE, "conv: unknown kind %d", d->kind);
    }
    p->G = 1;
    if (d->groups > 1) {
        if (d->kind != AGX_CONV_PADDED) return fail(AGX_ERR_UNSUPPORTED, "conv: groups only with AGX_CONV_PADDED");
        if (d->c_in % d->groups || d->c_out % d->groups)
            return fail(AGX_ERR_BAD_SHAPE, "conv: groups=%d does not divide Cin=%d / Cout=%d", d->groups, d->c_in, d->c_out);
        p->G = d->groups;
    }
    p->M = p->q * p->Cout;
    p->oshift = 0;
    p->mask = nullptr;
    p->kh = 1;
    p->sh = 1;
    p->ph = 0;
    p->Tin = p->Tout = 1;
    p->ncv = p->cin_real = p->Cin;
    p->pm_R = p->pm_WF = 0;
    p->qh = 1;
    p->oshift_h = 0;
    p->Tt = 1;
    p->prec = d->impl == AGX_IMPL_MFMA_BF16X3 ? 1 : 0;
    p->x_cstride = p->Lin;
    p->y_cstride = p->Lout;
    // forward packed image = [standard image][dim0 scale scratch][tile image, for the layers that have one]
    p->tile_off = -1;
    if (tile_image_eligible(*p, d->kind))
        p->tile_off = packed_weight_floats(p->Cin / p->G, p->J, p->M) + (d->kind == AGX_CONV_TRANSPOSED ? d->c_in : d->c_out);
    else if (b3_image_eligible(*p, d->kind))
        p->tile_off = packed_weight_floats_bf(p->Cin, p->J, p->M) + (d->kind == AGX_CONV_TRANSPOSED ? d->c_in : d->c_out);
    return AGX_OK;
}

// d(out)/d(in) of the forward plan  y[co, q t + p] = sum Wp[ci,j][co,p] x[ci, t s + j d - P]:
//   * s == 1 (stride-1 causal / same convs): correlation with the flipped kernel,
//       dx[ci, i] = sum_{co,j'} W[co,ci,J-1-j'] dy[co, i + j' d - ((J-1) d - P)]
//   * s > 1, q == 1 (strided down conv, d == 1): i + P = s t' + p'  =>  dx[s t' + p' - P] =
//       sum_m W[., p' + s m] dy[t' - m]   -- the transposed-conv phase form with an output shift of P
//   * q > 1 (polyphase upsample / transposed forward, s == d == 1): a strided conv over dy,
//       dx[ci, i] = sum_{co,k} Wp[p = k % q][ci][j = J-1 - k / q][co] dy[co, q i + k - q (J-1-P)]
int lower_conv_bwd_data(const agx_conv_desc *d, ConvPlan *b) {
    ConvPlan f;
    int rc = lower_conv(d, &f);
    if (rc != AGX_OK) return rc;
    if (f.Lvalid != f.Lin) return fail(AGX_ERR_UNSUPPORTED, "bwd_data: cropped inputs (negative right pad) not supported");
    if (f.G != 1) return fail(AGX_ERR_UNSUPPORTED, "bwd_data: grouped convs have no backward kernel");
    *b = f;
    b->tile_off = -1;   // backward images (agx_conv_pack_bwd) carry no tile image
    b->Cin = f.Cout;
    b->Cout = f.Cin;
    b->Lin = f.Lout;   // the op reads dy
    b->Lvalid = f.Lout;
    b->Lout = f.Lin;   // and writes dx
    b->epilogue = 0;
    b->mask = nullptr;
    b->oshift = 0;
    if (f.q == 1 && f.s == 1) {
        b->q = 1;
        b->J = f.J;
        b->s = 1;
        b->d = f.d;
        b->P = (f.J - 1) * f.d - f.P;
        b->Lt = f.Lin;
    } else if (f.q == 1) {
        if (f.d != 1) return fail(AGX_ERR_UNSUPPORTED, "bwd_data: strided conv with dilation > 1");
        b->q = f.s;
        b->J = (f.J + f.s - 1) / f.s;
        b->s = 1;
        b->d = 1;
        b->P = b->J - 1;
        b->oshift = f.P;
        b->Lt = (f.Lin + f.P + f.s - 1) / f.s;  // t' up to the last i + P
        if (f.P < 0) return fail(AGX_ERR_UNSUPPORTED, "bwd_data: negative left pad");
    } else {
        b->q = 1;
        b->J = f.q * f.J;
        b->s = f.q;
        b->d = 1;
        b->P = f.q * (f.J - 1 - f.P);
        b->Lt = f.Lin;
    }
    b->M = b->q * b->Cout;
    b->x_cstride = b->Lin;
    b->y_cstride = b->Lout;
    b->ncv = b->cin_real = b->Cin;
    b->pm_R = b->pm_WF = 0;
    b->qh = 1;
    b->oshift_h = 0;
    b->Tt = 1;
    b->prec = 0;
    return AGX_OK;
}

}  // namespace agx

namespace agx {
int device_cu_count(int dev, int *n_cu) {
    static std::atomic<int> cache[64];
    int v = dev >= 0 && dev < 64 ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (v <= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(AGX_ERR_LAUNCH, "cannot query device %d", dev);
        v = prop.multiProcessorCount;
        if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
    }
    *n_cu = v;
    return AGX_OK;
}
}  // namespace agx

extern "C" {

int agx_version(void) { return AGX_VERSION; }
int32_t agx_sizeof_conv_desc(void) { return (int32_t)sizeof(agx_conv_desc); }
int32_t agx_sizeof_conv2d_desc(void) { return (int32_t)sizeof(agx_conv2d_desc); }

const char *agx_last_error(void) { return agx::last_error(); }

int agx_set_tuning(const char *name, int32_t value) {
    if (!name) return agx::fail(AGX_ERR_NULL_POINTER, "agx_set_tuning: NULL name");
    if (!strcmp(name, "rb_cc")) agx::tuning().rb_cc = value;
    else if (!strcmp(name, "rb_wgs")) agx::tuning().rb_wgs = value;
    else if (!strcmp(name, "rb_sched")) agx::tuning().rb_sched = value;
    else if (!strcmp(name, "rb_occ")) agx::tuning().rb_occ = value;
    else if (!strcmp(name, "rb_impl")) agx::tuning().rb_impl = value;
    else if (!strcmp(name, "b3_dbg")) {
#ifndef AGX_RVQ_PROBE
        if (value >= 7 && value <= 9)   // they weaken / repurpose the RVQ candidate bound: never in the product library
            return agx::fail(AGX_ERR_UNSUPPORTED, "agx_set_tuning: b3_dbg = %d exists in the probe build only (tools/rvq_stamps.py build)", value);
#endif
        agx::tuning().b3_dbg = value;
    } else if (!strcmp(name, "rvq_verify")) agx::tuning().rvq_verify = value;
    else if (!strcmp(name, "conv_impl")) agx::tuning().conv_impl = value;
    else if (!strcmp(name, "bf_sched")) agx::tuning().bf_sched = value;
    else if (!strcmp(name, "patch_tie")) agx::tuning().patch_tie = value;
    else if (!strcmp(name, "dw2_direct")) agx::tuning().dw2_direct = value;
    else if (!strcmp(name, "dw2_shared")) agx::tuning().dw2_shared = value;
    else if (!strcmp(name, "dw_wgs")) agx::tuning().dw_wgs = value;
    else if (!strcmp(name, "dw_xcd")) agx::tuning().dw_xcd = value;
    else if (!strcmp(name, "dw2_bf")) agx::tuning().dw2_bf = value;
    else if (!strcmp(name, "dw2_prepad")) agx::tuning().dw2_prepad = value;
    else if (!strcmp(name, "c2b3_sl")) agx::tuning().c2b3_sl = value;
    else if (!strcmp(name, "dw_direct")) agx::tuning().dw_direct = value;
    else if (!strcmp(name, "dw1_wgs")) agx::tuning().dw1_wgs = value;
    else if (!strcmp(name, "conv_cc")) agx::tuning().conv_cc = value;
    else if (!strcmp(name, "conv_shape")) agx::tuning().conv_shape = value;
    else if (!strcmp(name, "conv_short")) agx::tuning().conv_short = value;
    else return agx::fail(AGX_ERR_BAD_SHAPE, "agx_set_tuning: unknown knob '%s'", name);
    return AGX_OK;
}

int agx_get_tuning(const char *name) {
    if (!name) return agx::fail(AGX_ERR_NULL_POINTER, "agx_get_tuning: NULL name");
    if (!strcmp(name, "rb_cc")) return agx::tuning().rb_cc;
    if (!strcmp(name, "rb_wgs")) return agx::tuning().rb_wgs;
    if (!strcmp(name, "rb_sched")) return agx::tuning().rb_sched;
    if (!strcmp(name, "rb_occ")) return agx::tuning().rb_occ;
    if (!strcmp(name, "rb_impl")) return agx::tuning().rb_impl;
    if (!strcmp(name, "b3_dbg")) return agx::tuning().b3_dbg;
    if (!strcmp(name, "rvq_verify")) return agx::tuning().rvq_verify;
    if (!strcmp(name, "conv_impl")) return agx::tuning().conv_impl;
    if (!strcmp(name, "bf_sched")) return agx::tuning().bf_sched;
    if (!strcmp(name, "patch_tie")) return agx::tuning().patch_tie;
    if (!strcmp(name, "dw2_direct")) return agx::tuning().dw2_direct;
    if (!strcmp(name, "dw2_shared")) return agx::tuning().dw2_shared;
    if (!strcmp(name, "dw_wgs")) return agx::tuning().dw_wgs;
    if (!strcmp(name, "dw_xcd")) return agx::tuning().dw_xcd;
    if (!strcmp(name, "dw2_bf")) return agx::tuning().dw2_bf;
    if (!strcmp(name, "dw2_prepad")) return agx::tuning().dw2_prepad;
    if (!strcmp(name, "c2b3_sl")) return agx::tuning().c2b3_sl;
    if (!strcmp(name, "dw_direct")) return agx::tuning().dw_direct;
    if (!strcmp(name, "dw1_wgs")) return agx::tuning().dw1_wgs;
    if (!strcmp(name, "conv_cc")) return agx::tuning().conv_cc;
    if (!strcmp(name, "conv_shape")) return agx::tuning().conv_shape;
    if (!strcmp(name, "conv_short")) return agx::tuning().conv_short;
    return agx::fail(AGX_ERR_BAD_SHAPE, "agx_get_tuning: unknown knob '%s'", name);
}

int64_t agx_conv_out_len(const agx_conv_desc *d) {
    agx::ConvPlan p;
    int rc = agx::lower_conv(d, &p);
    return rc == AGX_OK ? int64_t(p.Lout) : int64_t(rc);
}

int64_t agx_conv_packed_floats(const agx_conv_desc *d) {
    agx::ConvPlan p;
    int rc = agx::lower_conv(d, &p);
    if (rc != AGX_OK) return rc;
    // + dim0 floats of scratch at the tail for the weight-norm scales
    const int dim0 = (d->kind == AGX_CONV_TRANSPOSED) ? d->c_in : d->c_out;
    if (p.prec) return agx::packed_weight_floats_bf(p.Cin, p.J, p.M) * (p.tile_off >= 0 ? 2 : 1) + dim0;
    const int64_t tile = p.tile_off >= 0 ? agx::tile_image_floats(p.Cin, p.J, p.M) : 0;
    return agx::packed_weight_floats(p.Cin / p.G, p.J, p.M) + dim0 + tile;
}

}  // extern "C"
