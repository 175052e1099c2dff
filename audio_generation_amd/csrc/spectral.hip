// Training-loop signal ops (torchaudio in the reference, training.py:151-156, 316-318, 333-334):
// pre-emphasis, low-pass biquad, and the mel spectrogram of the multi-resolution spectral loss.
//
// Framed DFT ("fdft"): a windowed, optionally one-sided STFT with hop H and n_fft = K H as a K-tap,
// stride-1 conv over H phase channels (disc.hip: the same formulation as the discriminator STFT) on the
// MFMA conv kernel; the weight image holds window[n] * twiddle(f, n) * norm.  Output stays channel-major
// (B, 2F, T): real rows then imaginary rows -- what the mel / power kernel below consumes.
// A window shorter than n_fft (the reference's short windows sit in n_fft = 512: training.py:51-78) is zero outside
// [left, left + W): only the taps j0 .. j0 + Ke - 1 that meet it are kept (Ke = 4 instead of up to 64 for hop = W / 4);
// the dropped products are exact zeros, so the sums do not change.
#include "common.hpp"

namespace agx {

int launch_conv_mfma(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res, float *y,
                     hipStream_t st);
int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias, const float *res,
                       float *y, hipStream_t st);
bool conv_mfma_supported(const ConvPlan &p);
// disc.hip: reflect-pad + polyphase transpose of `batch` rows (and its adjoint); ch_stride = channels allocated per item
void launch_stft_prep(const float *x, float *xc, int batch, int L, int N, int H, int Ttau, int ch_stride, int tau_off,
                      hipStream_t st);
void launch_stft_unprep(const float *dxc, float *dx, int batch, int L, int N, int H, int Ttau, int ch_stride, int tau_off,
                        hipStream_t st);

struct FdftGeom {
    int N, W, H, K, Hc, F, Fp, M;   // n_fft, win_length, hop, taps, channels (H rounded up to 16), bins, bins rounded up to 8, rows = 2 Fp
    int j0, Ke;                     // first tap that meets the window, taps kept
};

static int fdft_geom(int n_fft, int win_length, int hop, int onesided, FdftGeom *g) {
    if (n_fft < 16 || hop <= 0 || n_fft % hop || win_length <= 0 || win_length > n_fft)
        return fail(AGX_ERR_BAD_SHAPE, "fdft: need hop | n_fft and win_length <= n_fft");
    g->N = n_fft; g->W = win_length; g->H = hop; g->K = n_fft / hop;
    const int left = (n_fft - win_length) / 2;
    g->j0 = left / hop;
    g->Ke = ceil_div(left + win_length, hop) - g->j0;
    g->Hc = ceil_div(hop, kWG) * kWG;
    g->F = onesided ? n_fft / 2 + 1 : n_fft;
    g->Fp = ceil_div(g->F, 8) * 8;   // 2 Fp rows: a multiple of 16, so the backward plan's channels fill MFMA chunks
    g->M = 2 * g->Fp;
    return AGX_OK;
}

__device__ __forceinline__ double fdft_window(int n, int N, int W, int kind) {
    const int left = (N - W) / 2, k = n - left;      // torch.stft centres a short window in the frame
    if (k < 0 || k >= W) return 0.0;
    if (kind == 0) return 1.0;
    double sn, cs;
    sincospi(2.0 * double(k) / double(W), &sn, &cs);  // periodic Hann
    return 0.5 - 0.5 * cs;
}

// forward image: channel p (phase), tap j, row m = c * F + f   <-  w[n] * (cos | -sin)(2 pi f n / N) * scale, n = j H + p
// backward image (the conv's backward-data plan): channel m, tap jb <-> forward tap Ke-1-jb, row p   (taps counted from j0)
__global__ __launch_bounds__(256) void fdft_pack_kernel(float *__restrict__ packed, FdftGeom g, int window_kind,
                                                        float scale, int backward) {
    const int nch = backward ? g.M : g.Hc, nrow = backward ? g.Hc : g.M;
    const int64_t total = packed_weight_floats(nch, g.Ke, nrow);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int row = int((e / kWG) % nrow);
    const int gj = int(e / (int64_t(kWG) * nrow));
    const int jt = gj % g.Ke, ch = (gj / g.Ke) * kWG + c16;
    const int p = backward ? row : ch, m = backward ? ch : row, j = g.j0 + (backward ? g.Ke - 1 - jt : jt);
    float out = 0.f;
    const int c = m / g.Fp, f = m - c * g.Fp;
    if (p < g.H && m < g.M && f < g.F) {
        const int n = j * g.H + p;
        const long long k = (long long)f * n % g.N;
        double sn, cs;
        sincospi(2.0 * double(k) / double(g.N), &sn, &cs);
        out = float((c == 0 ? cs : -sn) * fdft_window(n, g.N, g.W, window_kind) * double(scale));
    }
    packed[e] = out;
}

// mel[b, m, t] = sum_f fb[f, m] (re[b,f,t]^2 + im[b,f,t]^2)      (MelScale(Spectrogram(power=2)))
// One thread = one (b, t) column x 8 mels; columns are contiguous across lanes.  A mel filter bank is banded (a bin
// feeds two triangles): the workgroup first finds the bins [lo, hi) where any of its 8 filters is non-zero and sums
// over those only -- the skipped products are exact zeros, the spectrum is read ~1.2 x instead of n_mels / 8 x.
__global__ __launch_bounds__(256) void melpower_kernel(const float *__restrict__ cv, const float *__restrict__ fb,
                                                       float *__restrict__ mel, int F, int Fp, int T, int n_mels) {
    __shared__ int band[2];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 8, b = blockIdx.z;
    if (threadIdx.x == 0) { band[0] = F; band[1] = 0; }
    __syncthreads();
    for (int e = threadIdx.x; e < F * 8; e += 256) {
        const int f = e >> 3, m = m0 + (e & 7);
        if (m < n_mels && fb[size_t(f) * n_mels + m] != 0.f) { atomicMin(&band[0], f); atomicMax(&band[1], f + 1); }
    }
    __syncthreads();
    if (t >= T) return;
    const int f_lo = band[0], f_hi = band[1];
    const float *re = cv + size_t(b) * 2 * Fp * T + t, *im = re + size_t(Fp) * T;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int f = f_lo; f < f_hi; ++f) {
        const float r = re[size_t(f) * T], i = im[size_t(f) * T];
        const float pw = fmaf(r, r, i * i);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fmaf(fb[size_t(f) * n_mels + min(m0 + u, n_mels - 1)], pw, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (m0 + u < n_mels) mel[(size_t(b) * n_mels + m0 + u) * T + t] = acc[u];
}

// dcv[b, f, t] = 2 cv[b, f, t] sum_m fb[f, m] dmel[b, m, t]   (both halves)
__global__ __launch_bounds__(256) void melpower_bwd_kernel(const float *__restrict__ cv, const float *__restrict__ fb,
                                                           const float *__restrict__ dmel, float *__restrict__ dcv,
                                                           int F, int Fp, int T, int n_mels) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;   // f < Fp: the padding rows get zeros
    if (t >= T) return;
    float g = 0.f;
    if (f < F)
        for (int m = 0; m < n_mels; ++m) {   // the filter bank is banded: the weight is uniform over the workgroup, zeros are skipped
            const float w = fb[size_t(f) * n_mels + m];
            if (w != 0.f) g = fmaf(w, dmel[(size_t(b) * n_mels + m) * T + t], g);
        }
    const size_t e = (size_t(b) * 2 * Fp + f) * T + t;
    dcv[e] = 2.f * cv[e] * g;
    dcv[e + size_t(Fp) * T] = 2.f * cv[e + size_t(Fp) * T] * g;
}

// torchaudio.functional.preemphasis: y[n] = x[n] - c x[n-1]; the adjoint: dx[n] = dy[n] - c dy[n+1]
__global__ __launch_bounds__(256) void preemphasis_kernel(const float *__restrict__ x, float *__restrict__ y, int L,
                                                          float coeff, int adjoint) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const float *row = x + size_t(blockIdx.y) * L;
    const int o = adjoint ? n + 1 : n - 1;
    y[size_t(blockIdx.y) * L + n] = row[n] - ((o >= 0 && o < L) ? coeff * row[o] : 0.f);
}

// torchaudio.functional.lowpass_biquad -> lfilter(clamp=True): direct form I.  The recurrence is linear, so a row is
// cut into 64 chunks (one wave per row, one chunk per lane): pass 1 runs every chunk from a zero state and, beside
// it, the two homogeneous responses (unit y[-1], unit y[-2]); a 64-step carry over the lanes turns those into each
// chunk's true entry state; pass 2 reruns the chunk from that state and writes the clamped output.  Sequential
// depth 2 L / 64 instead of L (32 x 72 000 samples: 12.8 ms as one thread per row).
__global__ __launch_bounds__(64) void biquad_kernel(const float *__restrict__ x, float *__restrict__ y, int L, float b0,
                                                    float b1, float b2, float a1, float a2) {
    const int lane = threadIdx.x;
    const float *xr = x + size_t(blockIdx.x) * L;
    float *yr = y + size_t(blockIdx.x) * L;
    const int Lc = (L + 63) / 64, n0 = min(L, lane * Lc), n1 = min(L, n0 + Lc);
    const float xm1 = n0 >= 1 ? xr[n0 - 1] : 0.f, xm2 = n0 >= 2 ? xr[n0 - 2] : 0.f;
    float x1 = xm1, x2 = xm2, y1 = 0.f, y2 = 0.f, p1 = 1.f, p2 = 0.f, q1 = 0.f, q2 = 1.f;
    for (int n = n0; n < n1; ++n) {
        const float xn = xr[n];
        const float yn = b0 * xn + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        const float pn = -a1 * p1 - a2 * p2, qn = -a1 * q1 - a2 * q2;
        x2 = x1; x1 = xn; y2 = y1; y1 = yn;
        p2 = p1; p1 = pn; q2 = q1; q1 = qn;
    }
    // entry state of every chunk: s(c+1) = zero-state end of chunk c + M s(c)
    float s1 = 0.f, s2 = 0.f, in1 = 0.f, in2 = 0.f;
    for (int c = 0; c < 64; ++c) {
        if (lane == c) { in1 = s1; in2 = s2; }
        const float z1 = __shfl(y1, c), z2 = __shfl(y2, c);
        const float m11 = __shfl(p1, c), m12 = __shfl(q1, c), m21 = __shfl(p2, c), m22 = __shfl(q2, c);
        const float t1 = z1 + m11 * s1 + m12 * s2, t2 = z2 + m21 * s1 + m22 * s2;
        s1 = t1; s2 = t2;
    }
    x1 = xm1; x2 = xm2; y1 = in1; y2 = in2;
    for (int n = n0; n < n1; ++n) {
        const float xn = xr[n];
        const float yn = b0 * xn + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        yr[n] = fminf(fmaxf(yn, -1.f), 1.f);
        x2 = x1; x1 = xn; y2 = y1; y1 = yn;
    }
}

static agx_conv_desc fdft_conv_desc(const FdftGeom &g, int batch, int Ttau) {
    return agx_conv_desc{AGX_CONV_PADDED, batch, g.Hc, g.M, Ttau, g.Ke, 1, 1, 0, 0.f, AGX_IMPL_AUTO, 1, 0};
}

}  // namespace agx

// torchaudio.transforms.Resample (training.py:554; applied per clip by utils.collator, utils.py:157-158): polyphase
// sinc interpolation.  y[n * nf + p] = sum_k table[p][k] xpad[n * of + k], xpad = x with `width` zeros in front.
// One output sample per thread; the (nf x K) table sits in LDS when it fits, the input comes through L1/L2
// (every sample is reused K * nf / of times by neighbouring threads).
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ x, const float *__restrict__ table,
                                                       float *__restrict__ y, int length, int of, int nf, int width,
                                                       int K, int out_len, int table_in_lds) {
    extern __shared__ float tab[];
    if (table_in_lds)
        for (int e = threadIdx.x; e < nf * K; e += 256) tab[e] = table[e];
    __syncthreads();
    const float *tb = table_in_lds ? tab : table;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= out_len) return;
    const float *xr = x + size_t(blockIdx.y) * length;
    const int n = i / nf, p = i - n * nf;
    const int first = n * of - width;              // x index of tap 0
    const int k0 = max(0, -first), k1 = min(K, length - first);
    const float *tp = tb + p * K;
    float acc = 0.f;
    for (int k = k0; k < k1; ++k) acc = fmaf(tp[k], xr[first + k], acc);
    y[size_t(blockIdx.y) * out_len + i] = acc;
}

extern "C" {

int64_t agx_fdft_frames(int32_t length, int32_t n_fft, int32_t hop) {
    if (hop <= 0 || length <= n_fft / 2) return agx::fail(AGX_ERR_BAD_SHAPE, "fdft: reflect padding needs length > n_fft / 2");
    return 1 + length / hop;
}

int64_t agx_fdft_packed_floats(int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, int32_t backward) {
    agx::FdftGeom g;
    int rc = agx::fdft_geom(n_fft, win_length, hop, onesided, &g);
    if (rc != AGX_OK) return rc;
    return backward ? agx::packed_weight_floats(g.M, g.Ke, g.Hc) : agx::packed_weight_floats(g.Hc, g.Ke, g.M);
}

int agx_fdft_pack(int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, int32_t window_kind,
                  int32_t norm_kind, int32_t backward, float *packed, void *stream) {
    using namespace agx;
    FdftGeom g;
    int rc = fdft_geom(n_fft, win_length, hop, onesided, &g);
    if (rc != AGX_OK) return rc;
    if (!packed) return fail(AGX_ERR_NULL_POINTER, "fdft_pack: NULL pointer");
    double scale = 1.0;
    if (norm_kind == 1) scale = 1.0 / sqrt(double(n_fft));
    if (norm_kind == 2) {   // 1 / sqrt(sum window^2)
        double e = 0.0;
        for (int k = 0; k < win_length; ++k) {
            const double w = window_kind == 0 ? 1.0 : 0.5 - 0.5 * cos(2.0 * M_PI * double(k) / double(win_length));
            e += w * w;
        }
        scale = 1.0 / sqrt(e);
    }
    const int64_t n = agx_fdft_packed_floats(n_fft, win_length, hop, onesided, backward);
    hipLaunchKernelGGL(fdft_pack_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), packed, g, window_kind, float(scale), backward);
    return check_launch("agx_fdft_pack");
}

int64_t agx_fdft_workspace_bytes(int32_t batch, int32_t length, int32_t n_fft, int32_t hop) {
    const int64_t T = agx_fdft_frames(length, n_fft, hop);
    if (T < 0) return T;
    const int K = n_fft / hop, Hc = agx::ceil_div(hop, agx::kWG) * agx::kWG;
    return int64_t(batch) * Hc * (T + K - 1) * int64_t(sizeof(float));
}

int agx_fdft_forward(const float *x, const float *packed, float *y, void *workspace, int32_t batch, int32_t length,
                     int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, void *stream) {
    using namespace agx;
    FdftGeom g;
    int rc = fdft_geom(n_fft, win_length, hop, onesided, &g);
    if (rc != AGX_OK) return rc;
    const int64_t T64 = agx_fdft_frames(length, n_fft, hop);
    if (T64 < 0) return int(T64);
    if (!x || !packed || !y || !workspace) return fail(AGX_ERR_NULL_POINTER, "fdft_forward: NULL pointer");
    if (batch <= 0 || batch > 32767) return fail(AGX_ERR_BAD_SHAPE, "fdft: batch out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int T = int(T64), Ttau = T + g.Ke - 1;
    float *xc = static_cast<float *>(workspace);
    if (g.Hc != g.H) hipMemsetAsync(xc, 0, size_t(batch) * g.Hc * Ttau * sizeof(float), st);   // padding channels
    launch_stft_prep(x, xc, batch, length, g.N, g.H, Ttau, g.Hc, g.j0, st);
    const agx_conv_desc d = fdft_conv_desc(g, batch, Ttau);
    ConvPlan p;
    rc = lower_conv(&d, &p);
    if (rc != AGX_OK) return rc;
    rc = conv_mfma_supported(p) ? launch_conv_mfma(p, xc, packed, nullptr, nullptr, y, st)
                                : launch_conv_direct(p, xc, packed, nullptr, nullptr, y, st);
    if (rc != AGX_OK) return rc;
    return check_launch("agx_fdft_forward");
}

int agx_fdft_backward(const float *dy, const float *packed_bwd, float *dx, void *workspace, int32_t batch,
                      int32_t length, int32_t n_fft, int32_t win_length, int32_t hop, int32_t onesided, void *stream) {
    using namespace agx;
    FdftGeom g;
    int rc = fdft_geom(n_fft, win_length, hop, onesided, &g);
    if (rc != AGX_OK) return rc;
    const int64_t T64 = agx_fdft_frames(length, n_fft, hop);
    if (T64 < 0) return int(T64);
    if (!dy || !packed_bwd || !dx || !workspace) return fail(AGX_ERR_NULL_POINTER, "fdft_backward: NULL pointer");
    if (batch <= 0 || batch > 32767) return fail(AGX_ERR_BAD_SHAPE, "fdft: batch out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int T = int(T64), Ttau = T + g.Ke - 1;
    float *dxc = static_cast<float *>(workspace);
    const agx_conv_desc d = fdft_conv_desc(g, batch, Ttau);
    ConvPlan p;
    rc = lower_conv_bwd_data(&d, &p);
    if (rc != AGX_OK) return rc;
    rc = conv_mfma_supported(p) ? launch_conv_mfma(p, dy, packed_bwd, nullptr, nullptr, dxc, st)
                                : launch_conv_direct(p, dy, packed_bwd, nullptr, nullptr, dxc, st);
    if (rc != AGX_OK) return rc;
    launch_stft_unprep(dxc, dx, batch, length, g.N, g.H, Ttau, g.Hc, g.j0, st);
    return check_launch("agx_fdft_backward");
}

int64_t agx_fdft_rows(int32_t n_fft, int32_t onesided) {   // rows per half of the channel-major spectrum (bins rounded up to 8)
    const int f = onesided ? n_fft / 2 + 1 : n_fft;
    return agx::ceil_div(f, 8) * 8;
}

int agx_melpower(const float *cv, const float *fb, float *mel, int32_t batch, int32_t bins, int32_t frames,
                 int32_t n_mels, void *stream) {
    using namespace agx;
    if (batch <= 0 || bins <= 0 || frames <= 0 || n_mels <= 0 || batch > 65535)
        return fail(AGX_ERR_BAD_SHAPE, "melpower: bad shape");
    if (!cv || !fb || !mel) return fail(AGX_ERR_NULL_POINTER, "melpower: NULL pointer");
    hipLaunchKernelGGL(melpower_kernel, dim3(ceil_div(frames, 256), ceil_div(n_mels, 8), batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cv, fb, mel, bins, ceil_div(bins, 8) * 8, frames, n_mels);
    return check_launch("agx_melpower");
}

int agx_melpower_backward(const float *cv, const float *fb, const float *dmel, float *dcv, int32_t batch, int32_t bins,
                          int32_t frames, int32_t n_mels, void *stream) {
    using namespace agx;
    if (batch <= 0 || bins <= 0 || frames <= 0 || n_mels <= 0 || batch > 65535 || bins > 65535)
        return fail(AGX_ERR_BAD_SHAPE, "melpower_backward: bad shape");
    if (!cv || !fb || !dmel || !dcv) return fail(AGX_ERR_NULL_POINTER, "melpower_backward: NULL pointer");
    hipLaunchKernelGGL(melpower_bwd_kernel, dim3(ceil_div(frames, 256), ceil_div(bins, 8) * 8, batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cv, fb, dmel, dcv, bins, ceil_div(bins, 8) * 8, frames, n_mels);
    return check_launch("agx_melpower_backward");
}

int agx_preemphasis(const float *x, float *y, int64_t rows, int32_t length, float coeff, int32_t adjoint,
                    void *stream) {
    using namespace agx;
    if (rows <= 0 || rows > 65535 || length <= 0) return fail(AGX_ERR_BAD_SHAPE, "preemphasis: bad shape");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "preemphasis: NULL pointer");
    hipLaunchKernelGGL(preemphasis_kernel, dim3(ceil_div(length, 256), unsigned(rows)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, length, coeff, adjoint);
    return check_launch("agx_preemphasis");
}

int agx_lowpass_biquad(const float *x, float *y, int64_t rows, int32_t length, float sample_rate, float cutoff_freq,
                       float q, void *stream) {
    using namespace agx;
    if (rows <= 0 || length <= 0 || sample_rate <= 0.f || cutoff_freq <= 0.f || q <= 0.f)
        return fail(AGX_ERR_BAD_SHAPE, "lowpass_biquad: bad argument");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "lowpass_biquad: NULL pointer");
    const double w0 = 2.0 * M_PI * double(cutoff_freq) / double(sample_rate);
    const double alpha = sin(w0) / 2.0 / double(q);
    const double b0 = (1 - cos(w0)) / 2, b1 = 1 - cos(w0), b2 = b0, a0 = 1 + alpha, a1 = -2 * cos(w0), a2 = 1 - alpha;
    hipLaunchKernelGGL(biquad_kernel, dim3(unsigned(rows)), dim3(64), 0, static_cast<hipStream_t>(stream), x, y, length, float(b0 / a0), float(b1 / a0), float(b2 / a0), float(a1 / a0), float(a2 / a0));
    return check_launch("agx_lowpass_biquad");
}

int64_t agx_resample_out_len(int64_t length, int32_t orig_freq, int32_t new_freq) {
    if (length < 0 || orig_freq <= 0 || new_freq <= 0) return agx::fail(AGX_ERR_BAD_SHAPE, "resample: bad argument");
    return (int64_t(new_freq) * length + orig_freq - 1) / orig_freq;     // ceil(new * length / orig)
}

int agx_resample(const float *x, const float *table, float *y, int64_t rows, int32_t length, int32_t orig_freq,
                 int32_t new_freq, int32_t width, void *stream) {
    using namespace agx;
    if (rows <= 0 || rows > 65535 || length <= 0 || orig_freq <= 0 || new_freq <= 0 || width < 0)
        return fail(AGX_ERR_BAD_SHAPE, "resample: bad argument");
    if (!x || !table || !y) return fail(AGX_ERR_NULL_POINTER, "resample: NULL pointer");
    const int64_t out_len = agx_resample_out_len(length, orig_freq, new_freq);
    if (out_len > INT32_MAX) return fail(AGX_ERR_BAD_SHAPE, "resample: output too long");
    const int K = 2 * width + orig_freq;
    const size_t tab_bytes = size_t(new_freq) * K * sizeof(float);
    const int in_lds = tab_bytes <= 64 * 1024;
    hipLaunchKernelGGL(resample_kernel, dim3(ceil_div(int(out_len), 256), unsigned(rows)), dim3(256), in_lds ? tab_bytes : 0,
                       static_cast<hipStream_t>(stream), x, table, y, length, orig_freq, new_freq, width, K, int(out_len), in_lds);
    return check_launch("agx_resample");
}

}  // extern "C"
