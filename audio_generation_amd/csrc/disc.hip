// Small kernels of the discriminators (networks/discriminator.py): spectral-norm sigma, AvgPool1d,
// the STFT front end (as a polyphase conv on the MFMA kernels) and the loss reductions.
// All are bandwidth- or latency-trivial next to the conv stacks; written for determinism
// (fixed reduction orders, no atomics).
#include "common.hpp"

namespace agx {

int launch_conv_mfma(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                     const float *res, float *y, hipStream_t st);
int launch_conv_direct(const ConvPlan &p, const float *x, const float *wp, const float *bias,
                       const float *res, float *y, hipStream_t st);
bool conv_mfma_supported(const ConvPlan &p);

__device__ __forceinline__ float block_sum_256(float v, float *sh4) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// ------------------------------------------------------------------ spectral norm
// t[c] = sum_r W[r, c] u[r]   (64 columns per block, rows in 4 interleaved slices)
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float *__restrict__ w, const float *__restrict__ u,
                                                      float *__restrict__ t, int rows, int cols) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.f;
    if (c < cols)
        for (int r = ry; r < rows; r += 4) acc = fmaf(w[size_t(r) * cols + c], u[r], acc);
    part[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) t[c] = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
}

// s[r] = sum_c W[r, c] v[c]   (one wave per row)
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float *__restrict__ w, const float *__restrict__ v,
                                                     float *__restrict__ s, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc = fmaf(w[size_t(r) * cols + c], v[c], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) s[r] = acc;
}

// dst = src / max(||src||, eps)   (F.normalize)   -- single block
__global__ __launch_bounds__(256) void sn_normalize_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int n, float eps) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc = fmaf(src[i], src[i], acc);
    const float nrm = fmaxf(sqrtf(block_sum_256(acc, sh)), eps);
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i] / nrm;
}

// sigma = u . s   -- single block
__global__ __launch_bounds__(256) void sn_dot_kernel(const float *__restrict__ u, const float *__restrict__ s,
                                                     int n, float *__restrict__ sigma) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc = fmaf(u[i], s[i], acc);
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) sigma[0] = tot;
}

// ------------------------------------------------------------------ AvgPool1d
__global__ __launch_bounds__(256) void avgpool1d_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                        int l_in, int l_out, int kernel, int stride, int padding) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= l_out) return;
    const float *row = x + size_t(blockIdx.y) * l_in;
    const int i0 = o * stride - padding;
    float acc = 0.f;
    for (int k = 0; k < kernel; ++k) {
        const int i = i0 + k;
        if (i >= 0 && i < l_in) acc += row[i];
    }
    y[size_t(blockIdx.y) * l_out + o] = acc / float(kernel);  // count_include_pad=True
}

// ------------------------------------------------------------------ STFT front end
// Polyphase view of the framed DFT: with hop H = N / 4 and n = j H + p,
//   Y[c, f, t] = sum_{p < H} sum_{j < 4} D_c[f, j H + p] xp[(t + j) H + p]
// = an unpadded K = 4 conv over H channels xc[p][tau] = xp[tau H + p] (xp = reflect-padded input).
// (tau_off: the first hop-column kept -- the framed DFT of a window shorter than n_fft skips the taps where the window is zero)
__global__ __launch_bounds__(256) void stft_prep_kernel(const float *__restrict__ x, float *__restrict__ xc,
                                                        int L, int N, int H, int Ttau, int chs, int tau_off) {
    // one block per (tau-tile of 64, batch); thread -> (p fastest over reads, tau fastest over writes)
    __shared__ float tile[64][65];
    const int b = blockIdx.z, tau0 = blockIdx.x * 64, p0 = blockIdx.y * 64;
    const int Lp = L + N;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int tt = e >> 6, pp = e & 63;  // consecutive threads: consecutive p = consecutive samples
        const int tau = tau0 + tt, p = p0 + pp;
        float v = 0.f;
        if (tau < Ttau && p < H) {
            const int i = (tau + tau_off) * H + p;
            if (i < Lp) {
                int src = i - N / 2;
                if (src < 0) src = -src;
                if (src >= L) src = 2 * (L - 1) - src;
                v = x[size_t(b) * L + src];
            }
        }
        tile[tt][pp] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int pp = e >> 6, tt = e & 63;
        const int tau = tau0 + tt, p = p0 + pp;
        if (tau < Ttau && p < H) xc[(size_t(b) * chs + p) * Ttau + tau] = tile[tt][pp];
    }
}

// conv output (B, 2N, T) -> (B, 2, T, N)
__global__ __launch_bounds__(256) void stft_transpose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                             int N, int T) {
    __shared__ float tile[64][65];
    const int bc = blockIdx.z;  // b * 2 + c
    const int f0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int ff = e >> 6, tt = e & 63;
        const int f = f0 + ff, t = t0 + tt;
        tile[ff][tt] = (f < N && t < T) ? src[(size_t(bc) * N + f) * T + t] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int tt = e >> 6, ff = e & 63;
        const int f = f0 + ff, t = t0 + tt;
        if (f < N && t < T) dst[(size_t(bc) * T + t) * N + f] = tile[ff][tt];
    }
}

// (B, 2, T, N) -> (B, 2N, T): the transpose back, for the adjoint
__global__ __launch_bounds__(256) void stft_untranspose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               int N, int T) {
    __shared__ float tile[64][65];
    const int bc = blockIdx.z;
    const int f0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int tt = e >> 6, ff = e & 63;
        const int f = f0 + ff, t = t0 + tt;
        tile[tt][ff] = (f < N && t < T) ? src[(size_t(bc) * T + t) * N + f] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int ff = e >> 6, tt = e & 63;
        const int f = f0 + ff, t = t0 + tt;
        if (f < N && t < T) dst[(size_t(bc) * N + f) * T + t] = tile[tt][ff];
    }
}

// adjoint of stft_prep_kernel: dx[n] = sum over the padded positions i that read x[n]
// (i = n + N/2 always; the reflected copies i = N/2 - n for 1 <= n <= N/2 and
//  i = N/2 + 2(L-1) - n for L-1-N/2 <= n <= L-2), with dxc[p][tau] = dxp[tau H + p]
__global__ __launch_bounds__(256) void stft_unprep_kernel(const float *__restrict__ dxc, float *__restrict__ dx, int L,
                                                          int N, int H, int Ttau, int chs, int tau_off) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const int b = blockIdx.y, Lp = L + N, half = N / 2;
    auto at = [&](int i) -> float {
        if (i < 0 || i >= Lp) return 0.f;
        const int th = i / H, p = i - th * H, tau = th - tau_off;
        return (tau >= 0 && tau < Ttau) ? dxc[(size_t(b) * chs + p) * Ttau + tau] : 0.f;
    };
    float acc = at(n + half);
    if (n >= 1 && n <= half) acc += at(half - n);
    if (n <= L - 2 && n >= L - 1 - half) acc += at(half + 2 * (L - 1) - n);
    dx[size_t(b) * L + n] = acc;
}

__global__ __launch_bounds__(256) void avgpool1d_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ add,
                                                            float *__restrict__ dx, int l_in, int l_out, int kernel,
                                                            int stride, int padding) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= l_in) return;
    const float *row = dy + size_t(blockIdx.y) * l_out;
    // outputs o with o*stride - padding <= i < o*stride - padding + kernel
    const int hi = (i + padding) / stride;
    int lo = (i + padding - kernel + stride) / stride;  // ceil((i + padding - kernel + 1) / stride)
    if (i + padding - kernel + 1 <= 0) lo = 0;
    float acc = 0.f;
    for (int o = max(lo, 0); o <= min(hi, l_out - 1); ++o) acc += row[o];
    const size_t e = size_t(blockIdx.y) * l_in + i;
    dx[e] = acc / float(kernel) + (add ? add[e] : 0.f);
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ s,
                                                          float *__restrict__ dz, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) dz[i] = dy[i] * s[i] * (1.f - s[i]);
}

// per-row <G, W>
__global__ __launch_bounds__(256) void sn_rowdot_kernel(const float *__restrict__ g, const float *__restrict__ w,
                                                        float *__restrict__ rowdot, int cols) {
    __shared__ float sh[4];
    const size_t base = size_t(blockIdx.x) * cols;
    float acc = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) acc = fmaf(g[base + c], w[base + c], acc);
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) rowdot[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sn_grad_kernel(float *__restrict__ g, const float *__restrict__ rowdot,
                                                      const float *__restrict__ sigma, const float *__restrict__ u,
                                                      const float *__restrict__ v, int rows, int cols) {
    __shared__ float tot_s;
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < rows; ++i) t += rowdot[i];
        tot_s = t;
    }
    __syncthreads();
    const float sg = sigma[0], coef = tot_s / (sg * sg) * u[blockIdx.x], inv = 1.f / sg;
    const size_t base = size_t(blockIdx.x) * cols;
    for (int c = threadIdx.x; c < cols; c += 256) g[base + c] = g[base + c] * inv - coef * v[c];
}

// packed image of the DFT "weights": row m = c * N + f, channel p, tap j  ->  D_c[f, j H + p] * scale
__global__ __launch_bounds__(256) void stft_pack_kernel(float *__restrict__ packed, int N, int H, float scale) {
    const int M = 2 * N;
    const int64_t total = packed_weight_floats(H, 4, M);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int m = int((e / kWG) % M);
    const int gj = int(e / (int64_t(kWG) * M));
    const int j = gj % 4, p = (gj / 4) * kWG + c16;
    const int c = m / N, f = m - c * N, n = j * H + p;
    const long long k = (long long)f * n % N;  // exact argument reduction
    double sn, cs;
    sincospi(2.0 * double(k) / double(N), &sn, &cs);
    packed[e] = float((c == 0 ? cs : -sn) * double(scale));
}

// image of the adjoint conv (backward-data plan of the DFT conv: channels = the 2N spectrum rows, rows = the
// H phase channels, tap jb <-> forward tap 3 - jb)
__global__ __launch_bounds__(256) void stft_pack_bwd_kernel(float *__restrict__ packed, int N, int H, float scale) {
    const int Mb = H;
    const int64_t total = packed_weight_floats(2 * N, 4, Mb);
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c16 = int(e % kWG);
    const int p = int((e / kWG) % Mb);
    const int gj = int(e / (int64_t(kWG) * Mb));
    const int jb = gj % 4, m = (gj / 4) * kWG + c16;
    const int c = m / N, f = m - c * N, n = (3 - jb) * H + p;
    const long long k = (long long)f * n % N;
    double sn, cs;
    sincospi(2.0 * double(k) / double(N), &sn, &cs);
    packed[e] = float((c == 0 ? cs : -sn) * double(scale));
}

// ------------------------------------------------------------------ loss reductions
__device__ __forceinline__ float loss_term(int mode, float a, float b) {
    switch (mode) {
        case 0: return a;
        case 1: return fminf(a - 1.f, 0.f);
        case 2: return fminf(-a - 1.f, 0.f);
        case 3: return fabsf(a - b);
        case 5: { const float t = logf(a + 1e-8f) - logf(b + 1e-8f); return t * t; }
        default: return fabsf(a + 1e-3f);
    }
}

// Partial sums of one loss term (PAIR = 0) or of the feature-matching pair |x - y| and |x + 1e-3| of one feature map in
// the same pass (PAIR = 1: part[0 .. nb) and part[nb .. 2 nb)).  16-byte loads, four of them in flight per operand and
// thread (one 4-byte load per iteration reached 1.75 TB/s on the 11.7 GB of feature maps of a config-5 step); the order
// of the sum is fixed by (n, grid), not by timing.
template <int PAIR>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             int64_t n, int mode, int vec, float *__restrict__ part) {
    __shared__ float sh[4];
    const bool two = PAIR || mode == 3 || mode == 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bcc[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = int64_t(gridDim.x) * 256, first = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n4 = vec ? n >> 2 : 0;
    // a workgroup sweeps 16 KB contiguous per operand and step, consecutive workgroups consecutive 16 KB (four loads of one
    // thread 4 MB apart -- the grid-stride form -- land in the same memory channel)
    auto term = [&](float a, float b, int u) {
        if (PAIR) { acc[u] += fabsf(a - b); bcc[u] += fabsf(a + 1e-3f); }
        else acc[u] += loss_term(mode, a, b);
    };
    auto quad = [&](const float4 a, const float4 b, int u) { term(a.x, b.x, u); term(a.y, b.y, u); term(a.z, b.z, u); term(a.w, b.w, u); };
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t i = int64_t(blockIdx.x) * 1024 + threadIdx.x;
    for (; i - threadIdx.x + 1024 <= n4; i += 4 * stride) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = x4[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = two ? y4[i + u * 256] : z4;
#pragma unroll
        for (int u = 0; u < 4; ++u) quad(a[u], b[u], u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)      // the workgroup that meets the end of the vector part: its last, partial 16 KB
        if (i - threadIdx.x < n4 && i + u * 256 < n4) quad(x4[i + u * 256], two ? y4[i + u * 256] : z4, u);
    for (int64_t e = 4 * n4 + first; e < n; e += stride) term(x[e], two ? y[e] : 0.f, 0);
    const float tot = block_sum_256((acc[0] + acc[1]) + (acc[2] + acc[3]), sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
    if (PAIR) {
        __syncthreads();
        const float tb = block_sum_256((bcc[0] + bcc[1]) + (bcc[2] + bcc[3]), sh);
        if (threadIdx.x == 0) part[gridDim.x + blockIdx.x] = tb;
    }
}

__global__ __launch_bounds__(256) void reduce_final_kernel(const float *__restrict__ part, int nparts, double inv_n,
                                                           float *__restrict__ out) {
    __shared__ double shd[256];
    double acc = 0.0;
    part += size_t(blockIdx.x) * nparts;   // (the feature-matching pair: one block per sum)
    out += blockIdx.x;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += double(part[i]);
    shd[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) shd[threadIdx.x] += shd[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = float(shd[0] * inv_n);
}

// d mean(term(x, y)) / dx * g  (and / dy for the L1 term), g a device scalar.  PAIR = 1: the gradient of the feature-matching
// pair in one pass, dx = g[0] sign(x - y) / n + g[1] sign(x + 1e-3) / n (each product exact, one rounding in the sum -- what
// adding the two separate gradients gives), dy = -g[0] sign(x - y) / n.  16-byte loads and stores when the pointers allow.
__device__ __forceinline__ float sign_of(float t) { return t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); }

template <int PAIR>
__global__ __launch_bounds__(256) void reduce_mean_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                              int64_t n, int mode, int vec, const float *__restrict__ g,
                                                              float inv_n, float *__restrict__ dx,
                                                              float *__restrict__ dy) {
    const float gs = g[0] * inv_n, gs2 = PAIR ? g[1] * inv_n : 0.f;
    const bool two = PAIR || mode == 3 || mode == 5;
    auto one = [&](float a, float b, float &ox, float &oy) {
        if (PAIR) {
            const float t1 = sign_of(a - b) * gs;
            ox = t1 + sign_of(a + 1e-3f) * gs2;
            oy = -t1;
            return;
        }
        float d;
        switch (mode) {
            case 0: d = 1.f; break;
            case 1: d = (a - 1.f < 0.f) ? 1.f : 0.f; break;       // torch.minimum(a - 1, 0): gradient to a where smaller
            case 2: d = (-a - 1.f < 0.f) ? -1.f : 0.f; break;
            case 3: d = sign_of(a - b); break;
            case 5: d = 2.f * (logf(a + 1e-8f) - logf(b + 1e-8f)); break;   // times 1/(a+eps) resp. -1/(b+eps) below
            default: d = sign_of(a + 1e-3f); break;
        }
        if (mode == 5) { ox = d * gs / (a + 1e-8f); oy = -d * gs / (b + 1e-8f); return; }
        ox = d * gs;
        oy = -d * gs;
    };
    const int64_t stride = int64_t(gridDim.x) * 256, first = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t n4 = vec ? n >> 2 : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    float4 *dx4 = reinterpret_cast<float4 *>(dx), *dy4 = reinterpret_cast<float4 *>(dy);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto quad = [&](int64_t i, const float4 a, const float4 b) {
        float4 ox, oy;
        one(a.x, b.x, ox.x, oy.x); one(a.y, b.y, ox.y, oy.y); one(a.z, b.z, ox.z, oy.z); one(a.w, b.w, ox.w, oy.w);
        if (dx) dx4[i] = ox;
        if (two && dy) dy4[i] = oy;
    };
    int64_t i = int64_t(blockIdx.x) * 512 + threadIdx.x;      // 8 KB contiguous per operand, workgroup and step (see above)
    for (; i - threadIdx.x + 512 <= n4; i += 2 * stride) {
        const float4 a0 = x4[i], a1 = x4[i + 256];
        const float4 b0 = two ? y4[i] : z4, b1 = two ? y4[i + 256] : z4;
        quad(i, a0, b0);
        quad(i + 256, a1, b1);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (i - threadIdx.x < n4 && i + u * 256 < n4) quad(i + u * 256, x4[i + u * 256], two ? y4[i + u * 256] : z4);
    for (int64_t e = 4 * n4 + first; e < n; e += stride) {
        float ox, oy;
        one(x[e], two ? y[e] : 0.f, ox, oy);
        if (dx) dx[e] = ox;
        if (two && dy) dy[e] = oy;
    }
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) y[i] = 1.f / (1.f + expf(-x[i]));
}

void launch_stft_prep(const float *x, float *xc, int batch, int L, int N, int H, int Ttau, int ch_stride, int tau_off,
                      hipStream_t st) {
    hipLaunchKernelGGL(stft_prep_kernel, dim3(ceil_div(Ttau, 64), ceil_div(H, 64), batch), dim3(256), 0, st, x, xc, L, N,
                       H, Ttau, ch_stride, tau_off);
}

void launch_stft_unprep(const float *dxc, float *dx, int batch, int L, int N, int H, int Ttau, int ch_stride, int tau_off,
                        hipStream_t st) {
    hipLaunchKernelGGL(stft_unprep_kernel, dim3(ceil_div(L, 256), batch), dim3(256), 0, st, dxc, dx, L, N, H, Ttau,
                       ch_stride, tau_off);
}

}  // namespace agx

extern "C" {

int agx_spectral_sigma(const float *w, int32_t rows, int32_t cols, float *u, float *v, int32_t power_iterations,
                       float eps, float *sigma, float *workspace, void *stream) {
    using namespace agx;
    if (rows <= 0 || cols <= 0 || power_iterations < 0) return fail(AGX_ERR_BAD_SHAPE, "spectral_sigma: bad shape");
    if (!w || !u || !v || !sigma || !workspace) return fail(AGX_ERR_NULL_POINTER, "spectral_sigma: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *t = workspace, *s = workspace + cols;
    for (int it = 0; it < power_iterations; ++it) {
        hipLaunchKernelGGL(sn_wt_u_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, w, u, t, rows, cols);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, st, t, v, cols, eps);
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, w, v, s, rows, cols);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, st, s, u, rows, eps);
    }
    if (power_iterations == 0)
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, w, v, s, rows, cols);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(1), dim3(256), 0, st, u, s, rows, sigma);
    return check_launch("agx_spectral_sigma");
}

int64_t agx_avgpool1d_out_len(int32_t l_in, int32_t kernel, int32_t stride, int32_t padding) {
    if (l_in <= 0 || kernel <= 0 || stride <= 0 || padding < 0 || l_in + 2 * padding < kernel)
        return agx::fail(AGX_ERR_BAD_SHAPE, "avgpool1d: bad shape");
    return (l_in + 2 * padding - kernel) / stride + 1;
}

int agx_avgpool1d(const float *x, float *y, int64_t rows, int32_t l_in, int32_t kernel, int32_t stride,
                  int32_t padding, void *stream) {
    using namespace agx;
    const int64_t l_out = agx_avgpool1d_out_len(l_in, kernel, stride, padding);
    if (l_out < 0) return int(l_out);
    if (rows <= 0 || rows > 65535) return fail(AGX_ERR_BAD_SHAPE, "avgpool1d: rows must be in [1, 65535]");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "avgpool1d: NULL pointer");
    hipLaunchKernelGGL(avgpool1d_kernel, dim3(ceil_div(int(l_out), 256), unsigned(rows)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, l_in, int(l_out), kernel, stride, padding);
    return check_launch("agx_avgpool1d");
}

static int stft_check(int32_t length, int32_t n_fft) {
    if (n_fft < 64 || (n_fft & (n_fft - 1))) return agx::fail(AGX_ERR_UNSUPPORTED, "stft: n_fft must be a power of two >= 64");
    if (length <= n_fft / 2) return agx::fail(AGX_ERR_BAD_SHAPE, "stft: reflect padding needs length > n_fft / 2");
    return AGX_OK;
}

int64_t agx_stft_frames(int32_t length, int32_t n_fft) {
    int rc = stft_check(length, n_fft);
    return rc != AGX_OK ? rc : 1 + length / (n_fft / 4);
}

int64_t agx_stft_packed_floats(int32_t n_fft) {
    if (n_fft < 64 || (n_fft & (n_fft - 1))) return agx::fail(AGX_ERR_UNSUPPORTED, "stft: n_fft must be a power of two >= 64");
    return agx::packed_weight_floats(n_fft / 4, 4, 2 * n_fft);
}

int agx_stft_pack(int32_t n_fft, int32_t normalized, float *packed, void *stream) {
    using namespace agx;
    const int64_t n = agx_stft_packed_floats(n_fft);
    if (n < 0) return int(n);
    if (!packed) return fail(AGX_ERR_NULL_POINTER, "stft_pack: NULL pointer");
    const float scale = normalized ? float(1.0 / sqrt(double(n_fft))) : 1.f;
    hipLaunchKernelGGL(stft_pack_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), packed, n_fft, n_fft / 4, scale);
    return check_launch("agx_stft_pack");
}

int64_t agx_stft_workspace_bytes(int32_t batch, int32_t length, int32_t n_fft) {
    const int64_t T = agx_stft_frames(length, n_fft);
    if (T < 0) return T;
    if (batch <= 0) return agx::fail(AGX_ERR_BAD_SHAPE, "stft: batch <= 0");
    return (int64_t(batch) * (n_fft / 4) * (T + 3) + int64_t(batch) * 2 * n_fft * T) * int64_t(sizeof(float));
}

int agx_stft_forward(const float *x, const float *packed, float *y, void *workspace, int32_t batch, int32_t length,
                     int32_t n_fft, void *stream) {
    using namespace agx;
    const int64_t T64 = agx_stft_frames(length, n_fft);
    if (T64 < 0) return int(T64);
    if (batch <= 0 || batch > 32767) return fail(AGX_ERR_BAD_SHAPE, "stft: batch out of range");
    if (!x || !packed || !y || !workspace) return fail(AGX_ERR_NULL_POINTER, "stft_forward: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = n_fft, H = N / 4, T = int(T64), Ttau = T + 3;
    float *xc = static_cast<float *>(workspace);
    float *cv = xc + size_t(batch) * H * Ttau;
    launch_stft_prep(x, xc, batch, length, N, H, Ttau, H, 0, st);
    agx_conv_desc d{AGX_CONV_PADDED, batch, H, 2 * N, Ttau, 4, 1, 1, 0, 0.f, AGX_IMPL_MFMA, 1, 0};
    ConvPlan p;
    int rc = lower_conv(&d, &p);
    if (rc != AGX_OK) return rc;
    rc = launch_conv_mfma(p, xc, packed, nullptr, nullptr, cv, st);
    if (rc != AGX_OK) return rc;
    hipLaunchKernelGGL(stft_transpose_kernel, dim3(ceil_div(T, 64), ceil_div(N, 64), batch * 2), dim3(256), 0, st, cv,
                       y, N, T);
    return check_launch("agx_stft_forward");
}

static inline int aligned16(const void *a, const void *b, const void *c, const void *d) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

int agx_reduce_mean(const float *x, const float *y, int64_t n, int32_t mode, float *out, float *workspace,
                    void *stream) {
    using namespace agx;
    if (n <= 0 || mode < 0 || mode > 5) return fail(AGX_ERR_BAD_SHAPE, "reduce_mean: bad n / mode");
    if (!x || !out || !workspace || ((mode == 3 || mode == 5) && !y)) return fail(AGX_ERR_NULL_POINTER, "reduce_mean: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = int(ceil_div64(n, 4096) < 1024 ? ceil_div64(n, 4096) : 1024);
    hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(nb), dim3(256), 0, st, x, y, n, mode, aligned16(x, y, nullptr, nullptr),
                       workspace);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, workspace, nb, 1.0 / double(n), out);
    return check_launch("agx_reduce_mean");
}

int agx_reduce_mean_backward(const float *x, const float *y, int64_t n, int32_t mode, const float *grad, float *dx,
                             float *dy, void *stream) {
    using namespace agx;
    if (n <= 0 || mode < 0 || mode > 5) return fail(AGX_ERR_BAD_SHAPE, "reduce_mean_backward: bad n / mode");
    if (!x || !grad || !dx || ((mode == 3 || mode == 5) && !y)) return fail(AGX_ERR_NULL_POINTER, "reduce_mean_backward: NULL pointer");
    const int nb = int(ceil_div64(n, 2048) < 8192 ? ceil_div64(n, 2048) : 8192);
    hipLaunchKernelGGL(reduce_mean_bwd_kernel<0>, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, mode,
                       aligned16(x, y, dx, dy), grad, float(1.0 / double(n)), dx, dy);
    return check_launch("agx_reduce_mean_backward");
}

int agx_feature_means(const float *x, const float *y, int64_t n, float *out, float *workspace, void *stream) {
    using namespace agx;
    if (n <= 0) return fail(AGX_ERR_BAD_SHAPE, "feature_means: n <= 0");
    if (!x || !y || !out || !workspace) return fail(AGX_ERR_NULL_POINTER, "feature_means: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = int(ceil_div64(n, 4096) < 1024 ? ceil_div64(n, 4096) : 1024);
    hipLaunchKernelGGL(reduce_partial_kernel<1>, dim3(nb), dim3(256), 0, st, x, y, n, 3, aligned16(x, y, nullptr, nullptr), workspace);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(2), dim3(256), 0, st, workspace, nb, 1.0 / double(n), out);
    return check_launch("agx_feature_means");
}

int agx_feature_means_backward(const float *x, const float *y, int64_t n, const float *grad, float *dx, float *dy,
                               void *stream) {
    using namespace agx;
    if (n <= 0) return fail(AGX_ERR_BAD_SHAPE, "feature_means_backward: n <= 0");
    if (!x || !y || !grad || (!dx && !dy)) return fail(AGX_ERR_NULL_POINTER, "feature_means_backward: NULL pointer");
    const int nb = int(ceil_div64(n, 2048) < 8192 ? ceil_div64(n, 2048) : 8192);
    hipLaunchKernelGGL(reduce_mean_bwd_kernel<1>, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, 3,
                       aligned16(x, y, dx, dy), grad, float(1.0 / double(n)), dx, dy);
    return check_launch("agx_feature_means_backward");
}

int agx_sigmoid(const float *x, float *y, int64_t n, void *stream) {
    using namespace agx;
    if (n <= 0) return fail(AGX_ERR_BAD_SHAPE, "sigmoid: n <= 0");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "sigmoid: NULL pointer");
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, n);
    return check_launch("agx_sigmoid");
}

// The adjoint runs the DFT conv's backward-data plan (core.hip:lower_conv_bwd_data: stride 1 -> the flipped
// kernel with the channel roles swapped) on the same MFMA kernel.
static agx_conv_desc stft_conv_desc(int batch, int n_fft, int Ttau) {
    return agx_conv_desc{AGX_CONV_PADDED, batch, n_fft / 4, 2 * n_fft, Ttau, 4, 1, 1, 0, 0.f, AGX_IMPL_MFMA, 1, 0};
}

int agx_stft_pack_bwd(int32_t n_fft, int32_t normalized, float *packed_bwd, void *stream) {
    using namespace agx;
    const int64_t n = agx_stft_packed_floats(n_fft);
    if (n < 0) return int(n);
    if (!packed_bwd) return fail(AGX_ERR_NULL_POINTER, "stft_pack_bwd: NULL pointer");
    const float scale = normalized ? float(1.0 / sqrt(double(n_fft))) : 1.f;
    hipLaunchKernelGGL(stft_pack_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), packed_bwd, n_fft, n_fft / 4, scale);
    return check_launch("agx_stft_pack_bwd");
}

int agx_stft_backward(const float *dy, const float *packed_bwd, float *dx, void *workspace, int32_t batch,
                      int32_t length, int32_t n_fft, void *stream) {
    using namespace agx;
    const int64_t T64 = agx_stft_frames(length, n_fft);
    if (T64 < 0) return int(T64);
    if (batch <= 0 || batch > 32767) return fail(AGX_ERR_BAD_SHAPE, "stft: batch out of range");
    if (!dy || !packed_bwd || !dx || !workspace) return fail(AGX_ERR_NULL_POINTER, "stft_backward: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = n_fft, H = N / 4, T = int(T64), Ttau = T + 3;
    float *dxc = static_cast<float *>(workspace);
    float *cv = dxc + size_t(batch) * H * Ttau;
    hipLaunchKernelGGL(stft_untranspose_kernel, dim3(ceil_div(T, 64), ceil_div(N, 64), batch * 2), dim3(256), 0, st, dy,
                       cv, N, T);
    const agx_conv_desc d = stft_conv_desc(batch, n_fft, Ttau);
    ConvPlan p;
    int rc = lower_conv_bwd_data(&d, &p);
    if (rc != AGX_OK) return rc;
    rc = conv_mfma_supported(p) ? launch_conv_mfma(p, cv, packed_bwd, nullptr, nullptr, dxc, st)
                                : launch_conv_direct(p, cv, packed_bwd, nullptr, nullptr, dxc, st);  // n_fft = 64: 16 rows
    if (rc != AGX_OK) return rc;
    launch_stft_unprep(dxc, dx, batch, length, N, H, Ttau, H, 0, st);
    return check_launch("agx_stft_backward");
}

int agx_avgpool1d_backward(const float *dy, const float *add, float *dx, int64_t rows, int32_t l_in, int32_t kernel,
                           int32_t stride, int32_t padding, void *stream) {
    using namespace agx;
    const int64_t l_out = agx_avgpool1d_out_len(l_in, kernel, stride, padding);
    if (l_out < 0) return int(l_out);
    if (rows <= 0 || rows > 65535) return fail(AGX_ERR_BAD_SHAPE, "avgpool1d_backward: rows must be in [1, 65535]");
    if (!dy || !dx) return fail(AGX_ERR_NULL_POINTER, "avgpool1d_backward: NULL pointer");
    hipLaunchKernelGGL(avgpool1d_bwd_kernel, dim3(ceil_div(l_in, 256), unsigned(rows)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, add, dx, l_in, int(l_out), kernel, stride, padding);
    return check_launch("agx_avgpool1d_backward");
}

int agx_sigmoid_backward(const float *dy, const float *s, float *dz, int64_t n, void *stream) {
    using namespace agx;
    if (n <= 0) return fail(AGX_ERR_BAD_SHAPE, "sigmoid_backward: n <= 0");
    if (!dy || !s || !dz) return fail(AGX_ERR_NULL_POINTER, "sigmoid_backward: NULL pointer");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, s, dz, n);
    return check_launch("agx_sigmoid_backward");
}

int agx_spectral_grad(float *g, const float *w, const float *sigma, const float *u, const float *v, int32_t rows,
                      int32_t cols, float *workspace, void *stream) {
    using namespace agx;
    if (rows <= 0 || cols <= 0) return fail(AGX_ERR_BAD_SHAPE, "spectral_grad: bad shape");
    if (!g || !w || !sigma || !u || !v || !workspace) return fail(AGX_ERR_NULL_POINTER, "spectral_grad: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3(rows), dim3(256), 0, st, g, w, workspace, cols);
    hipLaunchKernelGGL(sn_grad_kernel, dim3(rows), dim3(256), 0, st, g, workspace, sigma, u, v, rows, cols);
    return check_launch("agx_spectral_grad");
}

}  // extern "C"
