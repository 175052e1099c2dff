// Attention bottleneck kernels for gfx950, channel-major (B, C, T) layout.
//
// networks/transformers.py: LayerNorm (:159, :214), softmax(QK^T/sqrt(d) + ALiBi) V
// (:175-188, ALiBi :38-39, 62-75).  The Linear layers run as k=1 convolutions
// through conv_mfma.hip; this file holds the two ops that are not convolutions.
//
// Attention on MFMA without any transposition.  With channels-major q/k/v,
//   S^T[j][i] = sum_d K[d][j] Q[d][i]
// is a GEMM whose A operand (rows = keys j) and B operand (columns = queries i)
// are both contiguous in global memory, so QK^T needs no LDS at all.  Its 32x32
// accumulator holds the query on the lane and the keys in the registers, hence
// the softmax over keys is an in-lane reduction over registers plus ONE lane
// shuffle, and the probabilities are already the B operand of
//   O^T[dv][i] = sum_j V[dv][j] P^T[j][i]
// (k-step s <-> accumulator register s, as in resblock_mfma.hip).  Only V goes
// through LDS, because its fragment is strided in global memory.
// Exact fp32 (v_mfma_f32_32x32x2_f32); the reference computes fp32 too.
#include "mfma_tile.hpp"

namespace agx {

int launch_attention_flash(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, float scale_div,
                           int precision, hipStream_t st);   // attention_flash.hip

// ---------------------------------------------------------------------- LayerNorm
// Block = 64 time steps x all channels; wave w takes channels w, w+4, ...
__global__ __launch_bounds__(256) void layernorm_ct_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ weight,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ y, int C, int T, float eps) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int tc = min(t, T - 1);
    const float *xb = x + size_t(blockIdx.y) * C * T + tc;
    float *yb = y + size_t(blockIdx.y) * C * T + tc;

    float s = 0.f;
    for (int c = wave; c < C; c += 4) s += xb[size_t(c) * T];
    part[wave][lane] = s;
    __syncthreads();
    const float mean = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / float(C);
    __syncthreads();
    float v = 0.f;
    for (int c = wave; c < C; c += 4) {
        const float d = xb[size_t(c) * T] - mean;
        v = fmaf(d, d, v);
    }
    part[wave][lane] = v;
    __syncthreads();
    const float var = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / float(C);
    const float rstd = 1.f / sqrtf(var + eps);
    if (t >= T) return;
    for (int c = wave; c < C; c += 4) {
        const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
        yb[size_t(c) * T] = (xb[size_t(c) * T] - mean) * rstd * w + b;
    }
}

// One-pass form (round 3): block = 16 time steps x NG channel groups, a thread keeps its CPT channels (c = g + NG k) of
// one time step in registers -- x is read ONCE, every load is issued before the first use, statistics go through one
// 16 x 16 LDS exchange each (mean, then the centred sum of squares from the registers: the two-pass formula without
// the second pass over memory).  (B, 512, 225): 480 workgroups instead of 128, 14.7 MB moved once.
template <int CPT, int NG, int TS = 16>   // NG channel groups x TS time steps = TS NG threads; channel c = g + NG k, k < CPT
__global__ __launch_bounds__(TS * NG) void layernorm_ct_regs_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ weight,
                                                                    const float *__restrict__ bias,
                                                                    float *__restrict__ y, int C, int T, float eps) {
    __shared__ float part[NG][TS + 1];
    const int tl = threadIdx.x % TS, g = threadIdx.x / TS;
    const int t = blockIdx.x * TS + tl;
    const int tc = min(t, T - 1);
    const float *xb = x + size_t(blockIdx.y) * C * T + tc;
    float v[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + NG * k;
        v[k] = c < C ? xb[size_t(c) * T] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) s += v[k];
    part[g][tl] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) tot += part[j][tl];
    const float mean = tot / float(C);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const float d = (g + NG * k < C) ? v[k] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    part[g][tl] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) var += part[j][tl];
    const float rstd = 1.f / sqrtf(var / float(C) + eps);
    if (t >= T) return;
    float *yb = y + size_t(blockIdx.y) * C * T + t;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + NG * k;
        if (c < C) {
            const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
            yb[size_t(c) * T] = (v[k] - mean) * rstd * w + b;
        }
    }
}

// ---------------------------------------------------------------------- attention
// grid (ceil(T/128), H, B); 4 waves, each 32 queries against all T <= 32*NJ keys.
template <int NJ, int DVT>
__global__ __launch_bounds__(256) void attention_alibi_kernel(const float *__restrict__ qkv,
                                                              const float *__restrict__ slopes,
                                                              float *__restrict__ out, int H, int Dh, int T,
                                                              float scale_div) {
    extern __shared__ __attribute__((aligned(16))) float vs[];  // [32*DVT][TP]
    constexpr int TP = 32 * NJ + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int HD = H * Dh;
    const float *qb = qkv + (size_t(b) * 3 * HD + size_t(h) * Dh) * T;
    const float *kb = qb + size_t(HD) * T;
    const float *vb = kb + size_t(HD) * T;

    // ---- V tile -> LDS (rows >= Dh and columns >= T are zero) ----
    // (eight loads in flight per thread, on clamped addresses, masked afterwards: one conditional load per iteration was 64 serial
    // L2 round trips -- ~45 of the kernel's 110 us at T = 225)
    constexpr int VTOT = 32 * DVT * TP;
    for (int e0 = tid; e0 < VTOT; e0 += 256 * 8) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + 256 * u, VTOT - 1);
            const int dv = e / TP, j = e - dv * TP;
            v8[u] = vb[size_t(min(dv, Dh - 1)) * T + min(j, T - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u;
            if (e < VTOT) {
                const int dv = e / TP, j = e - dv * TP;
                vs[e] = (dv < Dh && j < T) ? v8[u] : 0.f;
            }
        }
    }
    __syncthreads();

    const int i = blockIdx.x * 128 + wave * 32 + li;  // this lane's query
    const int ic = min(i, T - 1);
    if (blockIdx.x * 128 + wave * 32 >= T) return;    // whole wave out of range (after the barrier)

    // ---- S^T = K^T Q : rows = keys, columns = queries ----
    f32x16 acc[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
    int kcol[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) kcol[jj] = min(jj * 32 + li, T - 1);
#pragma unroll 8
    for (int d0 = 0; d0 < Dh; d0 += 2) {
        const int d = min(d0 + lh, Dh - 1);
        const float sel = (d0 + lh < Dh) ? 1.f : 0.f;  // odd head_dim: second half of the last step is a zero term
        const float qv = qb[size_t(d) * T + ic] * sel;
        float kv[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) kv[jj] = kb[size_t(d) * T + kcol[jj]];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[jj], qv, acc[jj], 0, 0, 0);
    }

    // ---- scale, ALiBi, softmax over keys (registers + one shuffle) ----
    const float slope = slopes[h];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jj * 32 + acc_row(r, lh);
            float s = acc[jj][r] / scale_div;
            s += -fabsf(float(ic - j)) * slope;  // == M[h, i, j] of Alibi._create_M
            s = (j < T) ? s : -INFINITY;
            acc[jj][r] = s;
            m = fmaxf(m, s);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pexp = expf(acc[jj][r] - m);
            acc[jj][r] = pexp;
            l += pexp;
        }
    l += __shfl_xor(l, 32);

    // ---- O^T = V P^T : B operand = the probability registers ----
    f32x16 o[DVT];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = jj * 32 + acc_row(s, lh);
#pragma unroll
            for (int dt = 0; dt < DVT; ++dt) {
                const float av = vs[(dt * 32 + li) * TP + j];
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc[jj][s], o[dt], 0, 0, 0);
            }
        }

    const float inv = 1.f / l;
    float *ob = out + (size_t(b) * HD + size_t(h) * Dh) * T;
    if (i < T) {
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dv = dt * 32 + acc_row(r, lh);
                if (dv < Dh) ob[size_t(dv) * T + i] = o[dt][r] * inv;
            }
    }
}

template <int NJ, int DVT>
static int launch_attn(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T,
                       float scale_div, hipStream_t st) {
    const size_t lds = size_t(32 * DVT) * (32 * NJ + 1) * sizeof(float);
    auto kern = attention_alibi_kernel<NJ, DVT>;
    static DeviceOnce once;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, nullptr, "attention")) return rc;
    dim3 grid(ceil_div(T, 128), H, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, qkv, slopes, out, H, Dh, T, scale_div);
    return check_launch("attention_alibi");
}

// ---------------------------------------------------------------------- LayerNorm backward
// Same decomposition as the forward (block = 64 time steps x all channels).  Per column:
//   dxhat_c = dy_c w_c,  dx_c = rstd (dxhat_c - mean_c(dxhat) - xhat_c mean_c(dxhat xhat)) [+ add]
// and per block the partial sums of dy xhat / dy over its 64 columns (reduced by ln_param_reduce_kernel).
__global__ __launch_bounds__(256) void layernorm_ct_bwd_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ weight,
                                                               const float *__restrict__ dy,
                                                               const float *__restrict__ add, float *__restrict__ dx,
                                                               float *__restrict__ part, int C, int T, float eps) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const int tc = min(t, T - 1);
    const size_t base = size_t(blockIdx.y) * C * T + tc;
    auto colsum = [&](float v) -> float {
        __syncthreads();
        red[wave][lane] = v;
        __syncthreads();
        return (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    };
    float s = 0.f;
    for (int c = wave; c < C; c += 4) s += x[base + size_t(c) * T];
    const float mean = colsum(s) / float(C);
    float v = 0.f;
    for (int c = wave; c < C; c += 4) {
        const float d = x[base + size_t(c) * T] - mean;
        v = fmaf(d, d, v);
    }
    const float rstd = 1.f / sqrtf(colsum(v) / float(C) + eps);
    const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    float s1 = 0.f, s2 = 0.f;
    for (int c = wave; c < C; c += 4) {
        const float xh = (x[base + size_t(c) * T] - mean) * rstd;
        const float g = live ? dy[base + size_t(c) * T] : 0.f;
        const float dxh = g * (weight ? weight[c] : 1.f);
        s1 += dxh;
        s2 = fmaf(dxh, xh, s2);
        float pg = g * xh, pb = g;          // parameter gradients: sum over this block's columns
        for (int off = 32; off > 0; off >>= 1) {
            pg += __shfl_xor(pg, off);
            pb += __shfl_xor(pb, off);
        }
        if (lane == 0) {
            part[size_t(blk) * C + c] = pg;
            part[(size_t(nblk) + blk) * C + c] = pb;
        }
    }
    const float m1 = colsum(s1) / float(C);
    const float m2 = colsum(s2) / float(C);
    if (!live) return;
    for (int c = wave; c < C; c += 4) {
        const size_t e = base + size_t(c) * T;
        const float xh = (x[e] - mean) * rstd;
        const float dxh = dy[e] * (weight ? weight[c] : 1.f);
        dx[e] = rstd * (dxh - m1 - xh * m2) + (add ? add[e] : 0.f);
    }
}

__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float *__restrict__ part, int nblk, int C,
                                                              float *__restrict__ dweight, float *__restrict__ dbias) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, b = 0.f;
    for (int k = 0; k < nblk; ++k) {
        g += part[size_t(k) * C + c];
        b += part[(size_t(nblk) + k) * C + c];
    }
    if (dweight) dweight[c] = g;
    if (dbias) dbias[c] = b;
}

// ---------------------------------------------------------------------- attention backward
// One workgroup per (head, batch item); K and V (Dh x T each) stay in LDS, the queries are walked in blocks
// of QB.  Config 3 is 225 frames x 8 heads x 64: 8 GFLOP per batch -- a VALU kernel (the forward's
// register-resident MFMA formulation does not carry over: dK / dV accumulate across query blocks).
//   P = softmax(Q^T K / scale + alibi),  dP = dO^T V,  dS = P (dP - rowsum(dP P)),
//   dQ = dS K^T / scale,  dK = dS^T Q / scale,  dV = P^T dO
template <int QB>
__global__ __launch_bounds__(256) void attention_alibi_bwd_kernel(const float *__restrict__ qkv,
                                                                  const float *__restrict__ slopes,
                                                                  const float *__restrict__ dout,
                                                                  float *__restrict__ dqkv, int H, int Dh, int T,
                                                                  float scale_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ks = sm;                   // [Dh][T]
    float *Vs = Ks + Dh * T;          // [Dh][T]
    float *Qs = Vs + Dh * T;          // [Dh][QB]
    float *Os = Qs + Dh * QB;         // [Dh][QB]   (dO block)
    float *Ps = Os + Dh * QB;         // [QB][T]
    float *Ss = Ps + QB * T;          // [QB][T]    (dS / scale)
    __shared__ float rowa[QB][16], rowb[QB][16];
    const int tid = threadIdx.x;
    const int h = blockIdx.x, b = blockIdx.y;
    const int HD = H * Dh;
    const float *qg = qkv + (size_t(b) * 3 * HD + h * Dh) * T;
    const float *kg = qg + size_t(HD) * T, *vg = kg + size_t(HD) * T;
    const float *og = dout + (size_t(b) * HD + h * Dh) * T;
    float *dqg = dqkv + (size_t(b) * 3 * HD + h * Dh) * T;
    float *dkg = dqg + size_t(HD) * T, *dvg = dkg + size_t(HD) * T;
    const float slope = slopes[h], inv = 1.f / scale_div;
    for (int e = tid; e < Dh * T; e += 256) {
        Ks[e] = kg[e];
        Vs[e] = vg[e];
    }
    constexpr int MAXE = 64;          // dK / dV elements per thread: Dh * T <= 64 * 256
    float dk[MAXE], dv[MAXE];
#pragma unroll
    for (int u = 0; u < MAXE; ++u) dk[u] = dv[u] = 0.f;
    const int rq = tid / 16, rl = tid % 16;  // softmax: 16 threads per query row (QB <= 16)
    for (int i0 = 0; i0 < T; i0 += QB) {
        __syncthreads();
        for (int e = tid; e < Dh * QB; e += 256) {
            const int d = e / QB, q = e - d * QB, i = min(i0 + q, T - 1);
            Qs[e] = qg[size_t(d) * T + i];
            Os[e] = (i0 + q < T) ? og[size_t(d) * T + i] : 0.f;
        }
        __syncthreads();
        // scores and dP
        for (int e = tid; e < QB * T; e += 256) {
            const int q = e / T, j = e - q * T;
            float sacc = 0.f, pacc = 0.f;
            for (int d = 0; d < Dh; ++d) {
                sacc = fmaf(Qs[d * QB + q], Ks[d * T + j], sacc);
                pacc = fmaf(Os[d * QB + q], Vs[d * T + j], pacc);
            }
            Ps[e] = sacc * inv - fabsf(float(i0 + q - j)) * slope;
            Ss[e] = pacc;
        }
        __syncthreads();
        // row softmax + delta = sum_j dP P
        if (rq < QB) {
            float mx = -3.0e38f;
            for (int j = rl; j < T; j += 16) mx = fmaxf(mx, Ps[rq * T + j]);
            rowa[rq][rl] = mx;
        }
        __syncthreads();
        if (rq < QB) {
            float mx = rowa[rq][0];
            for (int k = 1; k < 16; ++k) mx = fmaxf(mx, rowa[rq][k]);
            float sum = 0.f;
            for (int j = rl; j < T; j += 16) {
                const float pe = expf(Ps[rq * T + j] - mx);
                Ps[rq * T + j] = pe;
                sum += pe;
            }
            rowb[rq][rl] = sum;
        }
        __syncthreads();
        if (rq < QB) {
            float sum = 0.f;
            for (int k = 0; k < 16; ++k) sum += rowb[rq][k];
            const float rs = 1.f / sum;
            float dl = 0.f;
            for (int j = rl; j < T; j += 16) {
                const float pn = Ps[rq * T + j] * rs;
                Ps[rq * T + j] = pn;
                dl = fmaf(pn, Ss[rq * T + j], dl);
            }
            rowa[rq][rl] = dl;     // rowa (the row maxima) was last read before the previous barrier
        }
        __syncthreads();
        if (rq < QB) {
            float dl = 0.f;
            for (int k = 0; k < 16; ++k) dl += rowa[rq][k];
            const bool qlive = i0 + rq < T;
            for (int j = rl; j < T; j += 16) {
                const float pn = qlive ? Ps[rq * T + j] : 0.f;
                Ps[rq * T + j] = pn;                                   // rows past T contribute nothing
                Ss[rq * T + j] = pn * (Ss[rq * T + j] - dl) * inv;     // dS / scale
            }
        }
        __syncthreads();
        // dQ[d][q] = sum_j dS[q][j] K[d][j]
        for (int e = tid; e < Dh * QB; e += 256) {
            const int d = e / QB, q = e - d * QB;
            if (i0 + q >= T) continue;
            float acc = 0.f;
            for (int j = 0; j < T; ++j) acc = fmaf(Ss[q * T + j], Ks[d * T + j], acc);
            dqg[size_t(d) * T + i0 + q] = acc;
        }
        // dK[d][j] += sum_q dS[q][j] Q[d][q],  dV[d][j] += sum_q P[q][j] dO[d][q]
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = tid + u * 256;
            if (e < Dh * T) {
                const int d = e / T, j = e - d * T;
                float ak = dk[u], av = dv[u];
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    ak = fmaf(Ss[q * T + j], Qs[d * QB + q], ak);
                    av = fmaf(Ps[q * T + j], Os[d * QB + q], av);
                }
                dk[u] = ak;
                dv[u] = av;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXE; ++u) {
        const int e = tid + u * 256;
        if (e < Dh * T) {
            dkg[e] = dk[u];
            dvg[e] = dv[u];
        }
    }
}

}  // namespace agx

extern "C" {

int agx_layernorm_ct_backward(const float *x, const float *weight, const float *dy, const float *add, float *dx,
                              float *dweight, float *dbias, float *workspace, int32_t batch, int32_t channels,
                              int32_t t, float eps, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || t <= 0) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct_backward: bad shape");
    if (!x || !dy || !dx || !workspace) return fail(AGX_ERR_NULL_POINTER, "layernorm_ct_backward: NULL pointer");
    if (batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct_backward: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(ceil_div(t, 64), batch);
    hipLaunchKernelGGL(layernorm_ct_bwd_kernel, grid, dim3(256), 0, st, x, weight, dy, add, dx, workspace, channels, t,
                       eps);
    if (dweight || dbias)
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ceil_div(channels, 256)), dim3(256), 0, st, workspace,
                           int(grid.x * grid.y), channels, dweight, dbias);
    return check_launch("agx_layernorm_ct_backward");
}

int agx_attention_alibi_backward(const float *qkv, const float *slopes, const float *dout, float *dqkv, int32_t batch,
                                 int32_t heads, int32_t head_dim, int32_t t, float scale_div, void *stream) {
    using namespace agx;
    if (batch <= 0 || heads <= 0 || head_dim <= 0 || t <= 0)
        return fail(AGX_ERR_BAD_SHAPE, "attention_alibi_backward: bad shape");
    if (!qkv || !slopes || !dout || !dqkv) return fail(AGX_ERR_NULL_POINTER, "attention_alibi_backward: NULL pointer");
    if (t > 256 || head_dim > 64) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi_backward: T <= 256, head_dim <= 64");
    if (heads > 65535 || batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "attention_alibi_backward: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto lds_of = [&](int qb) { return size_t(2 * head_dim * t + 2 * head_dim * qb + 2 * qb * t) * sizeof(float); };
    const dim3 grid(heads, batch);
    // the kernel also holds 2 KB of static LDS (row reductions): ask for 156 KB of dynamic space at most
    constexpr size_t kDynMax = 156 * 1024;
    auto run = [&](auto kern, int qb) -> int {
        if (lds_of(qb) > kDynMax) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi_backward: needs %zu B of LDS", lds_of(qb));
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(kDynMax));
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_of(qb), st, qkv, slopes, dout, dqkv, heads, head_dim, t, scale_div);
        return check_launch("agx_attention_alibi_backward");
    };
    return lds_of(16) <= 150 * 1024 ? run(attention_alibi_bwd_kernel<16>, 16) : run(attention_alibi_bwd_kernel<8>, 8);
}


int agx_layernorm_ct(const float *x, const float *weight, const float *bias, float *y, int32_t batch,
                     int32_t channels, int32_t t, float eps, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || t <= 0) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct: bad shape");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "layernorm_ct: NULL pointer");
    if (batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct: batch too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(ceil_div(t, 16), batch);
#define AGX_LN(N, NG) hipLaunchKernelGGL((layernorm_ct_regs_kernel<N, NG>), grid, dim3(16 * NG), 0, st, x, weight, bias, y, channels, t, eps)
    if (channels <= 64) AGX_LN(4, 16);
    else if (channels <= 128) AGX_LN(8, 16);
    else if (channels <= 256) AGX_LN(8, 32);
    else if (channels <= 512) AGX_LN(16, 32);          // 512 threads: 16 loads in flight per thread, one column block per workgroup
    else if (channels <= 1024) AGX_LN(32, 32);
    else if (channels <= 2048) AGX_LN(64, 32);
    else    // > 2048 channels: the three-pass kernel (no register budget for a whole column)
        hipLaunchKernelGGL(layernorm_ct_kernel, dim3(ceil_div(t, 64), batch), dim3(256), 0, st, x, weight, bias, y, channels, t,
                           eps);
#undef AGX_LN
    return check_launch("layernorm_ct");
}

int agx_attention_alibi(const float *qkv, const float *slopes, float *out, int32_t batch, int32_t heads,
                        int32_t head_dim, int32_t t, float scale_div, void *stream) {
    return agx_attention_alibi_ex(qkv, slopes, out, batch, heads, head_dim, t, scale_div, AGX_ATTN_FP32, 0, stream);
}

int agx_attention_alibi_ex(const float *qkv, const float *slopes, float *out, int32_t batch, int32_t heads,
                           int32_t head_dim, int32_t t, float scale_div, int32_t precision, int32_t flash, void *stream) {
    using namespace agx;
    if (batch <= 0 || heads <= 0 || head_dim <= 0 || t <= 0)
        return fail(AGX_ERR_BAD_SHAPE, "attention_alibi: bad shape B=%d H=%d Dh=%d T=%d", batch, heads, head_dim, t);
    if (!qkv || !slopes || !out) return fail(AGX_ERR_NULL_POINTER, "attention_alibi: NULL pointer");
    if (precision != AGX_ATTN_FP32 && precision != AGX_ATTN_BF16)
        return fail(AGX_ERR_BAD_SHAPE, "attention_alibi: unknown precision %d", precision);
    if (head_dim > 128) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi: head_dim=%d > 128", head_dim);
    if (heads > 65535 || batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "attention_alibi: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (t > 256 || precision == AGX_ATTN_BF16 || flash)   // online softmax over key blocks (attention_flash.hip)
        return launch_attention_flash(qkv, slopes, out, batch, heads, head_dim, t, scale_div, precision, st);
    const int nj = t <= 64 ? 2 : (t <= 128 ? 4 : 8);
    const int dvt = head_dim <= 32 ? 1 : (head_dim <= 64 ? 2 : 4);
#define AGX_ATTN(NJ, DVT) return launch_attn<NJ, DVT>(qkv, slopes, out, batch, heads, head_dim, t, scale_div, st)
    if (nj == 2) {
        if (dvt == 1) AGX_ATTN(2, 1);
        if (dvt == 2) AGX_ATTN(2, 2);
        AGX_ATTN(2, 4);
    }
    if (nj == 4) {
        if (dvt == 1) AGX_ATTN(4, 1);
        if (dvt == 2) AGX_ATTN(4, 2);
        AGX_ATTN(4, 4);
    }
    if (dvt == 1) AGX_ATTN(8, 1);
    if (dvt == 2) AGX_ATTN(8, 2);
    AGX_ATTN(8, 4);
#undef AGX_ATTN
}

}  // extern "C"
