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
    for (int e = tid; e < 32 * DVT * TP; e += 256) {
        const int dv = e / TP, j = e - dv * TP;
        vs[e] = (dv < Dh && j < T) ? vb[size_t(dv) * T + j] : 0.f;
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
#pragma unroll 4
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
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(ceil_div(T, 128), H, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, qkv, slopes, out, H, Dh, T, scale_div);
    return check_launch("attention_alibi");
}

}  // namespace agx

extern "C" {

int agx_layernorm_ct(const float *x, const float *weight, const float *bias, float *y, int32_t batch,
                     int32_t channels, int32_t t, float eps, void *stream) {
    using namespace agx;
    if (batch <= 0 || channels <= 0 || t <= 0) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct: bad shape");
    if (!x || !y) return fail(AGX_ERR_NULL_POINTER, "layernorm_ct: NULL pointer");
    if (batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "layernorm_ct: batch too large");
    hipLaunchKernelGGL(layernorm_ct_kernel, dim3(ceil_div(t, 64), batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, weight, bias, y, channels, t, eps);
    return check_launch("layernorm_ct");
}

int agx_attention_alibi(const float *qkv, const float *slopes, float *out, int32_t batch, int32_t heads,
                        int32_t head_dim, int32_t t, float scale_div, void *stream) {
    using namespace agx;
    if (batch <= 0 || heads <= 0 || head_dim <= 0 || t <= 0)
        return fail(AGX_ERR_BAD_SHAPE, "attention_alibi: bad shape B=%d H=%d Dh=%d T=%d", batch, heads, head_dim, t);
    if (!qkv || !slopes || !out) return fail(AGX_ERR_NULL_POINTER, "attention_alibi: NULL pointer");
    if (t > 256) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi: T=%d > 256 (single-pass kernel)", t);
    if (head_dim > 128) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi: head_dim=%d > 128", head_dim);
    if (heads > 65535 || batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "attention_alibi: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
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
