// Flash-form attention for the transformer bottleneck, any T (gfx950, channel-major (B, C, T) layout).
//
// networks/transformers.py:175-188:  softmax(Q K^T / sqrt(d) + M) V  with the ALiBi bias M[h,i,j] = -slope_h |i - j|
// (:38-39, 62-75), which the reference crops to any T <= context_x (:88-93).  attention.hip holds the single-pass
// kernel (every key of a query in registers, T <= 256); this file removes the length limit -- the reference's own
// inference caller runs 360 000 samples = 1125 frames (training.py:488-496) -- with the online-softmax recurrence
// over key blocks of 64:
//     m' = max(m, max_j s_j),  a = exp(m - m'),  l = a l + sum_j exp(s_j - m'),  O = a O + V P^T,  out = O / l
// The formulation of attention.hip carries over: S^T = K^T Q has the query on the lane and the keys in the
// accumulator registers, so the row statistics are in-lane reductions plus one shuffle and the probabilities are
// already the B operand of O^T = V P^T.
//
//   PREC 0  fp32-input MFMA for both contractions (exact fp32; the parity reference).  V block through LDS.
//   PREC 1  bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate), fp32 softmax -- BASELINE config 3's arithmetic.
//           Operands are rounded to bf16 in registers.  No LDS at all: the k-slot <-> key assignment of the PV product
//           is chosen so that the probability registers are the B operand as they stand (k-slot (lh, e) <-> key
//           4 lh + (e & 3) + 8 (e >> 2) of a 16-key half block) and the matching V fragment is two 16-byte global loads.
#include "mfma_tile.hpp"

namespace agx {

typedef __bf16 af_bf16x8 __attribute__((ext_vector_type(8)));

template <int DVT, int PREC>
__global__ __launch_bounds__(256) void attention_flash_kernel(const float *__restrict__ qkv,
                                                              const float *__restrict__ slopes,
                                                              float *__restrict__ out, int H, int Dh, int T,
                                                              float scale_div) {
    constexpr int KB = 64;         // keys per block (two 32-key accumulator tiles)
    constexpr int DH = 32 * DVT;   // head_dim rounded up to the tile
    constexpr int VP = KB + 1;     // LDS pitch of the V block (PREC 0)
    extern __shared__ __attribute__((aligned(16))) float vs[];   // PREC 0: [2][DH][VP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int HD = H * Dh;
    const float *qb = qkv + (size_t(b) * 3 * HD + size_t(h) * Dh) * T;
    const float *kb = qb + size_t(HD) * T;
    const float *vb = kb + size_t(HD) * T;
    const int i = blockIdx.x * 128 + wave * 32 + li;   // this lane's query
    const int ic = min(i, T - 1);
    const float slope = slopes[h], inv_scale = 1.f / scale_div;
    const int nblk = (T + KB - 1) / KB;

    // ---- the query fragment stays in registers for the whole key loop ----
    float qf[PREC == 0 ? DH / 2 : 1];
    af_bf16x8 qh[PREC == 1 ? DH / 16 : 1];
    if (PREC == 0) {
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) {
            const int d = 2 * s + lh;
            qf[s] = d < Dh ? qb[size_t(d) * T + ic] : 0.f;
        }
    } else {
#pragma unroll
        for (int kq = 0; kq < DH / 16; ++kq)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * kq + 8 * lh + e;
                qh[kq][e] = (__bf16)(d < Dh ? qb[size_t(d) * T + ic] : 0.f);
            }
    }

    f32x16 o[DVT];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    auto stage_v = [&](int blk, float *dst) {   // PREC 0: V[dv < Dh][64 keys of block blk] -> LDS, zeros outside
        for (int e = tid; e < DH * KB; e += 256) {
            const int dv = e / KB, jj = e - dv * KB, j = blk * KB + jj;
            dst[dv * VP + jj] = (dv < Dh && j < T) ? vb[size_t(dv) * T + j] : 0.f;
        }
    };
    if (PREC == 0) {
        stage_v(0, vs);
        __syncthreads();
    }

    for (int blk = 0; blk < nblk; ++blk) {
        const int j0 = blk * KB;
        float *vcur = vs + (blk & 1) * DH * VP;
        if (PREC == 0 && blk + 1 < nblk) stage_v(blk + 1, vs + ((blk + 1) & 1) * DH * VP);   // next block streams in meanwhile

        // ---- S^T = K^T Q for this block: rows = keys, columns = queries ----
        f32x16 acc[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
        int kcol[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) kcol[t2] = min(j0 + t2 * 32 + li, T - 1);
        if (PREC == 0) {
#pragma unroll 4
            for (int s = 0; s < DH / 2; ++s) {
                const int d = min(2 * s + lh, Dh - 1);
                float kv[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) kv[t2] = kb[size_t(d) * T + kcol[t2]];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[t2], qf[s], acc[t2], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kq = 0; kq < DH / 16; ++kq) {
                af_bf16x8 kh[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int d = 16 * kq + 8 * lh + e;
                        kh[t2][e] = (__bf16)(d < Dh ? kb[size_t(d) * T + kcol[t2]] : 0.f);
                    }
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[t2], qh[kq], acc[t2], 0, 0, 0);
            }
        }

        // ---- scale, ALiBi, online softmax (in-lane over the 32 registers + one shuffle) ----
        float bm = -INFINITY;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + t2 * 32 + acc_row(r, lh);
                float s = acc[t2][r] * inv_scale - fabsf(float(ic - j)) * slope;   // == M[h, i, j] of Alibi._create_M
                s = j < T ? s : -INFINITY;
                acc[t2][r] = s;
                bm = fmaxf(bm, s);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float mn = fmaxf(m, bm);            // finite: every block holds at least one key < T
        const float alpha = expf(m - mn);         // first block: exp(-inf) = 0
        float bl = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pe = expf(acc[t2][r] - mn);
                acc[t2][r] = pe;
                bl += pe;
            }
        bl += __shfl_xor(bl, 32);
        l = l * alpha + bl;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

        // ---- O^T += V P^T : B operand = the probability registers ----
        if (PREC == 0) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int jj = t2 * 32 + acc_row(s, lh);
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[(dt * 32 + li) * VP + jj], acc[t2][s], o[dt], 0, 0, 0);
                }
            __syncthreads();   // the next block's V has been written by everyone; this block's is free
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {   // 16-key half blocks: registers 8 hb .. 8 hb + 7
                    af_bf16x8 ph;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ph[e] = (__bf16)acc[t2][8 * hb + e];
                    // register 8 hb + e holds key j0 + 32 t2 + 16 hb + 4 lh + (e & 3) + 8 (e >> 2): two runs of 4 consecutive keys
                    const int jb = j0 + t2 * 32 + 16 * hb + 4 * lh;
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt) {
                        const int dv = min(dt * 32 + li, Dh - 1);
                        const float *vr = vb + size_t(dv) * T;
                        af_bf16x8 vh;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int j = jb + (e & 3) + 8 * (e >> 2);
                            vh[e] = (__bf16)((dt * 32 + li < Dh) ? vr[min(j, T - 1)] : 0.f);   // p = 0 for j >= T
                        }
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[dt], 0, 0, 0);
                    }
                }
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

template <int DVT, int PREC>
static int launch_flash(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, float scale_div,
                        hipStream_t st) {
    const size_t lds = PREC == 0 ? size_t(2) * 32 * DVT * 65 * sizeof(float) : 0;
    auto kern = attention_flash_kernel<DVT, PREC>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(AGX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(ceil_div(T, 128), H, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, qkv, slopes, out, H, Dh, T, scale_div);
    return check_launch("attention_flash");
}

int launch_attention_flash(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, float scale_div,
                           int precision, hipStream_t st) {
    const int dvt = Dh <= 32 ? 1 : (Dh <= 64 ? 2 : 4);
#define AGX_FL(DVT)                                                                                          \
    return precision ? launch_flash<DVT, 1>(qkv, slopes, out, B, H, Dh, T, scale_div, st)                    \
                     : launch_flash<DVT, 0>(qkv, slopes, out, B, H, Dh, T, scale_div, st)
    if (dvt == 1) AGX_FL(1);
    if (dvt == 2) AGX_FL(2);
    AGX_FL(4);
#undef AGX_FL
}

}  // namespace agx
