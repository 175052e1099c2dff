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

// ---------------------------------------------------------------------------------------------------------------
// bf16 arithmetic with K and V of one (head, item) staged ONCE per workgroup through LDS as bf16 (round 3; BASELINE
// config 3: T = 225).  The no-LDS kernel above reads every operand element with its own strided 4-byte load -- 128 per
// lane and key block, repeated by each of the 4 waves -- and its 16 MFMAs per block wait on them (0.9 % MFMA
// utilisation).  Here the workgroup's 256 threads read K and V once (8 coalesced loads -> one 16-byte LDS write per
// task), in the fragment layouts of the two products:
//     Ks[d / 16][lane half][key][8 d]                      S^T = K^T Q:  A fragment = one ds_read_b128
//     Vs[key / 16][lane half][dv][8 keys, PV slot order]   O^T = V P^T:  A fragment = one ds_read_b128
// (PV slot e of lane half lh = key 16 hb + 4 lh + (e & 3) + 8 (e >> 2): the probability registers are the B operand as
// they stand).  No barrier inside the key loop.  Same arithmetic as PREC 1 above (operands rounded to bf16, fp32
// accumulation and softmax).  Used when 2 * Tp * DH * 2 bytes fit the LDS (Tp = T rounded up to 64).
// (Round 3, second pass: 8 waves = 256 queries per workgroup -- ONE workgroup stages the K / V of a (head, item) with T <= 256
// instead of two; the staging loads are issued 32 per thread at a time on clamped addresses and masked afterwards: the
// `cond ? load : 0` form made every load wait for its predecessor, 17 serial round trips of ~1.5 us out of 62 us.)
constexpr int ABL_NT = 512;
template <int DVT>
__global__ __launch_bounds__(ABL_NT) void attention_bf16_lds_kernel(const float *__restrict__ qkv, const float *__restrict__ slopes,
                                                                 float *__restrict__ out, int H, int Dh, int T, int Tp,
                                                                 float scale_div) {
    constexpr int KB = 64, DH = 32 * DVT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Ks = smem;                              // [DH / 16][2][Tp][16 B]
    char *Vs = smem + size_t(DH / 16) * 2 * Tp * 16;   // [Tp / 16][2][DH][16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int HD = H * Dh;
    const float *qb = qkv + (size_t(b) * 3 * HD + size_t(h) * Dh) * T;
    const float *kb = qb + size_t(HD) * T;
    const float *vb = kb + size_t(HD) * T;
    // this lane's query operands are requested first, then K and V together: four K tasks + four V tasks = 64 loads in flight per
    // thread on top of the 8 DH / 16 query loads -- at T <= 256 and head_dim 64 the whole staging is ONE round trip
    const int i = blockIdx.x * (ABL_NT / 2) + wave * 32 + li;   // this lane's query
    const int ic = min(i, T - 1);
    const float slope = slopes[h], inv_scale = 1.f / scale_div;
    float qf[DH / 16][8];
#pragma unroll
    for (int kq = 0; kq < DH / 16; ++kq)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[kq][e] = qb[size_t(min(16 * kq + 8 * lh + e, Dh - 1)) * T + ic];
    // K task = (key j, group of 8 head dims); V task = (dv, 16-key half block hbk, lane half): keys 16 hbk + 4 lh + {0..3, 8..11}
    const int nk = Tp * (DH / 8), nv = DH * (Tp / 8);       // (equal)
    for (int u0 = tid; u0 < nk; u0 += ABL_NT * 4) {
        float kk[4][8], vv[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = min(u0 + ABL_NT * it, nk - 1);
            const int j = u % Tp, g8 = u / Tp;        // consecutive threads = consecutive keys: coalesced
#pragma unroll
            for (int e = 0; e < 8; ++e) kk[it][e] = kb[size_t(min(8 * g8 + e, Dh - 1)) * T + min(j, T - 1)];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = min(u0 + ABL_NT * it, nv - 1);
            const int g = u % (Tp / 8), dv = u / (Tp / 8);   // consecutive threads walk along the keys of one row
            const int hbk = g >> 1, vlh = g & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 16 * hbk + 4 * vlh + (e & 3) + 8 * (e >> 2);
                vv[it][e] = vb[size_t(min(dv, Dh - 1)) * T + min(j, T - 1)];
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = u0 + ABL_NT * it;
            if (u < nk) {
                const int j = u % Tp, g8 = u / Tp;
                af_bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)((8 * g8 + e < Dh && j < T) ? kk[it][e] : 0.f);
                *reinterpret_cast<af_bf16x8 *>(Ks + (size_t(g8) * Tp + j) * 16) = v;      // g8 = 2 (d / 16) + lane half
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = u0 + ABL_NT * it;
            if (u < nv) {
                const int g = u % (Tp / 8), dv = u / (Tp / 8);
                const int hbk = g >> 1, vlh = g & 1;
                af_bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = 16 * hbk + 4 * vlh + (e & 3) + 8 * (e >> 2);
                    v[e] = (__bf16)((dv < Dh && j < T) ? vv[it][e] : 0.f);
                }
                *reinterpret_cast<af_bf16x8 *>(Vs + ((size_t(hbk) * 2 + vlh) * DH + dv) * 16) = v;
            }
        }
    }
    af_bf16x8 qh[DH / 16];
#pragma unroll
    for (int kq = 0; kq < DH / 16; ++kq)
#pragma unroll
        for (int e = 0; e < 8; ++e) qh[kq][e] = (__bf16)(16 * kq + 8 * lh + e < Dh ? qf[kq][e] : 0.f);
    __syncthreads();
    if (blockIdx.x * (ABL_NT / 2) + wave * 32 >= T) return;     // (after the barrier) a wave without queries

    f32x16 o[DVT];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int nblk = (T + KB - 1) / KB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int j0 = blk * KB;
        f32x16 acc[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
#pragma unroll
        for (int kq = 0; kq < DH / 16; ++kq)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const af_bf16x8 kh = *reinterpret_cast<const af_bf16x8 *>(Ks + (size_t(2 * kq + lh) * Tp + j0 + 32 * t2 + li) * 16);
                acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[kq], acc[t2], 0, 0, 0);
            }
        float bm = -INFINITY;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + t2 * 32 + acc_row(r, lh);
                float sv = acc[t2][r] * inv_scale - fabsf(float(ic - j)) * slope;
                sv = j < T ? sv : -INFINITY;
                acc[t2][r] = sv;
                bm = fmaxf(bm, sv);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float mn = fmaxf(m, bm);
        const float alpha = expf(m - mn);
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
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                af_bf16x8 ph;
#pragma unroll
                for (int e = 0; e < 8; ++e) ph[e] = (__bf16)acc[t2][8 * hb + e];
                const int hbk = (j0 + 32 * t2 + 16 * hb) >> 4;
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) {
                    const af_bf16x8 vh = *reinterpret_cast<const af_bf16x8 *>(Vs + ((size_t(hbk) * 2 + lh) * DH + dt * 32 + li) * 16);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[dt], 0, 0, 0);
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

template <int DVT>
static int launch_bf16_lds(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, int Tp, float scale_div,
                           hipStream_t st) {
    const size_t lds = size_t(2) * Tp * 32 * DVT * 2;
    auto kern = attention_bf16_lds_kernel<DVT>;
    static DeviceOnce once;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, 160 * 1024, nullptr, "attention_flash")) return rc;
    hipLaunchKernelGGL(kern, dim3(ceil_div(T, ABL_NT / 2), H, B), dim3(ABL_NT), lds, st, qkv, slopes, out, H, Dh, T, Tp, scale_div);
    return check_launch("attention_bf16_lds");
}

template <int DVT, int PREC>
static int launch_flash(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, float scale_div,
                        hipStream_t st) {
    const size_t lds = PREC == 0 ? size_t(2) * 32 * DVT * 65 * sizeof(float) : 0;
    auto kern = attention_flash_kernel<DVT, PREC>;
    static DeviceOnce once;
    if (int rc = prepare_kernel(reinterpret_cast<const void *>(kern), once, lds > 48 * 1024 ? 96 * 1024 : 0, nullptr, "attention_flash")) return rc;
    dim3 grid(ceil_div(T, 128), H, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, qkv, slopes, out, H, Dh, T, scale_div);
    return check_launch("attention_flash");
}

int launch_attention_flash(const float *qkv, const float *slopes, float *out, int B, int H, int Dh, int T, float scale_div,
                           int precision, hipStream_t st) {
    const int dvt = Dh <= 32 ? 1 : (Dh <= 64 ? 2 : 4);
    const int Tp = (T + 63) / 64 * 64;
    if (precision && size_t(2) * Tp * 32 * dvt * 2 <= 128 * 1024) {   // K and V of a (head, item) fit the LDS as bf16
        if (dvt == 1) return launch_bf16_lds<1>(qkv, slopes, out, B, H, Dh, T, Tp, scale_div, st);
        if (dvt == 2) return launch_bf16_lds<2>(qkv, slopes, out, B, H, Dh, T, Tp, scale_div, st);
        return launch_bf16_lds<4>(qkv, slopes, out, B, H, Dh, T, Tp, scale_div, st);
    }
#define AGX_FL(DVT)                                                                                          \
    return precision ? launch_flash<DVT, 1>(qkv, slopes, out, B, H, Dh, T, scale_div, st)                    \
                     : launch_flash<DVT, 0>(qkv, slopes, out, B, H, Dh, T, scale_div, st)
    if (dvt == 1) AGX_FL(1);
    if (dvt == 2) AGX_FL(2);
    AGX_FL(4);
#undef AGX_FL
}

}  // namespace agx

// ---------------------------------------------------------------------------------------------------------------
// Backward for any T (and head_dim <= 128): attention.hip's backward holds K and V of one (head, item) in LDS and the
// dK / dV accumulators in registers -- T <= 256, head_dim <= 64.  Beyond that the work is split the flash way, in
// three deterministic VALU kernels (no atomics; a bottleneck of 1125 frames is 10 GFLOP per batch of 32):
//   stats   lse_i = log sum_j exp(s_ij)  (online over key blocks),  delta_i = sum_d dO[d,i] O[d,i]  (= sum_j P_ij dP_ij)
//   dq      per 16-query block, loop over key blocks:  P = exp(s - lse), dP = dO^T V, dS = P (dP - delta),  dQ += dS K^T / scale
//   dkv     per 64-key block, loop over query blocks:  dK += dS^T Q / scale,  dV += P^T dO   (accumulators in registers)
// s_ij = q_i . k_j / scale - slope |i - j|  (transformers.py:177-183).
namespace agx {

constexpr int AB_QB = 16;    // queries per block
constexpr int AB_KB = 64;    // keys per block

// one workgroup per (query block, head, item): lse and delta of its 16 queries
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const float *__restrict__ qkv, const float *__restrict__ slopes,
                                                             const float *__restrict__ out, const float *__restrict__ dout,
                                                             float *__restrict__ lse, float *__restrict__ delta, int H, int Dh,
                                                             int T, float scale_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Qs = sm;                 // [Dh][QB]
    float *Ks = Qs + Dh * AB_QB;    // [Dh][KB]
    float *Ss = Ks + Dh * AB_KB;    // [QB][KB]
    __shared__ float red[AB_QB][16];
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * AB_QB;
    const int HD = H * Dh;
    const float *qg = qkv + (size_t(b) * 3 * HD + h * Dh) * T, *kg = qg + size_t(HD) * T;
    const float *og = out + (size_t(b) * HD + h * Dh) * T, *dg = dout + (size_t(b) * HD + h * Dh) * T;
    const float slope = slopes[h], inv = 1.f / scale_div;
    for (int e = tid; e < Dh * AB_QB; e += 256) {
        const int d = e / AB_QB, q = e - d * AB_QB;
        Qs[e] = qg[size_t(d) * T + min(i0 + q, T - 1)];
    }
    const int rq = tid / 16, rl = tid % 16;   // 16 threads per query row
    float m = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < T; j0 += AB_KB) {
        __syncthreads();
        for (int e = tid; e < Dh * AB_KB; e += 256) {
            const int d = e / AB_KB, j = e - d * AB_KB;
            Ks[e] = kg[size_t(d) * T + min(j0 + j, T - 1)];
        }
        __syncthreads();
        for (int e = tid; e < AB_QB * AB_KB; e += 256) {
            const int q = e / AB_KB, j = e - q * AB_KB;
            float s = 0.f;
            for (int d = 0; d < Dh; ++d) s = fmaf(Qs[d * AB_QB + q], Ks[d * AB_KB + j], s);
            Ss[e] = (j0 + j < T) ? s * inv - fabsf(float(i0 + q - (j0 + j))) * slope : -3.0e38f;
        }
        __syncthreads();
        float bm = -3.0e38f;
        for (int j = rl; j < AB_KB; j += 16) bm = fmaxf(bm, Ss[rq * AB_KB + j]);
        red[rq][rl] = bm;
        __syncthreads();
        bm = red[rq][0];
        for (int k = 1; k < 16; ++k) bm = fmaxf(bm, red[rq][k]);
        const float mn = fmaxf(m, bm);
        float bs = 0.f;
        for (int j = rl; j < AB_KB; j += 16) bs += expf(Ss[rq * AB_KB + j] - mn);
        __syncthreads();
        red[rq][rl] = bs;
        __syncthreads();
        bs = 0.f;
        for (int k = 0; k < 16; ++k) bs += red[rq][k];
        l = l * expf(m - mn) + bs;
        m = mn;
    }
    // delta_i = sum_d dO[d,i] O[d,i]
    float dl = 0.f;
    const int iq = min(i0 + rq, T - 1);
    for (int d = rl; d < Dh; d += 16) dl = fmaf(dg[size_t(d) * T + iq], og[size_t(d) * T + iq], dl);
    __syncthreads();
    red[rq][rl] = dl;
    __syncthreads();
    if (rl == 0 && i0 + rq < T) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[rq][k];
        const size_t o = (size_t(b) * H + h) * T + i0 + rq;
        lse[o] = m + logf(l);
        delta[o] = s;
    }
}

// one workgroup per (query block, head, item): dQ of its 16 queries, keys in blocks of 64
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float *__restrict__ qkv, const float *__restrict__ slopes,
                                                          const float *__restrict__ dout, const float *__restrict__ lse,
                                                          const float *__restrict__ delta, float *__restrict__ dqkv, int H,
                                                          int Dh, int T, float scale_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Qs = sm;                  // [Dh][QB]
    float *Os = Qs + Dh * AB_QB;     // [Dh][QB]  dO
    float *Ks = Os + Dh * AB_QB;     // [Dh][KB]
    float *Vs = Ks + Dh * AB_KB;     // [Dh][KB]
    float *Ss = Vs + Dh * AB_KB;     // [QB][KB]  dS / scale
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * AB_QB;
    const int HD = H * Dh;
    const float *qg = qkv + (size_t(b) * 3 * HD + h * Dh) * T, *kg = qg + size_t(HD) * T, *vg = kg + size_t(HD) * T;
    const float *dg = dout + (size_t(b) * HD + h * Dh) * T;
    float *dqg = dqkv + (size_t(b) * 3 * HD + h * Dh) * T;
    const float slope = slopes[h], inv = 1.f / scale_div;
    const size_t so = (size_t(b) * H + h) * T;
    for (int e = tid; e < Dh * AB_QB; e += 256) {
        const int d = e / AB_QB, q = e - d * AB_QB, i = min(i0 + q, T - 1);
        Qs[e] = qg[size_t(d) * T + i];
        Os[e] = dg[size_t(d) * T + i];
    }
    constexpr int MAXA = 8;          // dQ elements per thread: Dh * 16 <= 128 * 16 = 8 * 256
    float dq[MAXA];
#pragma unroll
    for (int u = 0; u < MAXA; ++u) dq[u] = 0.f;
    for (int j0 = 0; j0 < T; j0 += AB_KB) {
        __syncthreads();
        for (int e = tid; e < Dh * AB_KB; e += 256) {
            const int d = e / AB_KB, j = e - d * AB_KB, jc = min(j0 + j, T - 1);
            Ks[e] = kg[size_t(d) * T + jc];
            Vs[e] = vg[size_t(d) * T + jc];
        }
        __syncthreads();
        for (int e = tid; e < AB_QB * AB_KB; e += 256) {
            const int q = e / AB_KB, j = e - q * AB_KB, i = i0 + q;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < Dh; ++d) {
                s = fmaf(Qs[d * AB_QB + q], Ks[d * AB_KB + j], s);
                dp = fmaf(Os[d * AB_QB + q], Vs[d * AB_KB + j], dp);
            }
            float ds = 0.f;
            if (i < T && j0 + j < T) {
                const float pn = expf(s * inv - fabsf(float(i - (j0 + j))) * slope - lse[so + i]);
                ds = pn * (dp - delta[so + i]) * inv;
            }
            Ss[e] = ds;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXA; ++u) {
            const int e = tid + u * 256;
            if (e < Dh * AB_QB) {
                const int d = e / AB_QB, q = e - d * AB_QB;
                float a = dq[u];
                for (int j = 0; j < AB_KB; ++j) a = fmaf(Ss[q * AB_KB + j], Ks[d * AB_KB + j], a);
                dq[u] = a;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXA; ++u) {
        const int e = tid + u * 256;
        if (e < Dh * AB_QB) {
            const int d = e / AB_QB, q = e - d * AB_QB;
            if (i0 + q < T) dqg[size_t(d) * T + i0 + q] = dq[u];
        }
    }
}

// one workgroup per (key block, head, item): dK and dV of its 64 keys, queries in blocks of 16
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float *__restrict__ qkv, const float *__restrict__ slopes,
                                                           const float *__restrict__ dout, const float *__restrict__ lse,
                                                           const float *__restrict__ delta, float *__restrict__ dqkv, int H,
                                                           int Dh, int T, float scale_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ks = sm;                  // [Dh][KB]
    float *Vs = Ks + Dh * AB_KB;     // [Dh][KB]
    float *Qs = Vs + Dh * AB_KB;     // [Dh][QB]
    float *Os = Qs + Dh * AB_QB;     // [Dh][QB]
    float *Ps = Os + Dh * AB_QB;     // [QB][KB]
    float *Ss = Ps + AB_QB * AB_KB;  // [QB][KB]
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z, j0 = blockIdx.x * AB_KB;
    const int HD = H * Dh;
    const float *qg = qkv + (size_t(b) * 3 * HD + h * Dh) * T, *kg = qg + size_t(HD) * T, *vg = kg + size_t(HD) * T;
    const float *dg = dout + (size_t(b) * HD + h * Dh) * T;
    float *dkg = dqkv + (size_t(b) * 3 * HD + size_t(HD) + h * Dh) * T, *dvg = dkg + size_t(HD) * T;
    const float slope = slopes[h], inv = 1.f / scale_div;
    const size_t so = (size_t(b) * H + h) * T;
    for (int e = tid; e < Dh * AB_KB; e += 256) {
        const int d = e / AB_KB, j = e - d * AB_KB, jc = min(j0 + j, T - 1);
        Ks[e] = kg[size_t(d) * T + jc];
        Vs[e] = vg[size_t(d) * T + jc];
    }
    constexpr int MAXE = 32;         // dK / dV elements per thread: Dh * 64 <= 128 * 64 = 32 * 256
    float dk[MAXE], dv[MAXE];
#pragma unroll
    for (int u = 0; u < MAXE; ++u) dk[u] = dv[u] = 0.f;
    for (int i0 = 0; i0 < T; i0 += AB_QB) {
        __syncthreads();
        for (int e = tid; e < Dh * AB_QB; e += 256) {
            const int d = e / AB_QB, q = e - d * AB_QB, i = min(i0 + q, T - 1);
            Qs[e] = qg[size_t(d) * T + i];
            Os[e] = (i0 + q < T) ? dg[size_t(d) * T + i] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < AB_QB * AB_KB; e += 256) {
            const int q = e / AB_KB, j = e - q * AB_KB, i = i0 + q;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < Dh; ++d) {
                s = fmaf(Qs[d * AB_QB + q], Ks[d * AB_KB + j], s);
                dp = fmaf(Os[d * AB_QB + q], Vs[d * AB_KB + j], dp);
            }
            float pn = 0.f, ds = 0.f;
            if (i < T && j0 + j < T) {
                pn = expf(s * inv - fabsf(float(i - (j0 + j))) * slope - lse[so + i]);
                ds = pn * (dp - delta[so + i]) * inv;
            }
            Ps[e] = pn;
            Ss[e] = ds;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = tid + u * 256;
            if (e < Dh * AB_KB) {
                const int d = e / AB_KB, j = e - d * AB_KB;
                float ak = dk[u], av = dv[u];
#pragma unroll
                for (int q = 0; q < AB_QB; ++q) {
                    ak = fmaf(Ss[q * AB_KB + j], Qs[d * AB_QB + q], ak);
                    av = fmaf(Ps[q * AB_KB + j], Os[d * AB_QB + q], av);
                }
                dk[u] = ak;
                dv[u] = av;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXE; ++u) {
        const int e = tid + u * 256;
        if (e < Dh * AB_KB) {
            const int d = e / AB_KB, j = e - d * AB_KB;
            if (j0 + j < T) {
                dkg[size_t(d) * T + j0 + j] = dk[u];
                dvg[size_t(d) * T + j0 + j] = dv[u];
            }
        }
    }
}

int launch_attention_flash_backward(const float *qkv, const float *slopes, const float *out, const float *dout, float *dqkv,
                                    float *workspace, int B, int H, int Dh, int T, float scale_div, hipStream_t st) {
    float *lse = workspace, *delta = workspace + size_t(B) * H * T;
    const dim3 gq(ceil_div(T, AB_QB), H, B), gk(ceil_div(T, AB_KB), H, B);
    const size_t l_stats = size_t(Dh * AB_QB + Dh * AB_KB + AB_QB * AB_KB) * sizeof(float);
    const size_t l_dq = size_t(2 * Dh * AB_QB + 2 * Dh * AB_KB + AB_QB * AB_KB) * sizeof(float);
    const size_t l_dkv = size_t(2 * Dh * AB_KB + 2 * Dh * AB_QB + 2 * AB_QB * AB_KB) * sizeof(float);
    static DeviceOnce once[3];
    {
        const void *ks[3] = {reinterpret_cast<const void *>(attn_bwd_stats_kernel), reinterpret_cast<const void *>(attn_bwd_dq_kernel),
                             reinterpret_cast<const void *>(attn_bwd_dkv_kernel)};
        for (int i = 0; i < 3; ++i)
            if (int rc = prepare_kernel(ks[i], once[i], 96 * 1024, nullptr, "attention_flash_backward")) return rc;   // head_dim 128: 90 KB
    }
    hipLaunchKernelGGL(attn_bwd_stats_kernel, gq, dim3(256), l_stats, st, qkv, slopes, out, dout, lse, delta, H, Dh, T, scale_div);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), l_dq, st, qkv, slopes, dout, lse, delta, dqkv, H, Dh, T, scale_div);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, gk, dim3(256), l_dkv, st, qkv, slopes, dout, lse, delta, dqkv, H, Dh, T, scale_div);
    return check_launch("attention_flash_backward");
}

}  // namespace agx

extern "C" {

size_t agx_attention_backward_workspace_bytes(int32_t batch, int32_t heads, int32_t t) {
    if (batch <= 0 || heads <= 0 || t <= 0) return 0;
    return size_t(2) * batch * heads * t * sizeof(float);
}

int agx_attention_alibi_backward_ex(const float *qkv, const float *slopes, const float *out, const float *dout, float *dqkv,
                                    float *workspace, size_t workspace_bytes, int32_t batch, int32_t heads, int32_t head_dim,
                                    int32_t t, float scale_div, void *stream) {
    using namespace agx;
    if (batch <= 0 || heads <= 0 || head_dim <= 0 || t <= 0)
        return fail(AGX_ERR_BAD_SHAPE, "attention_alibi_backward_ex: bad shape");
    if (!qkv || !slopes || !out || !dout || !dqkv || !workspace)
        return fail(AGX_ERR_NULL_POINTER, "attention_alibi_backward_ex: NULL pointer");
    if (head_dim > 128) return fail(AGX_ERR_UNSUPPORTED, "attention_alibi_backward_ex: head_dim=%d > 128", head_dim);
    if (heads > 65535 || batch > 65535) return fail(AGX_ERR_BAD_SHAPE, "attention_alibi_backward_ex: grid too large");
    if (workspace_bytes < agx_attention_backward_workspace_bytes(batch, heads, t))
        return fail(AGX_ERR_WORKSPACE, "attention_alibi_backward_ex: workspace too small");
    return launch_attention_flash_backward(qkv, slopes, out, dout, dqkv, workspace, batch, heads, head_dim, t, scale_div,
                                           static_cast<hipStream_t>(stream));
}

}  // extern "C"
